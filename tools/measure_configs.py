#!/usr/bin/env python3
"""Measure the sweeps on the BASELINE configurations other than the bench default (C2, C3, C4 with
1 % missing data, C5) and print one JSON line per configuration.  Development tool: bench.py is the
contractual benchmark; these are the rows of the table in DESIGN.md."""

import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bench import synthetic_thresholds  # noqa: E402
from ferromic_amd import _abi, device  # noqa: E402

CONFIGS = {
    # name: (sites, haplotypes, populations, kind, missing_rate)
    "C2": (1_000_000, 1_000, 2, "hudson", 0.0),
    "C3": (5_000_000, 2_500, 4, "wc", 0.0),
    "C4": (10_000_000, 5_000, 2, "hudson", 0.0),
    "C4m": (10_000_000, 5_000, 2, "hudson", 0.01),
    "C5": (2_000_000, 10_000, 2, "hudson", 0.0),
    "C3h": (5_000_000, 2_500, 4, "summaries", 0.0),
    "C2x10": (10_000_000, 1_000, 2, "hudson", 0.0),  # C2 rows, 10x the sites: separates fixed launch cost from per-row efficiency
    "NARROW": (10_000_000, 640, 2, "hudson", 0.0),  # 40 vectors per row: the per-row instruction stream a bit-packed C4 row would have
    "NARROWm": (10_000_000, 640, 2, "hudson", 0.01),
    "C4f": (10_000_000, 5_000, 2, "pair", 0.0),  # the fused region sweep (summaries + both groups' diversity + Hudson, one read) at C4's width
    "C2f": (1_000_000, 1_000, 2, "pair", 0.0),
    "WIDE": (100_000, 200_000, 2, "hudson", 0.0),  # byte masks beyond the LDS budget: bit masks in LDS (FMH_MASK_MODE=1: global)
}


def main():
    names = sys.argv[1:] or ["C2", "C3", "C4m", "C5"]
    lib = _abi.load()
    for name in names:
        S, H, P, kind, miss = CONFIGS[name]
        N = H // 2
        seed = S + N
        pop_of_sample = np.minimum(np.arange(N) * P // N, P - 1).astype(np.uint8)
        poc = np.repeat(pop_of_sample, 2)
        base = synthetic_thresholds(S, 0, seed)
        thr = np.stack([base[p % 2] for p in range(P)])
        dm = device.DeviceMatrix.alloc(S, N, 2, with_missing=miss > 0, max_allele=1)
        dm.generate(seed, 0, thr, poc, int(miss * (1 << 24)))
        layout = os.environ.get("MEASURE_LAYOUT", "packed")
        if layout == "packed":
            dm.pack(release_bytes=True)
        masks = np.stack([(poc == p) for p in range(P)]).astype(np.uint8)
        g = device.Groups(dm, masks)
        bufs = []

        def buf(nbytes):
            b = device.DeviceBuffer(0, nbytes)
            bufs.append(b)
            return b.ptr

        if kind == "hudson":
            sites = _abi.HudsonSites(None, buf(8 * S), buf(8 * S), buf(8 * S), buf(8 * S), buf(8 * S), buf(8 * S), buf(8 * S))
            tot = _abi.HudsonTotals()
            w_out = 56
            mode = os.environ.get("MEASURE_NO_SITE_OUTPUTS")
            if mode == "1":
                sites = _abi.HudsonSites()
            elif mode == "f64only":
                sites.d_alt = None
                sites.d_called = None
            elif mode == "u32only":
                sites = _abi.HudsonSites(None, None, None, None, None, None, sites.d_alt, sites.d_called)
            elif mode == "one":
                sites = _abi.HudsonSites(None, sites.d_dxy, None, None, None, None, None, None)

            def step():
                _abi.check(lib.fmh_hudson_sweep(dm._h, g._h, 0, S, _abi.FORMULA_DENSE, C.byref(sites), C.byref(tot), None))
        elif kind == "pair":
            sites = _abi.HudsonSites(buf(8 * S), buf(8 * S), buf(8 * S), buf(8 * S), buf(8 * S), buf(8 * S), buf(8 * S), buf(8 * S))
            div = _abi.PairDiversitySites(buf(16 * S), buf(16 * S))
            tot = _abi.HudsonTotals()
            w_out = 64 + 32  # fst, dxy, pi1, pi2, num, den f64 + alt, called u32 x 2 groups; diversity pi, theta f64 x 2 groups

            def step():
                _abi.check(lib.fmh_pair_region_sweep(dm._h, g._h, 0, S, _abi.FORMULA_DENSE, _abi.FORMULA_SPARSE, C.byref(div), C.byref(sites), C.byref(tot), None))
        elif kind == "wc":
            nw = 1 + P * (P - 1) // 2
            pa, pb, ps, pn = buf(8 * nw * S), buf(8 * nw * S), buf(nw * S), buf(4 * P * S)
            mode = os.environ.get("MEASURE_NO_SITE_OUTPUTS")
            if mode == "1":  # totals only: how much of the time is the stores
                pa = pb = ps = pn = None
            elif mode == "nostate":
                ps = None
            elif mode == "stateonly":
                pa = pb = pn = None
            elif mode == "abonly":
                ps = pn = None
            elif mode == "aonly":
                pb = ps = pn = None
            tot = _abi.WcTotals()
            w_out = nw * 17 + 4 * P  # a, b f64 + state u8 per slot, called u32 per group

            def step():
                _abi.check(lib.fmh_wc_sweep(dm._h, g._h, 0, S, pa, pb, ps, pn, C.byref(tot), None))
        else:
            pa, pc = buf(4 * P * S), buf(4 * P * S)
            tot = (_abi.PopTotals * P)()
            w_out = 8 * P

            def step():
                _abi.check(lib.fmh_population_summaries(dm._h, g._h, 0, S, _abi.FORMULA_SUMMARY, pa, pc, tot, None))

        # Warm-up by TIME, not by count: the first ~10 ms of launches after an idle period run 10-20 % slower (clocks / power state ramping:
        # ab_env's first block of 20 C3 launches measured 0.50 ms, the following blocks 0.43), which is most of a short kernel's whole
        # measurement.  The GPU is kept busy with ANOTHER kernel (a summaries sweep of the same matrix, no outputs) so that a rocprofv3
        # kernel trace of this script sees the measured kernel only in its steady state: three launches of it, then the timed ten.
        warm_tot = (_abi.PopTotals * P)()
        t_w = time.perf_counter()
        n_w = 0
        while time.perf_counter() - t_w < float(os.environ.get("MEASURE_WARMUP_S", "0.05")) and n_w < 5000:
            _abi.check(lib.fmh_population_summaries(dm._h, g._h, 0, S, _abi.FORMULA_SUMMARY, None, None, warm_tot, None))
            n_w += 1
        for _ in range(3):
            step()
        lib.fmh_timing_enable(1)
        lib.fmh_timing_reset()
        steps = 10
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        el = time.perf_counter() - t0
        ms, n = C.c_double(), C.c_uint64()
        lib.fmh_timing_read(C.byref(ms), C.byref(n))
        lib.fmh_timing_enable(0)
        row_b = (H + 7) // 8 if layout == "packed" else H   # resident genotype bytes per site
        b_site = row_b + (((H + 7) // 8) if miss > 0 else 0) + w_out
        b_site_u8 = H + (((H + 7) // 8) if miss > 0 else 0) + w_out
        k_s = ms.value / 1e3 / n.value
        print(json.dumps({"config": name, "layout": layout, "sites": S, "haplotypes": H, "populations": P, "kind": kind, "missing": miss,
                          "sites_per_s": S * steps / el, "ms_per_step": el / steps * 1e3, "kernel_ms": k_s * 1e3,
                          "bytes_per_site": b_site, "achieved_GBs": b_site * S / k_s / 1e9,
                          "frac_of_8TBs": b_site * S / k_s / 8e12, "warmup_launches": n_w, "u8_layout_equivalent_GBs": b_site_u8 * S / k_s / 1e9}), flush=True)
        del bufs, g, dm


if __name__ == "__main__":
    main()
