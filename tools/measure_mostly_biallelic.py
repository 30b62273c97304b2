#!/usr/bin/env python3
"""Development measurement: what a few multi-allelic sites cost the other sweeps.  One cohort (MEASURE_SITES x MEASURE_HAPLOTYPES, default
400 000 x 5 000) as a biallelic matrix and as multi-allelic ones whose share MEASURE_MULTI_FRACTION (default 0.002) of the sites carries
alleles up to 2 / up to 7: W&C with four groups (all tracks), four-group summaries, the fused region sweep, by the library's HIP events."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ferromic_amd import _abi, device  # noqa: E402


def timed(lib, call, reps=5):
    for _ in range(2):
        call()
    lib.fmh_timing_enable(1)
    lib.fmh_timing_reset()
    for _ in range(reps):
        call()
    ms, n = C.c_double(), C.c_uint64()
    lib.fmh_timing_read(C.byref(ms), C.byref(n))
    lib.fmh_timing_enable(0)
    return ms.value / max(n.value, 1)


def main():
    lib = _abi.load()
    rng = np.random.default_rng(0)
    S, H = int(os.environ.get("MEASURE_SITES", "400000")), int(os.environ.get("MEASURE_HAPLOTYPES", "5000"))
    N = H // 2
    frac = float(os.environ.get("MEASURE_MULTI_FRACTION", "0.002"))
    base = rng.integers(0, 2, size=(S, H), dtype=np.uint8)
    pick = rng.random((S, 1)) < frac
    quarter = np.minimum(np.arange(H) * 4 // H, 3)
    masks4 = np.ascontiguousarray(np.stack([quarter == p for p in range(4)]).astype(np.uint8))
    masks2 = np.ascontiguousarray(np.stack([quarter < 2, quarter >= 2]).astype(np.uint8))
    for max_allele in (1, 2, 7):
        data = base if max_allele == 1 else np.where(pick, rng.integers(0, max_allele + 1, size=(S, H), dtype=np.uint8), base).astype(np.uint8)
        dm = device.DeviceMatrix.from_host(data, None, S, N, 2, int(data.max()))
        g4, g2 = device.Groups(dm, masks4), device.Groups(dm, masks2)
        out = {"sites": S, "haplotypes": H, "max_allele": int(data.max()), "multi_allelic_site_share": 0.0 if max_allele == 1 else frac}
        nw = 7
        bufs = [device.DeviceBuffer(0, 8 * nw * S), device.DeviceBuffer(0, 8 * nw * S), device.DeviceBuffer(0, nw * S), device.DeviceBuffer(0, 4 * 4 * S)]
        tot = _abi.WcTotals()
        out["wc_4_groups_ms"] = timed(lib, lambda: _abi.check(lib.fmh_wc_sweep(dm._h, g4._h, 0, S, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, C.byref(tot), None)))
        out["summaries_4_groups_dense_ms"] = timed(lib, lambda: device.population_summaries(dm, g4, device.FORMULA_DENSE, want_sites=False))
        hb = [device.DeviceBuffer(0, 8 * S) for _ in range(7)]
        sites = _abi.HudsonSites(None, *[b.ptr for b in hb])
        ht = _abi.HudsonTotals()
        out["hudson_dense_ms"] = timed(lib, lambda: _abi.check(lib.fmh_hudson_sweep(dm._h, g2._h, 0, S, _abi.FORMULA_DENSE, C.byref(sites), C.byref(ht), None)))
        print(json.dumps(out), flush=True)
        del bufs, hb
        dm.close()


if __name__ == "__main__":
    main()
