#!/usr/bin/env python3
"""Development measurement: W&C sweep time vs number of groups, fused kernel (G <= 8) against the counts path
(fmh_wc_sweep_many), 2 M sites x 2 500 haplotypes, biallelic, no missing data; outputs kept on the device."""
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


def main():
    lib = _abi.load()
    S, H = int(os.environ.get("MEASURE_SITES", "2000000")), int(os.environ.get("MEASURE_HAPLOTYPES", "2500"))
    N = H // 2
    for G in (int(x) for x in (sys.argv[1:] or ["2", "4", "5", "8", "12", "26"])):
        pop_of_sample = np.minimum(np.arange(N) * G // N, G - 1).astype(np.uint8)
        poc = np.repeat(pop_of_sample, 2)
        base = synthetic_thresholds(S, 0, 7)
        thr = np.stack([base[p % 2] for p in range(G)])
        dm = device.DeviceMatrix.alloc(S, N, 2, with_missing=False, max_allele=1)
        dm.generate(7, 0, thr, poc, 0)
        if os.environ.get("MEASURE_LAYOUT", "packed") == "packed":
            dm.pack(release_bytes=True)
        masks = np.ascontiguousarray(np.stack([(poc == p) for p in range(G)]).astype(np.uint8))
        nw = 1 + G * (G - 1) // 2
        bufs = [device.DeviceBuffer(0, 8 * nw * S), device.DeviceBuffer(0, 8 * nw * S), device.DeviceBuffer(0, nw * S), device.DeviceBuffer(0, 4 * G * S)]
        out = {"groups": G, "sites": S, "haplotypes": H}
        if G <= 8:
            g = device.Groups(dm, masks)
            tot = _abi.WcTotals()
            fused = lambda: _abi.check(lib.fmh_wc_sweep(dm._h, g._h, 0, S, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, C.byref(tot), None))
            fused(); t0 = time.perf_counter(); [fused() for _ in range(5)]; out["fused_ms"] = (time.perf_counter() - t0) / 5 * 1e3
        sa, sb, si = np.zeros(nw), np.zeros(nw), np.zeros(nw, dtype=np.uint64)
        many = lambda: _abi.check(lib.fmh_wc_sweep_many(dm._h, device._ptr(masks), G, 0, S, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr,
                                                        device._ptr(sa), device._ptr(sb), device._ptr(si), None))
        many(); t0 = time.perf_counter(); [many() for _ in range(3)]; out["counts_path_ms"] = (time.perf_counter() - t0) / 3 * 1e3
        ta, tb, ti = np.zeros(nw), np.zeros(nw), np.zeros(nw, dtype=np.uint64)
        totals = lambda: _abi.check(lib.fmh_wc_sweep_many(dm._h, device._ptr(masks), G, 0, S, None, None, None, None, device._ptr(ta), device._ptr(tb), device._ptr(ti), None))
        totals(); t0 = time.perf_counter(); [totals() for _ in range(3)]; out["counts_path_totals_only_ms"] = (time.perf_counter() - t0) / 3 * 1e3
        out["totals_only_agrees"] = bool(np.allclose(ta, sa, rtol=1e-9, atol=1e-9) and np.array_equal(ti, si))
        if G <= 8:
            fused_tot = lambda: _abi.check(lib.fmh_wc_sweep(dm._h, g._h, 0, S, None, None, None, None, C.byref(tot), None))
            fused_tot(); t0 = time.perf_counter(); [fused_tot() for _ in range(5)]; out["fused_totals_only_ms"] = (time.perf_counter() - t0) / 5 * 1e3
            out["sums_agree"] = bool(np.allclose(sa[:nw], np.array(tot.sum_a[:nw]), rtol=1e-9, atol=1e-9))
        print(json.dumps(out), flush=True)
        del bufs
        dm.close()


if __name__ == "__main__":
    main()
