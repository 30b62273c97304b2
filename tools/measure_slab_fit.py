#!/usr/bin/env python3
"""Where does the time of a SHORT sweep go (the per-rank slab of the 8-GPU run: 1.25 M sites x 5 000 haplotypes)?

    tools/measure_slab_fit.py [haplotypes]

Hudson sweeps (fmh_hudson_sweep, all per-site tracks) over row ranges of one resident cohort: the strong-scaling slab sizes (10 M / 2^k),
and row counts that are whole multiples of one ROUND of the persistent grid (grid waves x 64 rows: every wave sweeps exactly k tiles)
next to counts one tile past them.  Kernel time from the library's HIP events, best and mean of 3 x 20 launches, warm.  The line of the
whole-round sizes is time = a + b * rounds (a: launch ramp, mask staging, block reduction; b: one round of tiles), and the distance of
the off-round sizes from that line is the quantisation of the last round.  One JSON line per size, then the fit."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ferromic_amd import _abi, device  # noqa: E402
import bench  # noqa: E402


def main():
    H = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    N = H // 2
    S = int(os.environ.get("SLAB_FIT_SITES", "5000000"))
    lib = _abi.load()
    poc = np.zeros(H, dtype=np.uint8)
    poc[H // 2:] = 1
    masks = np.stack([(poc == 0), (poc == 1)]).astype(np.uint8)
    thr = bench.synthetic_thresholds(S, 0, S + N)
    dm = device.DeviceMatrix.alloc(S, N, 2, with_missing=False, max_allele=1, device=0)
    dm.generate(S + N, 0, thr, poc, 0)
    dm.pack(release_bytes=True)
    groups = device.Groups(dm, masks)
    bufs = [device.DeviceBuffer(0, 8 * S) for _ in range(7)]
    sites = _abi.HudsonSites(None, *(b.ptr for b in bufs))
    totals = _abi.HudsonTotals()

    def sweep(rows):
        _abi.check(lib.fmh_hudson_sweep(dm._h, groups._h, 0, rows, _abi.FORMULA_DENSE, C.byref(sites), C.byref(totals), None))

    # the persistent grid of this shape: read back from a long launch (tools see it through fmh_last_grid when the library exports it)
    grid_waves = int(os.environ.get("SLAB_FIT_GRID_WAVES", "3072"))
    round_rows = grid_waves * 64
    sizes = []
    for k in (1, 2, 3, 4, 6, 8, 12, 16, 24):
        if k * round_rows <= S:
            sizes += [(k * round_rows, f"{k} whole rounds"), (k * round_rows + 64, f"{k} rounds + 1 tile"), (k * round_rows + round_rows // 2, f"{k}.5 rounds")]
    for s in (10_000_000 // 32, 10_000_000 // 16, 10_000_000 // 8, 10_000_000 // 4, 10_000_000 // 2):
        if s <= S:
            sizes.append((s, "strong-scaling slab"))
    for _ in range(200):  # clocks up
        sweep(S // 4)
    out = []
    for rows, what in sorted(sizes):
        best, mean = 1e9, 0.0
        for rep in range(3):
            for _ in range(3):
                sweep(rows)
            lib.fmh_timing_enable(1)
            lib.fmh_timing_reset()
            for _ in range(20):
                sweep(rows)
            ms, n = C.c_double(), C.c_uint64()
            lib.fmh_timing_read(C.byref(ms), C.byref(n))
            kmin, kmax = C.c_double(), C.c_double()
            lib.fmh_timing_read_minmax(C.byref(kmin), C.byref(kmax))
            lib.fmh_timing_enable(0)
            best = min(best, kmin.value)
            mean += ms.value / max(n.value, 1) / 3
        tiles = (rows + 63) // 64
        rec = {"sites": rows, "haplotypes": H, "what": what, "tiles": tiles, "rounds": tiles / grid_waves, "kernel_ms_min": round(best, 5), "kernel_ms_mean": round(mean, 5)}
        out.append(rec)
        print(json.dumps(rec), flush=True)
    whole = [r for r in out if r["what"].endswith("whole rounds")]
    if len(whole) >= 2:
        x = np.array([r["rounds"] for r in whole])
        y = np.array([r["kernel_ms_mean"] for r in whole])
        b, a = np.polyfit(x, y, 1)
        print(json.dumps({"fit": "kernel_ms_mean = a + b * rounds over the whole-round sizes", "a_ms": round(float(a), 5), "b_ms_per_round": round(float(b), 5),
                          "grid_waves": grid_waves, "bytes_per_round": round_rows * (H // 8 + 56),
                          "b_as_GBs": round(round_rows * (H // 8 + 56) / (float(b) * 1e-3) / 1e9, 1)}), flush=True)


if __name__ == "__main__":
    main()
