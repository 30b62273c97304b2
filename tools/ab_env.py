#!/usr/bin/env python3
"""Same-process A/B of a library option (fmh_set_option: FMH_PACKED_NO_PREFETCH, FMH_PACKED_UNROLL, FMH_PACKED_LPR, FMH_MASK_MODE, ...):

    tools/ab_env.py FMH_PACKED_NO_PREFETCH=1

Hudson sweeps over row ranges of one resident cohort, alternating "unset" and the given setting, kernel time from the library's HIP events,
best of three blocks of 20 launches each.  Box-to-box and process-to-process spread of the same kernel is several per cent (1.21-1.38 ms for
the C4 sweep in round 2), larger than most effects worth measuring, while inside one process on one allocation repeats agree to 0.2 %: that
is where an A/B has to be made.  (Round 2 used it to reject an "evenly dealt last round of tiles": predicted -8 % at 1.25 M sites from the
round count, measured +0.9 % - the waves that finish early leave their bandwidth to the others.)"""
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
    if len(sys.argv) < 2 or "=" not in sys.argv[1]:
        raise SystemExit(__doc__)
    var, value = sys.argv[1].split("=", 1)
    lib = _abi.load()
    cases = [(10_000_000, 2500, "packed"), (2_000_000, 5000, "packed"), (5_000_000, 1250, "packed"), (4_000_000, 500, "packed"), (2_000_000, 2500, "bytes")]
    if len(sys.argv) > 2:  # tools/ab_env.py VAR=value SITESxSAMPLES[:layout] ...
        cases = [(int(c.split(":")[0].split("x")[0]), int(c.split(":")[0].split("x")[1]), c.split(":", 1)[1] if ":" in c else "packed") for c in sys.argv[2:]]
    for S, N, layout in cases:
        H = 2 * N
        poc = np.zeros(H, dtype=np.uint8)
        poc[H // 2:] = 1
        masks = np.stack([(poc == 0), (poc == 1)]).astype(np.uint8)
        thr = bench.synthetic_thresholds(S, 0, S + N)
        missing = layout.endswith(":m")  # SITESxSAMPLES:packed:m = 1 % missing calls
        layout = layout.split(":")[0]
        dm = device.DeviceMatrix.alloc(S, N, 2, with_missing=missing, max_allele=1, device=0)
        dm.generate(S + N, 0, thr, poc, int(0.01 * (1 << 24)) if missing else 0)
        if layout == "packed":
            dm.pack(release_bytes=True)
        kind = os.environ.get("AB_KIND", "hudson")  # hudson (two groups) | wc4 (Weir & Cockerham, four groups) | sum4 (summaries, four groups)
        if kind != "hudson":
            poc = (np.arange(H) * 4 // H).astype(np.uint8)
            masks = np.stack([(poc == p) for p in range(4)]).astype(np.uint8)
        groups = device.Groups(dm, masks)
        if kind == "hudson":
            bufs = [device.DeviceBuffer(0, 8 * S) for _ in range(7)]
            sites = _abi.HudsonSites(None, *(b.ptr for b in bufs))
            totals = _abi.HudsonTotals()
            def sweep(rows):
                _abi.check(lib.fmh_hudson_sweep(dm._h, groups._h, 0, rows, _abi.FORMULA_DENSE, C.byref(sites), C.byref(totals), None))
        elif kind == "wc4":
            bufs = [device.DeviceBuffer(0, 8 * 7 * S), device.DeviceBuffer(0, 8 * 7 * S), device.DeviceBuffer(0, 7 * S), device.DeviceBuffer(0, 4 * 4 * S)]
            totals = _abi.WcTotals()
            def sweep(rows):
                # tool-level switches (not library ones): TOOL_WC_NO_STATE / TOOL_WC_NO_AB / TOOL_WC_NO_COUNTS drop one family of per-site tracks
                e = os.environ
                _abi.check(lib.fmh_wc_sweep(dm._h, groups._h, 0, rows, None if e.get("TOOL_WC_NO_AB") else bufs[0].ptr, None if e.get("TOOL_WC_NO_AB") else bufs[1].ptr,
                                            None if e.get("TOOL_WC_NO_STATE") else bufs[2].ptr, None if e.get("TOOL_WC_NO_COUNTS") else bufs[3].ptr, C.byref(totals), None))
        else:
            bufs = [device.DeviceBuffer(0, 4 * 4 * S), device.DeviceBuffer(0, 4 * 4 * S)]
            totals = (_abi.PopTotals * 4)()
            def sweep(rows):
                _abi.check(lib.fmh_population_summaries(dm._h, groups._h, 0, rows, 0, bufs[0].ptr, bufs[1].ptr, totals, None))
        for rows in (S, S // 2, S // 4, S // 8, S // 10, S // 16):
            res = {}
            for rep in range(3):
                for name, env in (("unset", None), ("set", value)):
                    _abi.set_option(var, env)  # None = the value the process started with
                    for _ in range(3):
                        sweep(rows)
                    lib.fmh_timing_enable(1)
                    lib.fmh_timing_reset()
                    for _ in range(20):
                        sweep(rows)
                    ms, n = C.c_double(), C.c_uint64()
                    lib.fmh_timing_read(C.byref(ms), C.byref(n))
                    lib.fmh_timing_enable(0)
                    res.setdefault(name, []).append(ms.value / max(n.value, 1))
            _abi.set_option(var, None)
            a, b = min(res["unset"]), min(res["set"])
            print(json.dumps({"switch": sys.argv[1], "kind": kind, "sites": rows, "haplotypes": H, "layout": layout + (" +1% missing" if missing else ""), "unset_ms": round(a, 4), "set_ms": round(b, 4),
                              "set_over_unset": round(b / a, 4), "all_unset": [round(x, 4) for x in res["unset"]], "all_set": [round(x, 4) for x in res["set"]]}), flush=True)
        del bufs, groups, dm, sweep


if __name__ == "__main__":
    main()
