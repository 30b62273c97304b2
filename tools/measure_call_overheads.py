#!/usr/bin/env python3
"""Where the ~50 us of one small statistic go: the C-ABI calls of a Hudson pair on a tiny resident cohort, timed one by one."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ferromic_amd import _abi, device  # noqa: E402


def best(fn, n=200):
    fn()
    ts = []
    for _ in range(n):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    ts.sort()
    return round(ts[len(ts) // 10] * 1e6, 1)


def main():
    lib = _abi.load()
    out = {}
    for (S, N) in ((512, 48), (65536, 256)):
        rng = np.random.default_rng(1)
        data = (rng.random((S, 2 * N)) < 0.3).astype(np.uint8)
        dm = device.DeviceMatrix.from_host(data.reshape(-1), None, S, N, 2, 1)
        masks = np.zeros((2, 2 * N), dtype=np.uint8); masks[0, :N] = 1; masks[1, N:] = 1
        g = device.Groups(dm, masks)
        tot = _abi.HudsonTotals()
        r = {}
        def create_destroy():
            h = C.c_void_p()
            _abi.check(lib.fmh_groups_create(dm._h, device._ptr(masks), 2, C.byref(h)))
            lib.fmh_groups_destroy(h.value)
        r["groups_create_destroy_us"] = best(create_destroy)
        r["hudson_sweep_totals_us"] = best(lambda: _abi.check(lib.fmh_hudson_sweep(dm._h, g._h, 0, S, 2, None, C.byref(tot), None)))
        pt = (_abi.PopTotals * 2)()
        r["population_summaries_totals_us"] = best(lambda: _abi.check(lib.fmh_population_summaries(dm._h, g._h, 0, S, 2, None, None, pt, None)))
        buf = device.DeviceBuffer(0, 4096)
        host = np.zeros(512, dtype=np.float64)
        r["copy_to_host_512B_us"] = best(lambda: _abi.check(lib.fmh_copy_to_host(0, device._ptr(host), buf.ptr, 512, None)))
        out[f"{S}x{N}"] = r
    print(json.dumps(out))


if __name__ == "__main__":
    main()
