#!/usr/bin/env python3
"""PCIe-inclusive rate of the boundary that hands over HOST buffers (fmh_matrix_create): upload of a
reference-layout matrix + one fused Hudson sweep.  Reported in DESIGN.md section 6.1; never `value`."""

import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ferromic_amd import device  # noqa: E402


def main():
    S, N = 1_000_000, 2500  # 5 GB of genotypes, C4 width
    rng = np.random.default_rng(1)
    data = rng.integers(0, 2, size=S * 2 * N, dtype=np.uint8)
    poc = np.repeat((np.arange(N) >= N // 2).astype(np.uint8), 2)
    masks = np.stack([poc == 0, poc == 1]).astype(np.uint8)
    best_up, best_total = 1e9, 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        dm = device.DeviceMatrix.from_host(data, None, S, N, 2, 1)
        t1 = time.perf_counter()
        g = device.Groups(dm, masks)
        device.hudson_sweep(dm, g, device.FORMULA_DENSE, want_sites=False)
        t2 = time.perf_counter()
        best_up, best_total = min(best_up, t1 - t0), min(best_total, t2 - t0)
        dm.close()
    print(json.dumps({"sites": S, "haplotypes": 2 * N, "bytes": data.nbytes, "upload_s": best_up,
                      "upload_GBs": data.nbytes / best_up / 1e9, "upload_plus_sweep_s": best_total,
                      "pcie_inclusive_sites_per_s": S / best_total}))


if __name__ == "__main__":
    main()
