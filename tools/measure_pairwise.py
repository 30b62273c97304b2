#!/usr/bin/env python3
"""Development measurement: pairwise-differences Gram throughput (fmh_pairwise_differences) on random biallelic data."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ferromic_amd import device  # noqa: E402


def main():
    cases = [(int(a), int(b)) for a, b in (x.split("x") for x in sys.argv[1:])] or [(200_000, 500), (1_000_000, 2500)]
    rng = np.random.default_rng(5)
    for S, N in cases:
        data = rng.integers(0, 2, size=(S, 2 * N), dtype=np.uint8)
        dm = device.DeviceMatrix.from_host(data, None, S, N, 2, 1)
        device.pairwise_differences(dm, N)
        t0 = time.perf_counter()
        device.pairwise_differences(dm, N)
        dt = time.perf_counter() - t0
        n_pad = -(-N // 256) * 256
        tiles = (n_pad // 256) * (n_pad // 256 + 1) // 2
        planes = 2 if os.environ.get("FMH_PD_TWO_PLANES") else 1  # biallelic, nothing missing: one plane (+ an all-ones row)
        n_pad = -(-(N + (planes == 1)) // 256) * 256
        tiles = (n_pad // 256) * (n_pad // 256 + 1) // 2
        macs = tiles * 256 * 256 * S * planes
        print(json.dumps({"case": f"pairwise {S}x{N}", "seconds": dt, "sample_pair_sites_per_s": N * (N - 1) / 2 * S / dt,
                          "mfma_TMAC_per_s": macs / dt / 1e12, "planes": planes,
                          "operands": "int8" if os.environ.get("FMH_PD_INT8") else "fp4"}), flush=True)
        dm.close()


if __name__ == "__main__":
    main()
