#!/usr/bin/env python3
"""Development measurement: multi-allelic (GENERAL) sweep and pairwise-differences Gram throughput."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ferromic_amd import _abi, device  # noqa: E402


def timed_hudson(dm, g, S, label, bytes_per_site):
    lib = _abi.load()
    bufs = [device.DeviceBuffer(0, 8 * S) for _ in range(7)]
    sites = _abi.HudsonSites(None, *[b.ptr for b in bufs])
    tot = _abi.HudsonTotals()
    for _ in range(2):
        _abi.check(lib.fmh_hudson_sweep(dm._h, g._h, 0, S, _abi.FORMULA_DENSE, C.byref(sites), C.byref(tot), None))
    lib.fmh_timing_enable(1)
    lib.fmh_timing_reset()
    for _ in range(5):
        _abi.check(lib.fmh_hudson_sweep(dm._h, g._h, 0, S, _abi.FORMULA_DENSE, C.byref(sites), C.byref(tot), None))
    ms, n = C.c_double(), C.c_uint64()
    lib.fmh_timing_read(C.byref(ms), C.byref(n))
    lib.fmh_timing_enable(0)
    k = ms.value / n.value / 1e3
    print(json.dumps({"case": label, "kernel_ms": k * 1e3, "sites_per_s": S / k, "GBs": bytes_per_site * S / k / 1e9}), flush=True)


def main():
    rng = np.random.default_rng(0)
    S, N = 400_000, 2500
    H = 2 * N
    poc = np.repeat((np.arange(N) >= N // 2).astype(np.uint8), 2)
    masks = np.stack([poc == 0, poc == 1]).astype(np.uint8)
    for max_allele in (1, 2, 3, 7):
        data = rng.integers(0, 2, size=(S, H), dtype=np.uint8)
        if max_allele > 1:
            extra = rng.integers(0, max_allele + 1, size=(S, H), dtype=np.uint8)
            pick = rng.random((S, 1)) < float(os.environ.get("MEASURE_MULTI_FRACTION", "0.5"))  # share of multi-allelic sites (default: half)
            data = np.where(pick, extra, data).astype(np.uint8)
        dm = device.DeviceMatrix.from_host(data, None, S, N, 2, int(data.max()))
        timed_hudson(dm, device.Groups(dm, masks), S, f"hudson max_allele={max_allele}", H + 56)
        dm.close()
    # biallelic with missing calls concentrated in a share of the rows (MEASURE_GAP_FRACTION, default: every row): the called plane is read
    # for those rows only (MatrixView::row_gap)
    gap = float(os.environ.get("MEASURE_GAP_FRACTION", "1.0"))
    data = rng.integers(0, 2, size=(S, H), dtype=np.uint8)
    miss = np.zeros((S, H), dtype=bool)
    rows = np.nonzero(rng.random(S) < gap)[0]
    miss[rows] = rng.random((len(rows), H)) < 0.01
    data[miss] = 0
    bits = np.packbits(miss.reshape(-1), bitorder="little")
    words = np.frombuffer(np.concatenate([bits, np.zeros((-len(bits)) % 8, np.uint8)]).tobytes(), dtype="<u8").copy()
    dm = device.DeviceMatrix.from_host(data, words, S, N, 2, 1)
    timed_hudson(dm, device.Groups(dm, masks), S, f"hudson biallelic, 1 % missing in {gap:g} of the rows", H + H // 8 + 56)
    dm.close()
    if os.environ.get("MEASURE_SKIP_PAIRWISE"):
        return
    # pairwise differences, C2 shape scaled: 1000 haplotypes (500 samples)
    for S2, N2 in ((200_000, 500), (50_000, 2500), (1_000_000, 2500)):
        data = rng.integers(0, 2, size=(S2, 2 * N2), dtype=np.uint8)
        dm = device.DeviceMatrix.from_host(data, None, S2, N2, 2, 1)
        device.pairwise_differences(dm, N2)
        t0 = time.perf_counter()
        device.pairwise_differences(dm, N2)
        dt = time.perf_counter() - t0
        n_pad = -(-N2 // 256) * 256
        tiles = (n_pad // 256) * (n_pad // 256 + 1) // 2
        macs = tiles * 256 * 256 * S2 * 2  # two allele-count planes (no missing data: length/valid terms are constants)
        print(json.dumps({"case": f"pairwise {S2}x{N2}", "seconds": dt, "sample_pair_sites_per_s": N2 * (N2 - 1) / 2 * S2 / dt,
                          "mfma_TMAC_per_s": macs / dt / 1e12}), flush=True)
        dm.close()


if __name__ == "__main__":
    main()
