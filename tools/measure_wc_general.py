#!/usr/bin/env python3
"""Development measurement: W&C and summary sweeps over MULTI-ALLELIC cohorts with 5 and 8 groups (the --fst_populations shape on
real SNP data: 1000 Genomes super-populations), 2 M sites x 2 500 haplotypes, max_allele 2 / 3 (/ 7), with and without 1 % missing
calls.  One JSON line per case; kept under profiles/ per round."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ferromic_amd import _abi, device  # noqa: E402


def cohort(S, H, max_allele, missing, seed):
    """A 250 k-site random block tiled to S sites: half of the sites biallelic, half carrying alleles up to max_allele."""
    rng = np.random.default_rng(seed)
    block = min(S, 250_000)
    data = rng.integers(0, 2, size=(block, H), dtype=np.uint8)
    extra = rng.integers(0, max_allele + 1, size=(block, H), dtype=np.uint8)
    pick = rng.random((block, 1)) < 0.5
    data = np.where(pick, extra, data).astype(np.uint8)
    words = None
    reps = -(-S // block)
    data = np.tile(data, (reps, 1))[:S]
    if missing > 0:
        miss = rng.random((block, H)) < missing
        miss = np.tile(miss, (reps, 1))[:S]
        bits = np.packbits(miss.reshape(-1), bitorder="little")
        words = np.frombuffer(np.pad(bits, (0, (-bits.size) % 8)).tobytes(), dtype="<u8").copy()
    return np.ascontiguousarray(data), words


def main():
    lib = _abi.load()
    S, H = int(os.environ.get("MEASURE_SITES", 2_000_000)), 2_500
    N = H // 2
    cases = [(g, a, m) for g in (5, 8) for a in (2, 3) for m in (0.0, 0.01)]
    if os.environ.get("MEASURE_ALLELE7"):
        cases += [(g, 7, 0.0) for g in (2, 5, 8)]
    for G, max_allele, missing in cases:
        data, words = cohort(S, H, max_allele, missing, 100 * G + max_allele)
        dm = device.DeviceMatrix.from_host(data, words, S, N, 2, int(data.max()))
        del data, words
        pop_of_sample = np.minimum(np.arange(N) * G // N, G - 1)
        poc = np.repeat(pop_of_sample, 2)
        masks = np.ascontiguousarray(np.stack([(poc == p) for p in range(G)]).astype(np.uint8))
        g = device.Groups(dm, masks)
        nw = 1 + G * (G - 1) // 2
        bufs = [device.DeviceBuffer(0, 8 * nw * S), device.DeviceBuffer(0, 8 * nw * S), device.DeviceBuffer(0, nw * S), device.DeviceBuffer(0, 4 * G * S)]
        tot = _abi.WcTotals()
        pt = (_abi.PopTotals * G)()
        out = {"groups": G, "max_allele": max_allele, "missing": missing, "sites": S, "haplotypes": H,
               "layout": os.environ.get("FMH_LAYOUT", "packed")}

        def wc():
            _abi.check(lib.fmh_wc_sweep(dm._h, g._h, 0, S, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, C.byref(tot), None))

        def summaries():
            _abi.check(lib.fmh_population_summaries(dm._h, g._h, 0, S, _abi.FORMULA_SPARSE, bufs[3].ptr, None, pt, None))

        for name, fn in (("wc_ms", wc), ("summaries_ms", summaries)):
            fn()
            t0 = time.perf_counter()
            for _ in range(5):
                fn()
            out[name] = (time.perf_counter() - t0) / 5 * 1e3
        out["wc_sum_a0"], out["informative0"] = tot.sum_a[0], int(tot.informative_sites[0])
        print(json.dumps(out), flush=True)
        del bufs, g
        dm.close()


if __name__ == "__main__":
    main()
