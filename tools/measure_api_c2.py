#!/usr/bin/env python3
"""Development measurement: BASELINE config C2 through the Python drop-in API (1 M sites x 1 000 haplotypes,
2 populations): Population.from_numpy, per_site_diversity, hudson_fst, hudson_fst_with_sites - wall seconds each."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ferromic as fm  # noqa: E402


def main():
    S, N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 500
    rng = np.random.default_rng(S + N)
    f = rng.beta(0.8, 0.8, size=(S, 1, 1))
    g = (rng.random((S, N, 2)) < f).astype(np.uint8)
    pos = np.cumsum(rng.integers(1, 50, size=S)).astype(np.int64)
    L = int(pos[-1] - pos[0] + 1)
    haps = [(s, side) for s in range(N) for side in (0, 1)]
    out = {"sites": S, "haplotypes": 2 * N}

    def timed(name, fn):
        t0 = time.perf_counter()
        r = fn()
        out[name + "_s"] = round(time.perf_counter() - t0, 4)
        return r

    # HIP start-up (runtime initialisation, code-object load, first allocations) on a throw-away 8-site population, so that the C2 numbers
    # below are the path itself
    def warm():
        tiny = fm.Population.from_numpy("warm", g[:8].copy(), pos[:8].copy(), haps, int(pos[7] - pos[0] + 1))
        return tiny.segregating_sites()

    timed("hip_startup_and_first_tiny_statistic", warm)
    pop = timed("from_numpy", lambda: fm.Population.from_numpy("all", g, pos, haps, L))
    p1 = pop.with_haplotypes("p1", haps[:N])
    p2 = pop.with_haplotypes("p2", haps[N:])
    timed("segregating_sites_first_call", pop.segregating_sites)
    timed("nucleotide_diversity", pop.nucleotide_diversity)
    timed("hudson_fst", lambda: fm.hudson_fst(p1, p2))
    res = timed("hudson_fst_with_sites", lambda: fm.hudson_fst_with_sites(p1, p2, (int(pos[0]), int(pos[-1]))))
    out["n_site_records"] = len(res[1])
    # per_site_diversity takes Python variant records (the reference signature): coercion dominates
    Sd = min(S, 100_000)
    variants = [(int(pos[i]), g[i].tolist()) for i in range(Sd)]
    res = timed(f"per_site_diversity_{Sd}_sites_python_lists", lambda: fm.per_site_diversity(variants, haps, (int(pos[0]), int(pos[Sd - 1]))))
    out["n_diversity_records"] = len(res)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
