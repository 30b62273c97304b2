#!/usr/bin/env python3
"""hudson_fst / hudson_dxy between two Population.from_numpy objects that do NOT share a matrix (two arrays over the same positions), against
the same two populations cut from ONE from_numpy matrix.  The reference computes the first from the two populations' summaries
(stats.rs:1554-1623); until round 3 this build laid both matrices side by side on the host and uploaded the result on EVERY call."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ferromic as fm  # noqa: E402


def main():
    S, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1_000_000, 250)
    rng = np.random.default_rng(5)
    freq = rng.beta(0.8, 0.8, size=S)
    g = (rng.random((S, 2 * N, 2)) < freq[:, None, None]).astype(np.int8)
    pos = np.arange(1, S + 1, dtype=np.int64)
    haps = [(s, k) for s in range(N) for k in (0, 1)]
    names = [f"s{i}" for i in range(2 * N)]
    fm.Population.from_numpy("warm", g[:64], pos[:64], haps, 64).segregating_sites()  # HIP start-up
    out = {"sites": S, "samples_per_population": N}
    t = time.perf_counter()
    p1 = fm.Population.from_numpy(1, np.ascontiguousarray(g[:, :N]), pos, haps, S, sample_names=names[:N])
    p2 = fm.Population.from_numpy(2, np.ascontiguousarray(g[:, N:]), pos, haps, S, sample_names=names[N:])
    out["two_from_numpy_s"] = round(time.perf_counter() - t, 4)
    times = []
    for _ in range(5):
        t = time.perf_counter(); r = fm.hudson_fst(p1, p2); times.append(time.perf_counter() - t)
    out["two_matrices_hudson_fst_first_call_ms"] = round(times[0] * 1e3, 3)
    out["two_matrices_hudson_fst_repeat_ms"] = round(min(times[1:]) * 1e3, 3)
    t = time.perf_counter(); d = fm.hudson_dxy(p1, p2); out["two_matrices_hudson_dxy_ms"] = round((time.perf_counter() - t) * 1e3, 3)
    whole = fm.Population.from_numpy("all", g, pos, [(s, k) for s in range(2 * N) for k in (0, 1)], S, sample_names=names)
    q1 = whole.with_haplotypes(1, [(s, k) for s in range(N) for k in (0, 1)])
    q2 = whole.with_haplotypes(2, [(s, k) for s in range(N, 2 * N) for k in (0, 1)])
    times = []
    for _ in range(5):
        t = time.perf_counter(); r1 = fm.hudson_fst(q1, q2); times.append(time.perf_counter() - t)
    out["one_matrix_hudson_fst_first_call_ms"] = round(times[0] * 1e3, 3)
    out["one_matrix_hudson_fst_repeat_ms"] = round(min(times[1:]) * 1e3, 3)
    out["fst_two_matrices"], out["fst_one_matrix"] = r.fst, r1.fst
    out["agree_1e-12"] = bool(abs(r.fst - r1.fst) <= 1e-12 and abs(d.d_xy - fm.hudson_dxy(q1, q2).d_xy) <= 1e-12)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
