#!/usr/bin/env python3
"""The reference's own API benchmark recipe (src/pybenches/test_population_statistics_benchmarks.py:67-73, 113-261, 620-900) through the
drop-in module: the four synthetic cohorts (512 x 48, 4 096 x 96, 16 384 x 128, 65 536 x 256 variants x diploid samples; seed = variants +
samples, Beta(0.8, 0.8) base frequencies, N(0, scale) divergence, two equal populations, rows 0 and 1 forced informative, positions =
cumsum of 1..49 steps), built exactly as the reference builds them, then the metrics it times - population.segregating_sites(),
population.nucleotide_diversity(), per-population nucleotide_diversity() x2, watterson_theta(segregating_sites(), haplotypes, L),
hudson_fst(pop1, pop2), hudson_dxy(pop1, pop2) - with the reference's rounds (3 / 3 / 4 / 3), one iteration per round, mean / min / max.

`import ferromic` here is ferromic/_core (C++ host over libferromic_hip.so): every statistic that touches genotypes runs on the GPU.
HIP start-up is paid on a throw-away population first and reported separately; Population.from_numpy (upload) is timed on its own.
Beside each metric: the CPU restatement (oracle/dense_oracle.c on the cores this process may use) computing the same quantity from the same
array, and the check the reference applies to its own numbers (abs 1e-12, :29 and :402-411) between the two.  The scikit-allel leg of the
reference's comparison needs `allel`, which is not installed here; its CI artefact is what a maintainer would put next to this table.

    python tools/measure_api_pybench.py            # one JSON line per (cohort, metric)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ferromic as fm  # noqa: E402
from bench import usable_cores  # noqa: E402
from oracle import dense as D  # noqa: E402  (the checker; never the product)
from oracle import ferromic_ref as R  # noqa: E402

CONFIGS = (("pilot_panel", 512, 48, 0.02, 3), ("regional_panel", 4096, 96, 0.05, 3), ("chromosome_arm", 16384, 128, 0.08, 4), ("deep_cohort", 65536, 256, 0.1, 3))
ABS_TOL = 1e-12


def build(variant_count, sample_count, scale):
    """_build_synthetic_dataset (:113-158), draw for draw"""
    rng = np.random.default_rng(seed=variant_count + sample_count)
    half = sample_count // 2
    base = rng.beta(0.8, 0.8, size=variant_count)
    div = rng.normal(0.0, scale, size=variant_count)
    f1, f2 = np.clip(base + div, 0.001, 0.999), np.clip(base - div, 0.001, 0.999)
    h1 = rng.binomial(1, f1[:, None], size=(variant_count, half * 2)).astype(np.int8)
    h2 = rng.binomial(1, f2[:, None], size=(variant_count, half * 2)).astype(np.int8)
    g = np.concatenate([h1.reshape(variant_count, half, 2), h2.reshape(variant_count, half, 2)], axis=1)
    g[0, :half, :] = 0
    g[0, half:, :] = 1
    g[1, :half, 0] = 0
    g[1, :half, 1] = 1
    g[1, half:, :] = 1
    positions = np.cumsum(rng.integers(1, 50, size=variant_count, dtype=np.int64), dtype=np.int64)
    start, stop_excl = int(positions[0]), int(positions[-1]) + 1
    return g, positions, stop_excl - start


def timed(fn, rounds):
    ts, r = [], None
    for _ in range(rounds):
        t0 = time.perf_counter()
        r = fn()
        ts.append(time.perf_counter() - t0)
    return r, {"mean_s": float(np.mean(ts)), "min_s": min(ts), "max_s": max(ts), "rounds": rounds}


def main():
    cores = usable_cores()["cores_usable"]
    t0 = time.perf_counter()
    g0, p0, l0 = build(8, 4, 0.02)
    haps0 = [(s, k) for s in range(4) for k in (0, 1)]
    fm.Population.from_numpy("warm", g0, p0, haps0, l0, sample_names=[f"s{i}" for i in range(4)]).segregating_sites()
    print(json.dumps({"hip_startup_and_first_tiny_statistic_s": round(time.perf_counter() - t0, 4)}), flush=True)
    for label, V, N, scale, rounds in CONFIGS:
        g, positions, L = build(V, N, scale)
        half, H = N // 2, 2 * N
        haps = [(s, k) for s in range(N) for k in (0, 1)]
        names = [f"sample_{i}" for i in range(N)]
        t0 = time.perf_counter()
        pop = fm.Population.from_numpy("all_samples", g, positions, haps, L, sample_names=names)
        from_numpy_s = time.perf_counter() - t0
        pop1 = pop.with_haplotypes("population_1", [h for h in haps if h[0] < half])
        pop2 = pop.with_haplotypes("population_2", [h for h in haps if h[0] >= half])
        # the CPU restatement on the same array (u8 rows, reference host layout)
        data = np.ascontiguousarray(g.reshape(V, H).astype(np.uint8)).reshape(-1)
        all_off, o1, o2 = np.arange(H, dtype=np.uint64), np.arange(0, N, dtype=np.uint64), np.arange(N, H, dtype=np.uint64)
        (whole, cpu_whole) = timed(lambda: D.hudson_sweep(data, None, V, H, all_off, o1, cores, want_sites=False), rounds)
        (pair, cpu_pair) = timed(lambda: D.hudson_sweep(data, None, V, H, o1, o2, cores, want_sites=False), rounds)

        def pi_of(p):
            return p["pi_sum"] / float(L - p["uncallable_sites"])

        seg = int(whole.pop[0]["segregating_sites"])
        t = pair.totals
        oracle = {"segregating_sites": seg, "nucleotide_diversity": pi_of(whole.pop[0]), "nucleotide_diversity_population_1": pi_of(pair.pop[0]),
                  "nucleotide_diversity_population_2": pi_of(pair.pop[1]), "watterson_theta": R.calculate_watterson_theta(seg, H, L),
                  "hudson_fst": t["numerator_sum"] / t["denominator_sum"], "hudson_dxy": t["dxy_sum_all"] / float(L - t["dxy_uncallable_sites"])}
        cpu_time = {"segregating_sites": cpu_whole, "nucleotide_diversity": cpu_whole, "nucleotide_diversity_population_1": cpu_pair,
                    "nucleotide_diversity_population_2": cpu_pair, "watterson_theta": cpu_whole, "hudson_fst": cpu_pair, "hudson_dxy": cpu_pair}
        metrics = {
            "segregating_sites": pop.segregating_sites,
            "nucleotide_diversity": pop.nucleotide_diversity,
            "nucleotide_diversity_population_1": pop1.nucleotide_diversity,
            "nucleotide_diversity_population_2": pop2.nucleotide_diversity,
            "watterson_theta": lambda: fm.watterson_theta(pop.segregating_sites(), H, L),
            "hudson_fst": lambda: fm.hudson_fst(pop1, pop2).fst,
            "hudson_dxy": lambda: fm.hudson_dxy(pop1, pop2).d_xy,
        }
        for name, fn in metrics.items():
            value, stats = timed(fn, rounds)
            exp = oracle[name]
            ok = (value == exp) if isinstance(exp, int) else abs(float(value) - exp) <= ABS_TOL
            print(json.dumps({"dataset": label, "variants": V, "samples": N, "haplotypes": H, "metric": name, "value": value, "oracle_value": exp,
                              "equal_at_abs_1e-12": bool(ok), "ferromic_hip": stats, "from_numpy_s": round(from_numpy_s, 5),
                              "cpu_restatement": dict(cpu_time[name], cores=cores, what="oracle/dense_oracle.c: both summaries + Hudson totals of the same array in one threaded pass")}),
                  flush=True)
            if not ok:
                raise SystemExit(f"{label} {name}: {value!r} != oracle {exp!r}")
        del pop, pop1, pop2


if __name__ == "__main__":
    main()
