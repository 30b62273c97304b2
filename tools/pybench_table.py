#!/usr/bin/env python3
"""profiles/<round>/api_pybench.jsonl (tools/measure_api_pybench.py) -> the markdown table of BASELINE.md section 3."""
import json
import sys


def main():
    rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
    start = [r for r in rows if "hip_startup_and_first_tiny_statistic_s" in r]
    by = {}
    for r in rows:
        if "dataset" in r:
            by.setdefault(r["dataset"], {})[r["metric"]] = r

    def ms(x):
        return f"{x * 1e3:.3f}"

    print("| cohort (variants × diploid samples) | `from_numpy` (upload) | `segregating_sites()` first call / repeat | per-population `nucleotide_diversity()` first call ×2 | `hudson_fst(p1, p2)` | `hudson_dxy(p1, p2)` | `watterson_theta(seg, n, L)` | CPU restatement, 16 threads: whole-population pass / pair pass | values vs oracle at abs 1e-12 |")
    print("|---|---|---|---|---|---|---|---|---|")
    for ds, d in by.items():
        seg = d["segregating_sites"]
        ok = all(m["equal_at_abs_1e-12"] for m in d.values())
        print(f"| {ds} {seg['variants']} × {seg['samples']} | {ms(seg['from_numpy_s'])} | {ms(seg['ferromic_hip']['max_s'])} / {ms(seg['ferromic_hip']['min_s'])} | "
              f"{ms(d['nucleotide_diversity_population_1']['ferromic_hip']['max_s'])}, {ms(d['nucleotide_diversity_population_2']['ferromic_hip']['max_s'])} | "
              f"{ms(d['hudson_fst']['ferromic_hip']['mean_s'])} | {ms(d['hudson_dxy']['ferromic_hip']['mean_s'])} | {ms(d['watterson_theta']['ferromic_hip']['mean_s'])} | "
              f"{ms(seg['cpu_restatement']['mean_s'])} / {ms(d['hudson_fst']['cpu_restatement']['mean_s'])} | {'all 7 equal' if ok else 'MISMATCH'} |")
    if start:
        print(f"\n(all times in ms; HIP start-up + first tiny statistic once per process: {start[0]['hip_startup_and_first_tiny_statistic_s']:.2f} s)")


if __name__ == "__main__":
    main()
