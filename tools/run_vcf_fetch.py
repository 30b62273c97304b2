#!/usr/bin/env python3
"""How many bytes the sweeps of one run_vcf region read: `rocprofv3 --pmc FETCH_SIZE` (its own pass, counters only) around the run_vcf
binary on the synthetic 200 000-site x 5 000-haplotype VCF of tools/run_vcf_scale.py, for each binary given - round 3's (one fused
fmh_pair_region_sweep per variant set) and round 2's (fmh_population_summaries + 2 x fmh_diversity_sites + fmh_hudson_sweep), the latter
rebuilt from its commit against today's library:

    python tools/run_vcf_fetch.py [--sites N --samples M] NAME=path/to/run_vcf ...

Prints one JSON line per binary: FETCH_SIZE (KB, as the counter reports it; gfx950 halves wide coalesced reads, MI355X_MICROARCH.md) summed
over the sweep kernels, their dispatch count, and the same per kernel name."""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import run_vcf_scale  # noqa: E402


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--sites", type=int, default=200_000)
    ap.add_argument("--samples", type=int, default=2_500)
    ap.add_argument("bins", nargs="+")
    args = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix="run_vcf_fetch_", dir="/tmp")
    run_vcf_scale.write_inputs(tmp, args.sites, args.samples, 202_500, "none")
    env = dict(os.environ, TMPDIR="/tmp")
    for item in args.bins:
        name, path = item.split("=", 1)
        out_csv = os.path.join(tmp, "out_" + name, "results.csv")
        prof = os.path.join(tmp, "prof_" + name)
        cmd = ["rocprofv3", "--pmc", "FETCH_SIZE", "--output-format", "csv", "-d", prof, "-o", "f", "--", os.path.abspath(path),
               "--vcf_folder", os.path.join(tmp, "vcfs"), "--reference", os.path.join(tmp, "ref.fa"), "--gtf", os.path.join(tmp, "ann.gtf"),
               "--config_file", os.path.join(tmp, "config.tsv"), "--output_file", out_csv, "--fst"]
        res = subprocess.run(cmd, capture_output=True, text=True, env=dict(env, FERROMIC_FULL_TEARDOWN="1"), cwd="/tmp")
        if res.returncode != 0:
            print(res.stderr[-3000:], file=sys.stderr)
            return 1
        per = defaultdict(lambda: [0, 0.0])
        for f in glob.glob(os.path.join(prof, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if row["Counter_Name"] != "FETCH_SIZE":
                    continue
                k = row["Kernel_Name"].split("(")[0].strip()
                per[k][0] += 1
                per[k][1] += float(row["Counter_Value"])
        sweeps = {k: v for k, v in per.items() if "sweep_kernel" in k}
        row1 = open(out_csv).read().splitlines()[1]
        print(json.dumps({"binary": name, "sites": args.sites, "haplotypes": 2 * args.samples,
                          "packed_matrix_KB": args.sites * ((2 * args.samples + 7) // 8 + 15) // 16 * 16 / 1024,
                          "sweep_dispatches": sum(v[0] for v in sweeps.values()), "sweep_FETCH_SIZE_KB": sum(v[1] for v in sweeps.values()),
                          "per_kernel": {k: {"dispatches": v[0], "FETCH_SIZE_KB": v[1]} for k, v in sorted(sweeps.items())},
                          "csv_row_sha": __import__("hashlib").sha256(row1.encode()).hexdigest()[:12]}), flush=True)
    import shutil

    shutil.rmtree(tmp, ignore_errors=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
