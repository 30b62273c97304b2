#!/usr/bin/env python3
"""Development measurement: run_vcf on MANY SMALL regions (the usual shape of a config file: hundreds of inversion
regions of a few kb to a few hundred kb): per-region overhead of uploads, launches and syncs rather than bandwidth."""
import json
import os
import random
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import run_vcf_scale as RS  # noqa: E402


def main():
    sites, samples, regions = 60_000, 200, int(sys.argv[1]) if len(sys.argv) > 1 else 500
    tmp = tempfile.mkdtemp(prefix="run_vcf_regions_")
    geno, pos, length, vcf_bytes = RS.write_inputs(tmp, sites, samples, 77)
    names = [f"SYN{i:05d}" for i in range(samples)]
    rng = random.Random(5)
    cfg = "seqnames\tstart\tend\tPOS\torig_ID\tverdict\tcateg\t" + "\t".join(names) + "\n"
    for _ in range(regions):
        a = rng.randint(1, length - int(os.environ.get("RUN_VCF_REGION_MAX", "25000")) - 5_000)
        b = a + rng.randint(2_000, int(os.environ.get("RUN_VCF_REGION_MAX", "25000")))
        cells = [rng.choice(["0|0", "0|1", "1|0", "1|1"]) for _ in range(samples)]
        cfg += f"chr1\t{a}\t{b}\t{a}\tid\tpass\tinv\t" + "\t".join(cells) + "\n"
    open(os.path.join(tmp, "config.tsv"), "w").write(cfg)
    out_csv = os.path.join(tmp, "out", "results.csv")
    cmd = [RS.BIN, "--vcf_folder", os.path.join(tmp, "vcfs"), "--reference", os.path.join(tmp, "ref.fa"), "--gtf", os.path.join(tmp, "ann.gtf"),
           "--config_file", os.path.join(tmp, "config.tsv"), "--output_file", out_csv, "--fst"]
    if os.environ.get("RUN_VCF_WORKERS"):
        cmd += ["--workers_per_device", os.environ["RUN_VCF_WORKERS"]]
    if os.environ.get("RUN_VCF_DEVICES"):
        cmd += ["--devices", os.environ["RUN_VCF_DEVICES"]]
    if os.environ.get("RUN_VCF_PREFIX"):  # e.g. "rocprofv3 --hip-trace --stats --output-format csv -d DIR -o t --" (the binary itself follows the --)
        cmd = os.environ["RUN_VCF_PREFIX"].split() + cmd
    import resource
    ru0 = resource.getrusage(resource.RUSAGE_CHILDREN)
    t0 = time.perf_counter()
    res = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, FERROMIC_TIMING="1", FERROMIC_PROGRESS="0"))
    wall = time.perf_counter() - t0
    ru1 = resource.getrusage(resource.RUSAGE_CHILDREN)
    assert res.returncode == 0, res.stderr[-2000:]
    stages = {}
    for l in res.stderr.splitlines():
        if l.startswith("[TIMING]"):
            name, sec = l[len("[TIMING]"):].rsplit(" ", 1)
            stages[name.strip()] = stages.get(name.strip(), 0.0) + float(sec)
    rows = len(open(out_csv).read().splitlines()) - 1
    out_dir = os.path.dirname(out_csv)
    sizes = {f: os.path.getsize(os.path.join(out_dir, f)) for f in sorted(os.listdir(out_dir)) if f.endswith(".gz")}
    print(json.dumps({"regions": regions, "csv_rows": rows, "sites": sites, "samples": samples, "wall_s": wall,
                      "gz_bytes": sizes, "child_user_s": round(ru1.ru_utime - ru0.ru_utime, 3), "child_sys_s": round(ru1.ru_stime - ru0.ru_stime, 3),
                      "child_minor_faults": ru1.ru_minflt - ru0.ru_minflt, "child_vol_ctx_switches": ru1.ru_nvcsw - ru0.ru_nvcsw,
                      "ms_per_region": 1e3 * stages.get("regions_statistics_and_writers", 0.0) / max(regions, 1), "stages_s": stages}))


if __name__ == "__main__":
    main()
