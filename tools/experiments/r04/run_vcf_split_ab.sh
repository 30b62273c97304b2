#!/bin/bash
# Round 4: did splitting run_vcf.cpp into translation units cost anything?  The pre-split single-unit binary (build/variants/r04mono/run_vcf,
# built from the commit before the split) against the shipped one, alternating on one box: 500 small regions and the 3.5 GB VCF.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out/${1:-r04_split}
mkdir -p $O
cd /tmp
for rep in 1 2 3; do
  for b in ${AB_ORDER:-split mono}; do
    if [ $b = mono ]; then export RUN_VCF_BIN=$R/build/variants/r04mono/run_vcf; else unset RUN_VCF_BIN; fi
    python3 $R/tools/run_vcf_many_regions.py 2>/dev/null | tail -1 | sed "s/^{/{\"build\": \"$b\", /" >> $O/run_vcf_500_regions_split_vs_mono.jsonl
  done
done
for rep in 1 2; do
  for b in ${AB_ORDER:-split mono}; do
    if [ $b = mono ]; then export RUN_VCF_BIN=$R/build/variants/r04mono/run_vcf; else unset RUN_VCF_BIN; fi
    python3 $R/tools/run_vcf_scale.py --sites 200000 --samples 2500 --bin ${RUN_VCF_BIN:-$R/ferromic_amd/bin/run_vcf} 2>/dev/null | tail -1 | sed "s/^{/{\"build\": \"$b\", /" >> $O/run_vcf_scale_split_vs_mono.jsonl
  done
done
python3 - $O <<'PY'
import json, sys
for f in ("run_vcf_500_regions_split_vs_mono.jsonl", "run_vcf_scale_split_vs_mono.jsonl"):
    for l in open(sys.argv[1] + "/" + f):
        d = json.loads(l); print(f[:24], d["build"], d.get("wall_s") or d.get("run_vcf_wall_s"), d.get("ms_per_region"))
PY
