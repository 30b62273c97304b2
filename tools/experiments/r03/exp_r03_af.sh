#!/bin/bash
# run_vcf, 500 small regions: the tracks of a region formatted + deflated by its own worker (inline) or on the shared pool, by the number of region workers
set -o pipefail
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r03af
mkdir -p $O
for rep in 1 2; do
for w in 2 4 6 8 12 16; do
for pool in 0 1; do
  RUN_VCF_WORKERS=$w FERROMIC_TRACKS_POOL=$pool python tools/run_vcf_many_regions.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({'workers': $w, 'tracks_on_pool': $pool, 'wall_s': round(d['wall_s'],3), 'ms_per_region': d['ms_per_region'], 'user_s': d['child_user_s'], 'sys_s': d['child_sys_s']}))" | tee -a $O/tracks_pool_or_inline.jsonl
done
done
done
