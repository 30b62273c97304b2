#!/bin/bash
# run_vcf, 500 small regions, 16 and 8 region workers: user / system CPU, page faults, context switches; glibc malloc settings through the environment
set -o pipefail
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r03ad
mkdir -p $O
run() { python tools/run_vcf_many_regions.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({'setting': '$1', 'workers': '$RUN_VCF_WORKERS', 'wall_s': round(d['wall_s'],3), 'ms_per_region': d['ms_per_region'], 'user_s': d['child_user_s'], 'sys_s': d['child_sys_s'], 'minor_faults': d['child_minor_faults'], 'ctx': d['child_vol_ctx_switches']}))" | tee -a $O/malloc_settings.jsonl; }
for w in 16 8 4 1; do
  export RUN_VCF_WORKERS=$w
  run default
  MALLOC_MMAP_THRESHOLD_=33554432 run mmap_threshold_32M
  MALLOC_ARENA_MAX=64 MALLOC_MMAP_THRESHOLD_=33554432 run mmap_threshold_32M_arena_max_64
done
