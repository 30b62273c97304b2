#!/bin/bash
# where run_vcf's user CPU goes: 1 region against 500 regions, one and sixteen region workers, the track writer alone
set -o pipefail
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r03ae
mkdir -p $O
run() { python tools/run_vcf_many_regions.py $1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({'regions': $1, 'workers': '$RUN_VCF_WORKERS', 'env': '$2', 'wall_s': round(d['wall_s'],3), 'user_s': d['child_user_s'], 'sys_s': d['child_sys_s'], 'ctx': d['child_vol_ctx_switches'], 'stages': {k: round(v,3) for k,v in d['stages_s'].items() if v > 0.02}}))" | tee -a $O/cpu.jsonl; }
export RUN_VCF_WORKERS=1
run 1 -
run 500 -
export RUN_VCF_WORKERS=16
run 1 -
run 500 -
HIP_FORCE_DEV_KERNARG=1 GPU_MAX_HW_QUEUES=4 run 500 queues4
for t in 1 16; do FERROMIC_THREADS=$t ./ferromic_amd/bin/run_vcf --bench_tracks 2>&1 | tail -1; done
