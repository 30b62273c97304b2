#!/bin/bash
# the run-aware gzip writer against zlib level 1 on whole run_vcf runs: a sparse cohort (a variant every 100 bp) with 200 regions of 50 kb - 1 Mb,
# and the dense 500-region cohort; wall, user CPU and the sizes of the FALSTA files
set -o pipefail
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r03ah
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_run_vcf.py -x -q > $O/pytest_auto.log 2>&1; echo "run_vcf tests (writer by density): exit $?"; tail -1 $O/pytest_auto.log
FERROMIC_TRACK_WRITER=runs timeout -k 10 600 python -m pytest tests/test_gpu_run_vcf.py -x -q > $O/pytest_runs.log 2>&1; echo "run_vcf tests (run-aware writer everywhere): exit $?"; tail -1 $O/pytest_runs.log
show() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({'case': '$1', 'writer': '$2', 'wall_s': round(d['wall_s'],3), 'ms_per_region': d['ms_per_region'], 'user_s': d['child_user_s'], 'gz_bytes': d['gz_bytes']}))" | tee -a $O/writers.jsonl; }
for w in zlib runs auto; do
  export FERROMIC_TRACK_WRITER=$w; [ $w = auto ] && unset FERROMIC_TRACK_WRITER
  RUN_VCF_GAP_MAX=200 RUN_VCF_REGION_MAX=1000000 python tools/run_vcf_many_regions.py 200 2>/dev/null | tail -1 | show "sparse, 200 regions up to 1 Mb" $w
  python tools/run_vcf_many_regions.py 500 2>/dev/null | tail -1 | show "dense, 500 regions up to 25 kb" $w
done
