#!/bin/bash
# run_vcf, 500 small regions: wall against the number of region workers (the uploads, group masks and result copies of a worker now run on
# its own stream instead of the legacy default stream), run_vcf tests first
set -o pipefail
cd "$(dirname "$0")/../../.." || exit 1
R=$(pwd)
O=$R/gpurun_out/r03aa
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_run_vcf.py tests/test_gpu_api_dropin.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "tests: exit $rc"; tail -2 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
for w in 1 2 4 8 16; do
  RUN_VCF_WORKERS=$w python tools/run_vcf_many_regions.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({'workers': $w, 'wall_s': round(d['wall_s'],3), 'ms_per_region': d['ms_per_region'], 'pack_upload_s': round(d['stages_s'].get('region:pack_and_upload_matrices',0),3), 'sweeps_s': round(d['stages_s'].get('region:gpu_sweeps_and_host_statistics',0),3)}))" | tee -a $O/workers_own_stream.jsonl
done
done
python tools/run_vcf_scale.py --sites 200000 --samples 2500 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('200k x 2500:', d['run_vcf_wall_s'])"
