#!/bin/bash
# run_vcf: every track of a sweep in one device block, one copy back - tests, then the 500-small-regions run against the binary before
set -o pipefail
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r03z
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_run_vcf.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "tests: exit $rc"; tail -2 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2 3; do
  python tools/run_vcf_many_regions.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({'binary': 'now', 'wall_s': round(d['wall_s'],3), 'ms_per_region': d['ms_per_region'], 'stages_s': d['stages_s']}))" | tee -a $O/many_regions.jsonl | cut -c1-120
done
python tools/run_vcf_scale.py --sites 200000 --samples 2500 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['run_vcf_wall_s'])"
