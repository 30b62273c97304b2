#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r03ag
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_run_vcf.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "tests: exit $rc"; tail -2 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2 3; do
  python tools/run_vcf_many_regions.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({'wall_s': round(d['wall_s'],3), 'ms_per_region': d['ms_per_region'], 'user_s': d['child_user_s'], 'sys_s': d['child_sys_s']}))" | tee -a $O/many_regions.jsonl
done
python tools/run_vcf_many_regions.py 2>/dev/null | tail -1 > $O/run_vcf_500_regions.json
python tools/run_vcf_scale.py --sites 200000 --samples 2500 2>/dev/null | tail -1 | tee $O/run_vcf_scale_200k_x_2500.json | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('200k x 2500:', d['run_vcf_wall_s'], d.get('all_match'))"
for K in 5 26; do python tools/run_vcf_scale.py --sites 100000 --samples 2500 --populations $K 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({k: d[k] for k in ('sites','haplotypes','csv_populations','population_pairs','run_vcf_wall_s','all_match')}))" | tee -a $O/run_vcf_csv_populations.jsonl; done
