#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r03i
mkdir -p $O
FMH_GRID_BLOCKS=2 FMH_PIPE=1 timeout -k 10 600 python -m pytest tests/test_gpu_device_parity.py tests/test_gpu_api_fuzz.py tests/test_gpu_api_dropin.py tests/test_gpu_comm.py -x -q > $O/two_blocks_pipe_everywhere.log 2>&1; echo "two blocks, pipelined loop wherever built: exit $?"; tail -1 $O/two_blocks_pipe_everywhere.log
timeout -k 10 600 python -m pytest tests/test_gpu_run_vcf.py -x -q > $O/pytest_run_vcf.log 2>&1; echo "run_vcf tests exit $?"; tail -1 $O/pytest_run_vcf.log
for K in 5 26; do
  for b in build/variants/r02/run_vcf ferromic_amd/bin/run_vcf; do
    python tools/run_vcf_scale.py --sites 100000 --samples 2500 --populations $K --bin $b 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({k: d[k] for k in ('binary','sites','haplotypes','csv_populations','population_pairs','run_vcf_wall_s','all_match')}))" >> $O/run_vcf_csv_populations.jsonl
  done
done
cat $O/run_vcf_csv_populations.jsonl
