#!/bin/bash
# per-call costs of small statistics after: totals finalised straight into pinned host memory, group masks in one block / one copy
set -o pipefail
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r03v
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_device_parity.py tests/test_gpu_api_dropin.py tests/test_gpu_comm.py tests/test_gpu_run_vcf.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "tests: exit $rc"; tail -2 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
python tools/measure_call_overheads.py 2>/dev/null | tail -1 | tee $O/call_overheads.json
python tools/measure_api_pybench.py 2>/dev/null | grep '^{' > $O/api_pybench.jsonl; python tools/pybench_table.py $O/api_pybench.jsonl 2>/dev/null | tail -8
