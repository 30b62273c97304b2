#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r03w
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_api_dropin.py tests/test_gpu_api_fuzz.py tests/test_api_host.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "tests: exit $rc"; tail -2 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
python tools/measure_api_pybench.py 2>/dev/null | grep '^{' > $O/api_pybench.jsonl; python tools/pybench_table.py $O/api_pybench.jsonl 2>/dev/null | tail -8
