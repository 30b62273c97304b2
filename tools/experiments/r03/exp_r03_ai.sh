#!/bin/bash
# Population.from_numpy packing one-byte biallelic input straight to bit planes: API tests (both routes), then timings
set -o pipefail
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r03ai
mkdir -p $O
FERROMIC_FUZZ_NUMPY_CASES=600 FERROMIC_FUZZ_PAIR_CASES=200 timeout -k 10 900 python -m pytest tests/test_gpu_api_dropin.py tests/test_gpu_api_fuzz.py tests/test_gpu_scale.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "API tests (planes route): exit $rc"; tail -2 $O/pytest.log
[ $rc -eq 0 ] || { grep -E "Error|assert|FAILED" $O/pytest.log | head; exit $rc; }
FERROMIC_NUMPY_BYTES=1 timeout -k 10 900 python -m pytest tests/test_gpu_api_dropin.py tests/test_gpu_api_fuzz.py -x -q > $O/pytest_bytes.log 2>&1; echo "API tests (byte route): exit $?"; tail -1 $O/pytest_bytes.log
for v in planes bytes; do
  [ $v = bytes ] && export FERROMIC_NUMPY_BYTES=1
  python tools/measure_api_c2.py 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps(dict(route='$v', **{k: d[k] for k in ('from_numpy_s','segregating_sites_first_call_s','hudson_fst_s')})))" | tee -a $O/api_c2_routes.jsonl
  python tools/measure_api_pybench.py 2>/dev/null | grep '^{' > $O/api_pybench_$v.jsonl; python tools/pybench_table.py $O/api_pybench_$v.jsonl 2>/dev/null | tail -7 | head -4 | cut -c1-120
done
