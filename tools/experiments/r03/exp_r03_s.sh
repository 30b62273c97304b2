#!/bin/bash
# Hudson pair from two populations' count tables (populations of different matrices): tests, then API timing
set -o pipefail
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r03s
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_device_parity.py tests/test_gpu_api_fuzz.py tests/test_gpu_api_dropin.py tests/test_abi_library.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "tests: exit $rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || { grep -E "Error|assert|FAILED" $O/pytest.log | head -20; exit $rc; }
timeout -k 10 300 python tools/measure_api_two_matrices.py 1000000 250 > $O/api_two_matrices.jsonl 2>$O/api_two.err; cat $O/api_two_matrices.jsonl; tail -3 $O/api_two.err
timeout -k 10 300 python tools/measure_api_two_matrices.py 65536 128 >> $O/api_two_matrices.jsonl 2>>$O/api_two.err; tail -1 $O/api_two_matrices.jsonl
