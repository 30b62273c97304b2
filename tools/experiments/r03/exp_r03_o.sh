#!/bin/bash
# regional W&C sums of many groups straight from the count tables: tests, then time against groups
set -o pipefail
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r03o
mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_gpu_device_parity.py tests/test_gpu_device_fuzz.py tests/test_gpu_run_vcf.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "tests: exit $rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || { grep -E "Error|assert|FAILED" $O/pytest.log | head -20; exit $rc; }
timeout -k 10 300 python tools/measure_wc_groups.py 5 8 12 26 > $O/wc_groups.jsonl 2>$O/wc_groups.err; cat $O/wc_groups.jsonl
