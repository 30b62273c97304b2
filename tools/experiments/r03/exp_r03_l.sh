#!/bin/bash
# eight-group kernels (five to eight groups) with deep batches on one-batch rows: parity, then W&C time against the number of groups
set -o pipefail
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r03l
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_device_parity.py tests/test_gpu_api_fuzz.py tests/test_gpu_api_dropin.py -x -q > $O/parity.log 2>&1; rc=$?; echo "parity: exit $rc"; tail -2 $O/parity.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/measure_wc_groups.py 4 5 6 8 12 26 > $O/wc_groups.jsonl 2>$O/wc_groups.err; cat $O/wc_groups.jsonl
