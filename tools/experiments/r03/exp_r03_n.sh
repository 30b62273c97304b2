#!/bin/bash
# eight-group W&C kernels with their regional sums kept in the kernel (per-wave LDS transposition): whole GPU suite, then time against groups
set -o pipefail
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r03n
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "gpu suite: exit $rc"; tail -3 $O/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/measure_wc_groups.py 4 5 6 8 > $O/wc_groups.jsonl 2>$O/wc_groups.err; cat $O/wc_groups.jsonl
