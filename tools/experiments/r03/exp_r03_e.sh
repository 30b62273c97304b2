#!/bin/bash
# round 3, fifth GPU session: the tests touched since the last full pass, then the collection kept under profiles/r03/
set -o pipefail
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r03e
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_comm.py tests/test_gpu_bench_multirank.py -x -q > $O/pytest_comm_bench.log 2>&1; echo "comm+bench exit $?"; tail -3 $O/pytest_comm_bench.log
timeout -k 10 1000 bash tools/collect_profiles.sh r03 > $O/collect.log 2>&1; echo "collect exit $?"; tail -5 $O/collect.log
