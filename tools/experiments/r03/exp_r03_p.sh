#!/bin/bash
# regional W&C sums of many groups straight from the count tables: kernel trace of the 26-group call, then time against groups
set -o pipefail
cd "$(dirname "$0")/../../.." || exit 1
R=$(pwd)
O=$R/gpurun_out/r03p
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_device_parity.py -x -q -k "many_groups or wider" > $O/pytest.log 2>&1; rc=$?; echo "tests: exit $rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/measure_wc_groups.py 5 12 26 > $O/wc_groups.jsonl 2>$O/wc_groups.err; cat $O/wc_groups.jsonl
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/wc26 -o t -- python3 $R/tools/measure_wc_groups.py 26 > $O/wc26.out 2> $O/wc26.log; echo "exit $?"
f=$(find $O/wc26 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cut -c1-200 "$f"
