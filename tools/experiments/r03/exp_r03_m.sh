#!/bin/bash
# kernel trace of the W&C call for five and eight groups: how the time splits between the sweep and the slot-sum pass
set -o pipefail
cd "$(dirname "$0")/../../.." || exit 1
R=$(pwd)
O=$R/gpurun_out/r03m
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/wc58 -o t -- python3 $R/tools/measure_wc_groups.py 5 8 > $O/wc58.out 2> $O/wc58.log; echo "exit $?"
cat $O/wc58.out
f=$(find $O/wc58 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cut -c1-260 "$f"
