#!/bin/bash
# Round 4: clock-aligned read / write phases in the traffic-only microbenchmark
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out/${1:-r04_aligned}
mkdir -p $O
cd $R
B=build/microbench/traffic_ceiling
export CEILING_ALIGNED=1
{
echo '{"config": "C3 W&C traffic"}';        timeout -k 10 200 $B 5000000 320 14 4 7 4
echo '{"config": "C3 summaries traffic"}';  timeout -k 10 200 $B 5000000 320 0 8 0 4
echo '{"config": "C2x10 Hudson traffic"}';  timeout -k 10 200 $B 10000000 128 5 4 0 4
echo '{"config": "C4 Hudson traffic"}';     timeout -k 10 200 $B 10000000 640 5 4 0 3
} | tee $O/aligned.jsonl
