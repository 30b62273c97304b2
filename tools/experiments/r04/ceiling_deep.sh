#!/bin/bash
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out/${1:-r04_deep}
mkdir -p $O
cd $R
B=build/microbench/traffic_ceiling
{
echo '{"config": "C4 Hudson traffic"}';     timeout -k 10 200 $B 10000000 640 5 4 0 3 4
echo '{"config": "C3 summaries traffic"}';  timeout -k 10 200 $B 5000000 320 0 8 0 3 4
echo '{"config": "C3 W&C traffic"}';        timeout -k 10 200 $B 5000000 320 14 4 7 3 4
} | tee $O/traffic_ceiling_deep.jsonl
python3 tools/measure_configs.py C4 C3 C3h 2>/dev/null | grep '^{' | tee $O/configs_same_box.jsonl
