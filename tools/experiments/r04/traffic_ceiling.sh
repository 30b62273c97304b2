#!/bin/bash
# Round 4: the traffic-only ceiling of each configuration (tools/microbench/traffic_ceiling.hip), next to the real kernels on the same box.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out/${1:-r04_ceiling}
mkdir -p $O
cd $R
B=build/microbench/traffic_ceiling
[ -x $B ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $B tools/microbench/traffic_ceiling.hip
{
echo '{"config": "C2x10 Hudson traffic"}';  timeout -k 10 120 $B 10000000 128 5 4 0 3 4 8
echo '{"config": "C2 Hudson traffic"}';     timeout -k 10 120 $B 1000000 128 5 4 0 3 4 8
echo '{"config": "C3 W&C traffic"}';        timeout -k 10 120 $B 5000000 320 14 4 7 3 4 6
echo '{"config": "C3 summaries traffic"}';  timeout -k 10 120 $B 5000000 320 0 8 0 3 4 6
echo '{"config": "C4 Hudson traffic"}';     timeout -k 10 120 $B 10000000 640 5 4 0 3 4
} | tee $O/traffic_ceiling.jsonl
python3 tools/measure_configs.py C2 C2x10 C3 C3h C4 2>/dev/null | grep '^{' | tee $O/configs_same_box.jsonl
