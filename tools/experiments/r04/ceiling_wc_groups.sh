#!/bin/bash
# Round 4: the traffic-only ceiling of the W&C sweeps with five and eight groups (2 M x 2 500: 320-byte rows; tracks a, b f64 + state u8 per
# slot, called u32 per group) next to the real kernels on the same box.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out/${1:-r04_ceiling_wc}
mkdir -p $O $R/build/microbench
cd $R
B=build/microbench/traffic_ceiling
[ -x $B ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $B tools/microbench/traffic_ceiling.hip
{
echo '{"config": "W&C 4 groups traffic (7 slots)"}';  timeout -k 10 120 $B 2000000 320 14 4 7 3 4 6
echo '{"config": "W&C 5 groups traffic (11 slots)"}'; timeout -k 10 120 $B 2000000 320 22 5 11 3 4 6
echo '{"config": "W&C 8 groups traffic (29 slots)"}'; timeout -k 10 120 $B 2000000 320 58 8 29 3 4 6
} | tee $O/traffic_ceiling_wc_groups.jsonl
cd /tmp
python3 $R/tools/measure_wc_groups.py 4 5 8 2>/dev/null | grep '^{' | tee $O/wc_groups_same_box.jsonl
