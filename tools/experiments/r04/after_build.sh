#!/bin/bash
# Round 4: multi-allelic Hudson parity after the first-occurrence repair + flat-route parity + the traffic ceilings with deferred stores.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out/${1:-r04_b}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_device_parity.py tests/test_gpu_flat_route.py tests/test_gpu_device_fuzz.py -x -q 2>&1 | tail -15 | tee $O/pytest.log
bash tools/experiments/r04/traffic_ceiling.sh ${1:-r04_b}
