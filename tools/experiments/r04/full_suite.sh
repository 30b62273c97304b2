#!/bin/bash
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out/${1:-r04_full}
mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest_gpu.log
bash tools/experiments/r04/c2_grid.sh ${1:-r04_full}
python bench.py --steps 20 --warmup 3 2>$O/bench.err | grep '^{' > $O/bench.json; python -c "import json;d=json.load(open('$O/bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['kernel_ms_avg'])"
