#!/bin/bash
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out/${1:-r04_pdf2}
mkdir -p $O
cd $R
for int8 in 0 1; do
  FMH_PD_PHASED=1 FMH_PD_INT8=$int8 timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "pairwise" 2>&1 | tail -2 | tee -a $O/pytest_phased.log
done
bash tools/experiments/r04/pairwise_final.sh ${1:-r04_pdf2}
