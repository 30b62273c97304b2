#!/bin/bash
# Round 4, flat-tile route, register-staged variant: parity, then same-process A/Bs per deferral depth and occupancy cap.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out/${1:-r04_flat_rs}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_flat_route.py -x -q 2>&1 | tail -15 | tee $O/pytest_flat.log
for defer in 1 4 8; do
  for occ in 0 2 3; do
    export FMH_FLAT_DEFER=$defer FMH_MAX_OCC=$occ
    echo "== rs defer $defer occ $occ" | tee -a $O/ab.jsonl
    timeout -k 10 300 python tools/ab_env.py FMH_FLAT=1 10000000x500 1000000x500 2>/dev/null | grep '^{' | tee -a $O/ab.jsonl
    AB_KIND=wc4 timeout -k 10 300 python tools/ab_env.py FMH_FLAT=1 5000000x1250 2>/dev/null | grep '^{' | tee -a $O/ab.jsonl
    AB_KIND=sum4 timeout -k 10 300 python tools/ab_env.py FMH_FLAT=1 5000000x1250 2>/dev/null | grep '^{' | tee -a $O/ab.jsonl
  done
done
