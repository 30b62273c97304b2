#!/bin/bash
# Round 4, flat-tile route: compile-time variants (one wave per workgroup; the swizzle on the LDS write side) against the default build, each
# process A/B-ing flat vs four-lane on the same allocation (tools/ab_env.py); plus the slab-size fit of the sixteen-lane kernel.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out/${1:-r04_flat_var}
mkdir -p $O
cd $R
export FMH_FLAT_DEFER=8
for v in default f_w1 f_swz f_w1swz; do
  if [ $v = default ]; then unset FMH_LIB_PATH; else export FMH_LIB_PATH=$R/build/variants/$v/libferromic_hip.so; fi
  if [ $v != default ]; then timeout -k 10 600 python -m pytest tests/test_gpu_flat_route.py -x -q 2>&1 | tail -3 | tee -a $O/pytest_$v.log; fi
  echo "== $v" | tee -a $O/ab.jsonl
  timeout -k 10 300 python tools/ab_env.py FMH_FLAT=1 10000000x500 2>/dev/null | grep '^{' | head -3 | tee -a $O/ab.jsonl
  AB_KIND=wc4 timeout -k 10 300 python tools/ab_env.py FMH_FLAT=1 5000000x1250 2>/dev/null | grep '^{' | head -3 | tee -a $O/ab.jsonl
  AB_KIND=sum4 timeout -k 10 300 python tools/ab_env.py FMH_FLAT=1 5000000x1250 2>/dev/null | grep '^{' | head -3 | tee -a $O/ab.jsonl
done
unset FMH_LIB_PATH FMH_FLAT_DEFER
timeout -k 10 300 python tools/measure_slab_fit.py 5000 2>/dev/null | grep '^{' | tee $O/slab_fit_5000.jsonl
