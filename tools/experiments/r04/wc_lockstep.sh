#!/bin/bash
# Round 4: the fused W&C kernels with the two alleles of a slot in lockstep (default) against allele after allele (build/variants/nolockstep),
# same box: parity of the W&C suites on the default build, then wall times 4..8 groups (2 M x 2 500), the C3 configuration, and kernel traces.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out/${1:-r04_lockstep}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_device_parity.py tests/test_gpu_scale.py -x -q -k "wc or c3 or groups" 2>&1 | tail -4 | tee $O/pytest.log
cd /tmp; export TMPDIR=/tmp
for lib in default nolockstep; do
  if [ $lib = nolockstep ]; then export FMH_LIB_PATH=$R/build/variants/nolockstep/libferromic_hip.so; else unset FMH_LIB_PATH; fi
  for rep in 1 2; do
    python3 $R/tools/measure_wc_groups.py 4 5 6 7 8 2>/dev/null | grep '^{' | sed "s/^{/{\"build\": \"$lib\", /" >> $O/wc_groups_lockstep.jsonl
    python3 $R/tools/measure_configs.py C3 2>/dev/null | grep '^{' | sed "s/^{/{\"build\": \"$lib\", /" >> $O/c3_lockstep.jsonl
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o t -- python3 $R/tools/measure_wc_groups.py 4 5 8 > /dev/null 2> $O/tr.log
  python3 $R/tools/summarize_rocprof.py trace $O/tr $O/wc_groups_${lib}_kernel_stats.csv
  rm -rf $O/tr
done
cut -c1-330 $O/wc_groups_lockstep.jsonl
cut -c1-330 $O/c3_lockstep.jsonl
grep -h "sweep_kernel<[4-8], 8" $O/wc_groups_*_kernel_stats.csv
