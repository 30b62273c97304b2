#!/bin/bash
# More than eight W&C groups, regional sums only: the biallelic pair kernel against the general one (2 M x 2 500; 12, 26 and 40 groups):
# parity tests, then per option the kernel trace (avg of the pair kernel per group count) and the wall time of fmh_wc_sweep_many.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out/r04_wcbi
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 python3 -m pytest $R/tests/test_gpu_device_parity.py -x -q -k "biallelic_pair_totals or many_groups" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
rm -f $O/wc_many_groups_pair_kernels.jsonl $O/wc_many_groups_pair_kernel_trace.csv
for v in ${WCBI_VARIANTS:-FMH_WC_BI_TOTALS=0 FMH_WC_BI_TOTALS=1 FMH_WC_BI_REPLICAS=1 FMH_WC_BI_REPLICAS=2 FMH_WC_BI_REPLICAS=4}; do
  export $v
  for G in ${WCBI_GROUPS:-12 26 40}; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o t -- python3 $R/tools/measure_wc_groups.py $G 2> $O/tr.log | grep '^{' | sed "s/^{/{\"env\": \"$v\", /" >> $O/wc_many_groups_pair_kernels.jsonl
    python3 $R/tools/summarize_rocprof.py trace $O/tr $O/tr.csv
    grep "wc_pair_totals\|wc_overall_totals\|wc_slot_finalize_wave" $O/tr.csv | sed "s/^/$v,$G,/" >> $O/wc_many_groups_pair_kernel_trace.csv
    rm -rf $O/tr
  done
  unset ${v%%=*}
done
cut -c1-250 $O/wc_many_groups_pair_kernels.jsonl
cat $O/wc_many_groups_pair_kernel_trace.csv
