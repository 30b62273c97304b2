#!/bin/bash
# Issue / wait / LDS counters of the biallelic pair kernel at 26 groups, 2 M x 2 500, one and four replicas per pair.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out/r04_wcbi
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for rep in ${WCBI_REPLICAS:-1 4}; do
  export FMH_WC_BI_REPLICAS=$rep
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -o p -- python3 $R/tools/measure_wc_groups.py 26 > /dev/null 2> $O/p1.log
  python3 $R/tools/summarize_rocprof.py pmc $O/p1 $O/wc_26_groups_bi_r${rep}_pmc_issue_wait.csv
  rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $O/p2 -o p -- python3 $R/tools/measure_wc_groups.py 26 > /dev/null 2> $O/p2.log
  python3 $R/tools/summarize_rocprof.py pmc $O/p2 $O/wc_26_groups_bi_r${rep}_pmc_lds.csv
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o t -- python3 $R/tools/measure_wc_groups.py 26 > /dev/null 2> $O/t.log
  python3 $R/tools/summarize_rocprof.py trace $O/t $O/wc_26_groups_bi_r${rep}_kernel_stats.csv
  rm -rf $O/p1 $O/p2 $O/t
  grep -h "biallelic\|^kernel" $O/wc_26_groups_bi_r${rep}_pmc_issue_wait.csv $O/wc_26_groups_bi_r${rep}_pmc_lds.csv $O/wc_26_groups_bi_r${rep}_kernel_stats.csv
done
