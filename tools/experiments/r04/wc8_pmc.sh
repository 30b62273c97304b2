#!/bin/bash
# Round 4: issue / wait / instruction-mix / LDS counters of the fused W&C kernels with five and eight groups (2 M x 2 500).
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out/${1:-r04_wc8}
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -o p -- python3 $R/tools/measure_wc_groups.py 5 8 > /dev/null 2> $O/p1.log
python3 $R/tools/summarize_rocprof.py pmc $O/p1 $O/wc_5_8_groups_pmc_issue_wait.csv
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/p2 -o p -- python3 $R/tools/measure_wc_groups.py 5 8 > /dev/null 2> $O/p2.log
python3 $R/tools/summarize_rocprof.py pmc $O/p2 $O/wc_5_8_groups_pmc_inst_mix.csv
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_WAIT_INST_ANY --output-format csv -d $O/p3 -o p -- python3 $R/tools/measure_wc_groups.py 5 8 > /dev/null 2> $O/p3.log
python3 $R/tools/summarize_rocprof.py pmc $O/p3 $O/wc_5_8_groups_pmc_lds_vmem.csv || true
rm -rf $O/p1 $O/p2 $O/p3
grep -h "sweep_kernel<[58], 8" $O/wc_5_8_groups_pmc_*.csv
