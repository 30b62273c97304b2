#!/bin/bash
# Runs on the GPU box (via gpurun): counter passes that split SQ_WAIT_INST_ANY on the four-lane kernels (C2, C3, C3 summaries) against the
# sixteen-lane C4 kernel.  Each pass is its own rocprofv3 run (--pmc only), 8 SQ counters at most.
# Usage: bash tools/collect_r04_wait_inst.sh <tag> [extra env assignments for the measured process, e.g. FMH_FLAT=1]
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r04_wait}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp
export TMPDIR=/tmp
CFGS=${CFGS:-C2 C2x10 C3 C3h C4}
step() { echo "[$(date +%H:%M:%S)] $*"; }
pass() {  # name, counters...
  local name=$1; shift
  step "pmc pass $name: $*"
  rocprofv3 --pmc "$@" --output-format csv -d $O/$name -o p -- python3 $R/tools/measure_configs.py $CFGS > /dev/null 2> $O/$name.log
  python3 $R/tools/summarize_rocprof.py pmc $O/$name $O/pmc_${name}_summary.csv
  rm -rf $O/$name
}
pass issue_wait SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_WAVES
pass wait_split SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM
pass icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ SQC_TC_STALL SQ_WAVE_CYCLES
pass vmem SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM_RD
pass insts SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
step "kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o c -- python3 $R/tools/measure_configs.py $CFGS > $O/trace_lines.jsonl 2> $O/trace.log
python3 $R/tools/summarize_rocprof.py trace $O/trace $O/kernel_stats.csv
rm -rf $O/trace
python3 $R/tools/measure_configs.py $CFGS 2>/dev/null | grep '^{' > $O/configs.jsonl
ls -la $O
