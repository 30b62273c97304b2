#!/bin/bash
# Round 4: the phase-interleaved Gram kernel: parity of the pairwise suites with FMH_PD_PHASED=1 (both operand formats), then timings and a trace
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out/${1:-r04_pd}
mkdir -p $O
cd $R
for int8 in 0 1; do for slabs in 1 0; do
  FMH_PD_PHASED=1 FMH_PD_SLABS=$slabs FMH_PD_INT8=$int8 timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "pairwise" 2>&1 | tail -4 | tee -a $O/pytest_phased.log
done; done
for ph in 0 1; do for int8 in 0 1; do
  FMH_PD_PHASED=$ph FMH_PD_INT8=$int8 timeout -k 10 300 python tools/measure_pairwise.py 1000000x2500 200000x500 2>/dev/null | grep '^{' | sed "s/^{/{\"FMH_PD_PHASED\": $ph, /" | tee -a $O/pairwise_phased_vs_round3.jsonl
done; done
cd /tmp; export TMPDIR=/tmp
for ph in 0 1 2; do
  export FMH_PD_PHASED=$((ph > 0)) FMH_PD_SLABS=$((ph > 1)) FMH_PD_INT8=1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace$ph -o t -- python3 $R/tools/measure_pairwise.py 1000000x2500 > /dev/null 2> $O/trace$ph.log
  python3 $R/tools/summarize_rocprof.py trace $O/trace$ph $O/pairwise_int8_phased${ph}_kernel_stats.csv
  rm -rf $O/trace$ph
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc$ph -o t -- python3 $R/tools/measure_pairwise.py 1000000x2500 > /dev/null 2> $O/pmc$ph.log
  python3 $R/tools/summarize_rocprof.py pmc $O/pmc$ph $O/pairwise_int8_phased${ph}_pmc.csv
  rm -rf $O/pmc$ph
done
