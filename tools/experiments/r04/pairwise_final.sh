#!/bin/bash
# Round 4: round 3's Gram kernel against the phased one (slab epilogue), FP4 (default) and int8 operands, one and two planes: kernel traces + MFMA counters
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out/${1:-r04_pdf}
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for ph in 0 1; do for fmt in fp4 int8; do for planes in 1 2; do
  export FMH_PD_PHASED=$ph
  if [ $fmt = int8 ]; then export FMH_PD_INT8=1; else unset FMH_PD_INT8; fi
  if [ $planes = 2 ]; then export FMH_PD_TWO_PLANES=1; else unset FMH_PD_TWO_PLANES; fi
  tag=phased${ph}_${fmt}_planes${planes}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$tag -o t -- python3 $R/tools/measure_pairwise.py 1000000x2500 > $O/line_$tag.json 2> $O/t_$tag.log
  python3 $R/tools/summarize_rocprof.py trace $O/t_$tag $O/pairwise_${tag}_kernel_stats.csv
  rm -rf $O/t_$tag
done; done; done
export FMH_PD_PHASED=1 FMH_PD_INT8=1; unset FMH_PD_TWO_PLANES
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc -o t -- python3 $R/tools/measure_pairwise.py 1000000x2500 > /dev/null 2> $O/pmc.log
python3 $R/tools/summarize_rocprof.py pmc $O/pmc $O/pairwise_phased1_int8_planes1_pmc.csv
rm -rf $O/pmc
grep -h -i "gram\|slab\|planes" $O/pairwise_*_kernel_stats.csv
