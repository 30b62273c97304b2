#!/bin/bash
# Round 4: kernels held to the occupancy step just below their register count (sweep_min_blocks, FMH_OCC_STEPS) against the build without
# the rules (build/variants/occ0), same box, kernel traces: W&C 4 / 5 groups and the counting sweeps of 26 groups on 5 000-haplotype rows
# (sixteen lanes per row), W&C 6 groups on 2 500-haplotype rows (four lanes).
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out/${1:-r04_occ}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_device_parity.py tests/test_gpu_device_fuzz.py -x -q 2>&1 | tail -2 | tee $O/pytest.log
cd /tmp; export TMPDIR=/tmp
for lib in default occ0 default occ0; do
  if [ $lib = occ0 ]; then export FMH_LIB_PATH=$R/build/variants/occ0/libferromic_hip.so; else unset FMH_LIB_PATH; fi
  MEASURE_HAPLOTYPES=5000 MEASURE_SITES=1000000 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o t -- python3 $R/tools/measure_wc_groups.py 4 5 26 > /dev/null 2> $O/tr.log
  python3 $R/tools/summarize_rocprof.py trace $O/tr $O/tr.csv
  grep "sweep_kernel<[458], [18], false, false, 3, 16" $O/tr.csv | sed "s/^/$lib,1000000x5000,/" | tee -a $O/occupancy_steps_kernel_trace.csv
  rm -rf $O/tr
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o t -- python3 $R/tools/measure_wc_groups.py 6 > /dev/null 2> $O/tr.log
  python3 $R/tools/summarize_rocprof.py trace $O/tr $O/tr.csv
  grep "sweep_kernel<6, 8, false, false, 3, 4" $O/tr.csv | sed "s/^/$lib,2000000x2500,/" | tee -a $O/occupancy_steps_kernel_trace.csv
  rm -rf $O/tr
done
