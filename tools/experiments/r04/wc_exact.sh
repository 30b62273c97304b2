#!/bin/bash
# Round 4: exact five-to-seven-group W&C kernels: parity, then the same process timing the padded eight-group kernel against them; plus the
# slab-size fit under rocprofv3 (kernel-trace durations, then busy / active counters).
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out/${1:-r04_wc_exact}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_device_parity.py tests/test_gpu_flat_route.py tests/test_gpu_comm.py -x -q 2>&1 | tail -8 | tee $O/pytest.log
for e in 0 1; do
  FMH_WC_EXACT=$e timeout -k 10 300 python tools/measure_wc_groups.py 5 6 7 8 2>/dev/null | grep '^{' | sed "s/^{/{\"FMH_WC_EXACT\": $e, /" | tee -a $O/wc_groups_exact_vs_padded.jsonl
done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/slab_trace -o s -- python3 $R/tools/measure_slab_fit.py 5000 > $O/slab_fit_under_trace.jsonl 2> $O/slab_trace.log
python3 $R/tools/slab_fit_from_trace.py $O/slab_trace $O/slab_fit_under_trace.jsonl > $O/slab_fit_kernel_trace.jsonl
rm -rf $O/slab_trace
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/slab_pmc -o s -- python3 $R/tools/measure_slab_fit.py 5000 > /dev/null 2> $O/slab_pmc.log
python3 $R/tools/slab_fit_from_trace.py $O/slab_pmc $O/slab_fit_under_trace.jsonl pmc > $O/slab_fit_pmc.jsonl
rm -rf $O/slab_pmc
