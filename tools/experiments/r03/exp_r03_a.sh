#!/bin/bash
# round 3, first GPU session: the whole GPU suite on the refactored library (options through fmh_set_option, generic sharded sweeps),
# then the C3 W&C kernel: where its time goes (store families switched off) and the pipelined tile loop (FMH_PIPE=1) against the plain one
set -o pipefail
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r03a
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
for mode in "" 1 nostate stateonly abonly aonly; do
  MEASURE_NO_SITE_OUTPUTS=$mode timeout -k 10 120 python tools/measure_configs.py C3 2>/dev/null | grep '^{' | sed "s/^{/{\"no_site_outputs\": \"$mode\", /" >> $O/c3_store_families.jsonl
done
for mode in "" 1; do
  FMH_PIPE=1 MEASURE_NO_SITE_OUTPUTS=$mode timeout -k 10 120 python tools/measure_configs.py C3 2>/dev/null | grep '^{' | sed "s/^{/{\"variant\": 1, \"no_site_outputs\": \"$mode\", /" >> $O/c3_store_families.jsonl
done
cat $O/c3_store_families.jsonl
AB_KIND=wc4 timeout -k 10 200 python tools/ab_env.py FMH_PIPE=1 5000000x1250 2>/dev/null | grep '^{' > $O/ab_pipe_wc4.jsonl
AB_KIND=sum4 timeout -k 10 200 python tools/ab_env.py FMH_PIPE=1 5000000x1250 2>/dev/null | grep '^{' > $O/ab_pipe_sum4.jsonl
timeout -k 10 200 python tools/ab_env.py FMH_PIPE=1 10000000x500 5000000x1250 2>/dev/null | grep '^{' > $O/ab_pipe_hudson.jsonl
cat $O/ab_pipe_wc4.jsonl $O/ab_pipe_sum4.jsonl $O/ab_pipe_hudson.jsonl | cut -c1-260
