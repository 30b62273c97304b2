#!/bin/bash
# (needs the experimental build: tiles_pipelined with a DEPTH template parameter - a ring of DEPTH row buffers - and FMH_PIPE=4 selecting DEPTH 4; not kept in the tree)
# the pipelined tile loop with FOUR row steps of loads in flight (FMH_PIPE=4) against the default (two for one and two groups, plain loop for four)
set -o pipefail
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r03k
mkdir -p $O
FMH_PIPE=4 timeout -k 10 600 python -m pytest tests/test_gpu_device_parity.py -x -q > $O/parity_pipe4.log 2>&1; rc=$?; echo "parity with four row steps in flight: exit $rc"; tail -2 $O/parity_pipe4.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/ab_env.py FMH_PIPE=4 1000000x500 10000000x500 5000000x1250 1000000x1250 > $O/ab_pipe4_hudson.jsonl 2>$O/ab_hudson.err && cat $O/ab_pipe4_hudson.jsonl
AB_KIND=wc4 timeout -k 10 300 python tools/ab_env.py FMH_PIPE=4 5000000x1250 1000000x500 > $O/ab_pipe4_wc4.jsonl 2>$O/ab_wc4.err && cat $O/ab_pipe4_wc4.jsonl
AB_KIND=sum4 timeout -k 10 300 python tools/ab_env.py FMH_PIPE=4 5000000x1250 1000000x500 > $O/ab_pipe4_sum4.jsonl 2>$O/ab_sum4.err && cat $O/ab_pipe4_sum4.jsonl
