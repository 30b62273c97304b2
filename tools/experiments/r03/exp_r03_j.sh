#!/bin/bash
# (needs the experimental route: a sweep_packed8.hip with FMH_ROUTE_LPR 8 next to sweep_packed4.hip and FMH_PACKED_LPR=8 accepted in abi.hip - not kept in the tree)
# eight lanes per packed row (FMH_PACKED_LPR=8) against the default four on rows of 1 000 and 2 500 haplotypes: parity first, then same-process A/Bs
set -o pipefail
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r03j
mkdir -p $O
FMH_PACKED_LPR=8 timeout -k 10 600 python -m pytest tests/test_gpu_device_parity.py -x -q > $O/parity_lpr8.log 2>&1; rc=$?; echo "parity with eight lanes per row: exit $rc"; tail -2 $O/parity_lpr8.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/ab_env.py FMH_PACKED_LPR=8 1000000x500 10000000x500 5000000x1250 1000000x1250 > $O/ab_lpr8_hudson.jsonl 2>$O/ab_hudson.err && cat $O/ab_lpr8_hudson.jsonl
AB_KIND=wc4 timeout -k 10 300 python tools/ab_env.py FMH_PACKED_LPR=8 5000000x1250 1000000x500 > $O/ab_lpr8_wc4.jsonl 2>$O/ab_wc4.err && cat $O/ab_lpr8_wc4.jsonl
AB_KIND=sum4 timeout -k 10 300 python tools/ab_env.py FMH_PACKED_LPR=8 5000000x1250 1000000x500 > $O/ab_lpr8_sum4.jsonl 2>$O/ab_sum4.err && cat $O/ab_lpr8_sum4.jsonl
