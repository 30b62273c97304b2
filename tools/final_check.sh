#!/bin/bash
# what the driver runs at the end of a round, on the final tree: the GPU suite, smoke(), the default bench line
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/final
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "gpu suite: exit $rc"; tail -2 $O/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 300 python bench.py 2>/dev/null | grep '^{' | tee $O/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])"
