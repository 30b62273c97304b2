#!/bin/bash
# the same C4 kernel timed by measure_configs.py (ten launches after a 50-ms warm-up) and by bench.py's variants, alternating, on one box
set -o pipefail
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r03y
mkdir -p $O
one() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({'how': 'bench $*', 'kernel_ms_avg': round(d['roofline']['kernel_ms_avg'],4), 'ms_per_step': round(d['ms_per_step'],4), 'launches_timed': d['roofline'].get('kernel_launches_timed')}))" | tee -a $O/c4_by_process.jsonl; }
for rep in 1 2 3; do
  python tools/measure_configs.py C4 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({'how': 'measure_configs C4', 'kernel_ms_avg': round(d['kernel_ms'],4)}))" | tee -a $O/c4_by_process.jsonl
  one --steps 20 --warmup 3
  one --steps 20 --warmup 3 --u8-reference-steps 0
  one --steps 20 --warmup 3 --sync-steps
  one --steps 200 --warmup 3
  one --steps 20 --warmup 3 --timing-sample 1
done
