#!/bin/bash
# (needs two measurement flags in bench.py that are not kept: --tracks late = free and re-allocate the seven tracks after the u8 rows are released;
# --track-stagger B = start track k at k x B bytes into its block)
# where the per-site tracks sit: allocated next to the u8 rows (bench's order so far), after their release, and staggered
set -o pipefail
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r03x
mkdir -p $O
for rep in 1 2 3; do
for v in "--tracks early" "--tracks late" "--tracks late --track-stagger 4352" "--tracks late --track-stagger 69888"; do
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --u8-reference-steps 1 $v 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({'variant': '$v', 'kernel_ms_avg': round(d['roofline']['kernel_ms_avg'],4), 'ms_per_step': round(d['ms_per_step'],4), 'frac': round(d['roofline']['frac'],4), 'fst': d.get('results',{}).get('hudson_fst')}))" | tee -a $O/track_placement.jsonl
done
done
