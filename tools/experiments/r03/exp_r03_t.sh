#!/bin/bash
# eight-group kernels on SIXTEEN-lane rows (5 000 and 10 000 haplotypes): one deep batch per row against the shallow trips
set -o pipefail
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r03t
mkdir -p $O
for H in 5000 10000; do
for rep in 1 2; do
MEASURE_SITES=1000000 MEASURE_HAPLOTYPES=$H timeout -k 10 300 python tools/measure_wc_groups.py 5 8 12 | python -c "import sys,json; [print(json.dumps(dict(json.loads(l), rows_loop='one batch per row'))) for l in sys.stdin]" >> $O/wc_groups.jsonl
MEASURE_SITES=1000000 MEASURE_HAPLOTYPES=$H FMH_PACKED_NO_PREFETCH=1 timeout -k 10 300 python tools/measure_wc_groups.py 5 8 12 | python -c "import sys,json; [print(json.dumps(dict(json.loads(l), rows_loop='shallow trips'))) for l in sys.stdin]" >> $O/wc_groups.jsonl
done
done
python - <<'PY'
import json
for l in open('gpurun_out/r03t/wc_groups.jsonl'):
    d=json.loads(l); print(d['haplotypes'], d['rows_loop'], d['groups'], round(d.get('fused_ms',0),3), round(d.get('fused_totals_only_ms',0),3), round(d['counts_path_ms'],3), round(d['counts_path_totals_only_ms'],3))
PY
