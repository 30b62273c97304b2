#!/bin/bash
# eight-group kernels: one batch of loads per row (deep batches, masks re-read from LDS per row) against the row loop of several shallow
# trips (FMH_PACKED_NO_PREFETCH=1), both skipping the padded groups, five to eight groups
set -o pipefail
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r03q
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_device_parity.py tests/test_gpu_device_fuzz.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "tests: exit $rc"; tail -2 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
timeout -k 10 300 python tools/measure_wc_groups.py 5 6 8 12 | python -c "import sys,json; [print(json.dumps(dict(json.loads(l), rows_loop='one batch per row'))) for l in sys.stdin]" >> $O/wc_groups.jsonl
FMH_PACKED_NO_PREFETCH=1 timeout -k 10 300 python tools/measure_wc_groups.py 5 6 8 12 | python -c "import sys,json; [print(json.dumps(dict(json.loads(l), rows_loop='shallow trips'))) for l in sys.stdin]" >> $O/wc_groups.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r03q/wc_groups.jsonl'):
    d=json.loads(l); print(d['rows_loop'], d['groups'], round(d.get('fused_ms',0),3), round(d.get('fused_totals_only_ms',0),3), round(d['counts_path_ms'],3), round(d['counts_path_totals_only_ms'],3))
PY
