#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r03f
mkdir -p $O; rm -f $O/wide.jsonl
for rep in 1 2; do
for v in old new; do
  if [ $v = new ]; then L=ferromic_amd/lib/libferromic_hip.so; else L=build/variants/r03base/libferromic_hip.so; fi
  FMH_LIB_PATH=$L timeout -k 10 300 python tools/measure_configs.py WIDE C5 C4 C4m $( [ $v = new ] && echo C4f ) 2>/dev/null | grep '^{' | sed "s/^{/{\"lib\": \"$v\", /" >> $O/wide.jsonl
done
done
python - <<'PY'
import json
for l in open('gpurun_out/r03f/wide.jsonl'):
    x=json.loads(l); print(x['lib'], x['config'], round(x['kernel_ms'],4), round(x['frac_of_8TBs'],3))
PY
timeout -k 10 900 python -m pytest tests/test_gpu_device_parity.py tests/test_gpu_device_fuzz.py tests/test_gpu_scale.py -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
