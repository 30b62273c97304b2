#!/bin/bash
# round 3, third GPU session: deferral on four-lane rows (variant e) by depth; warm measurement of C2 / C3 / fused kernels; the API benchmark
set -o pipefail
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r03c
mkdir -p $O
L=build/variants/e/libferromic_hip.so
for d in 1 2 4 8; do
  for kind in wc4 sum4; do
    FMH_LIB_PATH=$L AB_KIND=$kind timeout -k 10 120 python tools/ab_env.py FMH_DEFER_TILES=$d 5000000x1250 2>/dev/null | grep '^{' | head -3 | sed "s/^{/{\"lib\": \"e\", /" >> $O/defer4.jsonl
  done
  FMH_LIB_PATH=$L timeout -k 10 200 python tools/ab_env.py FMH_DEFER_TILES=$d 10000000x500 5000000x1250 2>/dev/null | grep '^{' | grep -E '"sites": (10000000|5000000|1250000|1000000|625000),' | sed "s/^{/{\"lib\": \"e\", /" >> $O/defer4.jsonl
done
cut -c1-200 $O/defer4.jsonl
for v in base bd; do
  if [ $v = base ]; then L=ferromic_amd/lib/libferromic_hip.so; else L=build/variants/$v/libferromic_hip.so; fi
  for pv in 0 1; do
    FMH_LIB_PATH=$L FMH_PIPE=$pv timeout -k 10 300 python tools/measure_configs.py C2 C3 C3h C2x10 C4 C4f C2f 2>/dev/null | grep '^{' | sed "s/^{/{\"lib\": \"$v\", \"pipe\": $pv, /" >> $O/configs_warm.jsonl
  done
done
cut -c1-330 $O/configs_warm.jsonl
timeout -k 10 600 python tools/measure_api_pybench.py > $O/api_pybench.jsonl 2> $O/api_pybench.err; echo "pybench exit $?"; tail -3 $O/api_pybench.err; cut -c1-300 $O/api_pybench.jsonl | tail -30
