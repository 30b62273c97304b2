#!/bin/bash
# round 3, second GPU session: compile-time variants of the packed counting cores / the W&C epilogue (tools/build_variant.sh: base, b = one-instruction
# popcount-accumulate, c = b + scalar store bases, d = c + W&C shape constants in LDS), each with the plain and the pipelined four-lane tile loop;
# then the new tests (fused region sweep, slab failure, the self-verifying N > 1 bench line) and the parity suites on variant d
set -o pipefail
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r03b
mkdir -p $O
for rep in 1 2; do
for v in base b c d; do
  if [ $v = base ]; then L=ferromic_amd/lib/libferromic_hip.so; else L=build/variants/$v/libferromic_hip.so; fi
  for kind in wc4 sum4; do
    FMH_LIB_PATH=$L AB_KIND=$kind timeout -k 10 120 python tools/ab_env.py FMH_PIPE=1 5000000x1250 2>/dev/null | grep '^{' | head -2 | sed "s/^{/{\"lib\": \"$v\", /" >> $O/variants.jsonl
  done
  FMH_LIB_PATH=$L timeout -k 10 200 python tools/ab_env.py FMH_PIPE=1 10000000x2500 10000000x500 2>/dev/null | grep '^{' | grep -E '"sites": (10000000|1250000|1000000),' | sed "s/^{/{\"lib\": \"$v\", /" >> $O/variants.jsonl
done
done
cut -c1-230 $O/variants.jsonl
timeout -k 10 600 python -m pytest tests/test_gpu_device_parity.py -x -q -k "fused_region" > $O/pytest_fused.log 2>&1; echo "fused exit $?"; tail -3 $O/pytest_fused.log
timeout -k 10 600 python -m pytest tests/test_gpu_run_vcf.py -x -q -k "slab or large_region" > $O/pytest_slab.log 2>&1; echo "slab exit $?"; tail -3 $O/pytest_slab.log
timeout -k 10 600 python -m pytest tests/test_gpu_bench_multirank.py -x -q > $O/pytest_bench.log 2>&1; echo "bench exit $?"; tail -3 $O/pytest_bench.log
FMH_LIB_PATH=build/variants/d/libferromic_hip.so timeout -k 10 900 python -m pytest tests/test_gpu_device_parity.py tests/test_gpu_scale.py tests/test_gpu_device_fuzz.py -x -q > $O/pytest_variant_d.log 2>&1; echo "variant d parity exit $?"; tail -3 $O/pytest_variant_d.log
