#!/bin/bash
# round 3, fourth GPU session: the new default library (one-instruction popcount chain, W&C shapes in LDS, pipelined tile loop for one and two
# groups on four-lane rows, three workgroups per CU on sixteen-lane rows) against the library it replaces, on one box; the whole GPU suite;
# the API benchmark; the fixed cost of a pipelined step, eager against a captured hipGraph
set -o pipefail
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r03d
mkdir -p $O
for rep in 1 2; do
for v in old new; do
  if [ $v = new ]; then L=ferromic_amd/lib/libferromic_hip.so; else L=build/variants/r03base/libferromic_hip.so; fi
  FMH_LIB_PATH=$L timeout -k 10 300 python tools/measure_configs.py C2 C3 C3h C2x10 C4 C5 $( [ $v = new ] && echo C4f C2f ) 2>/dev/null | grep '^{' | sed "s/^{/{\"lib\": \"$v\", /" >> $O/old_vs_new.jsonl
done
done
cut -c1-300 $O/old_vs_new.jsonl
for sw in FMH_GRID_PER_CU=4 FMH_GRID_PER_CU=2 FMH_PIPE=0; do timeout -k 10 200 python tools/ab_env.py $sw 10000000x2500 10000000x500 2>/dev/null | grep '^{' | grep -E '"sites": (10000000|1250000|1000000),' >> $O/ab_new.jsonl; done
cut -c1-200 $O/ab_new.jsonl
timeout -k 10 600 python tools/measure_api_pybench.py > $O/api_pybench.jsonl 2> $O/api_pybench.err; echo "pybench exit $?"; tail -3 $O/api_pybench.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench exit $?"; cut -c1-600 $O/bench_default.json
for S in 10000000 1250000 312500; do
  for mode in "--explicit-stream" "--graph"; do
    python bench.py --sites $S --steps 200 --warmup 20 --no-cpu-baseline --u8-reference-steps 0 --timing-sample 1000000 $mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({'sites': d['config']['total_sites'], 'mode': '$mode', 'ms_per_step': d['ms_per_step'], 'value': d['value'], 'fst': d['results']['hudson_fst']}))" >> $O/graph_vs_eager.jsonl
  done
done
cat $O/graph_vs_eager.jsonl
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -4 $O/pytest_gpu.log
