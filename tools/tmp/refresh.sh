set -eo pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02m; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for wg in 3 4; do $R/build/micro/store_bursts 10000000 7 640 $wg >> $O/store_bursts_packed_pitch.jsonl; done
$R/build/micro/store_bursts 10000000 7 5008 4 >> $O/store_bursts_u8_pitch.jsonl
python3 $R/bench.py --steps 20 --warmup 3 2> $O/c4_bench.err | grep '^{' > $O/c4_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_trace -o c4 -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline 2> $O/c4_trace.log | grep '^{' > $O/c4_bench_under_rocprof.json
python3 $R/tools/summarize_rocprof.py trace $O/c4_trace $O/c4_kernel_stats.csv
for s in 10000000 5000000 2500000 1250000; do
  python3 $R/bench.py --sites $s --steps 50 --warmup 5 --no-cpu-baseline --u8-reference-steps 0 --sync-steps 2>/dev/null | grep '^{' >> $O/strong_scaling_step_sizes_blocking.jsonl
  python3 $R/bench.py --sites $s --steps 50 --warmup 5 --no-cpu-baseline --u8-reference-steps 0 2>/dev/null | grep '^{' >> $O/strong_scaling_step_sizes_pipelined_local.jsonl
  python3 $R/bench.py --sites $s --steps 50 --warmup 5 --no-cpu-baseline --u8-reference-steps 0 --force-collective 2>/dev/null | grep '^{' >> $O/strong_scaling_step_sizes_sharded_path.jsonl
done
rm -rf $O/c4_trace
cat $O/store_bursts_packed_pitch.jsonl
