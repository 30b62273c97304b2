#!/bin/bash
# Runs on the GPU box (via gpurun): the measurements and rocprofv3 passes whose summaries are kept under profiles/<round>/.
# Usage: bash tools/collect_profiles.sh <tag>     (writes gpurun_out/<tag>/; copy what is to be judged into profiles/<tag>/)
# Counter passes (--pmc) are separate runs from the kernel trace, as MI355X_MICROARCH.md prescribes; the profiled program is always
# `python3 <script>` itself (no env / shell hop between rocprofv3 and the process that touches the GPU).
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r04}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp
export TMPDIR=/tmp
step() { echo "[$(date +%H:%M:%S)] $*"; }

step "pmc FETCH_SIZE"; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c4_pmc_fetch -o f -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/c4_pmc_fetch.log
python3 $R/tools/summarize_rocprof.py pmc $O/c4_pmc_fetch $O/c4_pmc_fetch_summary.csv
step "pmc WRITE_SIZE"; rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c4_pmc_write -o w -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/c4_pmc_write.log
python3 $R/tools/summarize_rocprof.py pmc $O/c4_pmc_write $O/c4_pmc_write_summary.csv
step "the u8-row layout: FETCH_SIZE, WRITE_SIZE (its bench line follows the counters)"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c4_pmc_fetch8 -o f -- python3 $R/bench.py --steps 3 --warmup 1 --layout bytes --no-cpu-baseline > /dev/null 2> $O/c4_pmc_fetch8.log
python3 $R/tools/summarize_rocprof.py pmc $O/c4_pmc_fetch8 $O/c4_pmc_fetch_summary_u8_layout.csv
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c4_pmc_write8 -o w -- python3 $R/bench.py --steps 3 --warmup 1 --layout bytes --no-cpu-baseline > /dev/null 2> $O/c4_pmc_write8.log
python3 $R/tools/summarize_rocprof.py pmc $O/c4_pmc_write8 $O/c4_pmc_write_summary_u8_layout.csv
python3 - "$O" "$R" "$TAG" <<'PY'
import csv, json, sys
o, root, tag = sys.argv[1:4]
sys.path.insert(0, root)
import bench
def sweep_mean(path, mask_mode):
    rows = [r for r in csv.DictReader(open(path)) if "sweep_kernel<2, 3, false, false, %d, 16" % mask_mode in r["kernel"]]
    if not rows:
        raise SystemExit("no Hudson sweep kernel with mask mode %d in %s" % (mask_mode, path))
    return float(rows[0]["mean"])
out = {}
for layout, suffix, mode, b_site in (("packed", "", 3, 681), ("bytes", "_u8_layout", 0, 5056)):
    fetch, write = sweep_mean(o + "/c4_pmc_fetch_summary%s.csv" % suffix, mode), sweep_mean(o + "/c4_pmc_write_summary%s.csv" % suffix, mode)
    out["10000000x5000:" + layout] = {
        "hbm_bytes_per_launch": (2 * fetch + write) * 1024, "fetch_size_kb": fetch, "write_size_kb": write,
        "correction": "(2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md HBM section (gfx950 FETCH_SIZE halves wide coalesced reads); separate --pmc passes",
        "source": "profiles/%s/c4_pmc_fetch_summary%s.csv, c4_pmc_write_summary%s.csv" % (tag, suffix, suffix),
        "kernel_source_sha": bench.kernel_source_sha(),
        "algorithmic_bytes": b_site * 10_000_000}
json.dump(out, open(o + "/pmc_traffic.json", "w"), indent=1)
PY
cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json   # on this box's copy of the repo: the bench lines below report the traffic just measured
step "bench.py (C4, N = 1)"; python3 $R/bench.py --steps 20 --warmup 3 2> $O/c4_bench.err | grep '^{' > $O/c4_bench.json
step "kernel trace of the same command"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_trace -o c4 -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline 2> $O/c4_trace.log | grep '^{' > $O/c4_bench_under_rocprof.json
python3 $R/tools/summarize_rocprof.py trace $O/c4_trace $O/c4_kernel_stats.csv
python3 $R/bench.py --steps 10 --warmup 2 --layout bytes --no-cpu-baseline 2>/dev/null | grep '^{' > $O/c4_bench_u8_layout.json
step "strong-scaling step sizes on one GPU (the per-rank slab of 1, 2, 4, 8 GPUs): blocking fmh_hudson_sweep, the pipelined begin / end path on a local communicator (bench's N = 1 default) and on a one-rank RCCL group"
for s in 10000000 5000000 2500000 1250000; do
  python3 $R/bench.py --sites $s --steps 50 --warmup 5 --no-cpu-baseline --u8-reference-steps 0 --sync-steps 2>/dev/null | grep '^{' >> $O/strong_scaling_step_sizes_blocking.jsonl
  python3 $R/bench.py --sites $s --steps 50 --warmup 5 --no-cpu-baseline --u8-reference-steps 0 2>/dev/null | grep '^{' >> $O/strong_scaling_step_sizes_pipelined_local.jsonl
  python3 $R/bench.py --sites $s --steps 50 --warmup 5 --no-cpu-baseline --u8-reference-steps 0 --force-collective 2>/dev/null | grep '^{' >> $O/strong_scaling_step_sizes_sharded_path.jsonl
done
step "same-process A/B of the library options at the C4 shape and on four-lane rows (tools/ab_env.py: unset vs set, one allocation, best of 3 x 20 launches)"
for sw in FMH_DEFER_TILES=1 FMH_DEFER_TILES=8 FMH_GRID_PER_CU=2 FMH_GRID_PER_CU=4 FMH_PACKED_UNROLL=4; do
  python3 $R/tools/ab_env.py $sw 10000000x2500 2>/dev/null | grep '^{' >> $O/ab_switches_c4_shape.jsonl
done
python3 $R/tools/ab_env.py FMH_PIPE=0 10000000x500 5000000x1250 2>/dev/null | grep '^{' >> $O/ab_pipelined_tile_loop.jsonl
for k in wc4 sum4; do AB_KIND=$k python3 $R/tools/ab_env.py FMH_PIPE=1 5000000x1250 2>/dev/null | grep '^{' >> $O/ab_pipelined_tile_loop.jsonl; done
step "other configs (every kernel warmed for 50 ms before it is timed)"; python3 $R/tools/measure_configs.py C2 C2x10 C3 C3h C4 C4m C5 WIDE C4f C2f 2>/dev/null | grep '^{' > $O/other_configs.jsonl
step "kernel trace of C2, C3, C3 summaries, C2x10 and the fused region sweep"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg_trace -o c -- python3 $R/tools/measure_configs.py C2 C3 C3h C2x10 C4f > $O/cfg_trace_lines.jsonl 2> $O/cfg_trace.log
python3 $R/tools/summarize_rocprof.py trace $O/cfg_trace $O/c2_c3_kernel_stats.csv
step "FETCH_SIZE / WRITE_SIZE of the C3 W&C kernel and the fused region sweep"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c3_fetch -o f -- python3 $R/tools/measure_configs.py C3 C4f > /dev/null 2> $O/c3_fetch.log
python3 $R/tools/summarize_rocprof.py pmc $O/c3_fetch $O/c3_c4f_pmc_fetch_summary.csv
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c3_write -o w -- python3 $R/tools/measure_configs.py C3 C4f > /dev/null 2> $O/c3_write.log
python3 $R/tools/summarize_rocprof.py pmc $O/c3_write $O/c3_c4f_pmc_write_summary.csv
MEASURE_LAYOUT=bytes python3 $R/tools/measure_configs.py C2 C3 C3h C4 C4m C5 WIDE 2>/dev/null | grep '^{' > $O/other_configs_u8_layout.jsonl
step "C5 counting routes on u8 rows: v_dot4 / int8 MFMA (4 and 2 K steps in flight)"
for v in 0 1 2; do FMH_COUNTS_MFMA=$v MEASURE_LAYOUT=bytes python3 $R/tools/measure_configs.py C5 2>/dev/null | grep '^{' | sed "s/^{/{\"FMH_COUNTS_MFMA\": $v, /" >> $O/c5_counting_routes_u8.jsonl; done
export MEASURE_LAYOUT=bytes
export FMH_COUNTS_MFMA=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_trace -o c5 -- python3 $R/tools/measure_configs.py C5 > /dev/null 2> $O/c5_trace.log
python3 $R/tools/summarize_rocprof.py trace $O/c5_trace $O/c5_mfma_kernel_stats.csv
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/c5_pmc -o c5 -- python3 $R/tools/measure_configs.py C5 > /dev/null 2> $O/c5_pmc.log
python3 $R/tools/summarize_rocprof.py pmc $O/c5_pmc $O/c5_mfma_pmc_summary.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c5_pmc_fetch -o c5 -- python3 $R/tools/measure_configs.py C5 > /dev/null 2> $O/c5_pmc_fetch.log
python3 $R/tools/summarize_rocprof.py pmc $O/c5_pmc_fetch $O/c5_mfma_pmc_fetch_summary.csv
unset FMH_COUNTS_MFMA
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/c5_pmc_dot4 -o c5 -- python3 $R/tools/measure_configs.py C5 > /dev/null 2> $O/c5_pmc_dot4.log
python3 $R/tools/summarize_rocprof.py pmc $O/c5_pmc_dot4 $O/c5_dot4_pmc_summary.csv
unset MEASURE_LAYOUT
step "issue / wait counters of the C2, C3, C3-summaries and C4 kernels (two passes of 8 SQ counters)"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/cfg_pmc1 -o p -- python3 $R/tools/measure_configs.py C2 C3 C3h C4 C4f > /dev/null 2> $O/cfg_pmc1.log
python3 $R/tools/summarize_rocprof.py pmc $O/cfg_pmc1 $O/c2_c3_c4_pmc_issue_wait_summary.csv
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT --output-format csv -d $O/cfg_pmc2 -o p -- python3 $R/tools/measure_configs.py C2 C3 C3h C4 C4f > /dev/null 2> $O/cfg_pmc2.log
python3 $R/tools/summarize_rocprof.py pmc $O/cfg_pmc2 $O/c2_c3_c4_pmc_inst_mix_summary.csv
step "multi-allelic paths"; python3 $R/tools/measure_general.py 2>/dev/null | grep '^{' | grep hudson > $O/general_path.jsonl
for hi in 1 0; do for f in 0.02 0.002; do FMH_ROW_HI=$hi MEASURE_MULTI_FRACTION=$f python3 $R/tools/measure_general.py 2>/dev/null | grep hudson | sed "s/^{/{\"FMH_ROW_HI\": $hi, \"multi_fraction\": $f, /" >> $O/general_path_mostly_biallelic.jsonl; done; done
MEASURE_ALLELE7=1 python3 $R/tools/measure_wc_general.py 2>/dev/null | grep '^{' > $O/wc_general_5_8_groups.jsonl
python3 $R/tools/measure_wc_groups.py 2 4 5 8 12 26 2>/dev/null | grep '^{' > $O/wc_groups.jsonl
step "pairwise"; python3 $R/tools/measure_pairwise.py 1000000x2500 200000x500 2>/dev/null | grep '^{' > $O/pairwise.jsonl
step "upload, API, run_vcf"
python3 $R/tools/measure_h2d.py 2>/dev/null | grep '^{' > $O/h2d.json
python3 $R/tools/measure_api_c2.py 2>/dev/null | grep '^{' > $O/api_c2.json
python3 $R/tools/measure_api_pybench.py 2>/dev/null | grep '^{' > $O/api_pybench.jsonl
for c in "1000000 250" "65536 128"; do python3 $R/tools/measure_api_two_matrices.py $c 2>/dev/null | grep '^{' >> $O/api_two_matrices.jsonl; done
python3 $R/tools/run_vcf_scale.py --sites 200000 --samples 2500 2>/dev/null | tail -1 > $O/run_vcf_scale_200k_x_2500.json
python3 $R/tools/run_vcf_many_regions.py 2>/dev/null | tail -1 > $O/run_vcf_500_regions.json
for c in gzip bgzf; do python3 $R/tools/run_vcf_scale.py --sites 50000 --samples 2500 --compress $c 2>/dev/null | tail -1 >> $O/run_vcf_compressed_inputs.jsonl; done
step "bytes the sweeps of one run_vcf region read: round 2's binary (four sweeps per variant set) against round 3's (one fused sweep)"
[ -x $R/build/variants/r02/run_vcf ] && python3 $R/tools/run_vcf_fetch.py r02=$R/build/variants/r02/run_vcf r03=$R/ferromic_amd/bin/run_vcf 2>/dev/null | grep '^{' > $O/run_vcf_region_fetch_size.jsonl
python3 $R/tools/kernel_resources.py > $O/kernel_resources.txt || true
rm -rf $O/cfg_trace $O/c3_fetch $O/c3_write $O/c4_trace $O/c4_pmc_fetch $O/c4_pmc_write $O/c4_pmc_fetch8 $O/c4_pmc_write8 $O/c5_trace $O/c5_pmc $O/c5_pmc_fetch $O/c5_pmc_dot4 $O/cfg_pmc1 $O/cfg_pmc2
ls -la $O
