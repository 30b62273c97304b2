#!/bin/bash
# Runs on the GPU box (via gpurun): the measurements and rocprofv3 passes whose summaries are kept under profiles/.
# Usage: bash tools/collect_profiles.sh <tag>     (writes gpurun_out/<tag>/)
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r01}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp
export TMPDIR=/tmp
echo "[1/8] bench.py (C4)"; python3 $R/bench.py --steps 20 --warmup 3 > $O/c4_bench.json
echo "[2/8] kernel trace"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_trace -o c4 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/c4_bench_under_rocprof.json 2> $O/c4_trace.log
python3 $R/tools/summarize_rocprof.py trace $O/c4_trace $O/c4_kernel_stats.csv
echo "[3/8] pmc FETCH_SIZE"; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c4_pmc_fetch -o f -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/c4_pmc_fetch.log
python3 $R/tools/summarize_rocprof.py pmc $O/c4_pmc_fetch $O/c4_pmc_fetch_summary.csv
echo "[4/8] pmc WRITE_SIZE"; rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c4_pmc_write -o w -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/c4_pmc_write.log
python3 $R/tools/summarize_rocprof.py pmc $O/c4_pmc_write $O/c4_pmc_write_summary.csv
python3 - "$O" <<'PY'
import csv, json, sys
o = sys.argv[1]
def sweep_mean(path):
    # the packed sweep (mask mode 3) - the run also holds the five u8-row reference sweeps (mask mode 0) of bench.py
    rows = [r for r in csv.DictReader(open(path)) if "sweep_kernel" in r["kernel"]]
    packed = [r for r in rows if r["kernel"].rstrip('"').rstrip().endswith(", 3, 16>")]
    if not packed:
        raise SystemExit("no packed sweep kernel in " + path)
    return float(packed[0]["mean"])
fetch, write = sweep_mean(o + "/c4_pmc_fetch_summary.csv"), sweep_mean(o + "/c4_pmc_write_summary.csv")
json.dump({"10000000x5000:packed": {
    "hbm_bytes_per_launch": (2 * fetch + write) * 1024, "fetch_size_kb": fetch, "write_size_kb": write,
    "correction": "(2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md HBM section (gfx950 FETCH_SIZE halves wide coalesced reads); separate --pmc passes",
    "source": "c4_pmc_fetch_summary.csv, c4_pmc_write_summary.csv of the same collection",
    "algorithmic_bytes": 681 * 10_000_000}}, open(o + "/pmc_traffic_packed.json", "w"), indent=1)
PY
echo "[4b/8] the u8-row layout for comparison"; python3 $R/bench.py --steps 10 --warmup 2 --layout bytes --no-cpu-baseline > $O/c4_bench_u8_layout.json
echo "[5/8] other configs"; python3 $R/tools/measure_configs.py C2 C3 C4m C5 C3h WIDE 2>/dev/null | grep '^{' > $O/other_configs.jsonl
MEASURE_LAYOUT=bytes python3 $R/tools/measure_configs.py C2 C3 C4m C5 C3h WIDE 2>/dev/null | grep '^{' > $O/other_configs_u8_layout.jsonl
echo "[6/8] multi-allelic path"; python3 $R/tools/measure_general.py 2>/dev/null | grep '^{' | grep hudson > $O/general_path.jsonl
echo "[7/8] pairwise"; python3 $R/tools/measure_pairwise.py 1000000x2500 200000x500 2>/dev/null | grep '^{' > $O/pairwise.jsonl
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pd_trace -o pd -- python3 $R/tools/measure_pairwise.py 1000000x2500 > /dev/null 2> $O/pd_trace.log
python3 $R/tools/summarize_rocprof.py trace $O/pd_trace $O/pairwise_kernel_stats.csv
echo "[8/8] run_vcf at reduced C4 scale"; python3 $R/tools/run_vcf_scale.py --sites 200000 --samples 2500 2>/dev/null | tail -1 > $O/run_vcf_scale_200k_x_2500.json
rm -rf $O/c4_trace $O/c4_pmc_fetch $O/c4_pmc_write $O/pd_trace
ls -la $O
