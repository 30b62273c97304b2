set -eo pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02m; mkdir -p $O
for wg in 1 2 3 4; do $R/build/micro/store_bursts 10000000 7 640 $wg | tee -a $O/store_bursts_packed_pitch_v2.jsonl; done
