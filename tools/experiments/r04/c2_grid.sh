#!/bin/bash
# Round 4: the C2 launch (1 M x 1 000: 15 625 tiles, five rounds and a bit of the default grid) under grids that deal whole rounds
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out/${1:-r04_c2grid}
mkdir -p $O
cd $R
for gb in 0 652 768 782 977 1024 1303 1536 1954 2048; do
  FMH_GRID_BLOCKS=$gb timeout -k 10 120 python tools/measure_configs.py C2 C2 2>/dev/null | grep '^{' | sed "s/^{/{\"FMH_GRID_BLOCKS\": $gb, /" | tee -a $O/c2_grid_blocks.jsonl
done
for gb in 0 6511 7813; do
  FMH_GRID_BLOCKS=$gb timeout -k 10 120 python tools/measure_configs.py C2x10 2>/dev/null | grep '^{' | sed "s/^{/{\"FMH_GRID_BLOCKS\": $gb, /" | tee -a $O/c2_grid_blocks.jsonl
done
