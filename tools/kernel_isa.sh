#!/bin/bash
# tools/kernel_isa.sh <unit> <mangled-name-fragment> [out.s]: the gfx950 ISA of one kernel of ferromic_amd/csrc/<unit>.hip (device side only)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
U=$1; K=$2; O=${3:-/tmp/kernel.s}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math --cuda-device-only -S -o /tmp/${U}_dev.s $R/ferromic_amd/csrc/$U.hip 2>/dev/null
awk -v k="$K" '$0 ~ "^_Z.*" k ".*:" {f=1} f{print} f && $0 ~ "\\.amdhsa_kernel _Z.*" k {exit}' /tmp/${U}_dev.s > $O
echo "$(wc -l < $O) lines -> $O; scratch ops: $(grep -c scratch_ $O || true)"
grep "$K.*\.num_vgpr\|$K.*\.private_seg_size" /tmp/${U}_dev.s | sed 's/^.*\.\(num_vgpr\|private_seg_size\)/\1/'
