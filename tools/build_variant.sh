#!/bin/bash
# tools/build_variant.sh NAME "-DFMH_X_..." : libferromic_hip.so with extra compile flags into build/variants/NAME/ (same-box A/Bs of
# compile-time kernel variants: FMH_LIB_PATH=build/variants/NAME/libferromic_hip.so selects it for ferromic_amd._abi)
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
NAME=$1; shift
mkdir -p $R/build/variants/$NAME/obj
make -j8 -C $R/ferromic_amd/csrc OBJ=$R/build/variants/$NAME/obj OUT=$R/build/variants/$NAME EXTRA="$*" $R/build/variants/$NAME/libferromic_hip.so 2>&1 | grep -E "error|warning: " || true
ls -la $R/build/variants/$NAME/libferromic_hip.so
