#!/bin/bash
# run_vcf, 500 small regions: HIP API summary of one run (normal process exit so that the profiler can write its files)
set -o pipefail
cd "$(dirname "$0")/../../.." || exit 1
R=$(pwd)
O=$R/gpurun_out/r03ac
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export FERROMIC_FULL_TEARDOWN=1
RUN_VCF_WORKERS=1 RUN_VCF_PREFIX="rocprofv3 --hip-trace --stats --output-format csv -d $O/hip -o t --" python $R/tools/run_vcf_many_regions.py 2>$O/trace.err | tail -1 | cut -c1-200
find $O/hip -name '*.csv' | head
f=$(find $O/hip -name '*hip_api_stats.csv' | head -1); [ -n "$f" ] && head -30 "$f" | cut -c1-160
