#!/bin/bash
# Round 4, the measurements next to tools/collect_profiles.sh (on the GPU box, via gpurun): the W&C pair kernels of more than eight groups
# (trace per option and group count, counters of the shipped kernel) and the pairwise Gram routes (trace + MFMA / LDS counters).
# Outputs: gpurun_out/r04_wcbi/, gpurun_out/<pairwise tag>/ - copy what is to be judged into profiles/r04/.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
WCBI_VARIANTS="FMH_WC_BI_TOTALS=0 FMH_WC_BI_TOTALS=1" WCBI_GROUPS="12 26 40" bash $R/tools/experiments/r04/wc_many_bi.sh
WCBI_REPLICAS="0" bash $R/tools/experiments/r04/wc_many_bi_pmc.sh
bash $R/tools/experiments/r04/pairwise_final.sh r04_pairwise
