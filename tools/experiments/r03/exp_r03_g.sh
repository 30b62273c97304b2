#!/bin/bash
# round 3, final GPU session: the whole GPU suite, smoke, the default bench line, then the collection kept under profiles/r03/
set -o pipefail
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r03g
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?"; tail -1 $O/smoke.log
AB_KIND=wc4 TOOL_WC_NO_STATE=1 python tools/ab_env.py FMH_GRID_BLOCKS=0 5000000x1250 2>/dev/null | grep '^{' | head -1 | sed 's/^{/{"tracks": "no state", /' >> $O/c3_store_families_warm.jsonl
AB_KIND=wc4 TOOL_WC_NO_AB=1 python tools/ab_env.py FMH_GRID_BLOCKS=0 5000000x1250 2>/dev/null | grep '^{' | head -1 | sed 's/^{/{"tracks": "no a, b", /' >> $O/c3_store_families_warm.jsonl
AB_KIND=wc4 TOOL_WC_NO_COUNTS=1 python tools/ab_env.py FMH_GRID_BLOCKS=0 5000000x1250 2>/dev/null | grep '^{' | head -1 | sed 's/^{/{"tracks": "no group counts", /' >> $O/c3_store_families_warm.jsonl
AB_KIND=wc4 TOOL_WC_NO_COUNTS=1 TOOL_WC_NO_AB=1 TOOL_WC_NO_STATE=1 python tools/ab_env.py FMH_GRID_BLOCKS=0 5000000x1250 2>/dev/null | grep '^{' | head -1 | sed 's/^{/{"tracks": "none", /' >> $O/c3_store_families_warm.jsonl
AB_KIND=wc4 python tools/ab_env.py FMH_GRID_BLOCKS=0 5000000x1250 2>/dev/null | grep '^{' | head -1 | sed 's/^{/{"tracks": "all", /' >> $O/c3_store_families_warm.jsonl
cut -c1-200 $O/c3_store_families_warm.jsonl
timeout -k 10 900 bash tools/collect_profiles.sh r03 > $O/collect.log 2>&1; echo "collect exit $?"; tail -3 $O/collect.log
