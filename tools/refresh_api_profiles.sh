#!/bin/bash
# the API / run_vcf half of collect_profiles.sh alone (the kernel sources did not change: the kernel half stays)
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
R=$(pwd)
O=$R/gpurun_out/r03api
rm -rf $O; mkdir -p $O
python3 $R/tools/measure_h2d.py 2>/dev/null | grep '^{' > $O/h2d.json
python3 $R/tools/measure_api_c2.py 2>/dev/null | grep '^{' > $O/api_c2.json
python3 $R/tools/measure_api_pybench.py 2>/dev/null | grep '^{' > $O/api_pybench.jsonl
for c in "1000000 250" "65536 128"; do python3 $R/tools/measure_api_two_matrices.py $c 2>/dev/null | grep '^{' >> $O/api_two_matrices.jsonl; done
python3 $R/tools/run_vcf_scale.py --sites 200000 --samples 2500 2>/dev/null | tail -1 > $O/run_vcf_scale_200k_x_2500.json
python3 $R/tools/run_vcf_many_regions.py 2>/dev/null | tail -1 > $O/run_vcf_500_regions.json
for c in gzip bgzf; do python3 $R/tools/run_vcf_scale.py --sites 50000 --samples 2500 --compress $c 2>/dev/null | tail -1 >> $O/run_vcf_compressed_inputs.jsonl; done
python3 $R/tools/pybench_table.py $O/api_pybench.jsonl | tail -8
cat $O/api_c2.json; cat $O/api_two_matrices.jsonl | cut -c1-330
python3 -c "
import json
for f in ('run_vcf_scale_200k_x_2500','run_vcf_500_regions'):
    d=json.loads(open('$O/'+f+'.json').read()); print(f, d.get('run_vcf_wall_s', d.get('wall_s')), d.get('ms_per_region'))
for l in open('$O/run_vcf_compressed_inputs.jsonl'):
    d=json.loads(l); print(d.get('vcf_storage'), d.get('run_vcf_wall_s'))
"
