set -eo pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02m; mkdir -p $O
python3 -m pytest $R/tests/test_gpu_run_vcf.py -x -q 2>&1 | tail -3
python3 $R/tools/run_vcf_scale.py --sites 200000 --samples 2500 2>$O/rv.err | tail -1 > $O/run_vcf_scale_200k_x_2500.json
python3 $R/tools/run_vcf_many_regions.py 2>/dev/null | tail -1 > $O/run_vcf_500_regions.json
python3 -c "
import json;d=json.load(open('$O/run_vcf_scale_200k_x_2500.json'));print(d['run_vcf_wall_s'],d['all_match']);print('\n'.join(l for l in d['run_vcf_timing'] if 'parse_block' not in l))
d=json.load(open('$O/run_vcf_500_regions.json'));print({k:v for k,v in d.items() if not isinstance(v,(list,dict))})"
