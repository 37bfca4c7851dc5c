#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
t0=$(date +%s)
python $root/bench.py > $out/r5j_bench.json 2> $out/r5j_bench.err
echo "bench.py wall seconds: $(( $(date +%s) - t0 ))"
tail -3 $out/r5j_bench.err | cut -c1-300
python - <<PY
import json
d=json.loads(open('$out/r5j_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','value_single_stream','value_at_b640') if k in d})
print('roofline', {k:v for k,v in d['roofline'].items() if k not in ('kernel','traffic_source','per_launch')})
print('per_launch', d['roofline'].get('per_launch'))
print('b640', d.get('roofline_at_b640'))
print('parity_grade', d.get('value_parity_grade'))
for k,v in d.get('supplementary',{}).items():
    if k=='ark_to_ark': print(k, json.dumps(v)[:900])
    else: print(k, {a:v.get(a) for a in ('value','frac','dominant_kernel','dominant_tflops','dominant_frac','error')})
print('cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('cores'))
PY
