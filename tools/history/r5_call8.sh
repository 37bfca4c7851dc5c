#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --no-header -p no:cacheprovider -k "p8" > $out/r5h_pytest_p8.txt 2>&1
echo "pytest rc=$?" >> $out/r5h_pytest_p8.txt; tail -8 $out/r5h_pytest_p8.txt | cut -c1-300
timeout 300 python tools/bench_pipeline.py --utts 50000 --precisions f32x,bf16 --paths stream,sharded > $out/r5h_ark.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5h_ark.json'))
for k,v in d['runs'].items(): print(k, {a:b for a,b in v.items() if a in ('loop_utts_per_s','end_to_end_seconds','consumer_thread_seconds')})
PY
ASV_AMD_REPORT_TIMING=1 python - <<'PY' 2>&1 | grep -i "loop\[" 
import subprocess, sys, os
sys.path.insert(0, "tools")
import bench_pipeline as bp
files = bp.prepare("/tmp/asv_pipe_x", 50000)
for prec in ("bf16", "f32x"):
    env = dict(os.environ, ASV_AMD_PRECISION=prec, ASV_AMD_REPORT_TIMING="1")
    r = subprocess.run([sys.executable, bp.SCRIPT, "--nnet-config", files["cfg"], "--use-gpu", "true", "--gpu-id", "0", files["params"], "ark:" + files["ark"], "ark:/tmp/asv_pipe_x/o.ark"], capture_output=True, text=True, env=env)
    print(prec, r.stdout[-600:])
PY
