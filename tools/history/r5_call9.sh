#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --no-header -p no:cacheprovider -k "p8" > $out/r5i_pytest_p8.txt 2>&1
echo "pytest rc=$?" >> $out/r5i_pytest_p8.txt; tail -4 $out/r5i_pytest_p8.txt | cut -c1-300
python - <<'PY' 2>&1 | grep -i "loop\[\|Extracted\|rror" 
import subprocess, sys, os
sys.path.insert(0, "tools")
import bench_pipeline as bp
files = bp.prepare("/tmp/asv_pipe_x", 50000)
for prec in ("bf16", "f32x"):
  for extra, rspec in ((["--sharded", "true"], "scp:" + files["scp"]), ([], "scp:" + files["scp"])):
    env = dict(os.environ, ASV_AMD_PRECISION=prec, ASV_AMD_REPORT_TIMING="1")
    r = subprocess.run([sys.executable, bp.SCRIPT, "--nnet-config", files["cfg"], "--use-gpu", "true", "--gpu-id", "0"] + extra + [files["params"], rspec, "ark:/tmp/asv_pipe_x/o.ark"], capture_output=True, text=True, env=env)
    print(prec, extra, r.stdout[-700:], r.stderr[-300:] if r.returncode else "")
PY
