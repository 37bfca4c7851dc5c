#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py tests/test_gpu_rccl.py -m gpu -q --no-header -p no:cacheprovider -x -k "p8 or pipeline or rccl or script or sharded" > $out/r5x_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r5x_pytest.txt; tail -6 $out/r5x_pytest.txt | cut -c1-300
P=$root/asv-subtools_amd/tools_p8_probe
( for shape in "52224 80 512 5 20 5" "77824 80 1024 5 10 5"; do timeout 120 $P $shape || echo "probe rc=$?"; done ) > $out/r5x_p8_tail_probe.txt 2>&1
grep -v "0.0 us\|^$" $out/r5x_p8_tail_probe.txt | cut -c1-200
for rep in 1 2; do
timeout 300 python tools/bench_pipeline.py --utts 50000 --precisions f32x,bf16 --paths stream,sharded > $out/r5x_ark_$rep.json 2>$out/r5x_ark_$rep.err
python - <<PY
import json
d=json.load(open("$out/r5x_ark_$rep.json"))
for k,v in d["runs"].items(): print(k, v.get("loop_utts_per_s"), v.get("end_to_end_seconds"), v.get("error","")[:300])
PY
done
