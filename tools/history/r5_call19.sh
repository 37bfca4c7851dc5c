#!/bin/bash
# sharded path in segments + header table: GPU pipeline tests, then ark -> ark through the script (stream / sharded, f32x / bf16), twice
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
tag=${1:-r5u}
cd $root
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_rccl.py -m gpu -q --no-header -p no:cacheprovider -x > $out/${tag}_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/${tag}_pytest.txt; tail -8 $out/${tag}_pytest.txt | cut -c1-400
if ! grep -q "pytest rc=0" $out/${tag}_pytest.txt; then echo "tests failed: stopping"; exit 0; fi
for rep in 1 2; do
timeout 300 python tools/bench_pipeline.py --utts 50000 --precisions f32x,bf16 --paths stream,sharded > $out/${tag}_ark_$rep.json 2>$out/${tag}_ark_$rep.err
python - <<PY
import json
d=json.load(open("$out/${tag}_ark_$rep.json"))
for k,v in d["runs"].items(): print(k, v.get("loop_utts_per_s"), v.get("end_to_end_seconds"), v.get("error","")[:300])
PY
done
ASV_AMD_SHARD_SEGMENT=0 timeout 300 python tools/bench_pipeline.py --utts 50000 --precisions f32x,bf16 --paths sharded > $out/${tag}_ark_one_gather.json 2>/dev/null
python - <<PY
import json
d=json.load(open("$out/${tag}_ark_one_gather.json"))
for k,v in d["runs"].items(): print("one gather:", k, v.get("loop_utts_per_s"), v.get("end_to_end_seconds"), v.get("error","")[:300])
PY
