#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 900 python -m pytest tests/test_gpu_xvector.py tests/test_gpu_devlib.py tests/test_gpu_full_size_parity.py tests/test_gpu_end_to_end.py -q --no-header -p no:cacheprovider > $out/r4f_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r4f_pytest.txt
tail -5 $out/r4f_pytest.txt | cut -c1-300
one="--streams 1 --cpu-seconds 0 --no-supplementary --eer-trials 0 --min-seconds 1.0"
: > $out/r4f_chain_tail_ab.txt
for i in 1 2 3; do
  for tail in 1 0; do
    line=$(ASV_AMD_CHAIN_TAIL=$tail python bench.py $one 2>/dev/null | tail -1)
    python - "$tail" "$line" >> $out/r4f_chain_tail_ab.txt <<'PY'
import json, sys
tail, line = sys.argv[1:3]
d = json.loads(line)
r = d.get("roofline", {})
print("ASV_AMD_CHAIN_TAIL=%s  b256 single stream %10.1f utt/s  gemm frac %s  per_launch us %s" % (tail, d["value"], r.get("frac"), [p.get("us") for p in r.get("per_launch", [])]))
PY
  done
done
for i in 1 2; do
  for tail in 1 0; do
    ASV_AMD_CHAIN_TAIL=$tail python bench.py --cpu-seconds 0 --no-supplementary --eer-trials 0 --no-profile --min-seconds 1.0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ASV_AMD_CHAIN_TAIL=$tail  b256 TWO streams', d['value'])" >> $out/r4f_chain_tail_ab.txt
  done
done
cat $out/r4f_chain_tail_ab.txt
for prec in f32x; do
  for st in 1 2; do
    python bench.py --precision $prec --streams $st --cpu-seconds 0 --no-supplementary --eer-trials 0 --no-profile 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('f32x b256 streams', $st, d['value'])" | tee -a $out/r4f_f32x_streams.txt
  done
done
