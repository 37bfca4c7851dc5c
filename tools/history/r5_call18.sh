#!/bin/bash
# res2_chain_kernel window forms (6 / 7 row fragments): parity tests, then model-level A/B on ECAPA bf16
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
tag=${1:-r5t}
cd $root
timeout 600 python -m pytest tests/test_gpu_ecapa.py -m gpu -q --no-header -p no:cacheprovider -x > $out/${tag}_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/${tag}_pytest.txt; tail -15 $out/${tag}_pytest.txt | cut -c1-400
if ! grep -q "pytest rc=0" $out/${tag}_pytest.txt; then echo "tests failed: stopping"; exit 0; fi
cd /tmp && export TMPDIR=/tmp
one="--streams 1 --cpu-seconds 0 --no-supplementary --eer-trials 0 --no-traffic --gate-seeds 0"
two="--cpu-seconds 0 --no-supplementary --eer-trials 0 --no-traffic --gate-seeds 0 --no-profile"
for rep in 1 2; do
for fr in 0 6 7; do
  for m in "ecapa bf16"; do
    set -- $m
    ASV_AMD_RES2_FR=$fr timeout 300 python $root/bench.py --model $1 --precision $2 $one 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{})
print('one stream  RES2_FR=$fr $1 $2 value', d['value'], 'ms/step', d['ms_per_step'], 'gemm frac', r.get('frac'))"
    ASV_AMD_RES2_FR=$fr timeout 300 python $root/bench.py --model $1 --precision $2 $two 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('two streams RES2_FR=$fr $1 $2 value', d['value'], 'ms/step', d['ms_per_step'])"
  done
done
done | tee $out/${tag}_res2_model_ab.txt
for fr in 6 7; do
ASV_AMD_RES2_FR=$fr timeout 300 python $root/bench.py --model ecapa --precision bf16 $one --per-op 2>$out/${tag}_ecapa_perop_fr$fr.txt >/dev/null
grep "res2" $out/${tag}_ecapa_perop_fr$fr.txt
ASV_AMD_RES2_FR=$fr ASV_AMD_RES2_DBG=1 timeout 300 python $root/bench.py --model ecapa --precision bf16 $one --steps 3 --warmup 1 --no-profile 2>&1 >/dev/null | grep "res2 dbg" | tail -3 > $out/${tag}_res2_dbg_fr$fr.txt
cat $out/${tag}_res2_dbg_fr$fr.txt | cut -c1-600
done
