cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /root/repo/gpurun_out/fbp1 -o p -- python /root/repo/tools/bench_fbank.py --iters 20 --cmn 0 >/dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d /root/repo/gpurun_out/fbp2 -o p -- python /root/repo/tools/bench_fbank.py --iters 20 --cmn 0 >/dev/null 2>&1
python - <<'PY'
import csv, collections
for d in ('fbp1','fbp2'):
    acc=collections.defaultdict(list)
    try:
        for r in csv.DictReader(open('/root/repo/gpurun_out/%s/p_counter_collection.csv'%d)):
            if 'fbank512' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    except Exception as e: print(d, e)
    for k,v in acc.items(): print(d, k, sum(v)/len(v), len(v))
PY
