set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python $root/bench.py --model ecapa --cpu-seconds 0 --per-op > $out/r2a_ecapa.json 2> $out/r2a_ecapa_perop.txt
python $root/bench.py --model resnet --cpu-seconds 0 --per-op > $out/r2a_resnet.json 2> $out/r2a_resnet_perop.txt
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/r2a_kt_ecapa -- python $root/bench.py --model ecapa --cpu-seconds 0 --no-profile > /dev/null 2>&1
cp $out/r2a_kt_ecapa/*/*kernel_stats.csv $out/r2a_ecapa_kernel_stats.csv
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/r2a_kt_resnet -- python $root/bench.py --model resnet --cpu-seconds 0 --no-profile > /dev/null 2>&1
cp $out/r2a_kt_resnet/*/*kernel_stats.csv $out/r2a_resnet_kernel_stats.csv
rm -rf $out/r2a_kt_ecapa $out/r2a_kt_resnet
tail -1 $out/r2a_ecapa.json | cut -c1-400; tail -1 $out/r2a_resnet.json | cut -c1-400
