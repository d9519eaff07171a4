set -e
R=$PWD
export TMPDIR=/tmp GRAPPA_HEAD_STREAMS=1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3h_prof -- python3 $R/bench.py --no-cpu-baseline --no-extras --alt-precision "" --workload C3-espaloma-b1024 --act-dtype bf16 --steps 3 --warmup 1 > $R/gpurun_out/r3h_bench_c3bf16.json 2> $R/gpurun_out/r3h_rocprof.err
cd $R
cp $(find gpurun_out/r3h_prof -name "*kernel_stats.csv" | head -1) gpurun_out/r3h_c3bf16_kernel_stats.csv
rm -rf gpurun_out/r3h_prof
