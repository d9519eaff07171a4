set -e
R=$PWD
export TMPDIR=/tmp GRAPPA_HEAD_STREAMS=1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3d_prof -- python3 $R/bench.py --no-cpu-baseline --no-extras --alt-precision "" --bwd-precision "" --steps 5 > $R/gpurun_out/r3d_bench_under_rocprof.json 2> $R/gpurun_out/r3d_rocprof.err
cd $R
cp $(find gpurun_out/r3d_prof -name "*kernel_stats.csv" | head -1) gpurun_out/r3d_kernel_stats.csv
rm -rf gpurun_out/r3d_prof
