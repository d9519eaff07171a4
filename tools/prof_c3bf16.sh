#!/bin/bash
# GPU box: rocprofv3 kernel stats of the C3 train step in the bf16 storage configuration on ONE stream -> gpurun_out/${1}_c3bf16_kernel_stats.csv
set -e
TAG=${1:-r6}
R=$PWD
export TMPDIR=/tmp GRAPPA_HEAD_STREAMS=1 GRAPPA_WGRADS_ASIDE=0 GRAPPA_PLAN_TAILS=0
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -- python3 $R/bench.py --no-cpu-baseline --no-extras --alt-precision "" --workload C3-espaloma-b1024 --act-dtype bf16 --steps 3 --warmup 1 > $R/gpurun_out/${TAG}_c3bf16_under_rocprof.json 2> $R/gpurun_out/${TAG}_rocprof.err
cd $R
cp $(find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_c3bf16_kernel_stats.csv
rm -rf gpurun_out/${TAG}_prof
head -25 gpurun_out/${TAG}_c3bf16_kernel_stats.csv | cut -c1-220
