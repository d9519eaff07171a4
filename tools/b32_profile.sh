#!/bin/bash
# on the GPU box: kernel summary of the recorded b32 train step (tools/b32_graph_profile.py)
set -e
R=$PWD
mkdir -p gpurun_out
python tools/b32_graph_profile.py 200 > gpurun_out/r6_b32_wall.txt 2>&1
cat gpurun_out/r6_b32_wall.txt | tail -2
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/b32_prof -- python3 $R/tools/b32_graph_profile.py 100 > $R/gpurun_out/r6_b32_under_rocprof.txt 2>&1
cd $R
cp $(find /tmp/b32_prof -name "*kernel_stats.csv" | head -1) gpurun_out/r6_b32_kernel_stats.csv
python tools/kstats.py gpurun_out/r6_b32_kernel_stats.csv | head -30
