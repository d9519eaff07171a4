#!/bin/bash
# on the GPU box: kernel summary of the C5 inference forward (tools/c5_profile.py), four head streams and one
set -e
R=$PWD
mkdir -p gpurun_out
python tools/c5_profile.py 10 > gpurun_out/r6_c5_wall.txt 2>&1
GRAPPA_HEAD_STREAMS=1 python tools/c5_profile.py 10 >> gpurun_out/r6_c5_wall.txt 2>&1
cat gpurun_out/r6_c5_wall.txt | grep "C5 forward"
export TMPDIR=/tmp GRAPPA_HEAD_STREAMS=1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5_prof -- python3 $R/tools/c5_profile.py 10 > $R/gpurun_out/r6_c5_under_rocprof.txt 2>&1
cd $R
cp $(find /tmp/c5_prof -name "*kernel_stats.csv" | head -1) gpurun_out/r6_c5_kernel_stats.csv
python tools/kstats.py gpurun_out/r6_c5_kernel_stats.csv | head -32
