#!/bin/bash
# tools/recorded_profile.sh TAG -- on the GPU box: kernel-trace summary of the RECORDED batch-32 train step and the phase split of a recorded predict
set -e
TAG=$1
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof_b32 -- python3 $R/tools/recorded_profile.py train 32 > $R/gpurun_out/${TAG}_b32_recorded.txt 2> $R/gpurun_out/${TAG}_b32_recorded.err
cp $(find $R/gpurun_out/${TAG}_prof_b32 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_b32_recorded_kernel_stats.csv
rm -rf $R/gpurun_out/${TAG}_prof_b32
cd $R
python tools/recorded_profile.py train 32 >> gpurun_out/${TAG}_b32_recorded.txt 2>&1
python tools/recorded_profile.py predict > gpurun_out/${TAG}_predict_recorded.txt 2>&1
cat gpurun_out/${TAG}_b32_recorded.txt gpurun_out/${TAG}_predict_recorded.txt
