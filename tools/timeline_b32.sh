#!/bin/bash
# tools/timeline_b32.sh TAG -- on the GPU box: kernel trace of the recorded batch-32 train step summarised by tools/timeline_gaps.py
set -e
TAG=$1
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_tl32 -- python3 $R/tools/recorded_profile.py train 32 > $R/gpurun_out/${TAG}_tl32.txt 2> $R/gpurun_out/${TAG}_tl32.err
cd $R
python tools/timeline_gaps.py $(find gpurun_out/${TAG}_tl32 -name "*kernel_trace.csv" | head -1) 0.6 > gpurun_out/${TAG}_timeline_b32_recorded.txt
rm -rf gpurun_out/${TAG}_tl32
cat gpurun_out/${TAG}_timeline_b32_recorded.txt
