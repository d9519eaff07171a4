#!/bin/bash
# tools/timeline.sh TAG [env...] -- on the GPU box: kernel trace of the default (four-stream) C2 bench and of the recorded batch-32 step, summarised by tools/timeline_gaps.py
set -e
TAG=$1
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_tl -- python3 $R/bench.py --no-cpu-baseline --no-extras --alt-precision "" --steps 6 --warmup 3 > $R/gpurun_out/${TAG}_tl_bench.json 2> $R/gpurun_out/${TAG}_tl.err
cd $R
python tools/timeline_gaps.py $(find gpurun_out/${TAG}_tl -name "*kernel_trace.csv" | head -1) 0.5 > gpurun_out/${TAG}_timeline_c2.txt
rm -rf gpurun_out/${TAG}_tl
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_tl32 -- python3 $R/tools/recorded_profile.py train 32 > $R/gpurun_out/${TAG}_tl32.txt 2> $R/gpurun_out/${TAG}_tl32.err
cd $R
python tools/timeline_gaps.py $(find gpurun_out/${TAG}_tl32 -name "*kernel_trace.csv" | head -1) 0.6 > gpurun_out/${TAG}_timeline_b32_recorded.txt
rm -rf gpurun_out/${TAG}_tl32
cat gpurun_out/${TAG}_timeline_c2.txt gpurun_out/${TAG}_timeline_b32_recorded.txt
