#!/bin/bash
# tools/predict_kernel_stats.sh TAG -- on the GPU box: kernel-trace summary of recorded `Grappa.predict` calls on one 40-atom molecule
set -e
TAG=$1
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof_predict -- python3 $R/tools/recorded_profile.py predict > $R/gpurun_out/${TAG}_predict_recorded.txt 2> $R/gpurun_out/${TAG}_predict_recorded.err
cp $(find $R/gpurun_out/${TAG}_prof_predict -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_predict_kernel_stats.csv
rm -rf $R/gpurun_out/${TAG}_prof_predict
cd $R
python tools/predict_host_profile.py 2>&1 | grep "predict:"
