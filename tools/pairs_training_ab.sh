#!/bin/bash
# tools/pairs_training_ab.sh TAG -- on the GPU box: rocprofv3 kernel-trace summaries of the C2 train step with the pair format as the
# operands' storage format (GRAPPA_TRAINING_PAIRS=1, the default) and with fp32 operands (=0), one stream (per-kernel durations are the kernels' own)
set -e
TAG=$1
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp GRAPPA_HEAD_STREAMS=1 GRAPPA_WGRADS_ASIDE=0 GRAPPA_PLAN_TAILS=0
cd /tmp
for P in 1 0; do
  GRAPPA_TRAINING_PAIRS=$P rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof_p$P -- python3 $R/bench.py --no-cpu-baseline --no-extras --alt-precision "" --steps 5 > $R/gpurun_out/${TAG}_bench_p$P.json 2> $R/gpurun_out/${TAG}_rocprof_p$P.err
  cp $(find $R/gpurun_out/${TAG}_prof_p$P -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_kernel_stats_pairs$P.csv
  rm -rf $R/gpurun_out/${TAG}_prof_p$P
done
cd $R
head -c 300 gpurun_out/${TAG}_bench_p1.json; echo
