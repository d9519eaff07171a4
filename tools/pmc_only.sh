#!/bin/bash
# GPU box: only the two PMC passes (HBM traffic) of the C2 and C3 train steps; settings as in tools/profile_round.sh
set -e
TAG=$1
R=$PWD
export TMPDIR=/tmp GRAPPA_HEAD_STREAMS=1 GRAPPA_WGRADS_ASIDE=0 GRAPPA_PLAN_TAILS=0
for wl in ${WORKLOADS:-c2 c3 c3_bf16}; do
  W=""; [ $wl = c3 ] && W="--workload C3-espaloma-b1024"; [ $wl = c3_bf16 ] && W="--workload C3-espaloma-b1024 --act-dtype bf16"
  cd /tmp
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_$wl/FETCH_SIZE -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --alt-precision "" $W > /dev/null 2> $R/gpurun_out/${TAG}_pmc_${wl}_fetch.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_$wl/WRITE_SIZE -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --alt-precision "" $W > /dev/null 2> $R/gpurun_out/${TAG}_pmc_${wl}_write.err
  cd $R
  python tools/pmc_traffic.py gpurun_out/${TAG}_pmc_$wl/FETCH_SIZE gpurun_out/${TAG}_pmc_$wl/WRITE_SIZE > gpurun_out/${TAG}_pmc_traffic_$wl.json
  rm -rf gpurun_out/${TAG}_pmc_$wl
done
head -c 400 gpurun_out/${TAG}_pmc_traffic_c2.json
