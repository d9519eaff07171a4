#!/bin/bash
# tools/profile_round.sh TAG [a|b] -- on the GPU box.
#   part a: the rocprofv3 kernel-trace summaries of the bench command (C2, C3, C3 in the bf16 storage configuration) and the two PMC passes
#           (HBM traffic) of each, written under gpurun_out/.  Copy gpurun_out/TAG_pmc_traffic_*.json to profiles/pmc_traffic_*.json afterwards:
#   part b: the default bench line (bench.py reports `roofline.traffic` only from files whose kernel-source hash matches the running sources),
#           then the per-shape table of the dense products.
# Everything runs on ONE stream while profiling (GRAPPA_HEAD_STREAMS=1, GRAPPA_WGRADS_ASIDE=0): per-kernel durations are then the kernels' own;
# GRAPPA_PLAN_TAILS=0 keeps the products' plans those of the shipped four-stream configuration (no split-K tail launches).
set -e
TAG=$1
PART=${2:-a}
R=$PWD
mkdir -p gpurun_out
if [ $PART = b ]; then
  python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
  head -c 700 gpurun_out/${TAG}_bench.json; echo
  python bench.py --no-cpu-baseline --no-extras --alt-precision "" --shape-table gpurun_out/${TAG}_gemm_shapes.txt > /dev/null 2> gpurun_out/${TAG}_shapes.err
  head -12 gpurun_out/${TAG}_gemm_shapes.txt
  exit 0
fi
export TMPDIR=/tmp GRAPPA_HEAD_STREAMS=1 GRAPPA_WGRADS_ASIDE=0 GRAPPA_PLAN_TAILS=0
for wl in c2 c3 c3_bf16; do
  W=""; S="--steps 5"; [ $wl = c3 ] && W="--workload C3-espaloma-b1024" && S="--steps 3 --warmup 1"; [ $wl = c3_bf16 ] && W="--workload C3-espaloma-b1024 --act-dtype bf16" && S="--steps 3 --warmup 1"
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof_$wl -- python3 $R/bench.py --no-cpu-baseline --no-extras --alt-precision "" $W $S > $R/gpurun_out/${TAG}_bench_${wl}_under_rocprof.json 2> $R/gpurun_out/${TAG}_rocprof_$wl.err
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_$wl/FETCH_SIZE -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --alt-precision "" $W > /dev/null 2> $R/gpurun_out/${TAG}_pmc_${wl}_fetch.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_$wl/WRITE_SIZE -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --alt-precision "" $W > /dev/null 2> $R/gpurun_out/${TAG}_pmc_${wl}_write.err
  cd $R
  cp $(find gpurun_out/${TAG}_prof_$wl -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_bench_${wl}_kernel_stats.csv
  python tools/pmc_traffic.py gpurun_out/${TAG}_pmc_$wl/FETCH_SIZE gpurun_out/${TAG}_pmc_$wl/WRITE_SIZE > gpurun_out/${TAG}_pmc_traffic_$wl.json
  rm -rf gpurun_out/${TAG}_pmc_$wl gpurun_out/${TAG}_prof_$wl        # raw traces are large; the summaries stay
  echo "$wl done"
done
head -c 400 gpurun_out/${TAG}_pmc_traffic_c2.json; echo; head -5 gpurun_out/${TAG}_bench_c2_kernel_stats.csv | cut -c1-200
