#!/bin/bash
# tools/profile_round.sh TAG -- on the GPU box: the rocprofv3 kernel-trace summary of the bench command, the two PMC passes (HBM
# traffic) and then the default bench line, written under gpurun_out/ (copy what is to be judged into profiles/).
# Everything runs on ONE stream while profiling (GRAPPA_HEAD_STREAMS=1, GRAPPA_WGRADS_ASIDE=0): per-kernel durations are then the kernels' own;
# GRAPPA_PLAN_TAILS=0 keeps the products' plans those of the shipped four-stream configuration (no split-K tail launches).
set -e
TAG=$1
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp GRAPPA_HEAD_STREAMS=1 GRAPPA_WGRADS_ASIDE=0 GRAPPA_PLAN_TAILS=0
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -- python3 $R/bench.py --no-cpu-baseline --no-extras --alt-precision "" --steps 5 > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/${TAG}_rocprof.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc/FETCH_SIZE -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --alt-precision "" > /dev/null 2> $R/gpurun_out/${TAG}_pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc/WRITE_SIZE -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --alt-precision "" > /dev/null 2> $R/gpurun_out/${TAG}_pmc_write.err
cd $R
python tools/pmc_traffic.py gpurun_out/${TAG}_pmc/FETCH_SIZE gpurun_out/${TAG}_pmc/WRITE_SIZE > gpurun_out/${TAG}_pmc_traffic_c2.json
# the default bench line last, with the traffic file of THIS build in place (bench.py reports `roofline.traffic` only when the file's
# kernel-source hash matches the running sources)
cp gpurun_out/${TAG}_pmc_traffic_c2.json profiles/pmc_traffic_c2.json
( unset GRAPPA_HEAD_STREAMS GRAPPA_WGRADS_ASIDE GRAPPA_PLAN_TAILS; python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err )
cp $(find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_bench_c2_kernel_stats.csv
# BASELINE configs[2] in the bf16 storage configuration: kernel-trace summary of the same train step
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof_c3bf16 -- python3 $R/bench.py --no-cpu-baseline --no-extras --alt-precision "" --workload C3-espaloma-b1024 --act-dtype bf16 --steps 3 --warmup 1 > $R/gpurun_out/${TAG}_bench_c3_bf16_under_rocprof.json 2> $R/gpurun_out/${TAG}_rocprof_c3bf16.err
cd $R
cp $(find gpurun_out/${TAG}_prof_c3bf16 -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_bench_c3_bf16_kernel_stats.csv
# BASELINE configs[2]'s batch in the headline (fp32-grade) arithmetic: kernel-trace summary + the PMC passes for the GAT kernels' HBM traffic
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof_c3 -- python3 $R/bench.py --no-cpu-baseline --no-extras --alt-precision "" --workload C3-espaloma-b1024 --steps 3 --warmup 1 > $R/gpurun_out/${TAG}_bench_c3_under_rocprof.json 2> $R/gpurun_out/${TAG}_rocprof_c3.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_c3/FETCH_SIZE -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --alt-precision "" --workload C3-espaloma-b1024 > /dev/null 2> $R/gpurun_out/${TAG}_pmc_c3_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_c3/WRITE_SIZE -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --alt-precision "" --workload C3-espaloma-b1024 > /dev/null 2> $R/gpurun_out/${TAG}_pmc_c3_write.err
cd $R
cp $(find gpurun_out/${TAG}_prof_c3 -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_bench_c3_kernel_stats.csv
python tools/pmc_traffic.py gpurun_out/${TAG}_pmc_c3/FETCH_SIZE gpurun_out/${TAG}_pmc_c3/WRITE_SIZE > gpurun_out/${TAG}_pmc_traffic_c3.json
rm -rf gpurun_out/${TAG}_pmc gpurun_out/${TAG}_prof gpurun_out/${TAG}_prof_c3bf16 gpurun_out/${TAG}_prof_c3 gpurun_out/${TAG}_pmc_c3        # raw traces are large; the summaries stay
head -c 600 gpurun_out/${TAG}_bench.json; echo; head -5 gpurun_out/${TAG}_bench_c2_kernel_stats.csv | cut -c1-200
