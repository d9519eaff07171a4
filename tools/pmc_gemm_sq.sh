# SQ counters of the dense-product kernels on the workload's shapes (tools/gemm_f16x3_check.py --bench-only), two rocprofv3 --pmc
# passes, summarised by tools/pmc_summary.py -> gpurun_out/<tag>_pmc_gemm_sq_summary.txt
#   bash tools/pmc_gemm_sq.sh <tag>
set -e
TAG=${1:-r2y}
R=$PWD
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/${TAG}_sq1 -- python3 $R/tools/gemm_f16x3_check.py --bench-only > $R/gpurun_out/${TAG}_sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/${TAG}_sq2 -- python3 $R/tools/gemm_f16x3_check.py --bench-only > $R/gpurun_out/${TAG}_sq2.log 2>&1
cd $R
mkdir -p gpurun_out/${TAG}_sq && cp -r gpurun_out/${TAG}_sq1 gpurun_out/${TAG}_sq2 gpurun_out/${TAG}_sq/ 
python3 tools/pmc_summary.py gpurun_out/${TAG}_sq gemm_bf16x > gpurun_out/${TAG}_pmc_gemm_sq_summary.txt
rm -rf gpurun_out/${TAG}_sq gpurun_out/${TAG}_sq1 gpurun_out/${TAG}_sq2
