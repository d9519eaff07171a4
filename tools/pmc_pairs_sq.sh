# SQ counters of the fp32-operand fp16-split kernel beside the pair-format kernel on the workload's shapes (tools/gemm_pairs_check.py
# --timing-only): two rocprofv3 --pmc passes -> gpurun_out/<tag>_pmc_pairs_sq_summary.txt (per kernel and grid: counter means, the share of
# the matrix pipe SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs), vector instructions per wavefront)
#   bash tools/pmc_pairs_sq.sh <tag>
set -e
TAG=${1:-r3}
R=$PWD
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/${TAG}_psq1 -- python3 $R/tools/gemm_pairs_check.py --timing-only > $R/gpurun_out/${TAG}_psq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES --output-format csv -d $R/gpurun_out/${TAG}_psq2 -- python3 $R/tools/gemm_pairs_check.py --timing-only > $R/gpurun_out/${TAG}_psq2.log 2>&1
cd $R
mkdir -p gpurun_out/${TAG}_psq && cp -r gpurun_out/${TAG}_psq1 gpurun_out/${TAG}_psq2 gpurun_out/${TAG}_psq/
python3 - gpurun_out/${TAG}_psq > gpurun_out/${TAG}_pmc_pairs_sq_summary.txt <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        name = row["Kernel_Name"]
        if "gemm_pairs_kernel" not in name and "gemm_bf16x_kernel" not in name:
            continue
        m = re.search(r"(\w+_kernel)<([^>]*)>", name)
        acc[f"{m.group(1)}<{m.group(2)}> grid={row['Grid_Size']} wg={row['Workgroup_Size']}"][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("# product kernels of tools/gemm_pairs_check.py --timing-only (split = gemm_bf16x_kernel<512, 103, ...>, pairs = gemm_pairs_kernel<BN>)")
print("# mfma_pipe_share = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); valu_per_wave = SQ_INSTS_VALU / SQ_WAVES")
for k in sorted(acc):
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    share = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(c.get("GRBM_GUI_ACTIVE", 1) / 8 * 1024, 1)
    vpw = c.get("SQ_INSTS_VALU", 0) / max(c.get("SQ_WAVES", 1), 1)
    print(f"{k}\n    launches {len(next(iter(acc[k].values())))}  mfma_pipe_share {share:.3f}  valu_per_wave {vpw:.0f}  " +
          "  ".join(f"{n}={v:.4g}" for n, v in sorted(c.items())))
PY
rm -rf gpurun_out/${TAG}_psq gpurun_out/${TAG}_psq1 gpurun_out/${TAG}_psq2
cat gpurun_out/${TAG}_pmc_pairs_sq_summary.txt | cut -c1-260
