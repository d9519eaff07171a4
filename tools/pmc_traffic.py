"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py` into per-kernel-family HBM traffic per launch.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc/bench_FETCH_SIZE -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --alt-precision '' --bwd-precision ''
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc/bench_WRITE_SIZE -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --alt-precision '' --bwd-precision ''
    python tools/pmc_traffic.py gpurun_out/pmc/bench_FETCH_SIZE gpurun_out/pmc/bench_WRITE_SIZE > gpurun_out/pmc_traffic_c2.json

Units and corrections (MI355X_MICROARCH.md, HBM section): the counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of the
bytes of wide coalesced (16 B/lane) streaming reads, so it is doubled; WRITE_SIZE is exact for 16 B/lane stores.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# "dense_products": every kernel a dense product launches (main, grouped, split-K reduction, row-maxima passes and combines): bench.py
# divides its bytes PER STEP by the product calls per step, the unit of roofline.achieved / algorithmic_bytes_per_launch
# (round 5: every kernel whose name carries `gemm_` -- rounds 3 and 4 listed the kernels by name and left the pair-format kernels, gemm_pairs_kernel
# / gemm_wpairs_kernel, out of the sum: their "1.09 x algorithmic" undercounted)
FAMILIES = {"dense_products": r"gemm_|amax_|group_upload|group_index|split_pairs|split_planes",
            "adam": r"adam_kernel",
            "gemm_f32": r"gemm_f32_kernel|gemm_bf16x_kernel", "gemm_bf16x": r"gemm_bf16x_kernel", "gemm_pairs_il": r"gemm_pairs_il", "gemm_wpairs_il": r"gemm_wpairs_il",
            "gemm_wgrad_grouped": r"gemm_bf16x_grouped_kernel", "gemm_splitk_reduce": r"gemm_splitk_reduce_kernel", "gat_fwd": r"gat_fwd_kernel",
            "gat_bwd": r"gat_bwd_kernel|gat_delta_kernel", "layernorm": r"layernorm_", "seqattn": r"seqattn_",
            # round 6: the fused writer-head layer (csrc/writer_layer.hip) and, for whole-step comparisons, every kernel of the step
            "writer_layer_fwd": r"writer_layer_fwd", "writer_layer_bwd": r"writer_layer_bwd", "writer_layer": r"writer_layer_|writer_pack", "whole_step": r"."}


def collect(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    tot, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        for fam, pat in FAMILIES.items():
            if re.search(pat, r["Kernel_Name"]):
                tot[fam] += float(r["Counter_Value"])
                cnt[fam] += 1
    return tot, cnt


def main():
    fdir, wdir = sys.argv[1], sys.argv[2]
    ft, fc = collect(fdir, "FETCH_SIZE")
    wt, wc = collect(wdir, "WRITE_SIZE")
    from bench import kernel_source_hash
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py --steps 2 --warmup 1",
           "kernel_source_hash": kernel_source_hash(),
           "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024", "families": {}}
    steps = fc["adam"]                      # one optimiser step per train step
    out["steps"] = steps
    for fam in FAMILIES:
        if fc[fam] == 0 or fam == "adam":
            continue
        n = fc[fam]
        fetch_b, write_b = 2.0 * ft[fam] * 1024.0, wt[fam] * 1024.0 * (n / max(wc[fam], 1))
        out["families"][fam] = {"launches": n, "fetch_bytes_per_launch": fetch_b / n, "write_bytes_per_launch": write_b / n,
                                "hbm_bytes_per_launch": (fetch_b + write_b) / n,
                                "hbm_bytes_per_step": (fetch_b + write_b) / max(steps, 1)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
