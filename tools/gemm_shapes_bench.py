"""Per-shape timing of the GEMM on the GPU box: every distinct (M,N,K,layout) the production model issues in one train step of the C2
workload, timed with HIP events (10 repetitions after 2 warm-ups).  Prints TFLOP/s per shape and the step-weighted total.
    python tools/gemm_shapes_bench.py --precisions=f32_bf16x6,f32            # arithmetic modes side by side, one process
    python tools/gemm_shapes_bench.py --precisions=f32_bf16x6,f32_bf16x6@5   # "@cfg": force a tile configuration (plan override)
    GRAPPA_HIP_LIB=/path/to/variant.so python tools/gemm_shapes_bench.py ... # A/B a kernel variant built into another library"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


DEV = "cuda"


def record_shapes():
    """run one train step with a recording wrapper around the backend's gemm"""
    import golden_utils as gu
    from grappa_amd import Energy, MolwiseLoss, get_default_model_config, model_from_config
    from grappa_amd.backend import get_backend
    from grappa_amd.datasets import build_workload
    be = get_backend()
    shapes = collections.Counter()
    orig = be.gemm

    def rec(a, b, out, *, M, N, K, a_kcontig=True, b_kcontig=True, **kw):
        shapes[(M, N, K, int(a_kcontig), int(b_kcontig), (a.stride(0) if a.dim() == 2 and a.shape[0] > 1 else 0) if a is not None else K,
                b.stride(0) if b.shape[0] > 1 else 0, out.stride(0) if out.shape[0] > 1 else 0)] += 1
        return orig(a, b, out, M=M, N=N, K=K, a_kcontig=a_kcontig, b_kcontig=b_kcontig, **kw)

    be.gemm = rec
    model = model_from_config(get_default_model_config())
    model.load_state_dict(gu.keyed_state_dict(model))
    model = model.to(DEV).train()
    n_mols = int(os.environ.get("GRAPPA_TUNE_MOLECULES", "0"))
    if n_mols:                               # the first molecules of the workload only (e.g. 32: the reference's own batch size)
        from grappa_amd.datasets import build_batch_from_pool, workload_molecule_ids
        g = build_batch_from_pool(workload_molecule_ids(WORKLOAD, seed=0)[:n_mols], n_confs=32, seed=0).to(DEV)
    else:
        g = build_workload(WORKLOAD, seed=0).to(DEV)
    loss = MolwiseLoss(param_weight=0.0)(Energy()(model(g)))
    loss.backward()
    if DEV == "cuda":
        torch.cuda.synchronize()
    be.gemm = orig
    return shapes


WORKLOAD = "C2-pubchem-b256"


def main():
    global DEV, WORKLOAD
    from grappa_amd.backend import get_backend
    shapes = record_shapes()
    be = get_backend()
    # --precisions f32,f32_bf16x6,...: A/B the arithmetic modes in ONE process (same clocks, same buffers)
    modes = ["f32"]
    for a in sys.argv:
        if a.startswith("--precisions="):
            modes = a.split("=", 1)[1].split(",")
    rows = []
    for (M, N, K, ak, bk, lda, ldb, ldc), cnt in sorted(shapes.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][2] * kv[1]):
        A = torch.randn((M, K) if ak else (K, M), device="cuda")
        B = torch.randn((N, K) if bk else (K, N), device="cuda")
        C = torch.empty((M, N), device="cuda")
        ms = []
        for mode_cfg in modes:
            mode, _, cfg = mode_cfg.partition("@")       # "f32_bf16x6@7": force tile configuration 7 (plan override)
            be.plan_override = (int(cfg), 0, -1) if cfg else None
            for _ in range(2):
                be.gemm(A, B, C, M=M, N=N, K=K, a_kcontig=bool(ak), b_kcontig=bool(bk), precision=mode)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                be.gemm(A, B, C, M=M, N=N, K=K, a_kcontig=bool(ak), b_kcontig=bool(bk), precision=mode)
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1) / 10)
            be.plan_override = None
        rows.append((M, N, K, ak, bk, cnt, ms, 2.0 * M * N * K))
    print(f"{'M':>7} {'N':>5} {'K':>7} ak bk  cnt " + " ".join(f"{m + ' ms':>14} {'TF':>6}" for m in modes))
    for M, N, K, ak, bk, cnt, ms, fl in sorted(rows, key=lambda r: -r[5] * r[6][0]):
        print(f"{M:7d} {N:5d} {K:7d}  {ak}  {bk} {cnt:4d} " + " ".join(f"{t:14.3f} {fl / t / 1e9:6.1f}" for t in ms))
    tot_fl = sum(r[5] * r[7] for r in rows)
    for i, m in enumerate(modes):
        tot_ms = sum(r[5] * r[6][i] for r in rows)
        print(f"total[{m}]: {tot_ms:.1f} ms/step, {tot_fl / 1e12:.2f} TFLOP/step, {tot_fl / tot_ms / 1e9:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
