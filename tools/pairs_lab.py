"""GPU: the pair-format GEMM under whatever GRAPPA_PAIRS_* environment the process was started with -- correctness of the multi-tile
(persistent) walk against the fp32-operand fp16-split kernel (bit for bit) on shapes with several tiles per workgroup, every straight-line
epilogue class, ragged edges and short K; then the timing of the C2 / C3 product shapes (pairs kernel only).
    GRAPPA_PAIRS_PERSIST=0 python tools/pairs_lab.py --tag base ; GRAPPA_PAIRS_STAGGER=0 python tools/pairs_lab.py --tag persist ..."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gemm_pairs_check as g  # noqa: E402

dev = "cuda"
WP = "--wpairs" in sys.argv or os.environ.get("LAB_MODE") == "wpairs"          # A = fp32 rows + weight pairs (gemm_wpairs_il.hip) instead of both operands in pairs
SPLIT = "--split" in sys.argv or os.environ.get("LAB_MODE") == "split"        # timing only: the fp32-operand fp16-split kernel (gemm_bf16x_impl.h MODE H3) on the same shapes


def check(M, N, K, gen, dgrad=False, **epi):
    A = torch.randn((M, K), generator=gen, device=dev)
    W = torch.randn((K, N) if dgrad else (N, K), generator=gen, device=dev) * 0.05
    am_a, am_b = g.amax(A), g.amax(W, rows=not dgrad)
    kw = {}
    if "bias" in epi:
        kw["bias"] = torch.randn(N, generator=gen, device=dev)
    if epi.get("act"):
        kw["act"] = 1
    if "res" in epi:
        kw["res"] = torch.randn((M, N), generator=gen, device=dev)
    if "aux" in epi:
        kw["aux"] = torch.randn((M, N), generator=gen, device=dev)
    if "drop" in epi:
        kw["drop_p"], kw["drop_seed"] = 0.3, 4321
    ap, bp = g.split_pairs(A, am_a), g.split_pairs(W, am_b, transpose=dgrad)
    o_pairs = torch.full((M, N), float("nan"), device=dev)
    o_split = torch.full((M, N), float("nan"), device=dev)
    am_p = am_s = None
    if epi.get("amax"):
        am_p = torch.zeros(M, dtype=torch.int32, device=dev)
        am_s = torch.zeros(M, dtype=torch.int32, device=dev)
    g.gemm(A if WP else ap, bp, o_pairs, M, N, K, am_a, am_b, "b" if WP else True, **kw, **({"out_amax": am_p.data_ptr()} if am_p is not None else {}))
    g.gemm(A, W, o_split, M, N, K, am_a, am_b, False, b_kcontig=not dgrad, **kw, **({"out_amax": am_s.data_ptr()} if am_s is not None else {}))
    torch.cuda.synchronize()
    same = torch.equal(o_pairs, o_split) and (am_p is None or torch.equal(am_p, am_s))
    nbad = int((o_pairs != o_split).sum())
    print(f"  M={M:6d} N={N:5d} K={K:5d} dgrad={int(dgrad)} epi={sorted(epi)}: bit-identical {same}" + ("" if same else f"  ({nbad} elements differ, nan {int(torch.isnan(o_pairs).sum())})"),
          flush=True)
    return same


def bench(M, N, K, gen, dgrad=False, **epi):
    A = torch.randn((M, K), generator=gen, device=dev)
    W = torch.randn((K, N) if dgrad else (N, K), generator=gen, device=dev) * 0.05
    am_a, am_b = g.amax(A), g.amax(W, rows=not dgrad)
    ap, bp = g.split_pairs(A, am_a), g.split_pairs(W, am_b, transpose=dgrad)
    if not (WP or SPLIT):
        del A
    del W
    kw = {}
    if "bias" in epi:
        kw["bias"] = torch.randn(N, generator=gen, device=dev)
    if epi.get("act"):
        kw["act"] = 1
    if "res" in epi:
        kw["res"] = torch.randn((M, N), generator=gen, device=dev)
    if "aux" in epi:
        kw["aux"] = torch.randn((M, N), generator=gen, device=dev)
    if "drop" in epi:
        kw["drop_p"], kw["drop_seed"] = 0.5, 1234
    out = torch.empty((M, N), device=dev)
    ws = g.ws_for(M, N, K)
    if SPLIT:
        Wf = torch.randn((N, K), generator=gen, device=dev) * 0.05
        return g.timeit(lambda: g.gemm(A, Wf, out, M, N, K, am_a, am_b, False, ws=ws, **kw), n=30)
    t = g.timeit(lambda: g.gemm(A if WP else ap, bp, out, M, N, K, am_a, am_b, "b" if WP else True, ws=ws, **kw), n=30)
    return t


SHAPES = [  # (M, N, K, epilogue, launches per C2 step fwd + dgrad through this shape)
    (83328, 512, 512, dict(bias=1, drop=1, res=1), 6), (83328, 512, 512, dict(bias=1, act=1), 6), (83328, 512, 512, dict(aux=1), 3), (83328, 512, 512, dict(), 3),
    (83328, 1536, 512, dict(bias=1), 2), (83328, 512, 1536, dict(res=1), 2),
    (44325, 512, 512, dict(bias=1, drop=1, res=1), 6), (44325, 512, 512, dict(bias=1, act=1), 6), (44325, 512, 512, dict(aux=1), 3), (44325, 512, 512, dict(), 3),
    (44325, 1536, 512, dict(bias=1), 2), (44325, 512, 1536, dict(res=1), 2),
    (28248, 512, 512, dict(bias=1, drop=1, res=1), 6), (28248, 512, 512, dict(bias=1, act=1), 6), (28248, 512, 512, dict(aux=1), 3), (28248, 512, 512, dict(), 3),
    (28248, 1536, 512, dict(bias=1), 3), (28248, 512, 1536, dict(res=1), 3),
    (17158, 512, 512, dict(bias=1, drop=1, res=1), 6), (17158, 512, 512, dict(bias=1, act=1), 6), (17158, 512, 512, dict(aux=1), 3), (17158, 512, 512, dict(), 3),
    (17158, 1536, 512, dict(bias=1), 3), (17158, 512, 1536, dict(res=1), 3),
    (8233, 2048, 512, dict(bias=1, act=1), 14), (8233, 512, 2048, dict(bias=1, drop=1, res=1), 14), (8233, 512, 512, dict(bias=1, drop=1, res=1), 28),
    (41664, 256, 2048, dict(bias=1, act=1), 1), (41664, 2048, 256, dict(), 1), (41664, 256, 256, dict(bias=1, act=1), 4),
]


def main():
    gen = torch.Generator(device=dev)
    gen.manual_seed(0)
    tag = sys.argv[sys.argv.index("--tag") + 1] if "--tag" in sys.argv else "?"
    env = {k: v for k, v in os.environ.items() if k.startswith("GRAPPA_")}
    print(f"== {tag}  {env}", flush=True)
    ok = True
    if "--no-check" not in sys.argv:
        for (M, N, K, dg, epi) in [(40000, 512, 512, False, dict(bias=1)), (40000, 512, 512, True, dict()), (70001, 512, 96, False, dict(bias=1, act=1)),
                                   (70001, 516, 48, False, dict(bias=1, drop=1, res=1)), (50000, 512, 40, False, dict(bias=1)), (45003, 768, 512, False, dict(aux=1)),
                                   (45003, 512, 1536, True, dict(res=1)), (40000, 512, 512, False, dict(bias=1, act=1, amax=1)),
                                   (40000, 512, 512, False, dict(aux=1, res=1)), (140000, 512, 64, False, dict(bias=1)), (140000, 640, 512, False, dict(bias=1, res=1)),
                                   (300, 200, 64, False, dict()), (1000, 512, 512, False, dict(bias=1))]:
            if WP and K % 16:                 # (fp32 rows carry no padding: the weight-pairs kernels take whole slabs only)
                continue
            ok &= check(M, N, K, gen, dgrad=dg, **epi)
        print(f"== {tag} all bit-identical: {ok}", flush=True)
    if "--no-timing" in sys.argv:
        return
    tot = 0.0
    lines = []
    for (M, N, K, epi, cnt) in SHAPES:
        t = bench(M, N, K, gen, **epi)
        tot += t * cnt
        lines.append(f"  M={M:6d} N={N:5d} K={K:5d} epi={''.join(sorted(k[0] for k in epi)) or '-':5s} x{cnt:2d}: {1e3 * t:7.1f} us {2.0 * M * N * K / t / 1e9:7.1f} TF")
    print("\n".join(lines))
    print(f"== {tag} weighted sum {tot:.3f} ms per step, check {ok}", flush=True)


if __name__ == "__main__":
    main()
