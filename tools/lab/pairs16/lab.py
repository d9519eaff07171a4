#!/usr/bin/env python3
"""tools/lab/pairs16/lab.py [quick|big] -- on the GPU box, after tools/lab/pairs16/build.sh: the pair-format product on v_mfma_f32_16x16x32_f16
(plan_cfg 9 of the LAB library) against the shipped pinned-pipeline 32x32x16 kernel (plan_cfg 6) on the layer shapes of C2 / C3 / C5: float64 error
of both, their largest difference, time per launch (HIP events, random data, 20 launches after 3).  profiles/r6_pairs16_lab.txt holds the round's run."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault("GRAPPA_HIP_LIB", os.path.join(HERE, "libgrappa_hip_pairs16.so"))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import torch  # noqa: E402

import gemm_pairs_check as gp  # noqa: E402

SHAPES = [(1000, 512, 512), (300, 200, 128), (777, 130, 160), (5000, 512, 1536), (83328, 512, 512), (83328, 1536, 512), (83328, 512, 1536), (44325, 1536, 512),
          (44325, 512, 512), (28248, 512, 512), (17158, 1536, 512), (41664, 256, 2048), (529948, 1536, 512), (529948, 512, 512)]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    SHAPES = SHAPES[:6]
if len(sys.argv) > 1 and sys.argv[1] == "big":
    SHAPES = [(83328, 512, 512), (83328, 1536, 512), (83328, 512, 1536)]
gen = torch.Generator(device="cuda").manual_seed(5)
print(f"{'M':>7} {'N':>5} {'K':>5}    err16     err32   max|16-32|   us16    us32   us32/us16")
for M, N, K in SHAPES:
    A = torch.randn((M, K), generator=gen, device="cuda") * torch.exp2(torch.randint(-3, 4, (M, 1), generator=gen, device="cuda").float())
    W = torch.randn((N, K), generator=gen, device="cuda") * 0.05
    bias = torch.randn(N, generator=gen, device="cuda")
    am_a, am_b = gp.amax(A), gp.amax(W)
    ap, bp = gp.split_pairs(A, am_a), gp.split_pairs(W, am_b)
    outs, times = {}, {}
    for cfg in (9, 6):
        out = torch.full((M, N), float("nan"), device="cuda")
        gp.EXTRA = {"plan_cfg": cfg + 1, "plan_nsplit": 1, "plan_tail": 2}
        try:
            ws = gp.gemm(ap, bp, out, M, N, K, am_a, am_b, True, bias=bias)
            torch.cuda.synchronize()
            for _ in range(3):
                gp.gemm(ap, bp, out, M, N, K, am_a, am_b, True, ws=ws, bias=bias)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                gp.gemm(ap, bp, out, M, N, K, am_a, am_b, True, ws=ws, bias=bias)
            e1.record()
            torch.cuda.synchronize()
            times[cfg] = e0.elapsed_time(e1) / 20 * 1e3
        finally:
            gp.EXTRA = {}
        outs[cfg] = out
    err = {}
    for cfg in (9, 6):
        e = 0.0
        for rows in (slice(0, min(M, 20000)), slice(max(0, M - 3000), M)):          # the head and the (ragged) tail of the table
            ref = A[rows].double() @ W.double().t() + bias.double()
            e = max(e, float(((outs[cfg][rows].double() - ref).abs() / ref.abs().amax(dim=1, keepdim=True)).max()))
        err[cfg] = e
    dd = float((outs[9] - outs[6]).abs().max())
    print(f"{M:7d} {N:5d} {K:5d}  {err[9]:9.2e} {err[6]:9.2e} {dd:10.2e} {times[9]:8.1f} {times[6]:7.1f}   {times[6] / times[9]:.3f}"
          + ("  NaN!" if bool(torch.isnan(outs[9]).any()) else ""))
    del A, W, ap, bp, outs
    torch.cuda.empty_cache()
