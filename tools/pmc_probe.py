"""Small PMC probe workloads for rocprofv3 --pmc passes (one process, a few launches each):
   python tools/pmc_probe.py gemm [precision]  -> 10 launches of the 83328x512x512 forward GEMM (+ dgrad + wgrad variants)
   python tools/pmc_probe.py gat    -> 10 launches of gat_fwd / gat_bwd on the C3-size graph (1024 molecules)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def gemm():
    from grappa_amd.backend import get_backend
    be = get_backend()
    precision = sys.argv[2] if len(sys.argv) > 2 else "f32"
    for (M, N, K, ak, bk) in [(83328, 512, 512, 1, 1), (83328, 512, 512, 1, 0), (512, 512, 83328, 0, 0)]:
        A = torch.randn((M, K) if ak else (K, M), device="cuda")
        B = torch.randn((N, K) if bk else (K, N), device="cuda")
        C = torch.empty((M, N), device="cuda")
        for _ in range(10):
            be.gemm(A, B, C, M=M, N=N, K=K, a_kcontig=bool(ak), b_kcontig=bool(bk), precision=precision)
    torch.cuda.synchronize()


def gat():
    from grappa_amd.backend import get_backend
    from grappa_amd.datasets import build_workload
    be = get_backend()
    g = build_workload("C3-espaloma-b1024", seed=0).to("cuda")
    plan = g.plan()
    H, D = 16, 32
    ft = torch.randn(plan.N, H * D, device="cuda")
    dout = torch.randn(plan.N, H * D, device="cuda")
    out, alpha, dft = torch.empty_like(ft), torch.empty(plan.E, H, device="cuda"), torch.empty_like(ft)
    for _ in range(10):
        be.gat_fwd(plan, ft, H, D, out, alpha)
        be.gat_bwd(plan, ft, out, alpha, dout, H, D, dft)
    torch.cuda.synchronize()
    print("atoms", plan.N, "edges", plan.E, "fwd algorithmic MB", (plan.E * (H * D * 4 + 4) + plan.N * (2 * H * D * 4 + 4)) / 1e6,
          "bwd algorithmic MB", (plan.E * (2 * H * D * 4 + 4) + plan.N * (4 * H * D * 4 + 4)) / 1e6)


def gemm3():
    """the three fp32-grade GEMM designs of round 2 on M x 512 x 512 (forward layout) and the wgrad layout: split in the kernel
    (gemm_bf16x_kernel), both operands in planes (gemm_planes_kernel), fp32 A + weight planes (gemm_wplanes_kernel)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gemm_planes_check as gp
    gen = torch.Generator(device="cuda").manual_seed(0)
    M, N, K = 83328, 512, 512
    A = torch.randn((M, K), generator=gen, device="cuda")
    B = torch.randn((N, K), generator=gen, device="cuda")
    out = torch.empty((M, N), device="cuda")
    ap, bp = gp.split_planes(A), gp.split_planes(B)
    for _ in range(6):
        gp.gemm(A, B, out, M, N, K, True, True, False)
        gp.gemm(ap, bp, out, M, N, K, True, True, True)
        gp.gemm(A, bp, out, M, N, K, True, True, "b")
    T = 83328
    dZ = torch.randn((T, 512), generator=gen, device="cuda")
    X = torch.randn((T, 512), generator=gen, device="cuda")
    dzp, xp = gp.split_planes(dZ, rows_pad=1), gp.split_planes(X, rows_pad=1)
    dW = torch.empty((512, 512), device="cuda")
    for _ in range(6):
        gp.gemm(dZ, X, dW, 512, 512, T, False, False, False)
        gp.gemm(dzp, xp, dW, 512, 512, T, False, False, True)
    torch.cuda.synchronize()


if __name__ == "__main__":
    {"gemm": gemm, "gat": gat, "gemm3": gemm3}[sys.argv[1]]()
