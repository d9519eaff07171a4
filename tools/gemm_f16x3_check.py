"""precision "f32_f16x3" (two fp16 pieces per fp32 operand, three products, power-of-two row scales) beside "f32_bf16x6" and the
native fp32 matrix instruction: error against float64 on ordinary, wide-range and extreme-range operands, whether the matrix cores
keep fp16 denormals, and launch times (product alone; maxima pass alone) on the workload's shapes.

    python tools/gemm_f16x3_check.py [--bench-only] > gpurun_out/f16x3.txt
"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from grappa_amd.backend import HipBackend  # noqa: E402

MODES = ("f32", "f32_bf16x6", "f32_f16x3")


def rowrel(out, exact):
    return ((out.double().cpu() - exact).abs().amax(dim=1) / exact.abs().amax(dim=1).clamp_min(1e-300)).max().item()


def run(be, A, B, M, N, K, ak, bk, mode):
    out = torch.empty(M, N, device="cuda")
    be.gemm(A.cuda(), B.cuda(), out, M=M, N=N, K=K, a_kcontig=bool(ak), b_kcontig=bool(bk), precision=mode)
    torch.cuda.synchronize()
    return out


def accuracy(be):
    print("# max row-relative error against float64 (rows of C compared with their own largest magnitude)")
    g = torch.Generator().manual_seed(5)
    cases = []
    for (M, N, K, ak, bk) in [(1000, 512, 512, 1, 1), (700, 384, 1536, 1, 0), (512, 512, 5000, 0, 0), (257, 511, 256, 1, 1), (1536, 512, 3001, 0, 0)]:
        for name, sigma in (("normal", 0.0), ("lognormal s=2", 2.0), ("lognormal s=6", 6.0)):
            A = torch.randn((M, K) if ak else (K, M), generator=g)
            B = torch.randn((N, K) if bk else (K, N), generator=g)
            if sigma:
                A = A * torch.exp(torch.randn(A.shape, generator=g) * sigma)
                B = B * torch.exp(torch.randn(B.shape, generator=g) * sigma * 0.5)
            cases.append((f"{M}x{N}x{K} ak={ak} bk={bk} {name}", A, B, M, N, K, ak, bk))
    # rows of A spread over 60 orders of magnitude (each row has its own scale); columns of the weight likewise
    M, N, K = 600, 256, 512
    A = torch.randn(M, K, generator=g) * torch.pow(10.0, torch.empty(M, 1).uniform_(-30, 30, generator=g))
    B = torch.randn(N, K, generator=g) * torch.pow(10.0, torch.empty(N, 1).uniform_(-6, 6, generator=g))
    cases.append((f"{M}x{N}x{K} rows scaled 1e-30..1e30", A, B, M, N, K, 1, 1))
    A = torch.randn(M, K, generator=g)
    A[::7] = 0.0                                     # all-zero rows
    A[3, 5] = 1e-41                                  # an fp32 denormal as a row's ... not its maximum
    A[7] = 0.0
    A[7, 9] = 3e-42                                  # a row whose maximum is an fp32 denormal
    cases.append((f"{M}x{N}x{K} zero rows / fp32 denormals", A, torch.randn(N, K, generator=g), M, N, K, 1, 1))
    for name, A, B, M, N, K, ak, bk in cases:
        exact = (A if ak else A.t()).double() @ (B if bk else B.t()).double().t()
        errs = {m: rowrel(run(be, A, B, M, N, K, ak, bk, m), exact) for m in MODES}
        print(f"  {name:48s} " + "  ".join(f"{m} {errs[m]:.2e}" for m in MODES))


def denormals(be):
    """a row whose large element meets a zero of B: the result is carried by elements 2^-18 below the row maximum, whose low fp16
    piece is an fp16 denormal.  Kept denormals: ~2^-20 relative; flushed: ~2^-12."""
    M, N, K = 256, 256, 512
    g = torch.Generator().manual_seed(9)
    A = (1.0 + torch.rand(M, K, generator=g)) * 2.0 ** -18
    A[:, 0] = 1.0
    B = torch.randn(N, K, generator=g)
    B[:, 0] = 0.0
    exact = A.double() @ B.double().t()
    for m in MODES:
        print(f"  small elements under a large one: {m} {rowrel(run(be, A, B, M, N, K, 1, 1, m), exact):.2e}")


def bench(be):
    shapes = [(83328, 512, 512, 1, 1), (83328, 1536, 512, 1, 1), (83328, 512, 1536, 1, 0), (44325, 512, 512, 1, 0), (28248, 512, 512, 1, 1),
              (17158, 512, 512, 1, 1), (8233, 2048, 512, 1, 1), (8233, 512, 2048, 1, 0), (8233, 512, 512, 1, 1),
              (512, 512, 83328, 0, 0), (1536, 512, 83328, 0, 0), (512, 512, 28248, 0, 0), (2048, 512, 8233, 0, 0), (512, 512, 8233, 0, 0)]
    print("# launch time of the product alone (maxima computed ahead), and of the maxima pass over the activation operand(s)")
    tot = {m: 0.0 for m in MODES}
    tot_amax = 0.0
    for (M, N, K, ak, bk) in shapes:
        A = torch.randn((M, K) if ak else (K, M), device="cuda")
        B = torch.randn((N, K) if bk else (K, N), device="cuda")
        out = torch.empty(M, N, device="cuda")
        line = f"  M {M:6d} N {N:5d} K {K:6d} {'k-contig' if ak else 'k-major '}:"
        for m in MODES:
            def call():
                be.gemm(A, B, out, M=M, N=N, K=K, a_kcontig=bool(ak), b_kcontig=bool(bk), precision=m)
            for _ in range(3):
                call()
            be.start_profile()
            for _ in range(10):
                call()
            prof = be.stop_profile()
            ms = prof["gemm_f32"][1] / 10
            tot[m] += ms
            line += f"  {m} {ms:.3f} ms {2.0 * M * N * K / ms / 1e9:6.1f} TF"
        if ak:
            def call2():
                be.gemm(A, B, out, M=M, N=N, K=K, a_kcontig=True, b_kcontig=bool(bk), precision="f32_f16x3", out_amax=True)
            for _ in range(3):
                call2()
            be.start_profile()
            for _ in range(10):
                call2()
            ms = be.stop_profile()["gemm_f32"][1] / 10
            line += f"  +out_amax {ms:.3f} ms"
        # the maxima pass: the activation operand(s) of this product (weights are cached per step)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ops = [A] if ak else [A, B]
        for _ in range(2):
            for t in ops:
                be._amax_launch(t, True, False)
        ev0.record()
        for _ in range(10):
            for t in ops:
                be._amax_launch(t, True, False)
        ev1.record()
        torch.cuda.synchronize()
        am = ev0.elapsed_time(ev1) / 10
        tot_amax += am
        nbytes = sum(t.numel() * 4 for t in ops)
        print(line + f"  | maxima {am:.3f} ms {nbytes / am / 1e9:6.2f} TB/s")
    print("  sum: " + "  ".join(f"{m} {tot[m]:.3f} ms" for m in MODES) + f"  maxima {tot_amax:.3f} ms")


def build_variants():
    """libraries with the timing knock-outs of csrc/gemm_bf16x_impl.h (GB_KNOCK); run each with GRAPPA_HIP_LIB=<so> ... --bench-only"""
    import subprocess
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    csrc = os.path.join(root, "grappa_amd", "csrc")
    outdir = os.path.join(root, "build", "variants")
    os.makedirs(outdir, exist_ok=True)
    mode_units = ("gemm_bf16x_h3", "gemm_bf16x_x6")           # the two arithmetics the tool times (one translation unit each)
    objs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".o") and f[:-2] not in mode_units]
    only = [a.split("=", 1)[1].split(",") for a in sys.argv if a.startswith("--only=")]
    for k, name in ((1, "nomfma"), (2, "nosplit"), (3, "noglobal"), (4, "nolds"), (5, "nobarrier"), (6, "noepi"), (7, "nostore"), (8, "nodescale")):
        if only and name not in only[0]:
            continue
        so = os.path.join(outdir, f"libgrappa_hip_bx_{name}.so")
        built = []
        for unit in mode_units:
            o = os.path.join(outdir, f"{unit}_{name}.o")
            subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950",
                            "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",          # as csrc/Makefile
                            f"-DGB_KNOCK={k}", "-c",
                            os.path.join(csrc, unit + ".hip"), "-o", o], check=True)
            built.append(o)
        subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", so, *built, *objs], check=True)
        print(so)


def main():
    if "--build-variants" in sys.argv:
        build_variants()
        return
    be = HipBackend()
    if "--bench-only" not in sys.argv:
        accuracy(be)
        denormals(be)
    bench(be)


if __name__ == "__main__":
    main()
