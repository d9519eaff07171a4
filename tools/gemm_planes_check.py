"""GPU: correctness and speed of the plane-format GEMM (csrc/gemm_planes.hip) against a float64 product and against the
split-in-kernel GEMM (csrc/gemm_bf16x.hip), straight through the C ABI.
    python tools/gemm_planes_check.py            # correctness on small / ragged shapes, then timing on the workload's big shapes"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from grappa_amd import _lib  # noqa: E402

lib = _lib.load()
dev = "cuda"


def split_planes(x, rows_pad=32, cols_pad=32):
    """fp32 (R, C) -> bf16 (3, Rp, Cp) zero padded"""
    R, Cc = x.shape
    Rp, Cp = (R + rows_pad - 1) // rows_pad * rows_pad, (Cc + cols_pad - 1) // cols_pad * cols_pad
    out = torch.zeros((3, Rp, Cp), dtype=torch.bfloat16, device=x.device)
    r = x.clone()
    for p in range(3):
        h = r.to(torch.bfloat16)
        out[p, :R, :Cc] = h
        r = r - h.float()
    return out


def ws_for(M, N, K):
    n = lib.grappa_gemm_f32_workspace_bytes(M, N, K)
    return torch.empty(max(n, 16), dtype=torch.uint8, device=dev)


def gemm(a, b, out, M, N, K, a_kcontig, b_kcontig, planes, precision="f32_bf16x6", **kw):
    d = _lib.GemmDesc()
    d.M, d.N, d.K, d.a_kcontig, d.b_kcontig = M, N, K, int(a_kcontig), int(b_kcontig)
    if planes == "b":
        d.A, d.lda = a.data_ptr(), a.stride(0)
        d.B, d.ldb, d.b_planes, d.b_plane_stride = b.data_ptr(), b.stride(1), 1, b.stride(0)
    elif planes:
        d.A, d.lda, d.a_planes, d.a_plane_stride = a.data_ptr(), a.stride(1), 1, a.stride(0)
        d.B, d.ldb, d.b_planes, d.b_plane_stride = b.data_ptr(), b.stride(1), 1, b.stride(0)
    else:
        d.A, d.lda, d.B, d.ldb = a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0)
    if out is not None:
        d.C, d.ldc = out.data_ptr(), out.stride(0)
    for k, v in kw.items():
        if k in ("Cp", "resp", "auxp"):
            setattr(d, k, v.data_ptr())
            setattr(d, "ld" + k.lower(), v.stride(1))
            setattr(d, k.lower() + "_plane_stride", v.stride(0))
        elif k in ("res", "aux"):
            setattr(d, k, v.data_ptr())
            setattr(d, "ld" + k, v.stride(0))
        elif k in ("bias", "a_colsum"):
            setattr(d, k, v.data_ptr())
        else:
            setattr(d, k, v)
    d.precision = _lib.GEMM_PRECISIONS[precision]
    ws = ws_for(M, N, K)
    rc = lib.grappa_gemm_f32(torch.cuda.current_stream().cuda_stream, C.byref(d), ws.data_ptr(), ws.numel())
    assert rc == 0, rc
    return ws


def merge(pl, R, Cc):
    return (pl[0, :R, :Cc].float() + pl[1, :R, :Cc].float()) + pl[2, :R, :Cc].float()


def check(M, N, K, kmajor, gen):
    A = torch.randn((M, K), generator=gen, device=dev)
    B = torch.randn((N, K), generator=gen, device=dev)
    ref = (A.double() @ B.double().t())
    if not kmajor:
        ap, bp = split_planes(A), split_planes(B)
    else:
        # k-major operands carry NO row padding (the kernel takes k-rows beyond K from a page of zeros)
        ap, bp = split_planes(A.t().contiguous(), rows_pad=1), split_planes(B.t().contiguous(), rows_pad=1)
    assert torch.equal(merge(ap, *(A.shape if not kmajor else A.t().shape)), A if not kmajor else A.t())
    out = torch.full((M, N), float("nan"), device=dev)
    gemm(ap, bp, out, M, N, K, not kmajor, not kmajor, True)
    torch.cuda.synchronize()
    scale = (A.double().abs() @ B.double().abs().t())
    err = float(((out.double() - ref).abs() / scale).max())
    # the split-in-kernel GEMM on the same operands
    out2 = torch.empty((M, N), device=dev)
    if not kmajor:
        gemm(A, B, out2, M, N, K, True, True, False)
    else:
        gemm(A.t().contiguous(), B.t().contiguous(), out2, M, N, K, False, False, False)
    torch.cuda.synchronize()
    err2 = float(((out2.double() - ref).abs() / scale).max())
    msg = ""
    if not kmajor and K % 32 == 0:
        out3 = torch.full((M, N), float("nan"), device=dev)
        gemm(A, bp, out3, M, N, K, True, True, "b")
        torch.cuda.synchronize()
        err3 = float(((out3.double() - ref).abs() / scale).max())
        msg = f"   fp32 A + weight planes err {err3:.2e} (bit-identical to split-in-kernel: {bool(torch.equal(out3, out2))})"
        assert err3 < 4e-7, err3
    print(f"  M {M:6d} N {N:5d} K {K:6d} {'k-major' if kmajor else 'k-contig'}: planes err {err:.2e}   split-in-kernel err {err2:.2e}{msg}", flush=True)
    assert err < 4e-7, err
    return err


def check_epilogue(gen):
    """bias + ELU + residual in planes + planes output + ELU' from planes, dropout, accumulate, column sums"""
    M, N, K = 1000, 512, 512
    A = torch.randn((M, K), generator=gen, device=dev)
    W = torch.randn((N, K), generator=gen, device=dev) / K ** 0.5
    bias = torch.randn((N,), generator=gen, device=dev)
    R = torch.randn((M, N), generator=gen, device=dev)
    ap, wp, rp = split_planes(A), split_planes(W), split_planes(R)
    cp = torch.zeros((3, (M + 31) // 32 * 32, N), dtype=torch.bfloat16, device=dev)
    gemm(ap, wp, None, M, N, K, True, True, True, bias=bias, act=1, resp=rp, Cp=cp)
    torch.cuda.synchronize()
    want = torch.nn.functional.elu(A.double() @ W.double().t() + bias.double()) + R.double()
    got = merge(cp, M, N)
    e = float((got.double() - want).abs().max() / want.abs().max())
    print(f"  epilogue bias+ELU+resp -> Cp: {e:.2e}", flush=True)
    assert e < 1e-6
    assert float(cp[:, M:].float().abs().max()) == 0.0            # pad rows untouched
    # ELU' from planes (aux) + fp32 out with accumulate
    U = torch.nn.functional.elu(torch.randn((M, N), generator=gen, device=dev))
    up = split_planes(U)
    out = torch.ones((M, N), device=dev)
    gemm(ap, wp, out, M, N, K, True, True, True, auxp=up, accumulate=1)
    torch.cuda.synchronize()
    want = (A.double() @ W.double().t()) * torch.where(U > 0, torch.ones_like(U), U + 1).double() + 1.0
    e = float((out.double() - want).abs().max() / want.abs().max())
    print(f"  epilogue auxp + accumulate: {e:.2e}", flush=True)
    assert e < 1e-6
    # wgrad layout with column sums and split-K
    T, No, Ko = 20000, 512, 512
    dZ = torch.randn((T, No), generator=gen, device=dev)
    X = torch.randn((T, Ko), generator=gen, device=dev)
    dzp, xp = split_planes(dZ, rows_pad=1), split_planes(X, rows_pad=1)
    dW = torch.zeros((No, Ko), device=dev)
    db = torch.zeros((No,), device=dev)
    gemm(dzp, xp, dW, No, Ko, T, False, False, True, a_colsum=db, accumulate=1)
    torch.cuda.synchronize()
    want = dZ.double().t() @ X.double()
    e = float((dW.double() - want).abs().max() / want.abs().max())
    eb = float((db.double() - dZ.double().sum(0)).abs().max() / dZ.double().sum(0).abs().max())
    print(f"  wgrad (k-major, split-K) {e:.2e}  bias gradient {eb:.2e}", flush=True)
    assert e < 1e-5 and eb < 1e-5


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def bench(gen):
    shapes = [(83328, 512, 512, 0), (83328, 1536, 512, 0), (83328, 512, 1536, 0), (44325, 512, 512, 0), (28248, 512, 512, 0), (17158, 512, 512, 0),
              (8233, 2048, 512, 0), (8233, 512, 2048, 0), (8233, 512, 512, 0),
              (512, 512, 83328, 1), (1536, 512, 83328, 1), (512, 512, 28248, 1), (2048, 512, 8233, 1), (512, 512, 8233, 1)]
    tot_p = tot_s = tot_w = 0.0
    for M, N, K, kmajor in shapes:
        A = torch.randn((M, K), generator=gen, device=dev)
        B = torch.randn((N, K), generator=gen, device=dev)
        out = torch.empty((M, N), device=dev)
        if kmajor:
            Af, Bf = A.t().contiguous(), B.t().contiguous()
            ap, bp = split_planes(Af), split_planes(Bf)
            tp = timeit(lambda: gemm(ap, bp, out, M, N, K, False, False, True))
            ts = timeit(lambda: gemm(Af, Bf, out, M, N, K, False, False, False))
        else:
            ap, bp = split_planes(A), split_planes(B)
            tp = timeit(lambda: gemm(ap, bp, out, M, N, K, True, True, True))
            ts = timeit(lambda: gemm(A, B, out, M, N, K, True, True, False))
            tw = timeit(lambda: gemm(A, bp, out, M, N, K, True, True, "b"))
        fl = 2.0 * M * N * K
        tot_p += tp
        tot_s += ts
        tot_w += tw if not kmajor else tp
        print(f"  M {M:6d} N {N:5d} K {K:6d} {'k-major ' if kmajor else 'k-contig'}: planes {tp:7.3f} ms {fl / tp / 1e9:6.1f} TF  split-in-kernel {ts:7.3f} ms {fl / ts / 1e9:6.1f} TF"
              + ("" if kmajor else f"  fp32A+Wplanes {tw:7.3f} ms {fl / tw / 1e9:6.1f} TF"), flush=True)
    print(f"  sum: planes {tot_p:.3f} ms, split-in-kernel {tot_s:.3f} ms, fp32A+Wplanes (k-major rows: planes) {tot_w:.3f} ms")


def bench_epilogues(gen):
    """cost of the fused epilogue pieces on the split-in-kernel GEMM and the weight-plane GEMM (M x 512 x 512)"""
    M, N, K = 83328, 512, 512
    A = torch.randn((M, K), generator=gen, device=dev)
    W = torch.randn((N, K), generator=gen, device=dev) / K ** 0.5
    wp = split_planes(W)
    R = torch.randn((M, N), generator=gen, device=dev)
    U = torch.randn((M, N), generator=gen, device=dev)
    bias = torch.randn((N,), generator=gen, device=dev)
    out = torch.empty((M, N), device=dev)
    out2 = torch.empty((M, N), device=dev)
    cases = {"plain": {}, "bias": dict(bias=bias), "bias+res": dict(bias=bias, res=R), "bias+drop+res": dict(bias=bias, res=R, drop_p=0.5, drop_seed=123),
             "aux (ELU')": dict(aux=U), "bias+ELU+drop+res+C2": dict(bias=bias, res=R, drop_p=0.5, drop_seed=123, act=1, C2=out2.data_ptr(), ldc2=N),
             "accumulate": dict(accumulate=1)}
    for name, kw in cases.items():
        ts = timeit(lambda: gemm(A, W, out, M, N, K, True, True, False, **kw))
        tw = timeit(lambda: gemm(A, wp, out, M, N, K, True, True, "b", **kw))
        print(f"  {name:24s} split-in-kernel {ts:7.3f} ms   fp32A+Wplanes {tw:7.3f} ms", flush=True)


def bench_bf16(gen):
    """the plane kernels with ONE plane per operand = a plain bf16 GEMM (fp32 accumulate), against the split-in-kernel GEMM in its
    bf16 mode (fp32 operands rounded in the kernel)"""
    for M, N, K, kmajor in [(83328, 512, 512, 0), (83328, 1536, 512, 0), (83328, 512, 1536, 0), (333000, 512, 512, 0), (512, 512, 83328, 1), (1536, 512, 333000, 1)]:
        A = torch.randn((M, K), generator=gen, device=dev)
        B = torch.randn((N, K), generator=gen, device=dev)
        out = torch.empty((M, N), device=dev)
        if kmajor:
            Af, Bf = A.t().contiguous(), B.t().contiguous()
            ap, bp = split_planes(Af), split_planes(Bf)
            tp = timeit(lambda: gemm(ap, bp, out, M, N, K, False, False, True, precision="bf16"))
            ts = timeit(lambda: gemm(Af, Bf, out, M, N, K, False, False, False, precision="bf16"))
        else:
            ap, bp = split_planes(A), split_planes(B)
            tp = timeit(lambda: gemm(ap, bp, out, M, N, K, True, True, True, precision="bf16"))
            ts = timeit(lambda: gemm(A, B, out, M, N, K, True, True, False, precision="bf16"))
        fl = 2.0 * M * N * K
        print(f"  M {M:6d} N {N:5d} K {K:6d} {'k-major ' if kmajor else 'k-contig'}: bf16 planes {tp:7.3f} ms {fl / tp / 1e9:6.1f} TF   fp32-in bf16 mode {ts:7.3f} ms {fl / ts / 1e9:6.1f} TF", flush=True)


def build_variants(only=None):
    """libraries with the knock-out switches of csrc/gemm_planes.hip (GP_KNOCK), for `--variants`"""
    import subprocess
    csrc = os.path.join(ROOT, "grappa_amd", "csrc")
    outdir = os.path.join(ROOT, "build", "variants")
    os.makedirs(outdir, exist_ok=True)
    objs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".o") and f != "gemm_planes.o"]
    libs = {}
    for name, flags in (("nodma", ["-DGP_KNOCK=1"]), ("nomfma", ["-DGP_KNOCK=2"]), ("nodma_noepi", ["-DGP_KNOCK=1", "-DGP_NOEPI=1"]),
                        ("noepi", ["-DGP_NOEPI=1"]), ("stagger8", ["-DGP_STAGGER=8"]), ("stagger24", ["-DGP_STAGGER=24"]),
                        ("nodma_stagger8", ["-DGP_KNOCK=1", "-DGP_STAGGER=8"]), ("persist", ["-DGP_PERSIST=1"])):
        if only and name not in only:
            continue
        o = os.path.join(outdir, f"gemm_planes_{name}.o")
        so = os.path.join(outdir, f"libgrappa_hip_{name}.so")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", *flags, "-c", os.path.join(csrc, "gemm_planes.hip"), "-o", o], check=True)
        subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", so, o, *objs], check=True)
        libs[name] = so
    return libs


if __name__ == "__main__":
    if "--build-variants" in sys.argv:
        only = [a.split("=", 1)[1].split(",") for a in sys.argv if a.startswith("--only=")]
        print(build_variants(only[0] if only else None))
        sys.exit(0)
    if "--bf16" in sys.argv:
        bench_bf16(torch.Generator(device=dev).manual_seed(0))
        sys.exit(0)
    if "--epilogues" in sys.argv:
        bench_epilogues(torch.Generator(device=dev).manual_seed(0))
        sys.exit(0)
    if "--bench-only" in sys.argv:
        bench(torch.Generator(device=dev).manual_seed(0))
        sys.exit(0)
    gen = torch.Generator(device=dev).manual_seed(0)
    print("correctness (max |err| / sum |a||b|):")
    for M, N, K, km in [(256, 128, 64, 0), (256, 128, 512, 0), (300, 200, 96, 0), (1000, 512, 512, 0), (83, 77, 160, 0), (8233, 1536, 512, 0),
                        (256, 128, 64, 1), (256, 128, 4096, 1), (512, 512, 8233, 1), (300, 200, 1000, 1), (1536, 512, 20011, 1)]:
        check(M, N, K, bool(km), gen)
    check_epilogue(gen)
    if "--no-bench" not in sys.argv:
        print("timing:")
        bench(gen)
