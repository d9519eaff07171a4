"""GPU: correctness and speed of the pair-format GEMM (csrc/gemm_pairs.hip) against a float64 product and against the fp32-operand
fp16-split GEMM (csrc/gemm_bf16x_impl.h MODE H3, the default arithmetic), straight through the C ABI.
    python tools/gemm_pairs_check.py            # correctness on small / ragged shapes, then timing on the workload's big shapes"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from grappa_amd import _lib  # noqa: E402

lib = _lib.load()
dev = "cuda"
F16X3 = _lib.GEMM_PRECISIONS["f32_f16x3"]


def stream():
    return torch.cuda.current_stream().cuda_stream


def amax(x, rows=True):
    """bit patterns of the row (or column) maxima of an fp32 matrix"""
    R, Cc = x.shape
    out = torch.empty(R if rows else Cc, dtype=torch.int32, device=dev)
    need = lib.grappa_amax_f32_workspace_bytes(R, Cc)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
    rc = lib.grappa_amax_f32(stream(), R, Cc, x.data_ptr(), x.stride(0), out.data_ptr() if rows else None, None if rows else out.data_ptr(),
                             ws.data_ptr(), ws.numel())
    assert rc == 0, rc
    return out


def split_pairs(x, am, transpose=False):
    """fp32 (R, C) + maxima of the output rows -> fp16 (rows, 2 * round_up(cols, 32)) in the pair layout, zero padded"""
    R, Cc = x.shape
    rows, cols = (Cc, R) if transpose else (R, Cc)
    out = torch.zeros((rows, 2 * ((cols + 31) // 32 * 32)), dtype=torch.float16, device=dev)
    rc = lib.grappa_split_pairs_f32(stream(), R, Cc, x.data_ptr(), x.stride(0), am.data_ptr(), out.data_ptr(), out.stride(0), int(transpose))
    assert rc == 0, rc
    return out


def ws_for(M, N, K):
    return torch.empty(max(lib.grappa_gemm_f32_workspace_bytes(M, N, K), 16), dtype=torch.uint8, device=dev)


EXTRA = {}          # descriptor fields added to every product (plan_cfg, plan_nsplit, plan_tail ...: C ABI 10's per-call options)


def gemm(a, b, out, M, N, K, a_amax, b_amax, pairs, b_kcontig=True, ws=None, **kw):
    d = _lib.GemmDesc()
    kw = {**EXTRA, **kw}
    d.M, d.N, d.K, d.a_kcontig, d.b_kcontig = M, N, K, 1, int(b_kcontig)
    if pairs == "b":                        # "weight pairs": fp32 A, the weight in pairs
        d.A, d.lda = a.data_ptr(), a.stride(0)
        d.B, d.ldb, d.b_planes = b.data_ptr(), b.stride(0), 1
        d.b_kcontig = 1
    elif pairs:
        d.A, d.lda, d.a_planes = a.data_ptr(), a.stride(0), 1
        d.B, d.ldb, d.b_planes = b.data_ptr(), b.stride(0), 1
        d.b_kcontig = 1
    else:
        d.A, d.lda, d.B, d.ldb = a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0)
    d.C, d.ldc = out.data_ptr(), out.stride(0)
    d.a_amax, d.b_amax = a_amax.data_ptr(), b_amax.data_ptr()
    for k, v in kw.items():
        if k in ("res", "aux"):
            setattr(d, k, v.data_ptr())
            setattr(d, "ld" + k, v.stride(0))
        elif k == "bias":
            d.bias = v.data_ptr()
        else:
            setattr(d, k, v)
    d.precision = F16X3
    if ws is None:
        ws = torch.empty(max(lib.grappa_gemm_f32_workspace_bytes_desc(C.byref(d)), 16), dtype=torch.uint8, device=dev)
    rc = lib.grappa_gemm_f32(stream(), C.byref(d), ws.data_ptr(), ws.numel())
    assert rc == 0, rc
    return ws


def check(M, N, K, gen, dgrad=False, scale_rows=False, **epi):
    A = torch.randn((M, K), generator=gen, device=dev)
    if scale_rows:
        A = A * torch.exp2(torch.randint(-40, 40, (M, 1), generator=gen, device=dev).float())
    # forward: W[N][K]; dgrad: W[K][N] read as B^T (row-contiguous B of the fp32 kernel = the pairs of W^T)
    W = torch.randn((K, N) if dgrad else (N, K), generator=gen, device=dev) * 0.05
    am_a = amax(A)
    am_b = amax(W, rows=not dgrad)
    ref = A.double() @ (W.double() if dgrad else W.double().t())
    kw = {}
    if "bias" in epi:
        kw["bias"] = torch.randn(N, generator=gen, device=dev)
        ref = ref + kw["bias"].double()
    if epi.get("act"):
        kw["act"] = 1
        ref = torch.nn.functional.elu(ref)
    if "res" in epi:
        kw["res"] = torch.randn((M, N), generator=gen, device=dev)
        ref = ref + kw["res"].double()
    ap, bp = split_pairs(A, am_a), split_pairs(W, am_b, transpose=dgrad)
    o_pairs = torch.full((M, N), float("nan"), device=dev)
    o_split = torch.full((M, N), float("nan"), device=dev)
    gemm(ap, bp, o_pairs, M, N, K, am_a, am_b, True, **kw)
    gemm(A, W, o_split, M, N, K, am_a, am_b, False, b_kcontig=not dgrad, **kw)
    o_wp = None
    if K % 16 == 0:                         # fp32 A + weight pairs: the same bits as the all-pairs product where their K cuts agree
        o_wp = torch.full((M, N), float("nan"), device=dev)
        gemm(A, bp, o_wp, M, N, K, am_a, am_b, "b", **kw)
    torch.cuda.synchronize()
    scale = ref.abs().amax(dim=1, keepdim=True).clamp_min(1e-300)
    e_pairs = ((o_pairs.double() - ref).abs() / scale).max().item()
    e_split = ((o_split.double() - ref).abs() / scale).max().item()
    same = torch.equal(o_pairs, o_split)
    if o_wp is not None:
        e_wp = ((o_wp.double() - ref).abs() / scale).max().item()
        assert e_wp < 3e-6, ("weight-pairs kernel", e_wp)
        same = same and torch.equal(o_wp, o_pairs)
    print(f"  M={M:6d} N={N:5d} K={K:5d} dgrad={int(dgrad)} rows_scaled={int(scale_rows)} epi={sorted(epi)}: pairs err {e_pairs:.2e}  split err {e_split:.2e}  "
          f"bit-identical {same}")
    assert e_pairs < 3e-6, e_pairs
    return same


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def bench(M, N, K, gen, dgrad=False, **epi):
    A = torch.randn((M, K), generator=gen, device=dev)
    W = torch.randn((K, N) if dgrad else (N, K), generator=gen, device=dev) * 0.05
    am_a, am_b = amax(A), amax(W, rows=not dgrad)
    ap, bp = split_pairs(A, am_a), split_pairs(W, am_b, transpose=dgrad)
    kw = {}
    if "bias" in epi:
        kw["bias"] = torch.randn(N, generator=gen, device=dev)
    if epi.get("act"):
        kw["act"] = 1
    if "res" in epi:
        kw["res"] = torch.randn((M, N), generator=gen, device=dev)
    if "drop" in epi:
        kw["drop_p"], kw["drop_seed"] = 0.5, 1234
    out = torch.empty((M, N), device=dev)
    ws = ws_for(M, N, K)
    t_split = timeit(lambda: gemm(A, W, out, M, N, K, am_a, am_b, False, b_kcontig=not dgrad, ws=ws, **kw))
    t_pairs = timeit(lambda: gemm(ap, bp, out, M, N, K, am_a, am_b, True, ws=ws, **kw))
    t_wp = timeit(lambda: gemm(A, bp, out, M, N, K, am_a, am_b, "b", ws=ws, **kw))
    fl = 2.0 * M * N * K
    print(f"  M={M:6d} N={N:5d} K={K:5d} dgrad={int(dgrad)} epi={sorted(epi)}: split {t_split:.3f} ms {fl / t_split / 1e9:7.1f} TFLOP/s | "
          f"pairs {t_pairs:.3f} ms {fl / t_pairs / 1e9:7.1f} TFLOP/s x{t_split / t_pairs:.2f} | fp32 A + weight pairs {t_wp:.3f} ms {fl / t_wp / 1e9:7.1f} TFLOP/s "
          f"x{t_split / t_wp:.2f}", flush=True)
    return t_split, t_pairs, t_wp


def main():
    gen = torch.Generator(device=dev)
    gen.manual_seed(0)
    if "--timing-only" in sys.argv:          # knock-out builds (tools/pairs_knockouts.sh): results are wrong by construction
        for (M, N, K, dg, epi) in [(83328, 512, 512, False, dict(bias=1)), (83328, 1536, 512, False, dict(bias=1)), (83328, 512, 1536, True, dict()),
                                   (65536, 512, 512, False, dict(bias=1)), (65536, 512, 4096, False, dict(bias=1))]:
            bench(M, N, K, gen, dgrad=dg, **epi)
        return
    print("correctness (error relative to the row's largest result; float64 reference)")
    ok = True
    for (M, N, K) in [(256, 128, 32), (300, 200, 64), (1000, 512, 512), (257, 129, 96), (4096, 1536, 512), (777, 256, 1536), (5000, 512, 2048)]:
        ok &= check(M, N, K, gen)
    ok &= check(1000, 512, 512, gen, dgrad=True)
    ok &= check(3000, 512, 1536, gen, dgrad=True)
    ok &= check(1000, 512, 512, gen, scale_rows=True)
    ok &= check(1000, 512, 512, gen, bias=1, act=1)
    ok &= check(1000, 512, 512, gen, bias=1, res=1)
    ok &= check(20000, 64, 512, gen)                 # split-K territory (few tiles)
    print("all bit-identical to the fp32-operand fp16-split kernel:", ok)
    if "--no-timing" in sys.argv:
        return
    print("timing (ms, TFLOP/s of algorithmic fp32 work)")
    tot_s = tot_p = tot_w = 0.0
    for M in (83328, 44325, 17158, 28248):
        for (N, K, dg, epi) in [(1536, 512, False, dict(bias=1)), (512, 512, False, dict(bias=1, drop=1, res=1)), (512, 512, False, dict(bias=1, act=1)),
                                (512, 512, True, dict()), (512, 1536, True, dict(res=1))]:
            s, p, w = bench(M, N, K, gen, dgrad=dg, **epi)
            tot_s += s
            tot_p += p
            tot_w += w
    print(f"sum over the shapes: split {tot_s:.2f} ms, pairs {tot_p:.2f} ms (x{tot_s / tot_p:.2f}), fp32 A + weight pairs {tot_w:.2f} ms (x{tot_s / tot_w:.2f})")
    for M in (8233,):
        for (N, K) in [(512, 512), (2048, 512), (512, 2048), (256, 512)]:
            bench(M, N, K, gen, bias=1, act=1)


if __name__ == "__main__":
    main()
