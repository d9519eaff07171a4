"""GPU: where do the ~20 us of a small dense product go?  Times the default product (f32_f16x3, fp32 operands) through the backend for small
M over K = 16 .. 512 (the K = 16 time is the fixed cost: launch, prologue, epilogue, split-K reduction launch if the plan cuts K), back to
back on one stream (HIP events around 200 launches), with and without the row-maxima output, with split-K forced off.
    python tools/small_gemm_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from grappa_amd.backend import get_backend
    be = get_backend()
    lib = be.lib

    def t(f, n=200):
        for _ in range(10):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        lib.grappa_launch_count(1)
        e0.record()
        for _ in range(n):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n, lib.grappa_launch_count(1) / n

    print("us per product (library launches per product)")
    for M in (40, 160, 1062, 2228, 5546):
        for N in (512, 1536):
            row = []
            for K in (16, 64, 128, 256, 512, 2048):
                A = torch.randn(M, K, device="cuda")
                W = torch.randn(N, K, device="cuda")
                Cc = torch.empty(M, N, device="cuda")
                b = torch.randn(N, device="cuda")
                sa = be.amax(A, rows=True)
                us, nl = t(lambda: be.gemm(A, W, Cc, M=M, N=N, K=K, bias=b, a_scales=sa))
                us2, nl2 = t(lambda: be.gemm(A, W, Cc, M=M, N=N, K=K, bias=b, a_scales=sa, out_amax=True))
                row.append(f"K={K}: {us:5.1f} ({nl:.1f}) +amax {us2:5.1f} ({nl2:.1f})")
            print(f"M={M:5d} N={N:4d}  " + " | ".join(row))
    # an empty-ish kernel for scale: the library's add of two small vectors
    x = torch.randn(1024, device="cuda")
    y = torch.empty_like(x)
    us, _ = t(lambda: lib.grappa_add_f32(torch.cuda.current_stream().cuda_stream, 1024, x.data_ptr(), x.data_ptr(), y.data_ptr()))
    print(f"grappa_add_f32 on 1024 floats: {us:.1f} us per launch (launch floor)")


if __name__ == "__main__":
    main()
