"""GPU: what does a small dense product cost on the DEVICE?  100 identical launches of the default product (f32_f16x3, fp32 operands)
recorded in a hipGraph and replayed: device time per product including its split-K reduction launch, free of the host's ~12 us per call.
Over M (one molecule ... a batch of 32), K (16: the fixed cost of a launch) and forced K cuts.
    python tools/small_gemm_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _plan(lib, M, N, K, prec, v, cfg=-1, ns=0, tail=-1):
    """the plan the library would choose under a forced configuration (C ABI 10: the force is a field of the product's descriptor)"""
    from grappa_amd import _lib as _L
    d = _L.GemmDesc()
    d.M, d.N, d.K, d.precision = M, N, K, prec
    d.plan_cfg, d.plan_nsplit, d.plan_tail = (cfg + 1 if cfg >= 0 else 0), max(ns, 0), (2 if tail == 0 else 3 if tail == 1 else 0)
    return lib.grappa_gemm_f32_plan_desc(C.byref(d), *[C.byref(x) for x in v])


def main():
    from grappa_amd.backend import get_backend
    be = get_backend()
    lib = be.lib
    st = torch.cuda.Stream()

    def replay_us(f, n=100):
        with torch.cuda.stream(st):
            for _ in range(3):
                f()
            st.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                for _ in range(n):
                    f()
            g.replay()
            st.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(5):
                g.replay()
            e1.record(st)
            st.synchronize()
        return e0.elapsed_time(e1) * 1e3 / (5 * n)

    print("device us per product, 100 dependent-free launches back to back in a hipGraph (plan: tile, K cuts)")
    v = [__import__("ctypes").c_int() for _ in range(5)]
    import ctypes as C
    for M in (40, 160, 1062, 5546):
        for N, K in ((512, 16), (512, 128), (512, 512), (1536, 512), (2048, 512), (512, 2048)):
            A = torch.randn(M, K, device="cuda")
            W = torch.randn(N, K, device="cuda")
            Cc = torch.empty(M, N, device="cuda")
            b = torch.randn(N, device="cuda")
            with torch.cuda.stream(st):
                sa = be.amax(A, rows=True)
            row = []
            for ns in (0, 1, 2, 4, 7, 15):
                be.plan_override = (-1, ns, -1)
                _plan(lib, M, N, K, 5, v, ns=ns)
                if ns and v[2].value != ns:
                    continue
                try:
                    us = replay_us(lambda: be.gemm(A, W, Cc, M=M, N=N, K=K, bias=b, a_scales=sa))
                except Exception as e:
                    row.append(f"ns{ns}: {type(e).__name__}")
                    continue
                row.append(f"{'model' if ns == 0 else 'ns' + str(ns)} ({v[0].value}x{v[1].value} ns{v[2].value}): {us:5.1f}")
            be.plan_override = None
            print(f"M={M:5d} N={N:4d} K={K:4d}  " + " | ".join(row))
    x = torch.randn(1024, device="cuda")
    y = torch.empty_like(x)
    us = replay_us(lambda: lib.grappa_add_f32(st.cuda_stream, 1024, x.data_ptr(), x.data_ptr(), y.data_ptr()))
    print(f"grappa_add_f32 on 1024 floats: {us:.1f} us per launch in the graph (launch floor)")


if __name__ == "__main__":
    main()
