import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from grappa_amd.backend import get_backend
be = get_backend()
def run(M, N, K, mode, ak=True, bk=True):
    A = torch.randn((M, K) if ak else (K, M), device="cuda")
    B = torch.randn((N, K) if bk else (K, N), device="cuda")
    C = torch.empty(M, N, device="cuda")
    for _ in range(3):
        be.gemm(A, B, C, M=M, N=N, K=K, a_kcontig=ak, b_kcontig=bk, precision=mode)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        be.gemm(A, B, C, M=M, N=N, K=K, a_kcontig=ak, b_kcontig=bk, precision=mode)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10
for (M, N, K, ak, bk) in ((65536, 512, 512, 1, 1), (65536, 512, 1024, 1, 1), (65536, 512, 2048, 1, 1), (65536, 512, 256, 1, 1)):
    for mode in ("f32_bf16x6", "dbg2_x6", "dbg4_x6", "f32"):
        ms = run(M, N, K, mode, bool(ak), bool(bk))
        print(f"{M}x{N}x{K} ({ak},{bk}) {mode:12s}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.1f} TF", flush=True)

C = torch.empty(65536, 512, device="cuda"); A = torch.randn(65536, 512, device="cuda")
for name, fn in (("fill", lambda: C.fill_(1.0)), ("copy", lambda: C.copy_(A)), ("add", lambda: torch.add(A, 1.0, out=C))):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    print(name, "134 MB:", e0.elapsed_time(e1) / 10, "ms")
