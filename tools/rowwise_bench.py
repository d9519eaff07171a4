import sys, torch
sys.path.insert(0, '/root/repo')
from grappa_amd.backend import get_backend
be = get_backend()
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M, W in ((83328, 512), (44325, 512), (8233, 512), (20832, 2048)):
    x = torch.randn(M, W, device='cuda'); g = torch.randn(W, device='cuda'); b = torch.randn(W, device='cuda')
    y = torch.empty_like(x); mean = torch.empty(M, device='cuda'); rstd = torch.empty(M, device='cuda')
    dy = torch.randn(M, W, device='cuda'); dx = torch.empty_like(x); dg = torch.zeros(W, device='cuda'); db = torch.zeros(W, device='cuda')
    us_f = t(lambda: be.layernorm_fwd(x, g, b, y, mean, rstd))
    us_b = t(lambda: be.layernorm_bwd(dy, x, mean, rstd, g, dx, dg, db, accumulate=True))
    dz = torch.empty_like(x)
    us_a = t(lambda: be.act_dropout_bwd(dy, y, 0.5, 1234, dz))
    by = M * W * 4
    print(f"M={M} W={W}: LN fwd {us_f:.1f} us = {2*by/us_f/1e6:.2f} TB/s | LN bwd {us_b:.1f} us = {3*by/us_b/1e6:.2f} TB/s | act_dropout_bwd {us_a:.1f} us = {3*by/us_a/1e6:.2f} TB/s")
