"""Op-level probe of the multi-stream ordering problem ("Streams"): four streams each run GEMM -> LayerNorm chains
of this library; every LayerNorm output is compared with the result of the same chain run alone.
    GRAPPA_GEMM_PRECISION=bf16x3 python tools/stream_order_probe.py      (bf16x3 makes the window widest)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grappa_amd.backend import get_backend
be = get_backend()
torch.manual_seed(0)
Ms, W = [83328, 44325, 28248, 17158], 512
Wm = torch.randn(512, 512, generator=torch.Generator().manual_seed(1)).cuda() / 22.6
data = []
for M in Ms:
    x = torch.randn(M, W, device="cuda") * 2 + 0.3; g = torch.randn(W, device="cuda"); b = torch.randn(W, device="cuda")
    ref = torch.empty_like(x); mean = torch.empty(M, device="cuda"); rstd = torch.empty(M, device="cuda"); t0 = torch.empty_like(x)
    be.gemm(x, Wm, t0, M=M, N=512, K=512, res=x)
    be.layernorm_fwd(t0, g, b, ref, mean, rstd)
    data.append((x, g, b, ref, t0))
torch.cuda.synchronize()
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in Ms[1:]]
bad = first = 0
for trial in range(10):
    outs = []
    for st in streams[1:]:
        st.wait_stream(streams[0])
    for (x, g, b, ref, t0), st in zip(data, streams):
        with torch.cuda.stream(st):
            for rep in range(4):
                t = torch.empty_like(x)
                be.gemm(x, Wm, t, M=x.shape[0], N=512, K=512, res=x)
                y = torch.empty_like(x); m2 = torch.empty(x.shape[0], device="cuda"); r2 = torch.empty(x.shape[0], device="cuda")
                be.layernorm_fwd(t0 if os.environ.get('PROBE_CONST_INPUT') else t, g, b, y, m2, r2)
                outs.append((y, ref, t, t0))
    torch.cuda.synchronize()
    for y, ref, t, t0 in outs:
        if not torch.equal(y, ref):
            bad += 1
            if bad <= 2:
                rows = ((y - ref).abs().max(1).values > 0).nonzero().flatten()
                print("  LayerNorm output differs:", tuple(y.shape), len(rows), "rows in", [int(rows.min()), int(rows.max())],
                      "| GEMM output differs from solo in", int(((t - t0).abs().max(1).values > 0).sum()), "rows")
print(f"{be.gemm_precision_name}: {bad} of {10 * 16} LayerNorm outputs differ from the solo result")
