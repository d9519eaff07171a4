"""What do the deviating rows of tools/stream_order_probe.py look like?  (test-only library: GRAPPA_HIP_LIB=build/variants/libgrappa_hip_tworow.so,
tools/ln_two_rows_variant.py --no-maxima.)  Four streams run GEMM -> LayerNorm(constant input) chains; for every LayerNorm output that
differs from the solo result: which rows, their role in the two-row kernel (first / second row of a wavefront's trip), how far off, whether
the row statistics (mean, rstd) the kernel stored differ too, and whether the row equals the solo result of some other row.
    GRAPPA_HIP_LIB=... GRAPPA_GEMM_PRECISION=f32_f16x3 python tools/stream_probe_rows.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grappa_amd.backend import get_backend  # noqa: E402

be = get_backend()
torch.manual_seed(0)
Ms, W = [83328, 44325, 28248, 17158], 512
Wm = torch.randn(512, 512, generator=torch.Generator().manual_seed(1)).cuda() / 22.6
data = []
for M in Ms:
    x = torch.randn(M, W, device="cuda") * 2 + 0.3
    g, b = torch.randn(W, device="cuda"), torch.randn(W, device="cuda")
    ref, mean, rstd, t0 = torch.empty_like(x), torch.empty(M, device="cuda"), torch.empty(M, device="cuda"), torch.empty_like(x)
    be.gemm(x, Wm, t0, M=M, N=512, K=512, res=x)
    be.layernorm_fwd(t0, g, b, ref, mean, rstd, amax=False)
    data.append((x, g, b, ref, t0, mean, rstd))
torch.cuda.synchronize()
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in Ms[1:]]
shown = 0
for trial in range(4):
    outs = []
    for st in streams[1:]:
        st.wait_stream(streams[0])
    for (x, g, b, ref, t0, mean, rstd), st in zip(data, streams):
        with torch.cuda.stream(st):
            for rep in range(4):
                t = torch.empty_like(x)
                be.gemm(x, Wm, t, M=x.shape[0], N=512, K=512, res=x)
                y, m2, r2 = torch.empty_like(x), torch.empty(x.shape[0], device="cuda"), torch.empty(x.shape[0], device="cuda")
                be.layernorm_fwd(t0, g, b, y, m2, r2, amax=False)
                outs.append((y, ref, m2, r2, mean, rstd, t0, g, b))
    torch.cuda.synchronize()
    for y, ref, m2, r2, mean, rstd, t0, g, b in outs:
        if torch.equal(y, ref) or shown >= 6:
            continue
        shown += 1
        M = y.shape[0]
        nwaves = min((M + 3) // 4, 2048) * 4
        d = (y - ref).abs()
        rows = (d.max(1).values > 0).nonzero().flatten()
        role = (rows // nwaves) % 2
        print(f"M {M}: {len(rows)} rows differ; role first/second of a trip: {int((role == 0).sum())}/{int((role == 1).sum())}; "
              f"mean differs in {int((m2 != mean).sum())} rows, rstd in {int((r2 != rstd).sum())} "
              f"(of them among the differing rows: {int((m2[rows] != mean[rows]).sum())}, {int((r2[rows] != rstd[rows]).sum())})")
        for r in rows[:4].tolist():
            rel = float(d[r].max() / ref[r].abs().max())
            ncols = int((d[r] > 0).sum())
            cols = (d[r] > 0).nonzero().flatten()
            other = None
            for cand in (r - nwaves, r + nwaves):
                if 0 <= cand < M and torch.equal(y[r], ref[cand]):
                    other = cand
            # recompute the row on the GPU with torch from the stored statistics: does y = (x - mean2) * rstd2 * g + b hold?
            recon = (t0[r] - m2[r]) * r2[r] * g + b
            print(f"   row {r} (wave {r % nwaves}, trip {r // nwaves}): {ncols} columns differ [{int(cols.min())}..{int(cols.max())}], max rel {rel:.2e}; "
                  f"equals the solo result of its trip partner: {other is not None}; mean {float(m2[r]):.6f} vs {float(mean[r]):.6f}, rstd {float(r2[r]):.6f} vs {float(rstd[r]):.6f}; "
                  f"|y - recomputed from ITS stored statistics| {float((y[r] - recon).abs().max()):.2e} (solo row: {float((ref[r] - ((t0[r] - mean[r]) * rstd[r] * g + b)).abs().max()):.2e})")
