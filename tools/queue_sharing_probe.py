"""Does a kernel of this library return the same bits when bf16x3 GEMMs of this library run on three other queues?  (DESIGN.md section 6,
"Streams": a removed two-rows-per-trip LayerNorm did not.)  Every shipped row-wise / tuple / graph kernel is run alone, then beside the
GEMMs, on constant inputs; outputs are compared bit for bit.    python tools/queue_sharing_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grappa_amd.backend import get_backend  # noqa: E402
from grappa_amd.datasets import build_workload  # noqa: E402

be = get_backend()
be.set_gemm_precision("bf16x3")
torch.manual_seed(0)
dev = "cuda"
T, s, Fd = 20832, 4, 512
M = s * T
x = torch.randn(M, Fd, device=dev)
gam, bet = torch.randn(Fd, device=dev), torch.randn(Fd, device=dev)
dy = torch.randn(M, Fd, device=dev)
qkv = torch.randn(M, 3 * Fd, device=dev)
g = build_workload("C2-pubchem-b256", seed=0).to(dev)
plan = g.plan()
N = plan.N
ft = torch.randn(N, 512, device=dev)
a_tok = torch.randn(N, 512, device=dev)
mean0, rstd0 = torch.empty(M, device=dev), torch.empty(M, device=dev)
y0 = torch.empty_like(x)
be.layernorm_fwd(x, gam, bet, y0, mean0, rstd0)


def new(*shape):
    return torch.empty(shape, device=dev)


def op_ln_fwd():
    y, m, r = new(M, Fd), new(M), new(M)
    be.layernorm_fwd(x, gam, bet, y, m, r)
    return [y, m, r]


def op_ln_bwd():
    dx, dg, db = new(M, Fd), torch.zeros(Fd, device=dev), torch.zeros(Fd, device=dev)
    be.layernorm_bwd(dy, x, mean0, rstd0, gam, dx, dg, db, accumulate=True)
    return [dx, dg, db]


def op_actdrop():
    dz = new(M, Fd)
    be.act_dropout_bwd(dy, y0, 0.3, 77, dz)
    return [dz]


def op_seqattn_fwd():
    o = new(M, Fd)
    be.seqattn_fwd(qkv, s, T, 8, o)
    return [o]


def op_seqattn_bwd():
    d = new(M, 3 * Fd)
    be.seqattn_bwd(qkv, dy, s, T, 8, d)
    return [d]


def op_gather_fwd():
    o = new(s * plan.T["n4"], 512)
    be.tuple_gather_fwd(a_tok, plan.idx32["n4"], 4, None, o)
    return [o]


def op_gather_bwd():
    da = new(N, 512)
    be.tuple_gather_bwd(plan.inv_ptr["n4"], plan.inv_rows["n4"], dy[: s * plan.T["n4"]].contiguous(), da, False, False)
    return [da]


def op_perm():
    z = new(2 * T, 4 * Fd)
    be.perm_concat_fwd(x, 4, T, [[0, 1, 2, 3], [3, 2, 1, 0]], z)
    return [z]


def op_gat():
    out, alpha, dft = new(N, 512), new(plan.E, 16), new(N, 512)
    be.gat_fwd(plan, ft, 16, 32, out, alpha)
    be.gat_bwd(plan, ft, out, alpha, ft, 16, 32, dft)
    return [out, alpha, dft]


def op_colsum():
    o = torch.zeros(Fd, device=dev)
    be.colsum(x, o, accumulate=False)
    return [o]


OPS = [("layernorm_fwd", op_ln_fwd), ("layernorm_bwd", op_ln_bwd), ("act_dropout_bwd", op_actdrop), ("seqattn_fwd", op_seqattn_fwd),
       ("seqattn_bwd", op_seqattn_bwd), ("tuple_gather_fwd", op_gather_fwd), ("tuple_gather_bwd", op_gather_bwd), ("perm_concat_fwd", op_perm),
       ("gat_fwd+bwd", op_gat), ("colsum", op_colsum)]
Wm = torch.randn(512, 512, device=dev) / 22.6
others = [(torch.randn(Mo, 512, device=dev), torch.empty(Mo, 512, device=dev)) for Mo in (44325, 28248, 17158)]
side = [torch.cuda.Stream() for _ in others]
main = torch.cuda.current_stream()
for name, op in OPS:
    ref = op()
    torch.cuda.synchronize()
    bad = 0
    for trial in range(6):
        for st, (xo, to) in zip(side, others):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                for _ in range(5):
                    be.gemm(xo, Wm, to, M=xo.shape[0], N=512, K=512, res=xo)
        outs = [op() for _ in range(3)]
        torch.cuda.synchronize()
        bad += sum(int(not all(torch.equal(a, b) for a, b in zip(o, ref))) for o in outs)
    print(f"{name:18s}: {bad} of 18 runs beside GEMMs on three other queues differ from the solo result")
