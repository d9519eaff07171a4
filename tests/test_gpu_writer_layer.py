"""GPU (-m gpu): the fused writer-head layer (C ABI 11 grappa_writer_head_fwd, csrc/writer_layer.hip) against the oracle's restatement of the
reference layer (oracle/ops_ref.py RefBackend.writer_layer: models/network_utils.py:112-133, :44-54) and against the unfused kernel sequence.

bf16 storage configuration: every tensor the unfused kernels store is rounded to bf16 at the same place by the fused kernel, so the two agree
up to the roundings that a different summation order of the fp32 accumulators flips (one bf16 step on a few elements); against the fp32 oracle
run with the same roundings (`rnd`) the same holds.  The tolerance is written with each assert.
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
F = 512


def _params(gen, scale=1.0):
    rn = lambda *s: torch.randn(s, generator=gen, device="cuda")      # noqa: E731
    k = scale / F ** 0.5
    return dict(n1_w=1 + 0.1 * rn(F), n1_b=0.1 * rn(F), w_in=rn(3 * F, F) * k, b_in=0.1 * rn(3 * F), w_o=rn(F, F) * k, b_o=0.1 * rn(F),
                nf_w=1 + 0.1 * rn(F), nf_b=0.1 * rn(F), w1=rn(F, F) * k, b1=0.1 * rn(F), w2=rn(F, F) * k, b2=0.1 * rn(F))


ORDER = ("n1_w", "n1_b", "w_in", "b_in", "w_o", "b_o", "nf_w", "nf_b", "w1", "b1", "w2", "b2")


def _close_bf16(got16, want32, what, frac=0.97, steps=2.0):
    """bf16 tensors of one arithmetic, two summation orders: most elements equal, the rest within `steps` bf16 steps (2^-7 relative) of the
    value -- or of the tensor's typical magnitude where a sum cancels"""
    got = got16.float()
    want16 = want32.to(BF).float()
    same = float((got == want16).float().mean())
    d = (got - want32).abs()
    bound = steps * 2.0 ** -7 * (want32.abs() + want32.pow(2).mean().sqrt())
    bad = int((d > bound).sum())
    assert same >= frac and bad == 0, f"{what}: equal {same:.4f}, outside {steps} steps: {bad}, worst {float((d / bound.clamp_min(1e-30)).max()):.2f}"


def test_pack_weight_layout_is_the_mfma_fragment_order():
    from grappa_amd import _lib
    from grappa_amd.backend import get_backend
    be = get_backend()
    N, K = 48, 96
    w = torch.arange(N * K, device="cuda", dtype=torch.float32).reshape(N, K) % 251      # exact in bf16
    for transposed in (False, True):
        src = w.t().contiguous() if transposed else w
        pk = torch.empty(N * K, dtype=BF, device="cuda")
        assert be.lib.grappa_writer_pack_bytes(N, K, _lib.WRITER_BF16) == N * K * 2
        rc = be.lib.grappa_writer_pack_weight(be._stream(), N, K, src.data_ptr(), src.stride(0), int(transposed), _lib.WRITER_BF16, pk.data_ptr())
        assert rc == 0
        got = pk.float().cpu().numpy().reshape(N // 16, K // 32, 64, 8)
        wn = w.cpu().numpy()
        for nb in range(N // 16):
            for ks in range(K // 32):
                for l in (0, 5, 16, 37, 63):
                    n, k0 = nb * 16 + (l & 15), ks * 32 + 8 * (l >> 4)
                    assert np.array_equal(got[nb, ks, l], wn[n, k0:k0 + 8]), (transposed, nb, ks, l)
    assert be.lib.grappa_writer_pack_weight(be._stream(), 40, 96, w.data_ptr(), K, 0, _lib.WRITER_BF16, pk.data_ptr()) != 0      # N % 16
    assert be.lib.grappa_writer_pack_bytes(48, 100, _lib.WRITER_BF16) == 0


@pytest.mark.parametrize("s,T", [(2, 1), (2, 32), (2, 33), (2, 1000), (3, 1), (3, 21), (3, 22), (3, 707), (4, 1), (4, 16), (4, 17), (4, 1501)])
@pytest.mark.parametrize("drop_p", [0.0, 0.3])
def test_fused_layer_matches_the_oracle_and_the_unfused_kernels(s, T, drop_p):
    if drop_p > 0 and T not in (33, 707, 1501):
        pytest.skip("dropout on the ragged big cases only")
    from oracle.ops_ref import RefBackend
    from grappa_amd.backend import get_backend
    be = get_backend()
    gen = torch.Generator(device="cuda").manual_seed(100 * s + T)
    P = _params(gen)
    M = s * T
    x16 = (torch.randn(M, F, generator=gen, device="cuda") * 1.5 + 0.2).to(BF)
    seed1, seed2 = 1234567 + T, 7654321 + s
    names = ("mean1", "rstd1", "meanf", "rstdf", "x1", "qkv", "att", "x2", "x3", "u")
    sv = {n: torch.full((M,) if n in names[:4] else (M, 3 * F if n == "qkv" else F), float("nan"), device="cuda", dtype=torch.float32 if n in names[:4] else BF)
          for n in names}
    out = torch.full((M, F), float("nan"), device="cuda", dtype=BF)
    args = [P[k] for k in ORDER]
    assert be.writer_layer_ok(x16, s, 8, *args)
    be.writer_layer_fwd(x16, s, T, 8, drop_p, seed1, seed2, *args, out, save=sv)
    torch.cuda.synchronize()
    # inference: the same result without the by-products
    out_i = torch.empty_like(out)
    be.writer_layer_fwd(x16, s, T, 8, drop_p, seed1, seed2, *args, out_i, save=None)
    assert torch.equal(out_i, out)
    # the oracle on the CPU in fp32, rounding where the storage configuration rounds; weights as the kernel sees them (bf16)
    ref = RefBackend()
    Pc = {k: (v.to(BF).float() if k.startswith("w") else v).cpu() for k, v in P.items()}
    want = ref.writer_layer(x16.float().cpu(), s, T, 8, drop_p, seed1, seed2, *[Pc[k] for k in ORDER], rnd=lambda t: t.to(BF).float())
    for n in ("mean1", "rstd1", "meanf", "rstdf"):
        a, b = sv[n].cpu(), want[n]
        tol = 2e-5 if n.endswith("1") else 2e-2      # the second LayerNorm's input carries flipped bf16 roundings
        assert torch.allclose(a, b, rtol=tol, atol=tol * float(b.abs().max())), (n, float((a - b).abs().max()))
    for n in ("x1", "qkv", "att", "x2", "x3", "u"):
        _close_bf16(sv[n].cpu(), want[n], n, frac=0.93 if n in ("x3", "u") else 0.97)
    _close_bf16(out.cpu(), want["out"], "out", frac=0.90, steps=3.0)
    # the unfused kernel sequence of the product on the same inputs (ops.TransformerLayerFn with the fused layer switched off); tables of at
    # most 32 rows take the native fp32 product there (fp32 copies of the operands: other roundings), so they are compared with the oracle only
    if M <= 32:
        return
    from grappa_amd import ops
    be.fused_writer_layer = False
    try:
        ops._INFERENCE["on"] = True
        with torch.no_grad():
            o2 = ops.TransformerLayerFn.apply(x16, s, T, 8, drop_p, seed1, seed2, *args)
    finally:
        be.fused_writer_layer = True
        ops._INFERENCE["on"] = False
    _close_bf16(out.cpu(), o2.float().cpu(), "out vs unfused", frac=0.90, steps=3.0)


def test_training_through_the_fused_forward_equals_the_unfused_layer():
    """forward fused (by-products saved), backward = the unfused kernels: loss and every gradient against the all-unfused layer"""
    from grappa_amd import ops
    from grappa_amd.backend import get_backend
    be = get_backend()
    gen = torch.Generator(device="cuda").manual_seed(5)
    s, T = 3, 500
    res = {}
    for fused in (True, False):
        g2 = torch.Generator(device="cuda").manual_seed(11)
        P = {k: v.clone().requires_grad_(True) for k, v in _params(g2).items()}
        x = (torch.randn(s * T, F, generator=torch.Generator(device="cuda").manual_seed(12), device="cuda")).to(BF).requires_grad_(True)
        be.fused_writer_layer = fused
        try:
            ops._INFERENCE["on"] = False
            y = ops.TransformerLayerFn.apply(x, s, T, 8, 0.1, 77, 88, *[P[k] for k in ORDER])
            wgt = torch.randn(y.shape, generator=torch.Generator(device="cuda").manual_seed(13), device="cuda").to(BF)
            (y.float() * wgt.float()).sum().backward()
        finally:
            be.fused_writer_layer = True
        torch.cuda.synchronize()
        res[fused] = (y.detach().float(), x.grad.float(), {k: v.grad.clone() for k, v in P.items()})
    ya, xa, ga = res[True]
    yb, xb, gb = res[False]
    _close_bf16(ya.to(BF), yb, "out", frac=0.90, steps=3.0)
    assert float((xa - xb).abs().max()) <= 3e-2 * float(xb.abs().max())
    for k in ORDER:
        d = float((ga[k] - gb[k]).abs().max())
        assert d <= 2e-2 * float(gb[k].abs().max()), (k, d, float(gb[k].abs().max()))


def test_fused_layer_refuses_what_it_cannot_run():
    from grappa_amd import _lib
    from grappa_amd.backend import get_backend
    be = get_backend()
    d = _lib.WriterLayerDesc()
    d.s, d.T, d.F, d.nheads, d.dtype = 2, 4, 256, 8, _lib.WRITER_BF16
    assert be.lib.grappa_writer_head_fwd(be._stream(), C.byref(d)) == -1      # F != 512
    d.F, d.s = 512, 5
    assert be.lib.grappa_writer_head_fwd(be._stream(), C.byref(d)) == -1      # s > 4
    d.s = 2
    assert be.lib.grappa_writer_head_fwd(be._stream(), C.byref(d)) == -1      # null pointers
    x32 = torch.zeros(8, 512, device="cuda")
    assert not be.writer_layer_ok(x32, 2, 8)                                  # fp32 storage: the unfused sequence


def _layer_grads(be, ops, s, T, p, fused_fwd, fused_bwd, seed=21):
    P = {k: v.clone().requires_grad_(True) for k, v in _params(torch.Generator(device="cuda").manual_seed(seed)).items()}
    x = (torch.randn(s * T, F, generator=torch.Generator(device="cuda").manual_seed(seed + 1), device="cuda") * 1.3).to(BF).requires_grad_(True)
    keep = (be.fused_writer_layer, be.fused_writer_layer_bwd)
    be.fused_writer_layer, be.fused_writer_layer_bwd = fused_fwd, fused_bwd
    try:
        ops._INFERENCE["on"] = False
        y = ops.TransformerLayerFn.apply(x, s, T, 8, p, 77, 88, *[P[k] for k in ORDER])
        wgt = torch.randn(y.shape, generator=torch.Generator(device="cuda").manual_seed(seed + 2), device="cuda").to(BF)
        (y.float() * wgt.float()).sum().backward()
        be.flush_wgrads() if hasattr(be, "flush_wgrads") else None
    finally:
        be.fused_writer_layer, be.fused_writer_layer_bwd = keep
    torch.cuda.synchronize()
    return y.detach().float(), x.grad.float(), {k: v.grad.clone() for k, v in P.items()}, (x.detach(), {k: v.detach() for k, v in P.items()}, wgt)


@pytest.mark.parametrize("s,T,p", [(2, 1, 0.0), (2, 33, 0.1), (2, 1000, 0.1), (3, 22, 0.1), (3, 707, 0.0), (4, 17, 0.3), (4, 1501, 0.1)])
def test_fused_backward_equals_the_unfused_backward(s, T, p):
    """grappa_writer_head_bwd (one launch: the whole input-gradient chain, operands of the weight gradients as by-products, LayerNorm parameter
    gradients as per-tile partials) against the unfused kernel sequence run over the SAME saved tensors (both forwards fused): dx within 3 bf16
    steps, every parameter gradient within 2e-2 of its largest magnitude (bf16 operands of K = s*T-long sums, two summation orders)."""
    from grappa_amd import ops
    from grappa_amd.backend import get_backend
    be = get_backend()
    ya, xa, ga, _ = _layer_grads(be, ops, s, T, p, True, True)
    yb, xb, gb, _ = _layer_grads(be, ops, s, T, p, True, False)
    assert torch.equal(ya, yb)
    _close_bf16(xa.to(BF), xb, "dx", frac=0.90 if s * T > 32 else 0.0, steps=3.0)      # (<= 32 rows: the unfused products run on fp32 copies)
    for k in ORDER:
        d, scale = float((ga[k] - gb[k]).abs().max()), float(gb[k].abs().max())
        assert d <= 2e-2 * scale, (k, d, scale)


@pytest.mark.parametrize("s,T", [(2, 40), (3, 300), (4, 16), (4, 333)])
def test_fused_layer_gradients_match_the_oracle_autograd(s, T):
    """fused forward + fused backward (bf16 storage) against torch.autograd through the oracle's layer (oracle/cpu_ref.py TransformerLayer =
    the reference's DottedAttWithMLP, models/network_utils.py:112-133) in float64 on the same bf16-rounded weights and input, dropout off:
    the bf16 configuration's gate of SURVEY 8(d), 2e-2 of each tensor's largest magnitude"""
    from grappa_amd import ops
    from grappa_amd.backend import get_backend
    from oracle import cpu_ref
    be = get_backend()
    y, dx, g, (x, P, wgt) = _layer_grads(be, ops, s, T, 0.0, True, True, seed=31)
    ref = cpu_ref.TransformerLayer(F, 8, F, 0.0).double()
    sd = {"norm1.weight": P["n1_w"], "norm1.bias": P["n1_b"], "attn.in_proj_weight": P["w_in"].to(BF), "attn.in_proj_bias": P["b_in"],
          "attn.out_proj.weight": P["w_o"].to(BF), "attn.out_proj.bias": P["b_o"], "ff.norm1.weight": P["nf_w"], "ff.norm1.bias": P["nf_b"],
          "ff.linear1.weight": P["w1"].to(BF), "ff.linear1.bias": P["b1"], "ff.linear2.weight": P["w2"].to(BF), "ff.linear2.bias": P["b2"]}
    ref.load_state_dict({k: v.double().cpu() for k, v in sd.items()})
    ref.eval()
    xr = x.double().cpu().view(s, T, F).clone().requires_grad_(True)
    yr = ref(xr)
    (yr * wgt.double().cpu().view(s, T, F)).sum().backward()
    names = {"n1_w": "norm1.weight", "n1_b": "norm1.bias", "w_in": "attn.in_proj_weight", "b_in": "attn.in_proj_bias", "w_o": "attn.out_proj.weight",
             "b_o": "attn.out_proj.bias", "nf_w": "ff.norm1.weight", "nf_b": "ff.norm1.bias", "w1": "ff.linear1.weight", "b1": "ff.linear1.bias",
             "w2": "ff.linear2.weight", "b2": "ff.linear2.bias"}
    rp = dict(ref.named_parameters())
    tol = 2e-2
    assert float((y.cpu().double() - yr.detach().view(s * T, F)).abs().max()) <= tol * float(yr.abs().max())
    assert float((dx.cpu().double() - xr.grad.view(s * T, F)).abs().max()) <= tol * float(xr.grad.abs().max())
    for k in ORDER:
        r = rp[names[k]].grad
        d = float((g[k].cpu().double() - r).abs().max())
        assert d <= tol * float(r.abs().max()), (k, d, float(r.abs().max()))


@pytest.mark.parametrize("s,p", [(3, 0.0), (4, 0.2)])
def test_first_layer_on_table_rows_through_the_gather_mode_equals_the_unfused_sequence(s, p):
    """ops.ProjFirstLayerFn (rep projector -> LayerNorm + q | k | v once per (atom, position) row -> first transformer layer) with everything
    behind the table-level products as ONE launch forward and ONE backward (the fused layer's GATHER mode: x1 and q | k | v of a token come from
    the table rows) against the unfused sequence: output within 3 bf16 steps, dh and every parameter gradient within 2e-2 of its largest magnitude"""
    from grappa_amd import ops
    from grappa_amd.backend import get_backend
    from grappa_amd.datasets import build_batch_from_pool
    be = get_backend()
    g = build_batch_from_pool(list(range(40, 72)), n_confs=1, seed=0).to("cuda")
    plan = g.plan()
    lvl = {3: "n3", 4: "n4"}[s]
    T, N = plan.T[lvl], plan.N
    assert T > 64 and 4 * N <= 3 * T * 2
    res = []
    for fused in (True, False):
        gen = torch.Generator(device="cuda").manual_seed(77)
        P = {k: v.clone().requires_grad_(True) for k, v in _params(gen).items()}
        w = (torch.randn(F - 1, 256, generator=gen, device="cuda") / 16).requires_grad_(True)
        b = (0.1 * torch.randn(F - 1, generator=gen, device="cuda")).requires_grad_(True)
        h = torch.randn(N, 256, generator=gen, device="cuda").requires_grad_(True)
        pe = torch.tensor([0., 1., 0.] if s == 3 else [0., 1., 1., 0.], device="cuda")
        keep = (be.fused_first_layer, be.fused_writer_layer_bwd)
        be.fused_first_layer = fused
        ops.set_activation_dtype("bf16")
        try:
            ops._INFERENCE["on"] = False
            y = ops.ProjFirstLayerFn.apply(h, w, b, plan.position_tables(lvl), s, T, pe, ops.act_dtype(), 8, p, 5, 6, *[P[k] for k in ORDER])
            wgt = torch.randn(y.shape, generator=torch.Generator(device="cuda").manual_seed(3), device="cuda").to(BF)
            (y.float() * wgt.float()).sum().backward()
            be.flush_wgrads()
        finally:
            ops.set_activation_dtype("f32")
            be.fused_first_layer, be.fused_writer_layer_bwd = keep
        torch.cuda.synchronize()
        res.append((y.detach().float(), h.grad.clone(), w.grad.clone(), b.grad.clone(), {k: v.grad.clone() for k, v in P.items()}))
    (ya, ha, wa, ba, ga), (yb, hb, wb, bb, gb) = res
    _close_bf16(ya.to(BF), yb, "out", frac=0.90, steps=3.0)
    for name, a_, b_ in (("dh", ha, hb), ("dW", wa, wb), ("db", ba, bb)):
        assert float((a_ - b_).abs().max()) <= 2e-2 * float(b_.abs().max()), name
    for k in ORDER:
        d, scale = float((ga[k] - gb[k]).abs().max()), float(gb[k].abs().max())
        assert d <= 2e-2 * scale, (k, d, scale)
