"""GPU (-m gpu): the bf16 STORAGE configuration (BASELINE configs[2]: "bf16, MFMA dense heads").

Activations and activation gradients live in HBM as bf16; every kernel computes in fp32 and rounds once on store.  So
  * op level: a *_bf16 kernel on bf16 inputs returns EXACTLY round_to_bf16(the *_f32 kernel on the same values) -- bit equality;
  * the dense products from bf16 operands (one-plane LDS-DMA kernels) against a float64 product of the same bf16 values;
  * end to end: parameters vs the oracle within SURVEY 8(d)'s bf16 gate (2e-2 relative, floors as in tests/test_gpu_configs.py),
    a full train step (finite loss and gradients, gradient direction = the fp32-grade one), and bit-reproducibility.
"""
import os

import numpy as np
import pytest
import torch

import golden_utils as gu

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _r(x):
    """what a bf16 store makes of fp32 values"""
    return x.to(BF)


def _one_ulp(a16, ref32):
    """kernels that read bf16 rows 8 elements per lane (graph attention, the attention over a tuple's tokens) sum their dot products
    in another order than the fp32 kernels: the fp32 results differ in the last bits, so a few bf16 roundings land one step away"""
    want = _r(ref32)
    same = (a16 == want).float().mean()
    assert float(same) > 0.98, float(same)
    d = (a16.float() - ref32).abs()
    assert bool((d <= ref32.abs() * 2.0 ** -7 + 1e-6 * ref32.abs().max()).all())      # one bf16 step; sums that cancel keep the absolute noise


def test_rowwise_graph_and_tuple_kernels_in_bf16_equal_rounded_fp32_kernels():
    from grappa_amd.backend import get_backend
    from grappa_amd.datasets import build_batch_from_pool
    be = get_backend()
    gen = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: torch.randn(s, generator=gen, device="cuda")      # noqa: E731
    M, W = 1000, 512
    x16 = _r(rnd(M, W) * 2 + 0.3)
    x32 = x16.float()
    gamma, beta = rnd(W) * 0.1 + 1, rnd(W) * 0.1
    # LayerNorm forward / backward
    y16, y32 = torch.empty_like(x16), torch.empty_like(x32)
    st = [torch.empty(M, device="cuda") for _ in range(4)]
    be.layernorm_fwd(x16, gamma, beta, y16, st[0], st[1])
    be.layernorm_fwd(x32, gamma, beta, y32, st[2], st[3])
    assert torch.equal(y16, _r(y32)) and torch.equal(st[0], st[2]) and torch.equal(st[1], st[3])
    dy16 = _r(rnd(M, W))
    dx16, dx32 = torch.empty_like(x16), torch.empty_like(x32)
    dg = [torch.zeros(W, device="cuda") for _ in range(4)]
    be.layernorm_bwd(dy16, x16, st[0], st[1], gamma, dx16, dg[0], dg[1], accumulate=True)
    be.layernorm_bwd(dy16.float(), x32, st[2], st[3], gamma, dx32, dg[2], dg[3], accumulate=True)
    assert torch.equal(dx16, _r(dx32)) and torch.equal(dg[0], dg[2]) and torch.equal(dg[1], dg[3])
    # ELU' / dropout backward
    yelu = _r(torch.nn.functional.elu(rnd(M, W)))
    dz16, dz32 = torch.empty_like(x16), torch.empty_like(x32)
    be.act_dropout_bwd(dy16, yelu, 0.3, 99, dz16)
    be.act_dropout_bwd(dy16.float(), yelu.float(), 0.3, 99, dz32)
    assert torch.equal(dz16, _r(dz32))
    # attention over the s tokens of a tuple
    for s in (2, 3, 4):
        T, F = 700, 512
        qkv16 = _r(rnd(s * T, 3 * F) * 0.5)
        o16, o32 = torch.empty((s * T, F), dtype=BF, device="cuda"), torch.empty((s * T, F), device="cuda")
        be.seqattn_fwd(qkv16, s, T, 8, o16)
        be.seqattn_fwd(qkv16.float(), s, T, 8, o32)
        _one_ulp(o16, o32)
        do16 = _r(rnd(s * T, F))
        dq16, dq32 = torch.empty_like(qkv16), torch.empty((s * T, 3 * F), device="cuda")
        be.seqattn_bwd(qkv16, do16, s, T, 8, dq16)
        be.seqattn_bwd(qkv16.float(), do16.float(), s, T, 8, dq32)
        _one_ulp(dq16, dq32)
        perms = [list(range(s)), list(range(s))[::-1]]
        z16, z32 = torch.empty((2 * T, s * F), dtype=BF, device="cuda"), torch.empty((2 * T, s * F), device="cuda")
        be.perm_concat_fwd(o16, s, T, perms, z16)
        be.perm_concat_fwd(o16.float(), s, T, perms, z32)
        assert torch.equal(z16, _r(z32))
        dxa, dxb = torch.empty_like(o16), torch.empty_like(o32)
        be.perm_concat_bwd(z16, s, T, perms, dxa)
        be.perm_concat_bwd(z32, s, T, perms, dxb)
        assert torch.equal(dxa, _r(dxb))
    # graph attention and the tuple gather on a real batch plan
    g = build_batch_from_pool(list(range(300, 340)), n_confs=2, seed=1).to("cuda")
    plan = g.plan()
    N, H, D = plan.N, 16, 32
    ft16 = _r(rnd(N, H * D))
    m16, m32 = torch.empty_like(ft16), torch.empty((N, H * D), device="cuda")
    al16, al32 = torch.empty((plan.E, H), device="cuda"), torch.empty((plan.E, H), device="cuda")
    be.gat_fwd(plan, ft16, H, D, m16, al16)
    be.gat_fwd(plan, ft16.float(), H, D, m32, al32)
    _one_ulp(m16, m32)
    assert float((al16 - al32).abs().max()) < 1e-6
    dm16 = _r(rnd(N, H * D))
    df16, df32 = torch.empty_like(ft16), torch.empty_like(m32)
    be.gat_bwd(plan, ft16, m16, al16, dm16, H, D, df16)
    be.gat_bwd(plan, ft16.float(), m16.float(), al16, dm16.float(), H, D, df32)
    _one_ulp(df16, df32)
    a16 = _r(rnd(N, 512))
    pe = torch.tensor([0.0, 1.0, 1.0, 0.0], device="cuda")
    T4 = plan.T["n4"]
    xa, xb = torch.empty((4 * T4, 512), dtype=BF, device="cuda"), torch.empty((4 * T4, 512), device="cuda")
    be.tuple_gather_fwd(a16, plan.idx32["n4"], 4, pe, xa)
    be.tuple_gather_fwd(a16.float(), plan.idx32["n4"], 4, pe, xb)
    assert torch.equal(xa, _r(xb))
    da16, da32 = torch.empty_like(a16), torch.empty((N, 512), device="cuda")
    be.tuple_gather_bwd(plan.inv_ptr["n4"], plan.inv_rows["n4"], xa, da16, True, False)
    be.tuple_gather_bwd(plan.inv_ptr["n4"], plan.inv_rows["n4"], xb, da32, True, False)
    assert torch.equal(da16, _r(da32))


def test_dense_products_from_bf16_operands():
    """forward (bf16 activations x weight), dgrad (transposed weight planes) and wgrad (both operands bf16 activations, ragged K, fused
    bias gradient) with bf16 / fp32 outputs and bf16 epilogue tensors, against float64 on the same bf16 values"""
    from grappa_amd.backend import get_backend
    be = get_backend()
    gen = torch.Generator(device="cuda").manual_seed(2)
    rnd = lambda *s: torch.randn(s, generator=gen, device="cuda")      # noqa: E731
    M, N, K = 5000, 512, 512
    x = _r(rnd(M, K))
    w = (rnd(N, K) / K ** 0.5).requires_grad_(True)
    bias, res = rnd(N), _r(rnd(M, N))
    w16 = w.detach().to(BF).double()
    # forward: bf16 out with bias + ELU + bf16 residual
    out = torch.empty((M, N), dtype=BF, device="cuda")
    be.gemm(x, w, out, M=M, N=N, K=K, bias=bias, act=1, res=res)
    want = torch.nn.functional.elu(x.double() @ w16.t() + bias.double()) + res.double()
    assert float((out.double() - want).abs().max() / want.abs().max()) < 6e-3          # one bf16 rounding of the result
    out32 = torch.empty((M, N), device="cuda")
    be.gemm(x, w, out32, M=M, N=N, K=K, bias=bias)
    want = x.double() @ w16.t() + bias.double()
    assert float((out32.double() - want).abs().max() / want.abs().max()) < 2e-6        # fp32 accumulation of exact bf16 products
    # dgrad with ELU' from a bf16 tensor
    dy = _r(rnd(M, N))
    u = _r(torch.nn.functional.elu(rnd(M, K)))
    dx = torch.empty((M, K), dtype=BF, device="cuda")
    be.gemm(dy, w, dx, M=M, N=K, K=N, b_kcontig=False, aux=u)
    want = (dy.double() @ w16) * torch.where(u > 0, torch.ones_like(u), u + 1).double()
    assert float((dx.double() - want).abs().max() / want.abs().max()) < 6e-3
    # wgrad: ragged K (no padding anywhere), fp32 accumulate into the gradient, fused bias gradient
    T = 33333
    dz, xx = _r(rnd(T, 512)), _r(rnd(T, 256))
    dW, db = torch.ones((512, 256), device="cuda"), torch.zeros(512, device="cuda")
    be.gemm(dz, xx, dW, M=512, N=256, K=T, a_kcontig=False, b_kcontig=False, accumulate=True, a_colsum=db)
    want = dz.double().t() @ xx.double() + 1.0
    assert float((dW.double() - want).abs().max() / want.abs().max()) < 2e-6
    assert float((db.double() - dz.double().sum(0)).abs().max() / dz.double().sum(0).abs().max()) < 2e-6
    # (pre-dropout copy, final value) pair in bf16
    pre, fin = torch.empty((M, N), dtype=BF, device="cuda"), torch.empty((M, N), dtype=BF, device="cuda")
    be.gemm(x, w, pre, M=M, N=N, K=K, bias=bias, act=1, drop_p=0.5, drop_seed=7, res=res, out2=fin)
    y = torch.nn.functional.elu(x.double() @ w16.t() + bias.double())
    assert float((pre.double() - y).abs().max() / y.abs().max()) < 6e-3
    kept = ((fin.double() - res.double()).abs() > 1e-3)
    assert 0.4 < float(kept.double().mean()) < 0.6


def _bf16_mode(on: bool):
    from grappa_amd import ops
    from grappa_amd.backend import get_backend
    be = get_backend()
    ops.set_activation_dtype("bf16" if on else "f32")
    from grappa_amd.backend import DEFAULT_GEMM_PRECISION
    be.set_gemm_precision("bf16" if on else os.environ.get("GRAPPA_GEMM_PRECISION", DEFAULT_GEMM_PRECISION))      # off = the backend's own default


@pytest.mark.parametrize("n_conv", [0, 2])
def test_bf16_configuration_end_to_end_against_the_oracle(n_conv):
    """n_conv = 0: the production model (grappa-1.2, seven attention blocks); n_conv = 2: two SAGE blocks in front of them (the
    grappa-1.1 layout, reference models/graph_attention.py:383-415) -- the bf16 storage configuration covers both (round 3)"""
    from grappa_amd import Energy, MolwiseLoss, get_default_model_config, model_from_config, ops
    from grappa_amd.datasets import build_batch_from_pool
    from grappa_amd.optim import FlatParams
    from oracle import cpu_ref
    cfg = get_default_model_config()
    if n_conv:
        cfg.update(gnn_convolutions=n_conv, gnn_attentional_layers=3)
    model = model_from_config(cfg)
    sd = gu.keyed_state_dict(model)
    model.load_state_dict(sd)
    model = model.to("cuda").eval()
    flat = FlatParams(model)
    ids = list(range(600, 664))
    lk = dict(gradient_weight=0.8, energy_weight=1.0, param_weight=0.0, proper_regularisation=1e-3)
    ref = cpu_ref.RefGrappaModel(**cfg)
    ref.load_state_dict(sd)
    ref.eval()
    with torch.no_grad():
        rg = ref(build_batch_from_pool(ids, n_confs=4, seed=3))
    runs = {}
    try:
        for mode in (False, True, True):
            _bf16_mode(mode)
            flat.zero_grad()
            g = Energy()(model(build_batch_from_pool(ids, n_confs=4, seed=3).to("cuda")))
            loss = MolwiseLoss(**lk)(g)
            loss.backward()
            torch.cuda.synchronize()
            runs.setdefault(mode, []).append((g, loss.detach().clone(), flat.grad.clone()))
    finally:
        _bf16_mode(False)
    g16, loss16, grad16 = runs[True][0]
    assert g16.nodes["n1"].data["h"].dtype == torch.float32 and g16.nodes["n2"].data["k"].dtype == torch.float32
    # parameters vs the oracle: SURVEY 8(d)'s bf16 gate (2e-2 relative; floors = half the output scale of each head)
    floors = {"n2": 1.0, "n3": 1.0, "n4": 0.5, "n4_improper": 2.0}
    worst = {}
    for lvl in ["n2", "n3", "n4", "n4_improper"]:
        e = gu.rel_err(g16.nodes[lvl].data["k"].detach().cpu(), rg.nodes[lvl].data["k"].numpy(), floors[lvl])
        worst[lvl + "_k"] = e
        assert e < 2e-2, (lvl, e)
        if lvl in ("n2", "n3"):
            e = gu.rel_err(g16.nodes[lvl].data["eq"].detach().cpu(), rg.nodes[lvl].data["eq"].numpy(), 1e-4)
            worst[lvl + "_eq"] = e
            assert e < 2e-2, (lvl, e)
    print("bf16 storage configuration, worst relative parameter errors vs oracle:", {k: f"{v:.2e}" for k, v in worst.items()})
    # the train step: finite, and the gradient points where the fp32-grade gradient points
    _, loss32, grad32 = runs[False][0]
    assert torch.isfinite(loss16) and torch.isfinite(grad16).all()
    assert abs(float(loss16) - float(loss32)) < 5e-2 * abs(float(loss32))
    cos = float((grad16.double() * grad32.double()).sum() / (grad16.double().norm() * grad32.double().norm()))
    print("bf16 storage configuration: loss", float(loss16), "vs", float(loss32), "; cosine(gradient, fp32-grade gradient) =", cos)
    assert cos > 0.98
    # bit-reproducible
    assert torch.equal(runs[True][0][1], runs[True][1][1]) and torch.equal(runs[True][0][2], runs[True][1][2])
