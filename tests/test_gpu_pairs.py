"""GPU (-m gpu): the pair-format product (csrc/gemm_pairs.hip; include/grappa_hip.h, ABI 5) through the C ABI: operands split once by
grappa_split_pairs_f32 (rows of A, rows of W or -- for the input-gradient layout -- rows of W^T) give the SAME BITS as the fp32-operand
fp16-split product of the same K split, and float64-grade errors on ragged shapes, scaled rows and the fused epilogues.  Where K % 16 == 0
`check` also runs the "weight pairs" form (fp32 A as it is + the pairs of W) and asserts the all-pairs product's bits."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("M,N,K,dgrad,scaled,epi", [
    (256, 128, 32, False, False, {}), (300, 200, 64, False, False, {}), (257, 129, 96, False, False, {}), (1000, 512, 512, False, False, {}),
    (4096, 1536, 512, False, False, {}), (777, 256, 1536, False, False, {}), (1000, 512, 512, True, False, {}),
    (1000, 512, 512, False, True, {}), (1000, 512, 512, False, False, dict(bias=1, act=1)), (1000, 512, 512, False, False, dict(bias=1, res=1)),
    (33, 40, 48, False, False, {}), (20000, 64, 512, False, False, {}), (3000, 512, 1536, True, False, {})])
def test_pair_format_product_equals_the_fp32_operand_product_bit_for_bit(M, N, K, dgrad, scaled, epi):
    import gemm_pairs_check as gp
    gen = torch.Generator(device="cuda")
    gen.manual_seed(M * 7 + N)
    gp.check(M, N, K, gen, dgrad=dgrad, scale_rows=scaled, **epi)                 # the shipped plans: asserts the float64 error inside
    # the two kernels plan their tiles and K cuts independently (the pair kernel: 256 x 128 or, since round 4, 128 x 128): with K in one
    # piece for both, the bits agree
    gp.EXTRA = {"plan_nsplit": 1}                                                    # (a per-call option of the descriptor since C ABI 10)
    try:
        gen.manual_seed(M * 7 + N)
        same = gp.check(M, N, K, gen, dgrad=dgrad, scale_rows=scaled, **epi)      # (returns bit equality)
    finally:
        gp.EXTRA = {}
    assert same


@pytest.mark.parametrize("cfg", [6, 7, 8])
@pytest.mark.parametrize("M,N,K,nsplit", [(300, 200, 160, 2), (1000, 512, 400, 3), (2000, 256, 1104, 5), (77, 130, 96, 2), (5000, 512, 512, 7)])
def test_split_k_ranges_of_the_pair_kernels_do_not_overlap(cfg, M, N, K, nsplit):
    """every tile of the pair kernels under a forced split-K whose ranges are NOT multiples of 32 columns when cut evenly (K / nsplit = 80, 133.3,
    220.8, 48, 73.1): the pinned-pipeline kernel walks pairs of 16-column slabs and once read 16 columns of the NEXT range (found in round 5 when the
    planner began to choose such cuts: wrong improper force constants in Grappa.predict); `check` asserts the float64-grade error inside"""
    import gemm_pairs_check as gp
    gen = torch.Generator(device="cuda")
    gen.manual_seed(K + nsplit)
    A = torch.randn((M, K), generator=gen, device="cuda")
    W = torch.randn((N, K), generator=gen, device="cuda") * 0.05
    bias = torch.randn(N, generator=gen, device="cuda")
    am_a, am_b = gp.amax(A), gp.amax(W)
    ap, bp = gp.split_pairs(A, am_a), gp.split_pairs(W, am_b)
    out = torch.full((M, N), float("nan"), device="cuda")
    gp.EXTRA = {"plan_cfg": cfg + 1, "plan_nsplit": nsplit, "plan_tail": 2}
    try:
        gp.gemm(ap, bp, out, M, N, K, am_a, am_b, True, bias=bias)
    finally:
        gp.EXTRA = {}
    torch.cuda.synchronize()
    ref = A.double() @ W.double().t() + bias.double()
    err = ((out.double() - ref).abs() / ref.abs().amax(dim=1, keepdim=True)).max().item()
    assert err < 3e-6, err


def test_randomised_shapes_formats_plans_and_epilogues_stay_inside_the_arithmetic_error():
    """48 seeded random cases (tools/gemm_fuzz.py: shape, operand format, forced tile / split-K / tail launch, epilogue) against float64"""
    import gemm_fuzz
    worst, refused, bad = gemm_fuzz.run(48, 0, verbose=False)
    assert not bad, bad
    assert refused < 24, refused


def test_pair_layout_and_rejections():
    """element (r, k): HI at r * ld + 32 * (k // 16) + k % 16, LO 16 further; (HI + LO) * 2^-s reproduces the value to 2^-22 of the row maximum"""
    import ctypes as C
    import gemm_pairs_check as gp
    from grappa_amd import _lib
    gen = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn((37, 50), generator=gen, device="cuda") * torch.exp2(torch.randint(-20, 20, (37, 1), generator=gen, device="cuda").float())
    am = gp.amax(x)
    p = gp.split_pairs(x, am)
    assert p.shape == (37, 2 * 64) and p.dtype == torch.float16
    k = torch.arange(50, device="cuda")
    col = 32 * (k // 16) + k % 16
    shift = 141 - ((am >> 23) & 0xff)
    back = (p[:, col].double() + p[:, col + 16].double()) * torch.exp2(-shift.double())[:, None]
    rowmax = x.abs().amax(dim=1, keepdim=True).double()
    assert float(((back - x.double()).abs() / rowmax).max()) < 2.0 ** -21
    assert float(p[:, col].abs().max()) < 2.0 ** 15 and float(p[:, col].abs().amax(dim=1).min()) >= 2.0 ** 14       # every row's largest element in [2^14, 2^15)
    pad = torch.ones(128, dtype=torch.bool, device="cuda")
    pad[col] = False
    pad[col + 16] = False
    assert float(p[:, pad].abs().max()) == 0.0                       # the padding beyond C stays zero
    # one operand in pairs, the other not: refused
    d = _lib.GemmDesc()
    out = torch.empty((37, 37), device="cuda")
    d.M, d.N, d.K, d.a_kcontig, d.b_kcontig = 37, 37, 50, 1, 1
    d.A, d.lda, d.a_planes = p.data_ptr(), p.stride(0), 1
    d.B, d.ldb = x.data_ptr(), x.stride(0)
    d.C, d.ldc = out.data_ptr(), out.stride(0)
    d.a_amax, d.b_amax = am.data_ptr(), am.data_ptr()
    d.precision = _lib.GEMM_PRECISIONS["f32_f16x3"]
    ws = gp.ws_for(37, 37, 50)
    assert gp.lib.grappa_gemm_f32(gp.stream(), C.byref(d), ws.data_ptr(), ws.numel()) == -1


@pytest.mark.parametrize("M,W", [(1000, 512), (77, 256), (4097, 2048), (300, 32), (5, 512)])
def test_layernorm_writes_its_rows_in_the_pair_format(M, W):
    """grappa_layernorm_fwd_pairs_f32: y, mean, rstd and the row maxima are the bits of the plain kernel; the pairs are the bits of
    grappa_split_pairs_f32 on that y; y = NULL writes the pairs alone"""
    import gemm_pairs_check as gp
    from grappa_amd.backend import get_backend
    be = get_backend()
    gen = torch.Generator(device="cuda").manual_seed(M + W)
    x = torch.randn((M, W), generator=gen, device="cuda") * torch.exp2(torch.randint(-6, 6, (M, 1), generator=gen, device="cuda").float())
    g, b = torch.randn(W, generator=gen, device="cuda"), torch.randn(W, generator=gen, device="cuda")
    y0, m0, r0 = torch.empty_like(x), torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    s0 = be.layernorm_fwd(x, g, b, y0, m0, r0, amax=True)
    y1, m1, r1 = torch.full_like(x, float("nan")), torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    s1 = be.layernorm_fwd(x, g, b, y1, m1, r1, pairs=True)
    assert torch.equal(y0, y1) and torch.equal(m0, m1) and torch.equal(r0, r1) and torch.equal(s0.row, s1.row)
    assert s1.pairs.shape == (M, 2 * W) and torch.equal(s1.pairs, gp.split_pairs(y0, s0.row))
    s2 = be.layernorm_fwd(x, g, b, None, None, None, pairs=True)
    assert torch.equal(s2.pairs, s1.pairs) and torch.equal(s2.row, s1.row)
    with pytest.raises(ValueError):
        be.layernorm_fwd(x[:, :24].contiguous(), g[:24].contiguous(), b[:24].contiguous(), None, None, None, pairs=True)


@pytest.mark.parametrize("s,T,heads,F", [(4, 500, 8, 512), (3, 77, 8, 512), (2, 1000, 4, 256), (1, 40, 2, 64), (4, 3, 8, 512)])
def test_tuple_attention_writes_its_output_in_the_pair_format(s, T, heads, F):
    import gemm_pairs_check as gp
    from grappa_amd.backend import get_backend
    be = get_backend()
    gen = torch.Generator(device="cuda").manual_seed(s * 1000 + T)
    qkv = torch.randn((s * T, 3 * F), generator=gen, device="cuda")
    att = torch.empty((s * T, F), device="cuda")
    s0 = be.seqattn_fwd(qkv, s, T, heads, att, amax=True)
    s1 = be.seqattn_fwd(qkv, s, T, heads, None, pairs=True)
    assert torch.equal(s0.row, s1.row) and torch.equal(s1.pairs, gp.split_pairs(att, s0.row))


def test_inference_through_the_pair_format_equals_the_fp32_operand_path():
    """eval / no_grad: LayerNorm and the tuple attention hand their rows to the product behind them in the pair format (the default);
    GRAPPA_INFERENCE_PAIRS=0 (be.inference_pairs = False) keeps fp32 operands.  Same values: the pair kernel computes the same partial
    products, bit for bit where both kernels cut K alike.  With gradients enabled the producers write pairs only where the training configuration allows them (be.training_pairs)."""
    from grappa_amd import get_default_model_config, model_from_config
    from grappa_amd.backend import get_backend
    from grappa_amd.datasets import build_batch_from_pool
    import golden_utils as gu
    be = get_backend()
    model = model_from_config(get_default_model_config())
    model.load_state_dict(gu.keyed_state_dict(model))
    model = model.to("cuda").eval()
    g_cpu = build_batch_from_pool(list(range(100, 164)), n_confs=2, seed=3)
    outs = {}
    taken = {"n": 0}
    ln = be.layernorm_fwd
    be.layernorm_fwd = lambda *a, **k: (taken.__setitem__("n", taken["n"] + bool(k.get("pairs"))), ln(*a, **k))[1]
    try:
        for flag in (True, False):
            be.inference_pairs = flag
            with torch.no_grad():
                g = model(g_cpu.to("cuda"))
            outs[flag] = {(lvl, k): g.nodes[lvl].data[k].clone() for lvl in ("n2", "n3", "n4", "n4_improper") for k in ("k", "eq") if k in g.nodes[lvl].data}
            outs[flag]["h"] = g.nodes["n1"].data["h"].clone()
    finally:
        be.inference_pairs = True
        del be.layernorm_fwd
    from grappa_amd import ops
    assert taken["n"] >= 40, (taken, be.gemm_precision_name, be.inference_pairs, ops.act_dtype(), ops._INFERENCE)      # the pair path was taken (and only with the flag on)
    for key, a in outs[True].items():
        b = outs[False][key]
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()), key
    # with gradients enabled: pairs where the training configuration allows them (round 4: be.training_pairs, tables of >= be.pairs_min_rows
    # rows), fp32 rows everywhere with the switch off
    for flag in (False, True):
        taken["n"] = 0
        be.layernorm_fwd = lambda *a, **k: (taken.__setitem__("n", taken["n"] + bool(k.get("pairs"))), ln(*a, **k))[1]
        be.training_pairs = flag
        try:
            model(g_cpu.to("cuda"))
        finally:
            del be.layernorm_fwd
            be.training_pairs = True
        assert (taken["n"] > 0) == flag, (flag, taken)


@pytest.mark.parametrize("M,N,K,drop", [(1000, 512, 512, 0.0), (5000, 256, 512, 0.3), (257, 512, 256, 0.0), (40, 64, 64, 0.0)])
def test_product_recomputes_a_layernorm_residual_in_its_epilogue(M, N, K, drop):
    """res_ln: the residual is given as the rows BEFORE a LayerNorm and the epilogue adds LayerNorm(rows) -- the bits of the product that
    is handed the normalised rows themselves (fast class, generic row walk and split-K reduction alike)"""
    from grappa_amd.backend import get_backend
    be = get_backend()
    gen = torch.Generator(device="cuda").manual_seed(M + N)
    a, w, bias = torch.randn((M, K), generator=gen, device="cuda"), torch.randn((N, K), generator=gen, device="cuda") * 0.05, torch.randn(N, generator=gen, device="cuda")
    x = torch.randn((M, N), generator=gen, device="cuda") * 3 + 1
    g, b = torch.randn(N, generator=gen, device="cuda"), torch.randn(N, generator=gen, device="cuda")
    y, mean, rstd = torch.empty_like(x), torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    be.layernorm_fwd(x, g, b, y, mean, rstd)
    want, got = torch.empty((M, N), device="cuda"), torch.empty((M, N), device="cuda")
    be.gemm(a, w, want, M=M, N=N, K=K, bias=bias, drop_p=drop, drop_seed=7, res=y)
    be.gemm(a, w, got, M=M, N=N, K=K, bias=bias, drop_p=drop, drop_seed=7, res=x, res_ln=(mean, rstd, g, b))
    assert torch.equal(want, got)
    if M > 64:                       # forced split-K: the reduction kernel applies the epilogue
        be.plan_override = (-1, 2, -1)
        try:
            be.gemm(a, w, got, M=M, N=N, K=K, bias=bias, drop_p=drop, drop_seed=7, res=x, res_ln=(mean, rstd, g, b))
            be.gemm(a, w, want, M=M, N=N, K=K, bias=bias, drop_p=drop, drop_seed=7, res=y)
        finally:
            be.plan_override = None
        assert torch.equal(want, got)
    if M > 64:                       # the generic row walk: activation + second output (the GNN's feed-forward form)
        pre_w, pre_g = torch.empty((M, N), device="cuda"), torch.empty((M, N), device="cuda")
        be.gemm(a, w, pre_w, M=M, N=N, K=K, bias=bias, act=1, drop_p=drop, drop_seed=7, res=y, out2=want)
        be.gemm(a, w, pre_g, M=M, N=N, K=K, bias=bias, act=1, drop_p=drop, drop_seed=7, res=x, res_ln=(mean, rstd, g, b), out2=got)
        assert torch.equal(want, got) and torch.equal(pre_w, pre_g)
    with pytest.raises(ValueError):
        be.gemm(a, w, got, M=M, N=N, K=K, res=None, res_ln=(mean, rstd, g, b))


@pytest.mark.parametrize("variant", ["plain", "no_layer_norm", "no_self_interaction", "conv_blocks", "wide_attention", "odd_width", "wrong_symmetry"])
def test_inference_pair_path_on_model_variants(variant):
    """the constructor options of the reference's model under no_grad with the pair path on and off: same parameters (the path must step
    aside by itself where a width, a missing LayerNorm or an attention wider than 512 columns rules it out)"""
    from grappa_amd import GrappaModel
    from grappa_amd.backend import get_backend
    from grappa_amd.datasets import build_batch_from_pool
    be = get_backend()
    cfg = dict(graph_node_features=64, in_feat_name=["atomic_number", "partial_charge", "ring_encoding", "degree", "charge_model"],
               gnn_width=128, gnn_attentional_layers=2, gnn_convolutions=0, gnn_attention_heads=4,
               **{f"{h}_{k}": v for h in ("bond", "angle", "proper", "improper")
                  for k, v in (("transformer_depth", 2), ("n_heads", 4), ("transformer_width", 128), ("symmetriser_depth", 3), ("symmetriser_width", 64))})
    if variant == "no_layer_norm":
        cfg["layer_norm"] = False
    elif variant == "no_self_interaction":
        cfg["self_interaction"] = False
    elif variant == "conv_blocks":
        cfg.update(gnn_convolutions=2, gnn_attentional_layers=1)
    elif variant == "wide_attention":
        cfg.update(bond_transformer_width=1024, bond_n_heads=8)          # attention rows of 1024 columns (> 512: no pair output); 2 x 1024 = the widest LayerNorm row
    elif variant == "odd_width":
        cfg.update(angle_transformer_width=80, angle_n_heads=5, angle_symmetriser_width=40)            # 80, 40: not multiples of 32 (head width 16)
    elif variant == "wrong_symmetry":
        cfg["wrong_symmetry"] = True
    torch.manual_seed(1)
    model = GrappaModel(**cfg).to("cuda").eval()
    g_cpu = build_batch_from_pool(list(range(200, 232)), n_confs=1, seed=5)
    outs = {}
    try:
        for flag in (True, False):
            be.inference_pairs = flag
            with torch.no_grad():
                g = model(g_cpu.to("cuda"))
            outs[flag] = [g.nodes[lvl].data[k].clone() for lvl in ("n2", "n3", "n4", "n4_improper") for k in ("k", "eq") if k in g.nodes[lvl].data]
    finally:
        be.inference_pairs = True
    for a, b in zip(outs[True], outs[False]):
        assert torch.isfinite(a).all() and float((a - b).abs().max()) <= 5e-6 * max(float(b.abs().max()), 1e-6), variant


@pytest.mark.parametrize("fmt", ["pz", "px", "both"])
def test_weight_gradient_from_pair_operands(fmt):
    """C ABI 8: the grouped weight-gradient product reads an operand that its producer wrote in the pair format (every token row under its own
    scale) and moves the rows onto the tensor's scale with exact fp16 multiplications: the result equals the fp32-operand product -- bit
    for bit while no element falls into the fp16 denormals under the tensor's scale, to fp32 rounding otherwise -- and so does the bias gradient"""
    from grappa_amd.backend import Amax, get_backend
    be = get_backend()
    torch.manual_seed(3)
    T, Np, Kp = 5000, 512, 256
    for spread in (0.0, 12.0):                      # rows within one binade of each other / spread over 2^-36 .. 1 of the largest
        dz = torch.randn(T, Np, device="cuda") * torch.exp2(-spread * 3 * torch.rand(T, 1, device="cuda"))
        x = torch.randn(T, Kp, device="cuda") * torch.exp2(-spread * torch.rand(T, 1, device="cuda"))
        want_w, want_b = torch.zeros(Np, Kp, device="cuda"), torch.zeros(Np, device="cuda")
        be.gemm_wgrad(dz, x, want_w, want_b)
        be.flush_wgrads()
        rz, rx = be.to_pairs(dz), be.to_pairs(x)
        got_w, got_b = torch.zeros(Np, Kp, device="cuda"), torch.zeros(Np, device="cuda")
        be.gemm_wgrad(None if fmt in ("pz", "both") else dz, None if fmt in ("px", "both") else x, got_w, got_b,
                      dz_scales=rz if fmt in ("pz", "both") else None, x_scales=rx if fmt in ("px", "both") else None)
        be.flush_wgrads()
        torch.cuda.synchronize()
        ref = dz.double().T @ x.double()
        scale = float(ref.abs().max())
        err_pairs, err_f32 = float((got_w.double() - ref).abs().max()) / scale, float((want_w.double() - ref).abs().max()) / scale
        assert err_pairs <= max(2.0 * err_f32, 2e-7), (fmt, spread, err_pairs, err_f32)
        # (not bit for bit: a LO half that is an fp16 denormal under its row's scale is rounded once more when it moves to the tensor's)
        assert float((got_w - want_w).abs().max()) <= 2e-7 * scale, (fmt, spread, float((got_w - want_w).abs().max()) / scale)
        bref = dz.double().sum(0)
        assert float((got_b.double() - bref).abs().max()) <= 2e-6 * float(bref.abs().max()), fmt


@pytest.mark.parametrize("min_rows,bwd", [(12288, False), (0, True), (12288, True)])
def test_training_through_the_pair_format_equals_the_fp32_operand_path(min_rows, bwd):
    """round 4: in training LayerNorm and the tuple attention (bwd: the dropout backward too, GRAPPA_BACKWARD_PAIRS) write the operands of the
    products behind them in the pair format ONLY; forward / input-gradient products read them by LDS-DMA (csrc/gemm_pairs.hip), weight-gradient
    products through C ABI 8.  Same loss and gradients as with fp32 operands (be.training_pairs = False) to fp32 rounding, with dropout on (same
    counter-based masks); min_rows = 0: the GNN's atom rows too"""
    from grappa_amd import Energy, MolwiseLoss, get_default_model_config, model_from_config, ops
    from grappa_amd.backend import get_backend
    from grappa_amd.datasets import build_batch_from_pool
    from grappa_amd.optim import FlatParams
    import golden_utils as gu
    be = get_backend()
    model = model_from_config(get_default_model_config())
    model.load_state_dict(gu.keyed_state_dict(model))
    model = model.to("cuda").train()
    flat = FlatParams(model)
    g_cpu = build_batch_from_pool(list(range(100, 164)), n_confs=4, seed=3)
    res = {}
    keep, keep_bwd = be.pairs_min_rows, be.backward_pairs
    try:
        for flag in (False, True):
            be.training_pairs, be.pairs_min_rows, be.backward_pairs = flag, min_rows, bwd
            ops.manual_seed(11)
            flat.zero_grad()
            g = Energy()(model(g_cpu.to("cuda")))
            loss = MolwiseLoss(gradient_weight=0.8, energy_weight=1.0, param_weight=0.0)(g)
            loss.backward()
            torch.cuda.synchronize()
            res[flag] = (float(loss), flat.grad.clone(), g.nodes["n4"].data["k"].detach().clone(), g.nodes["n2"].data["eq"].detach().clone())
    finally:
        be.training_pairs, be.pairs_min_rows, be.backward_pairs = True, keep, keep_bwd
    (l0, g0, k0, e0), (l1, g1, k1, e1) = res[False], res[True]
    assert abs(l1 - l0) <= 1e-5 * abs(l0), (l0, l1)
    assert float((k1 - k0).abs().max()) <= 1e-5 * float(k0.abs().max()) and float((e1 - e0).abs().max()) <= 1e-5 * float(e0.abs().max())
    # per parameter tensor, relative to its largest gradient entry
    for name, p in model.named_parameters():
        a, b = flat.grad_view(p) if hasattr(flat, "grad_view") else None, None
        break
    worst = 0.0
    off = 0
    for p in model.parameters():
        home = getattr(p, "_grappa_flat", None)
        lo = home[1]
        a, b = g1[lo:lo + p.numel()], g0[lo:lo + p.numel()]
        scale = float(b.abs().max())
        if scale > 0:
            worst = max(worst, float((a - b).abs().max()) / scale)
    assert worst <= 2e-5, worst


def test_grouped_weight_gradients_of_mixed_operand_formats_in_one_launch():
    """one grouped launch over products whose operands come as fp32 or as pairs, product by product (C ABI 8, the kernel reads each product's
    own flags): the same results as launching every product alone in its own format"""
    from grappa_amd.backend import get_backend
    be = get_backend()
    torch.manual_seed(5)
    T = 3000
    shapes = [(512, 512), (1536, 512), (256, 1024), (512, 256)]
    fmts = [(True, False), (False, True), (True, True), (False, False)]
    ops_, want = [], []
    for (Np, Kp), (pz, px) in zip(shapes, fmts):
        dz, x = torch.randn(T, Np, device="cuda"), torch.randn(T, Kp, device="cuda") * 3
        rz, rx = be.to_pairs(dz), be.to_pairs(x)
        alone = torch.zeros(Np, Kp, device="cuda")
        be.gemm_wgrad(None if pz else dz, None if px else x, alone, None, dz_scales=rz if pz else None, x_scales=rx if px else None)
        be.flush_wgrads()
        want.append(alone)
        ops_.append((dz, x, rz, rx, pz, px))
    outs = [torch.zeros_like(w) for w in want]
    items = []
    for (dz, x, rz, rx, pz, px), out in zip(ops_, outs):
        am = (rz if pz else be.amax(dz, None, rows=True), rx if px else be.amax(x, None, rows=True))
        items.append((None if pz else dz, None if px else x, out, None, am, rz.pairs if pz else None, rx.pairs if px else None))
    be._launch_wgrad_group(items)
    torch.cuda.synchronize()
    for got, w, (dz, x, *_r) in zip(outs, want, ops_):
        ref = dz.double().T @ x.double()
        scale = float(ref.abs().max())
        assert float((got.double() - ref).abs().max()) <= 2e-6 * scale  # (fp32 accumulation over 3,000 tokens: measured 5e-7)
        assert float((got - w).abs().max()) <= 2e-6 * scale          # (another K cut than alone: fp32 rounding, not bits)
