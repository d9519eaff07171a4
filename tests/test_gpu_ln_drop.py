"""GPU (-m gpu): the dropout backward written by the LayerNorm backward that produces its input (C ABI 9 grappa_layernorm_bwd_drop_f32,
ops._ln_bwd / ops._masked_grad) against the two launches it replaces -- bit for bit, kernel by kernel and through a train step."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,W", [(1, 4), (37, 256), (1000, 512), (5003, 512), (300, 1024), (129, 2048), (64, 84)])
@pytest.mark.parametrize("p", [0.1, 0.5])
def test_layernorm_backward_writes_the_dropout_backward_of_its_result(M, W, p):
    from grappa_amd.backend import get_backend
    be = get_backend()
    gen = torch.Generator(device="cuda").manual_seed(M * 31 + W)
    x = torch.randn(M, W, device="cuda", generator=gen) * 3 + 1
    dy = torch.randn(M, W, device="cuda", generator=gen)
    gamma = torch.rand(W, device="cuda", generator=gen) + 0.5
    mean = x.mean(1)
    rstd = (x.var(1, unbiased=False) + 1e-5).rsqrt()
    seed = 0x1234567 + M
    # the two launches
    dx0, dg0, db0 = torch.empty_like(x), torch.zeros(W, device="cuda"), torch.zeros(W, device="cuda")
    s0 = be.layernorm_bwd(dy, x, mean, rstd, gamma, dx0, dg0, db0, accumulate=False, amax=True)
    dz0 = torch.empty_like(x)
    sz0 = be.act_dropout_bwd(dx0, None, p, seed, dz0)
    # the one
    dx1, dg1, db1 = torch.empty_like(x), torch.zeros(W, device="cuda"), torch.zeros(W, device="cuda")
    keep = be.fuse_ln_drop
    be.fuse_ln_drop = True
    try:
        assert be.drop_fusable(x)
        s1, dz1, sz1 = be.layernorm_bwd(dy, x, mean, rstd, gamma, dx1, dg1, db1, accumulate=False, amax=True, drop=(p, seed))
    finally:
        be.fuse_ln_drop = keep
    torch.cuda.synchronize()
    assert torch.equal(dx0, dx1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)
    assert torch.equal(dz0, dz1)
    assert torch.equal(s0.row, s1.row) and torch.equal(sz0.row, sz1.row)
    # and the mask is the forward pass's: element (r, c) kept iff the host-side hash keeps index r * W + c
    kept = (dz1 != 0) | (dx1 == 0)
    lib = be.lib
    for r, c in [(0, 0), (M - 1, W - 1), (M // 2, W // 3)]:
        assert bool(kept[r, c]) == bool(lib.grappa_dropout_keep(seed, r * W + c, p)), (r, c)
    assert float((dz1[kept] - dx1[kept] / (1 - p)).abs().max()) <= 1e-6 * float(dx1.abs().max())


def test_fused_dropout_backward_is_refused_where_it_cannot_run():
    from grappa_amd.backend import get_backend
    be = get_backend()
    x = torch.randn(8, 64, device="cuda")
    args = (torch.randn_like(x), x, x.mean(1), torch.ones(8, device="cuda"), torch.ones(64, device="cuda"), torch.empty_like(x),
            torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda"))
    with pytest.raises(ValueError):
        be.layernorm_bwd(*args, accumulate=False, drop=(0.0, 1))
    with pytest.raises(ValueError):
        be.layernorm_bwd(*args, accumulate=False, drop=(1.0, 1))
    assert not be.drop_fusable(x.to(torch.bfloat16))


def test_train_step_with_the_fused_dropout_backward_equals_the_separate_launches():
    """same loss and bit-identical gradients with be.fuse_ln_drop on and off (the masked gradient is the same fp32 values either way, so every
    product behind it reads the same operand); five of the six dropout-backward launches of a three-layer head are gone"""
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
    g_cpu = build_batch_from_pool(list(range(200, 232)), n_confs=4, seed=5)
    res, launches = {}, {}
    keep = be.fuse_ln_drop
    try:
        for flag in (False, True):
            be.fuse_ln_drop = flag
            ops.manual_seed(13)
            flat.zero_grad()
            torch.cuda.synchronize()
            be.lib.grappa_launch_count(1)
            g = Energy()(model(g_cpu.to("cuda")))
            loss = MolwiseLoss(gradient_weight=0.8, energy_weight=1.0, param_weight=0.0)(g)
            loss.backward()
            torch.cuda.synchronize()
            launches[flag] = be.lib.grappa_launch_count(1)
            res[flag] = (float(loss.detach()), flat.grad.clone())
    finally:
        be.fuse_ln_drop = keep
    assert res[False][0] == res[True][0]
    assert torch.equal(res[False][1], res[True][1])
    assert float(res[True][1].abs().max()) > 0
    assert launches[True] <= launches[False] - 12, launches           # (four heads, at least three fused dropouts each)
