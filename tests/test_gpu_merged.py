"""GPU (-m gpu): the writer heads layer-locked (ops.MultiTransformerLayerFn / MultiSymmetriserFn: one grouped launch per product over the four
heads, C ABI 8 grappa_gemm_f32_group) against the head-by-head path: same parameters, loss and gradients to fp32 rounding (the grouped
launch fixes the tile and never cuts K; head by head the planner may), with dropout on (same counter-based masks: the seeds are drawn in the
same order)."""
import pytest
import torch

import golden_utils as gu

pytestmark = pytest.mark.gpu


def _step(model, g_cpu, flat, seed):
    from grappa_amd import Energy, MolwiseLoss, ops
    ops.manual_seed(seed)
    flat.zero_grad()
    g = Energy()(model(g_cpu.to("cuda")))
    loss = MolwiseLoss(gradient_weight=0.8, energy_weight=1.0, param_weight=0.0)(g)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss.detach()), flat.grad.clone(), {lvl: g.nodes[lvl].data["k"].detach().clone() for lvl in ("n2", "n3", "n4", "n4_improper")}


@pytest.mark.parametrize("train,n_mols", [(False, 6), (True, 6), (True, 64)])
def test_layer_locked_heads_equal_the_head_by_head_path(train, n_mols):
    from grappa_amd import get_default_model_config, model_from_config
    from grappa_amd.backend import get_backend
    from grappa_amd.datasets import build_batch_from_pool
    from grappa_amd.optim import FlatParams
    be = get_backend()
    model = model_from_config(get_default_model_config())
    model.load_state_dict(gu.keyed_state_dict(model))
    model = model.to("cuda")
    model.train(train)
    flat = FlatParams(model)
    g_cpu = build_batch_from_pool(list(range(200, 200 + n_mols)), n_confs=3, seed=1)
    pw = model.parameter_writer
    res = {}
    launches = {}
    for mode in ("0", "1"):
        pw.merged_heads = mode
        _step(model, g_cpu, flat, 3)                       # (caches warm)
        be.lib.grappa_launch_count(1)
        res[mode] = _step(model, g_cpu, flat, 3)
        launches[mode] = int(be.lib.grappa_launch_count(1))
    pw.merged_heads = "auto"
    (l0, g0, k0), (l1, g1, k1) = res["0"], res["1"]
    assert abs(l1 - l0) <= 1e-5 * abs(l0), (l0, l1)
    for lvl in k0:
        assert float((k1[lvl] - k0[lvl]).abs().max()) <= 1e-5 * max(float(k0[lvl].abs().max()), 1e-6), lvl
    worst = 0.0
    for p in model.parameters():
        lo = p._grappa_flat[1]
        a, b = g1[lo:lo + p.numel()], g0[lo:lo + p.numel()]
        scale = float(b.abs().max())
        if scale > 0:
            worst = max(worst, float((a - b).abs().max()) / scale)
    assert worst <= 1e-4, worst                             # (measured 5e-5 on the 6-molecule batch: other K cuts, fp32 rounding)
    assert launches["1"] < launches["0"] - 80, launches      # the products of the heads as grouped launches


def test_group_entry_refuses_what_it_cannot_launch_together():
    """grappa_gemm_f32_group validates every descriptor before it launches anything: mixed layouts come back GRAPPA_ERR_ARG and the backend
    launches the products one by one"""
    from grappa_amd.backend import get_backend
    be = get_backend()
    torch.manual_seed(0)
    a1, a2 = torch.randn(300, 128, device="cuda"), torch.randn(500, 64, device="cuda")
    w1, w2 = torch.randn(256, 128, device="cuda"), torch.randn(64, 96, device="cuda")
    o1, o2 = torch.empty(300, 256, device="cuda"), torch.empty(500, 96, device="cuda")
    be.gemm_group([((a1, w1, o1), dict(M=300, N=256, K=128)), ((a2, w2, o2), dict(M=500, N=96, K=64, b_kcontig=False))])
    torch.cuda.synchronize()
    assert float((o1 - a1 @ w1.T).abs().max()) <= 1e-4 * float(o1.abs().max())
    assert float((o2 - a2 @ w2).abs().max()) <= 1e-4 * float(o2.abs().max())
    # and a group that does go out together
    o1b, o3 = torch.empty_like(o1), torch.empty(500, 256, device="cuda")
    a3 = torch.randn(500, 128, device="cuda")
    be.gemm_group([((a1, w1, o1b), dict(M=300, N=256, K=128)), ((a3, w1, o3), dict(M=500, N=256, K=128))])
    torch.cuda.synchronize()
    assert float((o1b - a1 @ w1.T).abs().max()) <= 1e-4 * float(o1.abs().max())
    assert float((o3 - a3 @ w1.T).abs().max()) <= 1e-4 * float(o3.abs().max())
