"""GPU (-m gpu): the plane-format GEMMs (csrc/gemm_planes.hip) through the C ABI.

  * both operands in planes (forward layout and the k-major wgrad layout with split-K and the fused bias gradient), against a
    float64 product at the accuracy of the split-in-kernel GEMM;
  * the epilogue's plane tensors: residual and saved ELU output READ from planes, result WRITTEN as planes;
  * weight planes (fp32 activations x pre-split weight): bit-identical to the split-in-kernel product at op level; a whole
    train step under GRAPPA_WEIGHT_PLANES semantics matches the default to rounding, including after an optimiser step (stale
    planes must be refreshed).
"""
import os
import sys

import ctypes

import pytest
import torch

import golden_utils as gu

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_plane_gemms_against_float64():
    import gemm_planes_check as gp
    gen = torch.Generator(device="cuda").manual_seed(0)
    for M, N, K, km in [(256, 128, 64, 0), (300, 200, 96, 0), (1000, 512, 512, 0), (83, 77, 160, 0), (8233, 1536, 512, 0),
                        (256, 128, 64, 1), (512, 512, 8233, 1), (300, 200, 1000, 1), (1536, 512, 20011, 1)]:
        gp.check(M, N, K, bool(km), gen)           # asserts <= 4e-7 of sum |a||b| and bit-equality of the weight-plane product
    gp.check_epilogue(gen)


def test_split_planes_reconstruct_exactly():
    from grappa_amd.backend import get_backend
    be = get_backend()
    gen = torch.Generator(device="cuda").manual_seed(1)
    w = (torch.randn((511, 300), generator=gen, device="cuda") * torch.logspace(-6, 3, 300, device="cuda")).requires_grad_(True)
    for transposed in (False, True):
        pl = be._planes_of_weight(w, transposed)
        ref = w.detach().t() if transposed else w.detach()
        R, C = ref.shape
        assert pl.shape == (3, (R + 31) // 32 * 32, (C + 31) // 32 * 32)
        got = (pl[0, :R, :C].float() + pl[1, :R, :C].float()) + pl[2, :R, :C].float()
        assert torch.equal(got, ref)                                   # three round-to-nearest bf16 pieces carry all 24 bits
        assert float(pl[:, R:].float().abs().max()) == 0 and float(pl[:, :, C:].float().abs().max()) == 0
    # cache: same tensor, same version -> same planes; an in-place update -> refreshed
    a = be._planes_of_weight(w, False)
    assert be._planes_of_weight(w, False) is a
    with torch.no_grad():
        w.mul_(2.0)
    b = be._planes_of_weight(w, False)
    assert torch.equal((b[0, :511, :300].float() + b[1, :511, :300].float()) + b[2, :511, :300].float(), w.detach())


def test_train_steps_with_weight_planes_match_the_default():
    from grappa_amd import Energy, MolwiseLoss, get_default_model_config, model_from_config, ops
    from grappa_amd.backend import get_backend
    from grappa_amd.datasets import build_batch_from_pool
    from grappa_amd.optim import FlatParams, FusedAdam
    be = get_backend()
    res = []
    default_precision = be.gemm_precision_name
    try:
        be.set_gemm_precision("f32_bf16x6")                     # the plane format is a bf16 split: it rides on this arithmetic
        for use in (False, True):
            be.weight_planes = use
            model = model_from_config(get_default_model_config())
            model.load_state_dict(gu.keyed_state_dict(model))
            model = model.to("cuda").train()
            flat = FlatParams(model)
            opt = FusedAdam(flat, lr=1e-3, max_grad_norm=10.0)
            ops.manual_seed(11)
            out = []
            for step in range(2):                               # the second step runs on UPDATED weights: planes must follow
                opt.zero_grad()
                g = Energy()(model(build_batch_from_pool(list(range(200, 232)), n_confs=8, seed=3).to("cuda")))
                loss = MolwiseLoss(gradient_weight=0.8, energy_weight=1.0, param_weight=0.0, proper_regularisation=1e-3)(g)
                loss.backward()
                out.append((loss.detach().clone(), flat.grad.clone(), g.nodes["n4"].data["k"].detach().clone()))
                opt.step()
            torch.cuda.synchronize()
            res.append(out)
    finally:
        be.weight_planes = False
        be.set_gemm_precision(default_precision)
    # per product the two kernels are bit-identical (test_plane_gemms_against_float64); inside the model the planner may pick
    # another tile / split-K plan for the plane kernel (256 x 128 only), i.e. another summation order: fp32 rounding noise
    for (l0, g0, k0), (l1, g1, k1) in zip(*res):
        assert abs(float(l0) - float(l1)) <= 5e-6 * abs(float(l0))         # a few tens of fp32 ulps of a sum over ~1e5 terms
        assert float((k0 - k1).abs().max()) <= 2e-5 * float(k0.abs().max())
        assert float((g0 - g1).abs().max()) <= 2e-5 * float(g0.abs().max())
    assert not torch.equal(res[0][0][0], res[0][1][0])          # the two steps differ (the weights moved)


def test_grouped_weight_gradients_match_single_launches():
    """grappa_gemm_f32_grouped (the queued weight gradients of a backward pass as ONE grid + one reduction): every product equals its
    own grappa_gemm_f32 launch up to the summation order of the K chunks; bias gradients ride along; `accumulate` semantics kept."""
    from grappa_amd.backend import get_backend
    be = get_backend()
    gen = torch.Generator(device="cuda").manual_seed(5)
    rnd = lambda *s: torch.randn(s, generator=gen, device="cuda")      # noqa: E731
    shapes = [(17158, 512, 512), (44325, 1536, 512), (83328, 512, 512), (28248, 512, 2048), (8233, 2048, 512), (5000, 256, 300), (40, 64, 64)]
    probs = [(rnd(T, Np), rnd(T, Kp)) for T, Np, Kp in shapes]
    want, got = [], []
    old = be.defer_wgrads
    try:
        for defer in (False, True):
            be.defer_wgrads = defer
            outs = []
            for dz, x in probs:
                dw, db = torch.full((dz.shape[1], x.shape[1]), 0.5, device="cuda"), torch.full((dz.shape[1],), -1.0, device="cuda")
                be.gemm_wgrad(dz, x, dw, db)
                outs.append((dw, db))
            be.flush_wgrads()                              # (outside a backward pass every push is launched at once: a group of one)
            torch.cuda.synchronize()
            (got if defer else want).append(outs)
    finally:
        be.defer_wgrads = old
    for (dz, x), (w0, b0), (w1, b1) in zip(probs, want[0], got[0]):
        ref = dz.double().t() @ x.double() + 0.5
        s = float(ref.abs().max())
        assert float((w0.double() - ref).abs().max()) < 2e-6 * s and float((w1.double() - ref).abs().max()) < 2e-6 * s
        rb = dz.double().sum(0) - 1.0
        assert float((b1.double() - rb).abs().max()) < 2e-6 * float(rb.abs().max())
        assert float((b0 - b1).abs().max()) < 2e-6 * float(rb.abs().max())


def test_split_k_finished_inside_the_launch_equals_the_reduction_launch_bit_for_bit():
    """split-K products (single launches: a forward product with a tail, weight gradients with a fused bias gradient, row maxima of
    the output; and a grouped launch) summed by the last workgroup of each tile inside the product's launch: the same bits as the
    reduction kernel behind the launch (both add the slabs in the order 0, 1, ...), and the same bits run after run"""
    from grappa_amd.backend import get_backend
    be = get_backend()
    gen = torch.Generator(device="cuda").manual_seed(11)
    rnd = lambda *s: torch.randn(s, generator=gen, device="cuda")      # noqa: E731
    x, w, bias, res = rnd(50000, 512), rnd(512, 512) * 0.05, rnd(512), rnd(50000, 512)      # 196 x 4 tiles: 768 + a tail of 16, K cut 4 ways
    wg = [(rnd(T, Np), rnd(T, Kp)) for T, Np, Kp in [(17158, 512, 512), (44325, 1536, 512), (30000, 512, 2048), (5000, 256, 300)]]
    tm, tn, ns, tail_tiles, tail_ns = (ctypes.c_int() for _ in range(5))
    be.lib.grappa_gemm_f32_plan(50000, 512, 512, 5, ctypes.byref(tm), ctypes.byref(tn), ctypes.byref(ns), ctypes.byref(tail_tiles), ctypes.byref(tail_ns))
    assert ns.value > 1 or tail_ns.value > 1, "the forward shape of this test must use split-K"

    def run():
        outs = []
        y = torch.empty(50000, 512, device="cuda")
        sy = be.gemm(x, w, y, M=50000, N=512, K=512, bias=bias, act=1, res=res, out_amax=True)
        outs += [y, sy[1].row]
        old = be.defer_wgrads
        try:
            for defer in (False, True):
                be.defer_wgrads = defer
                for dz, xx in wg:
                    dw, db = torch.full((dz.shape[1], xx.shape[1]), 0.5, device="cuda"), torch.full((dz.shape[1],), -1.0, device="cuda")
                    be.gemm_wgrad(dz, xx, dw, db)
                    outs += [dw, db]
                be.flush_wgrads()
        finally:
            be.defer_wgrads = old
        torch.cuda.synchronize()
        return outs

    try:
        be.splitk_reduce = 1                       # a reduction launch behind the product
        want = run()
        be.splitk_reduce = 2                       # inside the product's own launch
        got = [run() for _ in range(3)]
    finally:
        be.splitk_reduce = 0
    for r in got:
        for a, b in zip(want, r):
            assert torch.equal(a, b)


def test_tail_launches_can_be_switched_off_and_the_model_does_so_on_four_streams():
    """grappa_gemm_desc.plan_tail (C ABI 10; a process-wide setter before): the planner stops cutting the K of the last partial round; the product is the same to fp32
    summation order; `WriteParameters` turns the tails off while its heads keep four streams busy and on again on one stream, unless pinned"""
    from grappa_amd import GrappaModel
    from grappa_amd.backend import get_backend
    from grappa_amd.datasets import build_batch_from_pool
    from test_host_train import TINY
    be = get_backend()

    def plan(M, N, K):
        from grappa_amd import _lib
        d = _lib.GemmDesc()
        d.M, d.N, d.K, d.precision = M, N, K, 5
        be._call_options(d)                      # the options this backend would send with the product
        v = [ctypes.c_int() for _ in range(5)]
        be.lib.grappa_gemm_f32_plan_desc(ctypes.byref(d), *[ctypes.byref(x) for x in v])
        return [x.value for x in v]

    try:
        assert plan(50000, 512, 512)[4] > 1                          # library default: a tail of 16 tiles, K cut 4 ways
        gen = torch.Generator(device="cuda").manual_seed(5)
        x, w = torch.randn(50000, 512, generator=gen, device="cuda"), torch.randn(512, 512, generator=gen, device="cuda") * 0.05
        with_tail, without = torch.empty(50000, 512, device="cuda"), torch.empty(50000, 512, device="cuda")
        be.gemm(x, w, with_tail, M=50000, N=512, K=512)
        be.pin_tail_launches(False)
        assert plan(50000, 512, 512)[3:] == [0, 0]
        be.gemm(x, w, without, M=50000, N=512, K=512)
        torch.cuda.synchronize()
        assert float((with_tail - without).abs().max()) <= 2e-6 * float(with_tail.abs().max())
        be.pin_tail_launches(None)
        model = GrappaModel(**TINY).to("cuda").eval()
        g = build_batch_from_pool([30, 31], n_confs=2, seed=1).to("cuda")
        with torch.no_grad():
            model.parameter_writer.head_streams = 4
            model(g)
            assert be._tails is False and plan(50000, 512, 512)[4] == 0
            model.parameter_writer.head_streams = 1
            model(g)
            assert be._tails is True and plan(50000, 512, 512)[4] > 1
            be.pin_tail_launches(True)
            model.parameter_writer.head_streams = 4
            model(g)
            assert be._tails is True and plan(50000, 512, 512)[4] > 1       # pinned: the model's request is ignored
    finally:
        be.pin_tail_launches(None)


def test_train_step_with_and_without_grouped_weight_gradients():
    from grappa_amd import Energy, MolwiseLoss, get_default_model_config, model_from_config, ops
    from grappa_amd.backend import get_backend
    from grappa_amd.datasets import build_batch_from_pool
    from grappa_amd.optim import FlatParams
    be = get_backend()
    model = model_from_config(get_default_model_config())
    model.load_state_dict(gu.keyed_state_dict(model))
    model = model.to("cuda").train()
    flat = FlatParams(model)
    res = []
    old = be.defer_wgrads
    try:
        for defer in (False, True, True):
            be.defer_wgrads = defer
            ops.manual_seed(3)
            flat.zero_grad()
            g = Energy()(model(build_batch_from_pool(list(range(900, 964)), n_confs=8, seed=3).to("cuda")))
            loss = MolwiseLoss(gradient_weight=0.8, energy_weight=1.0, param_weight=0.0, proper_regularisation=1e-3)(g)
            loss.backward()
            assert not be._wq                       # the end-of-pass callback launched everything that was queued
            torch.cuda.synchronize()
            res.append((loss.detach().clone(), flat.grad.clone()))
    finally:
        be.defer_wgrads = old
    assert torch.equal(res[0][0], res[1][0])
    assert float((res[0][1] - res[1][1]).abs().max()) <= 2e-5 * float(res[0][1].abs().max())
    assert torch.equal(res[1][1], res[2][1])           # grouped launches are bit-reproducible
