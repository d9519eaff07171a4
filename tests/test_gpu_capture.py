"""GPU (-m gpu): the hot path recorded in a hipGraph (grappa_amd/capture.py) against the eager path -- a train step with dropout and Adam
(device-side dropout salt, learning rate and step count), and `Grappa.predict` through the cache of captured forwards."""
import numpy as np
import pytest
import torch

import golden_utils as gu

pytestmark = pytest.mark.gpu

LK = dict(gradient_weight=0.8, energy_weight=1.0, param_weight=0.0)


@pytest.fixture(autouse=True)
def _no_salt_left_behind():
    """the dropout salt is state of the backend object (sent per call, C ABI 10): tests that compare masks with the host-side hash expect none"""
    yield
    from grappa_amd.backend import get_backend
    be = get_backend()
    if getattr(be, "_salt", None) is not None:
        be._salt.zero_()
        be.disable_dropout_salt()


def _setup(seed=0):
    from grappa_amd import Energy, GrappaModel, MolwiseLoss
    from grappa_amd.optim import FlatParams, FusedAdam
    fx = gu.load("ref_small_att.npz")
    model = GrappaModel(**gu.config_of(fx))
    model.load_state_dict(gu.state_dict_of(fx))
    model = model.to("cuda").train()
    flat = FlatParams(model)
    opt = FusedAdam(flat, lr=1e-3, max_grad_norm=10.0)
    g = gu.build_batch(gu.molecules_of(fx), 4, False).to("cuda")
    return model, flat, opt, g, Energy(), MolwiseLoss(**LK)


def test_captured_train_step_equals_the_eager_steps():
    """3 warm-up steps + 5 replays of the recorded step == 8 eager steps from the same state with the same dropout salts: every replay draws
    new masks (the salt is a node of the graph), Adam's bias corrections follow the device-side step count, the learning rate written between
    replays takes effect"""
    from grappa_amd import ops
    from grappa_amd.backend import get_backend
    from grappa_amd.capture import CapturedTrainStep, _drop_outputs
    be = get_backend()
    be.enable_dropout_salt()
    # ---- eager twin
    ops.manual_seed(7)
    be._salt.zero_()
    model, flat, opt, g, energy, loss_fn = _setup()
    opt.enable_dynamic()
    losses_e = []
    for i in range(8):
        if i == 6:
            opt.lr = 5e-4
        be.bump_dropout_salt()
        opt.zero_grad()
        _drop_outputs(g)
        loss = loss_fn(energy(model(g)))
        loss.backward()
        opt.step()
        losses_e.append(float(loss.detach()))
    want = flat.data.clone()
    # ---- captured
    ops.manual_seed(7)
    be._salt.zero_()
    model, flat, opt, g, energy, loss_fn = _setup()
    step = CapturedTrainStep(model, energy, loss_fn, opt, g, warmup=3)
    losses_c = []
    for i in range(3, 8):
        if i == 6:
            opt.lr = 5e-4
        losses_c.append(float(step()))
    torch.cuda.synchronize()
    assert opt.step_count == 8 and int(opt._step_t) == 8 and int(be._salt) == 8
    assert len(set(losses_c)) == len(losses_c)                       # the masks (and the weights) move from replay to replay
    # NOTE the dropout SEEDS of the recorded step are those drawn while recording (ops.next_seed), not the eager twin's per-step seeds: the
    # trajectories agree in distribution, not in bits -- what must agree exactly is a replay against the same recorded seeds (below)
    assert all(np.isfinite(losses_c)) and abs(losses_c[-1] - losses_e[-1]) < 0.5 * abs(losses_e[-1])
    assert torch.isfinite(flat.data).all() and float((flat.data - want).abs().max()) < 0.1
    be._salt.zero_()


def test_replays_are_deterministic_and_eval_replay_equals_eager():
    """no dropout (eval mode): the recorded step and the eager step are the same arithmetic -- parameters after 4 steps agree bit for bit"""
    from grappa_amd import ops
    from grappa_amd.backend import get_backend
    from grappa_amd.capture import CapturedTrainStep, _drop_outputs
    be = get_backend()
    model, flat, opt, g, energy, loss_fn = _setup()
    model.eval()
    opt.enable_dynamic()
    for _ in range(6):
        opt.zero_grad()
        _drop_outputs(g)
        loss_fn(energy(model(g))).backward()
        opt.step()
    want = flat.data.clone()
    model, flat, opt, g, energy, loss_fn = _setup()
    model.eval()
    step = CapturedTrainStep(model, energy, loss_fn, opt, g, warmup=2)
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    assert torch.equal(flat.data, want), float((flat.data - want).abs().max())


def test_predict_through_captured_forwards():
    """the second and later calls on a shape replay a recorded graph: same Parameters as the eager path; another molecule of another shape in
    between; new weights invalidate the recorded graphs"""
    from grappa_amd import Grappa, get_default_model_config, model_from_config
    from grappa_amd.datasets import molecule_from_pool
    model = model_from_config(get_default_model_config())
    model.load_state_dict(gu.keyed_state_dict(model))
    gr = Grappa(model, device="cuda")
    assert gr._graphs is not None
    mols = [molecule_from_pool(i) for i in (3, 50)]
    fields = ["bond_k", "bond_eq", "angle_k", "angle_eq", "proper_ks", "improper_ks", "proper_phases"]
    eager = []
    keep, gr._graphs = gr._graphs, None
    for m in mols:
        eager.append(gr.predict(m))
    gr._graphs = keep
    for rep in range(3):
        for m, want in zip(mols, eager):
            got = gr.predict(m)
            for f in fields:
                a, b = np.asarray(getattr(got, f)), np.asarray(getattr(want, f))
                assert a.shape == b.shape and np.abs(a - b).max() <= 1e-6 * max(np.abs(b).max(), 1e-6), (rep, f)
    assert len(gr._graphs.entries) == 2
    with torch.no_grad():
        next(model.parameters()).mul_(1.0)                          # an in-place write: the version counter moves
    gr.predict(mols[0])
    assert len(gr._graphs.entries) == 1                              # the stale graphs were dropped, this shape re-recorded


def test_predict_cache_follows_changing_weights():
    """weights that change between predict calls (training next to inference): the first change drops the recorded forwards, the ones recorded
    afterwards refresh the per-weight caches inside the graph and stay valid through further changes -- same parameters as the eager path"""
    import numpy as np
    from grappa_amd import Grappa, GrappaModel
    from grappa_amd.datasets import molecule_from_pool, pool_atom_counts
    fx = gu.load("ref_small_att.npz")
    model = GrappaModel(**gu.config_of(fx))
    model.load_state_dict(gu.state_dict_of(fx))
    gr = Grappa(model, device="cuda")
    mol = molecule_from_pool(int(np.argmin(np.abs(pool_atom_counts() - 30))))

    def eager():
        cache, gr._graphs = gr._graphs, None
        try:
            return gr.predict(mol)
        finally:
            gr._graphs = cache

    def same(a, b):
        for k in ("bond_k", "bond_eq", "angle_k", "angle_eq", "proper_ks", "improper_ks"):
            x, y = np.asarray(getattr(a, k)), np.asarray(getattr(b, k))
            assert np.allclose(x, y, rtol=1e-5, atol=1e-6 * max(1.0, float(np.abs(y).max()))), k

    for _ in range(3):
        p = gr.predict(mol)
    assert len(gr._graphs.entries) == 1 and not gr._graphs.refresh_weights
    same(p, eager())
    for step in range(3):
        with torch.no_grad():
            for q in model.parameters():
                q.mul_(1.0 + 0.01 * (step + 1))
        want = eager()
        got = gr.predict(mol)
        same(got, want)
        assert gr._graphs.refresh_weights and len(gr._graphs.entries) == 1
        ent = next(iter(gr._graphs.entries.values()))
        assert ent.refresh_weights and ent.valid()
    # and the parameters did move with the weights (the check above was not comparing two stale results)
    assert not np.allclose(np.asarray(p.bond_k), np.asarray(got.bond_k), rtol=1e-4)


def test_predict_cache_follows_backend_settings_and_replaced_parameters():
    """what a recorded forward bakes in besides the weights' values -- the backend's settings (arithmetic, pair routing, plan options)
    and WHICH tensors the parameters are.  Either changes -> the graph is dropped, in refresh mode too; the signature table stays bounded"""
    import numpy as np
    from grappa_amd import Grappa, GrappaModel
    from grappa_amd.backend import get_backend
    from grappa_amd.datasets import molecule_from_pool, pool_atom_counts
    fx = gu.load("ref_small_att.npz")
    model = GrappaModel(**gu.config_of(fx))
    model.load_state_dict(gu.state_dict_of(fx))
    gr = Grappa(model, device="cuda")
    be = get_backend()
    mol = molecule_from_pool(int(np.argmin(np.abs(pool_atom_counts() - 30))))

    def eager():
        cache, gr._graphs = gr._graphs, None
        try:
            return gr.predict(mol)
        finally:
            gr._graphs = cache

    def same(a, b):
        for k in ("bond_k", "bond_eq", "angle_k", "angle_eq", "proper_ks", "improper_ks"):
            x, y = np.asarray(getattr(a, k)), np.asarray(getattr(b, k))
            assert np.allclose(x, y, rtol=1e-5, atol=1e-6 * max(1.0, float(np.abs(y).max()))), k

    for _ in range(3):
        gr.predict(mol)
    ent = next(iter(gr._graphs.entries.values()))
    assert ent.valid()
    # a backend setting: the recorded launches are no longer what an eager call would make
    old = be.inference_pairs
    try:
        be.inference_pairs = not old
        assert not ent.valid()
        got = gr.predict(mol)                                        # dropped, eager for this call (the signature is seen again first)
        same(got, eager())
        assert ent not in gr._graphs.entries.values() and not gr._graphs.refresh_weights
    finally:
        be.inference_pairs = old
    for _ in range(3):
        gr.predict(mol)
    ent = next(iter(gr._graphs.entries.values()))
    assert ent.valid()
    # replaced parameter storage (p.data = ...): same values, other tensors -- the graph would go on reading the old ones
    with torch.no_grad():
        for q in model.parameters():
            q.data = q.data.clone() * 1.5
    assert not ent.valid()
    want = eager()
    for _ in range(3):
        got = gr.predict(mol)
    same(got, want)
    # the table of seen signatures is bounded
    cache = gr._graphs
    cache.max_seen = 4
    for i in range(10):
        cache.seen[("fake", i)] = 1
        if len(cache.seen) > cache.max_seen:
            cache.seen.pop(next(iter(cache.seen)))
    assert len(cache.seen) <= 4


class _CountingReducer:
    """stands in for dist.BucketedGradReducer on one GPU: finish() is where the RCCL all-reduce of the flat gradient buffer would run"""

    def __init__(self):
        self.calls = 0

    def finish(self):
        from grappa_amd.backend import get_backend
        get_backend().flush_wgrads()
        self.calls += 1


@pytest.mark.parametrize("preserve", [False, True])
def test_split_recorded_step_equals_the_single_graph_step(preserve):
    """data parallelism: the recorded step as TWO graphs -- zero_grad .. backward, and clip + Adam -- with the reducer called
    eagerly between their replays; same kernels in the same order as the one-graph step, so the same bits.  preserve_state: the warm-up and
    the recording run without the reducer (a rank recording a new shape issues no collective of its own), every call issues exactly one."""
    from grappa_amd import ops
    from grappa_amd.backend import get_backend
    from grappa_amd.capture import CapturedTrainStep
    be = get_backend()
    res = []
    for split in (False, True):
        ops.manual_seed(11)
        be.enable_dropout_salt()
        be._salt.zero_()
        model, flat, opt, g, energy, loss_fn = _setup()
        red = _CountingReducer() if split else None
        step = CapturedTrainStep(model, energy, loss_fn, opt, g, warmup=3, preserve_state=preserve, reducer=red)
        losses = [float(step()) for _ in range(4)]
        torch.cuda.synchronize()
        if split:
            assert step.graph_b is not None and red.calls == 4 + (0 if preserve else 3)
        res.append((losses, flat.data.clone(), opt.step_count))
        be._salt.zero_()
    (la, pa, ca), (lb, pb, cb) = res
    assert la == lb and ca == cb and torch.equal(pa, pb)
