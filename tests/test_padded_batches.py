"""Batches of a FIXED shape for recorded train steps: `DeviceDataset.collate(pad_to=...)` puts ONE padding molecule behind
the real ones; the loss runs over the real molecules only.  Pinned here:
  * the padded batch is the unpadded batch plus rows -- every table's leading rows, the conformation selection (same random draws) and
    the index plan of the real part are unchanged; the whole padded plan is what BatchPlan builds on the host for the same graph;
  * loss and every parameter gradient of the padded batch equal the unpadded batch's (train mode, dropout off: a mask is a hash of the row
    index in a position-major token table, which moves with T); the padding molecule's loss entry is exactly 0;
  * (GPU) `Trainer(recorded=True)`: two epochs of replayed hipGraphs give the eager trainer's epoch losses.
CPU: host logic through the test-only backend; GPU (-m gpu): the HIP kernels."""
import numpy as np
import pytest
import torch

from grappa_amd.batch import BatchPlan
from grappa_amd.constants import TUPLE_LEVELS
from grappa_amd.datasets import graph_from_pool
from grappa_amd.device_dataset import DeviceDataset, ShapeBuckets

from test_host_train import TINY


def _items(n=14, start=200):
    items = []
    for j in range(n):
        g = graph_from_pool(start + 3 * j, n_confs=[6, 9, 3, 7, 12][j % 5], seed=5, with_refs=True)
        items.append((g, f"ds{j % 3}"))
    return items


def _caps(ds, ids, extra):
    tot = ds.totals(ids)
    return {k: tot[k] + extra[k] for k in tot}


def _check_structure(device):
    items = _items()
    ds = DeviceDataset(items, device=device)
    assert ds.bonds_are_n2
    ds.enable_padding({"n1": 64, "n2": 200, "n3": 64, "n4": 64, "n4_improper": 64})
    assert len(ds) == len(items)                                  # the padding entry is not a molecule of the dataset
    for strategy, ids, extra in ((4, [0, 5, 2, 9], dict(n1=9, n2=12, n3=5, n4=0, n4_improper=7)),
                                 ("min", list(range(14)), dict(n1=4, n2=2, n3=0, n4=3, n4_improper=0)),
                                 (32, [13, 1, 1, 7], dict(n1=16, n2=100, n3=33, n4=64, n4_improper=1))):
        caps = _caps(ds, ids, extra)
        torch.manual_seed(77)
        g0, names0 = ds.collate(ids, strategy)
        r0 = torch.rand(1)
        torch.manual_seed(77)
        g1, names1 = ds.collate(ids, strategy, pad_to=caps)
        assert torch.equal(torch.rand(1), r0)                     # the same number of random draws
        assert names0 == names1
        p0, p1 = g0.plan(), g1.plan()
        assert p1.n_real_mols == len(ids) and p1.B == len(ids) + 1 and getattr(p0, "n_real_mols", None) is None
        assert {nt: g1.num_nodes(nt) for nt in caps} == caps and p1.E == 2 * caps["n2"]
        for nt in g0.ntypes:
            n0 = g0.num_nodes(nt)
            assert set(g0.nodes[nt].data) == set(g1.nodes[nt].data)
            for k, v in g0.nodes[nt].data.items():
                w = g1.nodes[nt].data[k]
                assert w.dtype == v.dtype and w.shape[1:] == v.shape[1:] and torch.equal(w[:n0], v), (nt, k)
        assert bool((g1.nodes["g"].data["is_dummy"][-1] == 1).all())
        N0, E0 = p0.N, p0.E
        assert torch.equal(p1.indptr[:N0 + 1], p0.indptr) and torch.equal(p1.indices[:E0], p0.indices) and torch.equal(p1.rev[:E0], p0.rev)
        assert torch.equal(p1.atom_molptr[:-1], p0.atom_molptr) and int(p1.atom_molptr[-1]) == caps["n1"]
        for lvl in TUPLE_LEVELS:
            assert torch.equal(p1.idx32[lvl][:p0.T[lvl]], p0.idx32[lvl]) and torch.equal(p1.mol_ptr[lvl][:-1], p0.mol_ptr[lvl])
            pad = p1.idx32[lvl][p0.T[lvl]:]
            assert pad.numel() == 0 or (int(pad.min()) >= N0 and int(pad.max()) < caps["n1"])      # the padding tuples stay inside the padding molecule
        # the whole plan: what the host builder makes of the same padded graph
        hp = BatchPlan(g1.cpu(), "cpu")
        for name in ("indptr", "indices", "rev", "atom_molptr", "inc_ptr", "inc_code"):
            assert torch.equal(getattr(p1, name).cpu(), getattr(hp, name)), (strategy, name)
        for lvl in TUPLE_LEVELS:
            for name in ("idx32", "mol_ptr", "inv_ptr", "inv_rows"):
                assert torch.equal(getattr(p1, name)[lvl].cpu(), getattr(hp, name)[lvl]), (strategy, lvl, name)
    # a gap no molecule fills: fewer than four atoms, more bonds than atom pairs, an atom without a bond
    tot = ds.totals([0, 1])
    for bad in (dict(n1=3, n2=3, n3=0, n4=0, n4_improper=0), dict(n1=4, n2=7, n3=0, n4=0, n4_improper=0), dict(n1=8, n2=3, n3=0, n4=0, n4_improper=0),
                dict(n1=8, n2=8, n3=-1, n4=0, n4_improper=0)):
        assert DeviceDataset.pad_sizes(tot, {k: tot[k] + bad[k] for k in tot}) is None
        with pytest.raises(ValueError):
            ds.collate([0, 1], 4, pad_to={k: tot[k] + bad[k] for k in tot})


def test_padded_collate_structure_cpu(ref_backend):
    _check_structure("cpu")


@pytest.mark.gpu
def test_padded_collate_structure_gpu():
    _check_structure("cuda")


def _check_training(device, tol):
    from grappa_amd import Energy, GrappaModel, MolwiseLoss, ops
    torch.manual_seed(0)
    # dropout off: the writers lay their tokens out position-major (row = pos * T + t) and a dropout mask is a hash of the row index, so
    # with dropout a batch of another T draws other masks for the same tokens -- by design, and nothing a padded batch could keep
    model = GrappaModel(**dict(TINY, gnn_dropout_attention=0.0, gnn_dropout_initial=0.0, gnn_dropout_conv=0.0, gnn_dropout_final=0.0, parameter_dropout=0.0)).to(device)
    model.train()
    ds = DeviceDataset(_items(), device=device)
    ds.enable_padding({"n1": 64, "n2": 200, "n3": 64, "n4": 64, "n4_improper": 64})
    ids = [3, 8, 11, 0, 6]
    caps = _caps(ds, ids, dict(n1=11, n2=15, n3=9, n4=20, n4_improper=6))
    loss_fn = MolwiseLoss(gradient_weight=0.8, energy_weight=1.0, param_weight=1e-3, proper_regularisation=1e-3, improper_regularisation=1e-3,
                          param_weights_by_dataset={"ds1": 0.5})
    energy = Energy()
    out = []
    for pad in (None, caps):
        torch.manual_seed(5)
        ops.manual_seed(9)
        g, names = ds.collate(ids, 4, pad_to=pad)
        for p in model.parameters():
            p.grad = None
        loss = loss_fn(energy(model(g)), list(names))
        loss.backward()
        if device == "cuda":
            torch.cuda.synchronize()
        out.append((float(loss), {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters() if p.grad is not None},
                    loss_fn.last_per_molecule.detach().cpu().clone(), {lvl: g.nodes[lvl].data["k"].detach().cpu() for lvl in TUPLE_LEVELS}, g))
    (l0, g0, m0, k0, b0), (l1, g1, m1, k1, b1) = out
    assert np.isfinite(l1) and abs(l1 - l0) <= tol * abs(l0), (l0, l1)
    assert m1.shape[0] == len(ids) + 1 and float(m1[-1]) == 0.0 and torch.allclose(m1[:-1], m0, rtol=10 * tol, atol=0)
    for lvl in TUPLE_LEVELS:
        assert torch.isfinite(k1[lvl]).all()
        assert torch.allclose(k1[lvl][:k0[lvl].shape[0]], k0[lvl], rtol=10 * tol, atol=10 * tol * float(k0[lvl].abs().max())), lvl
    assert set(g0) == set(g1)
    for k in g0:
        assert torch.isfinite(g1[k]).all(), k
        scale = max(float(g0[k].abs().max()), 1e-12)
        assert float((g1[k] - g0[k]).abs().max()) <= 20 * tol * scale + 1e-12, (k, float((g1[k] - g0[k]).abs().max()), scale)


def test_padded_batch_trains_like_the_unpadded_one_cpu(ref_backend):
    _check_training("cpu", 2e-6)


@pytest.mark.gpu
def test_padded_batch_trains_like_the_unpadded_one_gpu():
    # fp32-grade products: the weight gradients' operand scale is the whole tensor's largest magnitude, padding rows included -- the
    # roundings of the fp16 pieces move within the arithmetic's own error (~1e-6), nothing else does
    _check_training("cuda", 1e-5)


def test_shape_buckets_cover_their_epoch():
    from grappa_amd.trainer import epoch_batches

    class _Counts:                     # the part of a DeviceDataset that ShapeBuckets reads
        PAD_DIMS = DeviceDataset.PAD_DIMS

        def __init__(self, n):
            rng = np.random.default_rng(1)
            atoms = rng.integers(8, 60, size=n)
            rings = rng.integers(0, 4, size=n)
            self.count = {"n1": atoms, "n2": atoms - 1 + rings, "n3": 2 * atoms - 3 + 3 * rings, "n4": 3 * atoms - 8 + 6 * rings, "n4_improper": (atoms // 3) * 3}
            self.names = ["a"] * n

        def totals(self, ids):
            return {nt: int(self.count[nt][np.asarray(ids)].sum()) for nt in self.PAD_DIMS}

    ds = _Counts(1000)
    gen = torch.Generator().manual_seed(0)
    batches = epoch_batches(ds.names, 32, generator=gen)
    sb = ShapeBuckets(ds, batches, n_buckets=4)
    assert 1 <= len(sb.caps) <= 4
    waste = []
    for b in batches:
        c = sb.choose(ds.totals(b))
        assert c is not None and DeviceDataset.pad_sizes(ds.totals(b), c) is not None
        waste.append(sum(c.values()) / sum(ds.totals(b).values()) - 1.0)
    assert np.mean(waste) < 0.20, np.mean(waste)
    # another epoch's batches of full size fall into the buckets too (the last one carries a margin)
    later = [b for b in epoch_batches(ds.names, 32, generator=gen) if len(b) == 32]
    fit = [sb.choose(ds.totals(b)) is not None for b in later]
    assert np.mean(fit) > 0.9, np.mean(fit)
    assert all(sb.max_pad[k] >= max(c[k] for c in sb.caps) - min(ds.totals(b)[k] for b in batches) for k in sb.max_pad)


def _with_param_refs(g, seed):
    """reference parameters on every level (k_ref / eq_ref; some NaN: the loss masks them), as a dataset with classical parameters carries them"""
    rng = np.random.default_rng(seed)
    for lvl, name, lo, hi, cols in (("n2", "k", 200., 900., None), ("n2", "eq", 0.9, 1.6, None), ("n3", "k", 50., 200., None), ("n3", "eq", 1.6, 2.2, None),
                                    ("n4", "k", -2., 2., 6), ("n4_improper", "k", -2., 2., 3)):
        n = g.num_nodes(lvl)
        v = rng.uniform(lo, hi, size=(n,) if cols is None else (n, cols)).astype(np.float32)
        if n > 3 and name == "k":
            v[rng.integers(0, n)] = np.nan
        g.nodes[lvl].data[name + "_ref"] = torch.from_numpy(v)
    return g


@pytest.mark.gpu
def test_recorded_epochs_match_the_eager_trainer_in_the_production_like_setting():
    """the recorded trainer against the eager one where the default Grappa schedule actually trains -- a parameter loss with
    per-dataset weights (the graph's `plan.param_weight_rows` input, copied by `load`), reference parameters with NaNs, and start_qm_epochs = 1
    (epoch 0: parameter loss only; the schedule then moves the loss weights, i.e. the recorded steps' stamp, so new graphs are recorded);
    dropout off so that the two trainers are comparable step by step"""
    from grappa_amd import GrappaModel, ops
    from grappa_amd.trainer import Trainer
    cfg = dict(TINY, gnn_dropout_attention=0.0, gnn_dropout_initial=0.0, gnn_dropout_conv=0.0, gnn_dropout_final=0.0, parameter_dropout=0.0)
    items = [(_with_param_refs(graph_from_pool(300 + i, n_confs=4, seed=2), i), f"ds{i % 2}") for i in range(40)]
    hist, stats = [], None
    for recorded in (False, True):
        torch.manual_seed(0)
        ops.manual_seed(5)
        model = GrappaModel(**cfg).to("cuda")
        train = DeviceDataset(items, device="cuda")
        tr = Trainer(model, train, None, batch_size=8, conf_strategy=4, lr=2e-3, proper_regularisation=1e-3, start_qm_epochs=1, warmup_steps=2,
                     energy_weight=1.0, gradient_weight=0.8, param_weight=1e-3, param_weights_by_dataset={"ds1": 5e-3}, recorded=recorded, shape_buckets=2)
        h = tr.fit(3)
        hist.append([e["train_loss"] for e in h])
        if recorded:
            stats = dict(tr.recorded_stats)
    eager, rec = hist
    assert stats["eager"] + stats["replayed"] == 15 and stats["replayed"] >= 10, stats      # (a later epoch's batch may miss every bucket cut on epoch 0: it runs eagerly)
    assert all(np.isfinite(eager)) and eager[0] != eager[1]
    for a, b in zip(eager, rec):
        assert abs(a - b) <= 5e-5 * abs(a), (eager, rec)


@pytest.mark.gpu
def test_recorded_epochs_with_dropout_draw_new_masks_every_replay():
    """dropout on: finite losses, and the same batch replayed twice through one graph gives different losses (the device-side salt moves)"""
    from grappa_amd import GrappaModel, ops
    from grappa_amd.trainer import Trainer
    items = [(graph_from_pool(300 + i, n_confs=4, seed=2), "ds") for i in range(16)]
    torch.manual_seed(0)
    ops.manual_seed(5)
    model = GrappaModel(**dict(TINY, parameter_dropout=0.3)).to("cuda")
    tr = Trainer(model, DeviceDataset(items, device="cuda"), None, batch_size=8, conf_strategy=4, lr=0.0, start_qm_epochs=0, warmup_steps=2,
                 energy_weight=1.0, gradient_weight=0.8, param_weight=0.0, recorded=True, shape_buckets=1)
    tr.model.train()
    ids = np.arange(8)
    tr.calibrate_buckets([ids])
    a, b = float(tr.train_step_recorded(ids)), float(tr.train_step_recorded(ids))      # lr = 0: the weights stay, only the masks differ
    assert np.isfinite(a) and np.isfinite(b) and a != b and tr.recorded_stats["replayed"] >= 1


@pytest.mark.gpu
def test_recorded_epochs_match_the_eager_trainer():
    """two epochs, batches of 8 from 44 molecules (a short last batch included), dropout off (a recorded step salts its dropout seeds: with
    dropout the two trainers draw different masks by design): every step of the recorded trainer is a graph replay and the epoch losses are
    the eager trainer's; a handful of graphs serves all batches"""
    from grappa_amd import GrappaModel, ops
    from grappa_amd.trainer import Trainer
    cfg = dict(TINY, gnn_dropout_attention=0.0, gnn_dropout_initial=0.0, gnn_dropout_conv=0.0, gnn_dropout_final=0.0, parameter_dropout=0.0)
    items = [(graph_from_pool(300 + i, n_confs=4, seed=2), f"ds{i % 2}") for i in range(44)]
    val_items = [(graph_from_pool(360 + i, n_confs=4 + (i % 3), seed=2), f"ds{i % 2}") for i in range(14)]
    hist, vals, stats = [], [], None
    for recorded in (False, True):
        torch.manual_seed(0)
        ops.manual_seed(5)
        model = GrappaModel(**cfg).to("cuda")
        train, val = DeviceDataset(items, device="cuda"), DeviceDataset(val_items, device="cuda")
        tr = Trainer(model, train, val, batch_size=8, conf_strategy=4, val_batch_size=4, val_conf_strategy="max", lr=2e-3, proper_regularisation=1e-3,
                     start_qm_epochs=0, warmup_steps=2, energy_weight=1.0, gradient_weight=0.8, param_weight=0.0, recorded=recorded, shape_buckets=2)
        h = tr.fit(2)
        hist.append([e["train_loss"] for e in h])
        vals.append([e["val_metrics"] for e in h])
        if recorded:
            stats = dict(tr.recorded_stats)
    eager, rec = hist
    assert stats["eager"] == 0 and stats["replayed"] == 12 and 1 <= stats["graphs_recorded"] <= 6, stats
    assert 0 < stats["padding_rows"] < 0.6 * stats["real_rows"], stats
    for a, b in zip(eager, rec):
        assert abs(a - b) <= 2e-5 * abs(a), (eager, rec)
    # the validation passes too: every batch of the second epoch is a replay (the validation batches are the same every epoch), same metrics
    assert stats["eval_replayed"] == 8 and 1 <= stats["eval_graphs_recorded"] <= 4, stats
    for ve, vr in zip(*vals):
        for ds in ve:
            for k, v in ve[ds].items():
                if v is not None:
                    assert abs(vr[ds][k] - v) <= 2e-5 * abs(v), (ds, k, v, vr[ds][k])


@pytest.mark.gpu
def test_recorded_trainer_survives_a_failed_recording(monkeypatch):
    """a recording that fails (memory, a call a capture cannot hold) must not end the run: batches of that shape run eagerly -- the padded
    batch itself, whose padding molecule the loss skips -- and the epochs' losses are the eager trainer's"""
    import warnings

    import grappa_amd.capture as capture
    from grappa_amd import GrappaModel, ops
    from grappa_amd.trainer import Trainer

    class Failing:
        def __init__(self, *a, **k):
            raise RuntimeError("capture refused (test)")

    cfg = dict(TINY, gnn_dropout_attention=0.0, gnn_dropout_initial=0.0, gnn_dropout_conv=0.0, gnn_dropout_final=0.0, parameter_dropout=0.0)
    items = [(graph_from_pool(300 + i, n_confs=4, seed=2), f"ds{i % 2}") for i in range(24)]
    hist = []
    for recorded in (False, True):
        torch.manual_seed(0)
        ops.manual_seed(5)
        model = GrappaModel(**cfg).to("cuda")
        train = DeviceDataset(items, device="cuda")
        tr = Trainer(model, train, None, batch_size=8, conf_strategy=4, lr=2e-3, proper_regularisation=1e-3, start_qm_epochs=0, warmup_steps=2,
                     energy_weight=1.0, gradient_weight=0.8, param_weight=0.0, recorded=recorded, shape_buckets=2)
        if recorded:
            monkeypatch.setattr(capture, "CapturedTrainStep", Failing)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            hist.append([h["train_loss"] for h in tr.fit(2)])
        if recorded:
            assert tr.recorded_stats["replayed"] == 0 and tr.recorded_stats["eager"] == 6 and tr.recorded_stats["graphs_recorded"] == 0
            assert any("recording a train step failed" in str(x.message) for x in w)
            assert 1 <= len(tr._unrecordable) <= 4
    for a, b in zip(*hist):
        assert abs(a - b) <= 2e-5 * abs(a), hist
