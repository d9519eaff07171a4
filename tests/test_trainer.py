"""The training loop on a resident dataset (grappa_amd/trainer.py): sampling semantics of the reference's GraphDataLoader, loss
decreasing over epochs, validation metrics feeding the schedule, export in the reference's container format.
CPU: through the test-only backend (tiny model); GPU (-m gpu): the same loop on the HIP kernels."""
import os

import numpy as np
import pytest
import torch

from test_host_train import TINY


def _items(ids, n_confs=4):
    from grappa_amd.datasets import graph_from_pool
    return [(graph_from_pool(i, n_confs=n_confs + (j % 3), seed=2), f"ds{j % 2}") for j, i in enumerate(ids)]


def test_epoch_batches_follow_the_reference_samplers():
    from grappa_amd.trainer import epoch_batches
    names = ["a"] * 9 + ["b"] * 3
    g = torch.Generator().manual_seed(0)
    b = epoch_batches(names, 5, shuffle=True, generator=g)
    assert [len(x) for x in b] == [5, 5, 2] and sorted(np.concatenate(b).tolist()) == list(range(12))     # RandomSampler: a permutation
    assert np.concatenate(epoch_batches(names, 5, shuffle=False)).tolist() == list(range(12))
    # balance_factor 1: every dataset is drawn equally often (weights ~ 1 / occurrence, GraphDataLoader.py:106-131); -> 0: per molecule
    g = torch.Generator().manual_seed(1)
    draws = np.concatenate([np.concatenate(epoch_batches(names, 12, True, {}, 1.0, g)) for _ in range(400)])
    assert 0.45 < np.mean(draws >= 9) < 0.55                     # the 3 molecules of 'b' make up half of the draws
    draws = np.concatenate([np.concatenate(epoch_batches(names, 12, True, {}, 1e-9, g)) for _ in range(400)])
    assert 0.21 < np.mean(draws >= 9) < 0.29                     # 3 of 12
    draws = np.concatenate([np.concatenate(epoch_batches(names, 12, True, {"b": 3.0}, 0.0, g)) for _ in range(400)])
    assert 0.45 < np.mean(draws >= 9) < 0.55                     # weight 3 on 'b': 9 of 18
    with pytest.raises(ValueError):
        epoch_batches(names, 5, shuffle=False, weights={"a": 2.0})


def _run(device, epochs=4):
    from grappa_amd import GrappaModel, model_from_dict, ops
    from grappa_amd.device_dataset import DeviceDataset
    from grappa_amd.trainer import Trainer
    torch.manual_seed(0)
    ops.manual_seed(5)
    model = GrappaModel(**TINY).to(device)
    train = DeviceDataset(_items(list(range(300, 324))), device=device)
    val = DeviceDataset(_items(list(range(340, 348))), device=device)
    tr = Trainer(model, train, val, batch_size=8, conf_strategy=4, val_batch_size=4, val_conf_strategy="min", lr=2e-3,
                 proper_regularisation=1e-3, start_qm_epochs=0, warmup_steps=2, energy_weight=1.0, gradient_weight=0.8, param_weight=0.0,
                 patience=1, lr_decay=0.5)
    hist = tr.fit(epochs)
    assert len(hist) == epochs and all(np.isfinite(h["train_loss"]) for h in hist)
    assert hist[-1]["train_loss"] < hist[0]["train_loss"]
    m = hist[-1]["val_metrics"]
    assert set(m) == {"ds0", "ds1", "avg"} and m["avg"]["rmse_gradients"] > 0
    assert hist[0]["early_stopping_loss"] is None and hist[1]["early_stopping_loss"] > 0      # counted only after start_qm_epochs (:257)
    # export / reload in the reference's container format: identical predictions
    md = tr.model_dict()
    assert set(md) >= {"state_dict", "config"} and "gnn.blocks.0.layer_norm.weight" in md["state_dict"]
    clone = model_from_dict(md).to(device).eval()
    model.eval()
    g, _ = val.collate([0, 1, 2], "min")
    g2, _ = val.collate([0, 1, 2], "min")
    with torch.no_grad():
        a, b = model(g), clone(g2)
    assert torch.equal(a.nodes["n3"].data["k"], b.nodes["n3"].data["k"])
    return hist


def test_trainer_cpu(ref_backend):
    _run("cpu")


def test_recorded_trainer_without_a_gpu_is_the_eager_trainer(ref_backend):
    """`recorded=True` asks for hipGraph replays; on a machine without a GPU (this suite's test-only backend) the trainer runs the same
    epochs eagerly -- same batches, same losses -- instead of failing"""
    from grappa_amd import GrappaModel, ops
    from grappa_amd.device_dataset import DeviceDataset
    from grappa_amd.trainer import Trainer
    losses = []
    for recorded in (False, True):
        torch.manual_seed(0)
        ops.manual_seed(5)
        model = GrappaModel(**TINY)
        train = DeviceDataset(_items(list(range(300, 312))), device="cpu")
        tr = Trainer(model, train, None, batch_size=4, conf_strategy=4, lr=2e-3, start_qm_epochs=0, warmup_steps=2, energy_weight=1.0,
                     gradient_weight=0.8, param_weight=0.0, recorded=recorded)
        losses.append([h["train_loss"] for h in tr.fit(2)])
        assert tr.recorded_stats["replayed"] == 0
    assert losses[0] == losses[1]


@pytest.mark.gpu
def test_trainer_gpu():
    _run("cuda")


@pytest.mark.gpu
def test_training_trajectory_on_the_gpu_tracks_the_cpu_host_path():
    """four epochs of the same training run (train mode: dropout masks from the same counter-based hash, weighted sampling, schedule,
    fused Adam + clip) on the HIP path and on the CPU through the test-only backend: the epoch losses and the validation metrics of
    every epoch agree -- state that lives across steps (weight maxima / plane caches keyed on the parameters' versions, deferred
    reductions, the optimiser's flat buffers) is what a single-step comparison cannot see"""
    from grappa_amd import backend
    from oracle.ops_ref import RefBackend
    gpu = _run("cuda")
    old = backend._BACKEND
    backend.set_backend(RefBackend())
    try:
        cpu = _run("cpu")
    finally:
        backend.set_backend(old)
    for e, (a, b) in enumerate(zip(gpu, cpu)):
        assert abs(a["train_loss"] - b["train_loss"]) <= 2e-3 * abs(b["train_loss"]), (e, a["train_loss"], b["train_loss"])
        for ds in b["val_metrics"]:
            for k, v in b["val_metrics"][ds].items():
                if v is not None:
                    assert abs(a["val_metrics"][ds][k] - v) <= 5e-3 * abs(v), (e, ds, k, a["val_metrics"][ds][k], v)
        assert a["lr"] == b["lr"] if "lr" in a else True


def test_size_window_evens_out_the_work_per_batch_and_keeps_the_draws():
    from grappa_amd.trainer import epoch_batches
    rng = np.random.default_rng(0)
    sizes = rng.integers(5, 120, size=403)
    names = ["a"] * 403
    g1, g2 = torch.Generator().manual_seed(3), torch.Generator().manual_seed(3)
    plain = epoch_batches(names, 16, True, {}, 0.0, g1)
    even = epoch_batches(names, 16, True, {}, 0.0, g2, sizes=sizes, size_window=6)
    assert [len(b) for b in plain] == [len(b) for b in even] == [16] * 25 + [3]
    for w0 in range(0, 26, 6):                                   # every window of 6 batches holds exactly the molecules that were drawn for it
        assert sorted(np.concatenate(plain[w0:w0 + 6]).tolist()) == sorted(np.concatenate(even[w0:w0 + 6]).tolist())
    assert np.array_equal(plain[-1], even[-1])                   # the short last batch keeps its draws
    spread = lambda bs: np.std([sizes[b].sum() for b in bs if len(b) == 16])      # noqa: E731
    for w0 in range(0, 24, 6):                                   # within a window the batches carry (almost) the same number of atoms
        assert spread(even[w0:w0 + 6]) < 0.1 * spread(plain[w0:w0 + 6]), (w0, spread(even[w0:w0 + 6]), spread(plain[w0:w0 + 6]))
    assert spread(even) < 0.6 * spread(plain)                    # (what is left over the epoch is the difference between the windows' draws)
    # weighted sampling with replacement goes through the same path; size_window < 2 or no sizes: the reference's batches
    g1, g2 = torch.Generator().manual_seed(4), torch.Generator().manual_seed(4)
    a = epoch_batches(names, 16, True, {"a": 2.0}, 0.0, g1)
    b = epoch_batches(names, 16, True, {"a": 2.0}, 0.0, g2, sizes=sizes, size_window=1)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def _resume(device, tmp_path):
    """four epochs in one go == two epochs, checkpoint, a NEW trainer (fresh model, optimiser, generators) loading it, two more"""
    from grappa_amd import GrappaModel, ops
    from grappa_amd.device_dataset import DeviceDataset
    from grappa_amd.trainer import Trainer
    kw = dict(batch_size=8, conf_strategy=4, val_batch_size=4, val_conf_strategy="max", lr=2e-3, proper_regularisation=1e-3, start_qm_epochs=1,
              warmup_steps=2, energy_weight=1.0, gradient_weight=0.8, param_weight=1e-3, patience=0, lr_decay=0.5, weights={"ds1": 2.0})

    def make(seed):
        torch.manual_seed(seed)
        ops.manual_seed(5)
        model = GrappaModel(**TINY).to(device)
        train = DeviceDataset(_items(list(range(300, 324))), device=device)
        val = DeviceDataset(_items(list(range(340, 348))), device=device)
        return Trainer(model, train, val, **kw)

    whole = make(0)
    whole.fit(4)
    first = make(0)
    ck = str(tmp_path / "last.ckpt")
    first.fit(2, checkpoint=ck)
    assert os.path.exists(ck) and not os.path.exists(ck + ".tmp")
    second = make(123)                                  # other initial weights and generator states: everything must come from the file
    assert second.load_checkpoint(ck) == 2
    second.fit(4)
    assert [h["epoch"] for h in second.history] == [0, 1, 2, 3]
    for a, b in zip(whole.history, second.history):
        assert a["train_loss"] == b["train_loss"] and a["lr"] == b["lr"], (a, b)
        assert a["val_metrics"]["avg"] == b["val_metrics"]["avg"]
    assert torch.equal(whole.flat.data, second.flat.data) and torch.equal(whole.opt.m, second.opt.m) and whole.opt.step_count == second.opt.step_count
    with pytest.raises(ValueError, match="not a trainer checkpoint"):
        torch.save({"state_dict": {}}, str(tmp_path / "other.pth"))
        second.load_checkpoint(str(tmp_path / "other.pth"))


def test_trainer_resumes_bit_for_bit_cpu(ref_backend, tmp_path):
    _resume("cpu", tmp_path)


@pytest.mark.gpu
def test_trainer_resumes_bit_for_bit_gpu(tmp_path):
    _resume("cuda", tmp_path)


def test_epoch_batches_merge_a_short_tail_for_data_parallel_runs():
    from grappa_amd.trainer import epoch_batches
    names = ["a"] * 9
    assert [len(x) for x in epoch_batches(names, 4, shuffle=False)] == [4, 4, 1]
    assert [len(x) for x in epoch_batches(names, 4, shuffle=False, min_last=2)] == [4, 5]       # 9 % 4 == 1 < 2 ranks
    assert [len(x) for x in epoch_batches(names, 4, shuffle=False, min_last=1)] == [4, 4, 1]
    with pytest.raises(ValueError):
        epoch_batches(["a"], 4, shuffle=False, min_last=2)


def _dp_trainer_worker(rank, world, port, out_q):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    from grappa_amd import GrappaModel, backend, ops
    from grappa_amd.device_dataset import DeviceDataset
    from grappa_amd.dist import init_process_group_from_env
    from grappa_amd.trainer import Trainer
    from oracle.ops_ref import RefBackend
    import torch.distributed as dist
    backend.set_backend(RefBackend())
    init_process_group_from_env("gloo")
    torch.manual_seed(0)
    ops.manual_seed(5)
    model = GrappaModel(**TINY)
    train = DeviceDataset(_items(list(range(300, 309))), device="cpu")          # 9 molecules, batches of 4: 9 % 4 == 1 < 2 ranks
    val = DeviceDataset(_items(list(range(340, 347))), device="cpu")            # 7 molecules, batches of 2: four batches dealt to the two ranks
    tr = Trainer(model, train, val, batch_size=4, conf_strategy=4, val_batch_size=2, val_conf_strategy="max", lr=2e-3, start_qm_epochs=0,
                 warmup_steps=2, energy_weight=1.0, gradient_weight=0.8, param_weight=0.0)
    hist = tr.fit(2)
    # the sharded validation (each rank a share of the batches, squared-error sums added over the ranks) against this rank alone on all batches
    sharded, _ = tr.validate(2)
    keep = (tr.world, tr.rank)
    tr.world, tr.rank = 1, 0
    alone, _ = tr.validate(2)
    tr.world, tr.rank = keep
    for ds in alone:
        for k, v in alone[ds].items():
            assert abs(sharded[ds][k] - v) <= 1e-6 * abs(v), (ds, k, sharded[ds][k], v)
    out_q.put((rank, [h["train_loss"] for h in hist] + [hist[-1]["val_metrics"]["avg"]["rmse_gradients"], hist[-1]["early_stopping_loss"]],
               float(tr.flat.data.double().sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_two_ranks_gloo_with_a_tail_batch_smaller_than_the_world():
    """len(dataset) % batch_size == 1 with two ranks used to leave one rank without molecules (hang in the all-reduce).
    Both ranks must finish, log the same (all-reduced) epoch loss and hold identical parameters."""
    import torch.multiprocessing as mp
    from test_host_train import _free_port
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_trainer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    (_, l0, s0), (_, l1, s1) = res
    assert l0 == l1 and all(np.isfinite(l0)) and s0 == s1


def test_flat_gradient_views_survive_zero_grad_set_to_none(ref_backend):
    """`module.zero_grad()` (set_to_none) detaches p.grad from the flat gradient buffer; the next backward pass must
    write into the flat buffer again (fused Adam / the all-reduce read that buffer), starting from zero."""
    from grappa_amd import GrappaModel
    from grappa_amd.optim import FlatParams
    from test_host_train import _loss
    torch.manual_seed(0)
    model = GrappaModel(**TINY).eval()
    flat = FlatParams(model)
    _loss(model, [30, 31], 5).backward()
    want = flat.grad.clone()
    assert float(want.abs().max()) > 0
    model.zero_grad(set_to_none=True)
    assert all(p.grad is None for p in flat.params)
    _loss(model, [30, 31], 5).backward()
    assert torch.equal(flat.grad, want)                       # not doubled, not stale: restored views, slices zeroed first
    assert all(p.grad.data_ptr() == flat.grad[flat._offsets[id(p)][0]:].data_ptr() for p in flat.params)
    flat.params[0].grad = torch.zeros_like(flat.params[0])    # a foreign tensor is refused
    with pytest.raises(RuntimeError):
        _loss(model, [30, 31], 5).backward()
