"""CPU: train-mode host logic through the test-only backend: dropout masks are regenerated consistently in the
backward pass (finite-difference check along a random direction), flat parameter buffers + fused Adam, and the
data-parallel path (2 gloo ranks, one all-reduce of the flat gradient) reproduces the single-process gradient."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_utils as gu

TINY = dict(graph_node_features=16, in_feat_name=["atomic_number", "partial_charge", "ring_encoding", "degree", "charge_model"],
            gnn_width=32, gnn_attentional_layers=1, gnn_convolutions=1, gnn_attention_heads=2, gnn_dropout_attention=0.2,
            gnn_dropout_initial=0.1, gnn_dropout_conv=0.2, gnn_dropout_final=0.1, parameter_dropout=0.3,
            bond_transformer_depth=1, bond_n_heads=2, bond_transformer_width=32, bond_symmetriser_depth=2, bond_symmetriser_width=16,
            angle_transformer_depth=1, angle_n_heads=2, angle_transformer_width=32, angle_symmetriser_depth=2, angle_symmetriser_width=16,
            proper_transformer_depth=1, proper_n_heads=2, proper_transformer_width=32, proper_symmetriser_depth=3, proper_symmetriser_width=16,
            improper_transformer_depth=1, improper_n_heads=2, improper_transformer_width=32, improper_symmetriser_depth=1,
            improper_symmetriser_width=16, n_periodicity_proper=3, n_periodicity_improper=2, gated_torsion=True)
LK = dict(gradient_weight=0.8, energy_weight=1.0, param_weight=0.0, proper_regularisation=1e-3)


def _loss(model, ids, seed, global_b=None):
    from grappa_amd import Energy, MolwiseLoss, ops
    from grappa_amd.datasets import build_batch_from_pool
    ops.manual_seed(seed)
    g = build_batch_from_pool(ids, n_confs=3, seed=1)
    lf = MolwiseLoss(**LK)
    lf.global_batch_size = global_b
    return lf(Energy()(model(g)))


@pytest.mark.parametrize("learn", [False, True])
def test_train_mode_gradient_matches_finite_differences(ref_backend, learn):
    from grappa_amd import GrappaModel
    torch.manual_seed(0)
    model = GrappaModel(**TINY, learnable_statistics=learn).double().float()
    model.train()
    ids = [10, 11, 12]
    loss = _loss(model, ids, 7)
    loss.backward()
    params = [p for p in model.parameters() if p.grad is not None]
    gen = torch.Generator().manual_seed(1)
    dirs = [torch.randn(p.shape, generator=gen) for p in params]
    gd = sum(float((p.grad * d).sum()) for p, d in zip(params, dirs))
    gnorm = np.sqrt(sum(float((d * d).sum()) for d in dirs))
    eps = 2e-3 / gnorm * np.sqrt(sum(float((p * p).sum()) for p in params))
    with torch.no_grad():
        for p, d in zip(params, dirs):
            p.add_(eps * d)
        lp = float(_loss(model, ids, 7))
        for p, d in zip(params, dirs):
            p.add_(-2 * eps * d)
        lm = float(_loss(model, ids, 7))
    fd = (lp - lm) / (2 * eps)
    assert abs(fd - gd) / max(abs(fd), abs(gd)) < 3e-2, (fd, gd)


@pytest.mark.parametrize("learn", [False, True])       # learnable statistics: 0-dim parameters in the flat buffer
def test_flat_params_and_fused_adam_track_torch_adam(ref_backend, learn):
    from grappa_amd import GrappaModel
    from grappa_amd.optim import FlatParams, FusedAdam
    torch.manual_seed(0)
    model = GrappaModel(**TINY, learnable_statistics=learn).eval()
    twin = GrappaModel(**TINY, learnable_statistics=learn).eval()
    twin.load_state_dict(model.state_dict())
    flat = FlatParams(model)
    assert all(p.data_ptr() >= flat.data.data_ptr() for p in flat.params)
    # eps=1e-4: keeps entries whose gradient is pure rounding noise out of Adam's sign(g) regime
    opt = FusedAdam(flat, lr=1e-3, eps=1e-4, max_grad_norm=1.0)
    topt = torch.optim.Adam(twin.parameters(), lr=1e-3, eps=1e-4)
    for it in range(3):
        opt.zero_grad()
        _loss(model, [20, 21], 3).backward()
        opt.step()
        topt.zero_grad()
        _loss(twin, [20, 21], 3).backward()
        torch.nn.utils.clip_grad_norm_(twin.parameters(), 1.0)
        topt.step()
    for (k, a), (_, b) in zip(model.named_parameters(), twin.named_parameters()):
        assert torch.allclose(a, b, rtol=1e-4, atol=2e-6), k
    # state dict still has the reference layout after flattening
    assert list(model.state_dict().keys()) == list(twin.state_dict().keys())


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _dp_worker(rank, world, port, ids, sd, out_q):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    from grappa_amd import GrappaModel, backend
    from grappa_amd.datasets import pool_atom_counts
    from grappa_amd.dist import all_reduce_gradients, init_process_group_from_env, shard_indices
    from grappa_amd.optim import FlatParams
    from oracle.ops_ref import RefBackend
    backend.set_backend(RefBackend())
    init_process_group_from_env("gloo")
    model = GrappaModel(**TINY).eval()
    model.load_state_dict(sd)
    flat = FlatParams(model)
    sizes = [int(pool_atom_counts()[i]) for i in ids]
    mine = [ids[j] for j in shard_indices(sizes, world, rank)]
    _loss(model, mine, 5, global_b=len(ids)).backward()
    one_bucket = flat.grad.clone()
    all_reduce_gradients(one_bucket)
    # the same step again through the two-bucket reducer (writer-head slice sent from inside the backward pass)
    from grappa_amd.dist import BucketedGradReducer
    reducer = BucketedGradReducer(model, flat, overlap=True)
    flat.zero_grad()
    _loss(model, mine, 5, global_b=len(ids)).backward()
    assert reducer._heads_sent and len(reducer._work) == 1
    reducer.finish()
    assert torch.equal(flat.grad, one_bucket)
    # and without the overlap (both buckets after backward): the default since round 4 (the overlap is opt-in until an N >= 2
    # RCCL run has compared the two orders bit for bit)
    assert BucketedGradReducer(model, flat).overlap is False
    plain = BucketedGradReducer(model, flat, overlap=False)
    assert plain.overlap is False and model.on_heads_backward_done is None
    flat.zero_grad()
    _loss(model, mine, 5, global_b=len(ids)).backward()
    plain.finish()
    assert torch.equal(flat.grad, one_bucket)
    result = flat.grad.clone()
    # a shard processed in two chunks (gradient accumulation): the overlapped bucket must leave with the LAST backward pass only
    half = len(mine) // 2
    assert half >= 1
    flat.zero_grad()
    for part in (mine[:half], mine[half:]):
        _loss(model, part, 5, global_b=len(ids)).backward()
    want2 = flat.grad.clone()
    all_reduce_gradients(want2)
    chunked = BucketedGradReducer(model, flat, overlap=True)
    flat.zero_grad()
    chunked.begin_step(2)
    _loss(model, mine[:half], 5, global_b=len(ids)).backward()
    assert not chunked._heads_sent
    _loss(model, mine[half:], 5, global_b=len(ids)).backward()
    assert chunked._heads_sent
    chunked.finish()
    assert torch.equal(flat.grad, want2)
    if rank == 0:
        out_q.put(result.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_two_ranks_gloo(ref_backend):
    from grappa_amd import GrappaModel
    from grappa_amd.dist import shard_indices
    from grappa_amd.optim import FlatParams
    assert shard_indices([5, 9, 7, 3], 2, 0) == [1, 0][::-1] or True
    s0, s1 = shard_indices([5, 9, 7, 3], 2, 0), shard_indices([5, 9, 7, 3], 2, 1)
    assert sorted(s0 + s1) == [0, 1, 2, 3] and s0 == [0, 1] and s1 == [2, 3]
    torch.manual_seed(0)
    model = GrappaModel(**TINY).eval()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    ids = [30, 31, 32, 33, 34]
    flat = FlatParams(model)
    _loss(model, ids, 5).backward()
    want = flat.grad.clone().numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, ids, sd, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert np.abs(got - want).max() <= 1e-4 * max(np.abs(want).max(), 1e-6)


def _split_step_worker(rank, world, port, ids, sd, out_q):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    from grappa_amd import Energy, GrappaModel, MolwiseLoss, backend, ops
    from grappa_amd.capture import CapturedTrainStep
    from grappa_amd.datasets import build_batch_from_pool, pool_atom_counts
    from grappa_amd.dist import BucketedGradReducer, init_process_group_from_env, shard_indices
    from grappa_amd.optim import FlatParams, FusedAdam
    from oracle.ops_ref import RefBackend
    backend.set_backend(RefBackend())
    init_process_group_from_env("gloo")
    model = GrappaModel(**TINY).eval()
    model.load_state_dict(sd)
    flat = FlatParams(model)
    opt = FusedAdam(flat, lr=1e-3, eps=1e-4, max_grad_norm=1.0)
    sizes = [int(pool_atom_counts()[i]) for i in ids]
    mine = [ids[j] for j in shard_indices(sizes, world, rank)]
    lf = MolwiseLoss(**LK)
    lf.global_batch_size = len(ids)
    calls = {"n": 0}
    reducer = BucketedGradReducer(model, flat)
    inner = reducer.finish

    def counted():
        calls["n"] += 1
        inner()
    reducer.finish = counted
    # the recorded step's SPLIT sequence (zero_grad .. backward | all-reduce | clip + Adam) without graphs (record=False): what a rank runs per
    # call when Trainer(recorded=True) meets world > 1
    ops.manual_seed(5)
    step = CapturedTrainStep(model, Energy(), lf, opt, build_batch_from_pool(mine, n_confs=3, seed=1), reducer=reducer, record=False)
    for _ in range(2):
        step()
    assert calls["n"] == 2 and step.replays == 2          # ONE collective per call
    if rank == 0:
        out_q.put(flat.data.clone().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_split_recorded_step_two_ranks_gloo_equals_the_single_process_step(ref_backend):
    """the train step as `Trainer(recorded=True)` runs it under data parallelism -- two recorded halves with the eager
    all-reduce of the flat gradient buffer between them (capture.CapturedTrainStep reducer=) -- here in its graph-less form (record=False,
    the test backend) on two gloo ranks: two steps on the sharded batch == two single-process steps on the whole batch."""
    from grappa_amd import Energy, GrappaModel, MolwiseLoss, ops
    from grappa_amd.datasets import build_batch_from_pool
    from grappa_amd.optim import FlatParams, FusedAdam
    torch.manual_seed(0)
    model = GrappaModel(**TINY).eval()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    ids = [30, 31, 32, 33, 34]
    flat = FlatParams(model)
    opt = FusedAdam(flat, lr=1e-3, eps=1e-4, max_grad_norm=1.0)
    lf = MolwiseLoss(**LK)
    g = build_batch_from_pool(ids, n_confs=3, seed=1)
    ops.manual_seed(5)
    for _ in range(2):
        opt.zero_grad()
        for lvl in ("n2", "n3", "n4", "n4_improper"):
            for k in ("k", "eq"):
                g.nodes[lvl].data.pop(k, None)
        lf(Energy()(model(g))).backward()
        opt.step()
    want = flat.data.clone().numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_split_step_worker, args=(r, 2, port, ids, sd, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert np.abs(got - want).max() <= 2e-5 * max(np.abs(want).max(), 1e-6)
