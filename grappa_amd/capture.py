"""hipGraph capture of the hot path for batches of a FIXED shape.

The eager path issues ~450 kernel launches per forward pass and ~1,000 per train step from Python; at the reference's own operating
point (batch 32, single-molecule `predict`; reference training/config.py:49-52, grappa.py:36-57) the step is bound by that host work,
not by the GPU.  A captured step is ONE `hipGraphLaunch`: the kernels, their arguments, the cross-stream dependencies of the writer
heads and the grouped weight gradients are recorded once (torch.cuda.CUDAGraph drives hipStreamBeginCapture; the library's kernels are
launched on torch's capturing stream through the C ABI as always) and replayed.

What makes a TRAIN step replayable although every kernel argument is frozen at capture:
  * dropout: every seed is a constant of the graph; the library mixes a 64-bit word of DEVICE memory into them
    (`grappa_gemm_desc.drop_salt` and the row-wise kernels' `drop_salt`, C ABI 10: passed per call), and the graph's first node increments that word -- fresh masks per replay, forward and backward of one
    replay agreeing;
  * Adam: learning rate and step count are read from device memory (`grappa_adam_step_dyn_f32`); the graph increments the step count,
    the host writes the learning rate between replays (`FusedAdam.lr = ...`);
  * weights change inside the graph, so the per-weight caches (row / column maxima, pair splits) are refreshed INSIDE it: they are
    invalidated right before capture, which puts the batched refresh kernels at the head of the recorded step.

Shapes are part of the graph: a captured step serves the batch it was captured on and any batch copied into the same device tensors
(`StaticForward.load`).  Multi-GPU steps (the RCCL all-reduce) stay eager.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import torch

from .backend import get_backend


def _pinned_by_graph(be):
    """everything the backend allocated OUTSIDE the capture whose address the recorded kernels hold: per-weight maxima and pair splits, their
    device tables, the workspaces, the dropout salt.  The graph object must keep them alive (the backend's caches may drop them)."""
    return (list(be._wamax.values()), be._wtable, list(be._wpairs.values()), be._wptable, dict(be._wplanes), dict(be._ws), getattr(be, "_salt", None),
            dict(be._side_streams))


def _touch_weight_caches(be) -> None:
    """the recorded graph uses every cached per-weight entry at every replay: keep them from ageing out of the backend's tables"""
    for e in be._wamax.values():
        e[3] = be._wepoch
    for e in be._wpairs.values():
        e[3] = be._wepoch


def _drop_outputs(g) -> None:
    for lvl in ("n2", "n3", "n4", "n4_improper"):
        for k in ("k", "eq"):
            g.nodes[lvl].data.pop(k, None)


class _StaticInputs:
    """every input of a recorded step (index plan, position tables, features and references of every level) in ONE device buffer the recorded
    kernels read from; `load(g2)` copies another batch of the same shape signature in.  Users set self.g, self.plan, self.static_inputs."""

    def _input_slots(self, plan, g):
        slots = dict(CapturedForward._plan_tensors(plan))
        for nt, d in g._data.items():
            for k, t in d.items():
                if torch.is_tensor(t):
                    slots[f"data.{nt}.{k}"] = t
        slots["graph.src"], slots["graph.dst"] = g._src, g._dst
        return slots

    def _rehome_inputs(self) -> None:
        slots = self._input_slots(self.plan, self.g)
        self._layout, off = {}, 0
        for name, t in slots.items():
            nbytes = t.numel() * t.element_size()
            self._layout[name] = (off, nbytes, t.dtype, tuple(t.shape))
            off += (nbytes + 15) // 16 * 16
        self._in_dev = torch.zeros(max(off, 16), dtype=torch.uint8, device=self.g.device)
        self._views = {}
        for name, (o, nbytes, dtype, shape) in self._layout.items():
            v = self._in_dev[o:o + nbytes].view(dtype).view(shape)
            v.copy_(slots[name])
            self._views[name] = v
        for name, v in self._views.items():
            parts = name.split(".")
            if parts[0] == "data":
                self.g._data[parts[1]][name[len("data.") + len(parts[1]) + 1:]] = v
            elif parts[0] == "graph":
                setattr(self.g, "_" + parts[1], v)
            elif len(parts) == 1:
                setattr(self.plan, name, v)
            elif len(parts) == 2:
                getattr(self.plan, parts[0])[parts[1]] = v
            else:
                d = getattr(self.plan, parts[0])
                tup = list(d[parts[1]])
                tup[int(parts[2])] = v
                d[parts[1]] = tuple(tup)

    def load(self, g) -> None:
        """another batch of this step's signature (on the device) into the recorded step's inputs: one multi-tensor copy, stream-ordered
        behind the previous replay"""
        if not self.static_inputs:
            raise RuntimeError("load: this step was recorded on its batch's own tensors (static_inputs=False)")
        plan = g.plan()
        for lvl in self.plan.__dict__.get("_pos_tables", {}):
            plan.position_tables(lvl)
        src = self._input_slots(plan, g)
        if set(src) != set(self._layout):
            raise ValueError(f"load: the batch has other input tables than the recorded one ({sorted(set(src) ^ set(self._layout))})")
        dsts, srcs = [], []
        for name, (o, nbytes, dtype, shape) in self._layout.items():
            t = src[name]
            if tuple(t.shape) != shape or t.dtype != dtype:
                raise ValueError(f"load: {name} is {t.dtype} {tuple(t.shape)}, the recorded step holds {dtype} {shape}")
            if t.numel():
                dsts.append(self._views[name])
                srcs.append(t if t.is_contiguous() else t.contiguous())
        torch._foreach_copy_(dsts, srcs)
        # host-side mirrors the eager modules read (molecule counts per level; nothing the recorded kernels depend on)
        self.g._bnn = {k: v.copy() for k, v in g._bnn.items()}
        self.plan.n_real_mols = getattr(plan, "n_real_mols", None)
        self.plan.max_degree = plan.max_degree


class CapturedTrainStep(_StaticInputs):
    """zero_grad -> GrappaModel -> Energy -> MolwiseLoss -> backward -> clip + Adam on ONE resident batch, as a hipGraph.

        step = CapturedTrainStep(model, energy, loss_fn, opt, g)     # warms up (3 eager steps: these DO train) and captures
        loss = step()                                                # one replay = one optimiser step; loss: device tensor of the graph

    `opt` (optim.FusedAdam) is switched to device-side scalars (`opt.enable_dynamic()`); `opt.lr = x` between replays takes effect."""

    def __init__(self, model, energy, loss_fn, opt, g, warmup: int = 3, preserve_state: bool = False, static_inputs: bool = False, reducer=None,
                 record: bool = True):
        """preserve_state: parameters, Adam moments and step count are put back after the warm-up steps and the recording, so that making the
        graph does not train (a trainer that records in the middle of an epoch); the first replay is then that batch's one real step.
        static_inputs: every input of the step (index plan, position tables, features and references of all levels) is moved into ONE device
        buffer the recorded kernels read from, and `load(g2)` copies another batch of the same shape signature in (`DeviceDataset.collate(
        pad_to=...)` makes batches of one signature): the graph then serves every such batch.
        reducer (data parallelism, SURVEY 8(e): anything with `finish()`, e.g. dist.BucketedGradReducer): the step is recorded as TWO graphs --
        zero_grad .. backward, and clip + Adam -- and one call = replay the first, `reducer.finish()` EAGERLY (the RCCL all-reduce of the flat
        gradient buffer is not recorded), replay the second.  Every rank issues exactly ONE collective per call whatever it replays or
        records: with preserve_state the warm-up steps and the recording run WITHOUT the reducer (their effect is undone anyway), so a rank
        that records a new shape in the middle of an epoch stays in step with ranks that replay.
        record=False: no hipGraph at all -- the same object runs its phases eagerly (the fallback for a shape whose recording failed, and the
        CPU / gloo tests of the split sequence)."""
        self.model, self.energy, self.loss_fn, self.opt, self.g = model, energy, loss_fn, opt, g
        self.reducer = reducer
        self.record = bool(record)
        self.be = get_backend()
        self.static_inputs = bool(static_inputs)
        if not self.record:
            self.graph = self.graph_b = None
            self.replays = 0
            self.loss = None
            return
        if not torch.cuda.is_available():
            raise RuntimeError("CapturedTrainStep needs a GPU (record=False runs the same sequence eagerly)")
        if self.static_inputs:
            self.plan = g.plan()
            for lvl in ("n2", "n3", "n4", "n4_improper"):     # built eagerly: their construction sorts (host-synchronising torch ops)
                if self.plan.T[lvl]:
                    self.plan.position_tables(lvl)
            self.signature = train_signature(g)
            self._rehome_inputs()
        saved = None
        if preserve_state:
            saved = (opt.flat.data.clone(), opt.m.clone(), opt.v.clone(), opt.step_count)
        # the salt word lives on the graph's device and is passed with every call made while this object records; afterwards eager calls go
        # back to their plain seeds (the recorded kernels keep the address they were recorded with: ADVICE r4)
        self.be.enable_dropout_salt(g.device)
        opt.enable_dynamic()
        # a recorded step has no host cost per launch, and on the GPU four heads on four streams run their short dependent chains side by
        # side: measured 9.0 ms (head by head, four streams) against 9.5 (layer-locked, one stream) on the batch-32 step -- "auto" means
        # head by head here; the layer-locked heads are for the EAGER small-batch step, which is bound by the host's launches
        pw = getattr(model, "parameter_writer", None)
        self._merged_was = getattr(pw, "merged_heads", None)
        if self._merged_was == "auto":
            pw.merged_heads = "0"
        try:
            self.stream = torch.cuda.Stream(device=g.device)
            self.stream.wait_stream(torch.cuda.current_stream(g.device))
            collective = self.reducer is not None and not preserve_state       # (see the docstring: one collective per CALL on every rank)
            with torch.cuda.stream(self.stream):
                for _ in range(max(int(warmup), 1)):           # on the capturing stream: workspaces and side streams are keyed by it
                    self.be.bump_dropout_salt()
                    self._eager_a()
                    if collective:
                        self.reducer.finish()
                    self._eager_b()
            self.stream.synchronize()
            # (the last warm-up step's optimiser left every per-weight cache stale: the recorded step starts by refreshing them, in the graph)
            self.graph = torch.cuda.CUDAGraph()
            self.graph_b = None
            if self.reducer is None:
                with torch.cuda.graph(self.graph, stream=self.stream):
                    self.be.bump_dropout_salt()
                    self.loss = self._eager()
            else:
                pool = torch.cuda.graph_pool_handle()
                with torch.cuda.graph(self.graph, stream=self.stream, pool=pool):
                    self.be.bump_dropout_salt()
                    self.loss = self._eager_a()
                self.graph_b = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_b, stream=self.stream, pool=pool):
                    self._eager_b()
            self.opt.step_count -= 1                            # (recorded, not executed: the device-side count did not move)
            self._pinned = _pinned_by_graph(self.be)
        finally:
            if self._merged_was == "auto":
                pw.merged_heads = "auto"
            self.be.disable_dropout_salt()
            if saved is not None:
                # also when the warm-up or the recording failed: the caller gets its training state back as it was and may go on eagerly
                stream = getattr(self, "stream", None)
                if stream is not None:
                    stream.synchronize()
                opt.flat.data.copy_(saved[0])
                opt.m.copy_(saved[1])
                opt.v.copy_(saved[2])
                opt.step_count = saved[3]
                opt.sync_dynamic()
                self.be.invalidate_weights()
        torch.cuda.current_stream(g.device).wait_stream(self.stream)
        self.replays = 0

    def _eager_a(self):
        """zero_grad .. backward (every queued weight-gradient product is launched by the end of the pass)"""
        self.opt.zero_grad()
        _drop_outputs(self.g)
        loss = self.loss_fn(self.energy(self.model(self.g)))
        loss.backward()
        if hasattr(self.be, "flush_wgrads"):
            self.be.flush_wgrads()
        return loss.detach()

    def _eager_b(self):
        self.opt.step()

    def _eager(self):
        loss = self._eager_a()
        self._eager_b()
        return loss

    def __call__(self) -> torch.Tensor:
        if not self.record:                                 # the same sequence without graphs
            self.loss = self._eager_a()
            if self.reducer is not None:
                self.reducer.finish()
            self._eager_b()
            self.replays += 1
            return self.loss
        self.graph.replay()
        if self.graph_b is not None:
            self.reducer.finish()                           # eager: RCCL's all-reduce of the flat gradient buffer, ordered behind the first graph
            self.graph_b.replay()
        self.replays += 1
        self.opt.step_count += 1                            # host mirror of the device-side counter
        self.be.invalidate_weights()                        # the weights moved under the host-side caches' feet
        _touch_weight_caches(self.be)
        return self.loss


class CapturedEvalStep(_StaticInputs):
    """GrappaModel -> Energy in eval mode under no_grad on batches of ONE shape signature, as a hipGraph (the validation pass of
    `Trainer(recorded=True)`): `load(g)` copies a batch in, `()` replays and returns the resident graph whose `energy` / `gradient` the evaluator
    reads.  The per-weight caches are refreshed INSIDE the graph (the weights change between two validation passes)."""

    def __init__(self, model, energy, g, warmup: int = 2):
        if not torch.cuda.is_available():
            raise RuntimeError("CapturedEvalStep needs a GPU")
        self.model, self.energy, self.g = model, energy, g
        self.be = get_backend()
        self.static_inputs = True
        self.plan = g.plan()
        for lvl in ("n2", "n3", "n4", "n4_improper"):
            if self.plan.T[lvl]:
                self.plan.position_tables(lvl)
        self.signature = train_signature(g)
        self._rehome_inputs()
        self.stream = torch.cuda.Stream(device=g.device)
        self.stream.wait_stream(torch.cuda.current_stream(g.device))
        with torch.cuda.stream(self.stream), torch.no_grad():
            for _ in range(max(int(warmup), 3)):
                # (three epochs of eager refreshes: weights of other models age out of the backend's tables and the tables are rebuilt NOW -- a
                # rebuild is a host-to-device copy, which a capture cannot hold; as CapturedForward(refresh_weights=True))
                self.be.invalidate_weights()
                self._eager()
        self.stream.synchronize()
        self.be.invalidate_weights()                         # every per-weight cache is stale: the recorded pass starts by refreshing them
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream), torch.no_grad():
            self._eager()
        self._pinned = _pinned_by_graph(self.be)
        torch.cuda.current_stream(g.device).wait_stream(self.stream)

    def _eager(self):
        _drop_outputs(self.g)
        return self.energy(self.model(self.g))

    def __call__(self):
        self.graph.replay()
        _touch_weight_caches(self.be)
        return self.g


def train_signature(g) -> tuple:
    """everything the recorded train step's kernel arguments and grids depend on: rows of every level, edges, molecules (incl. a padding
    molecule), real molecules (the loss's grid and its 1 / B), the shape of every input tensor"""
    plan = g.plan()
    return (tuple(int(g.num_nodes(nt)) for nt in ("n1", "n2", "n3", "n4", "n4_improper", "g")), int(g.num_edges()), getattr(plan, "n_real_mols", None),
            tuple(sorted((nt, k, tuple(v.shape), str(v.dtype)) for nt, d in g._data.items() for k, v in d.items() if torch.is_tensor(v))))


class CapturedForward:
    """GrappaModel forward (eval, no_grad) on ONE resident graph as a hipGraph; `load(g)` copies another graph of the same shape signature
    into the captured tensors.  Outputs (k / eq per level) are tensors of the graph: read them before the next replay."""

    def __init__(self, model, g, warmup: int = 2, refresh_weights: bool = False):
        """refresh_weights: the per-weight caches (row / column maxima, pair splits) are rebuilt INSIDE the graph at every replay (as in
        CapturedTrainStep), so the graph stays valid while the weights change between replays (predict calls interleaved with training);
        costs the two batched refresh launches per replay, hence only where the weights were seen to change (ForwardCache)"""
        self.model, self.g = model, g
        self.refresh_weights = bool(refresh_weights)
        self.be = get_backend()
        self.plan = g.plan()
        for lvl in ("n2", "n3", "n4", "n4_improper"):       # built eagerly: their construction sorts (host-synchronising torch ops)
            if self.plan.T[lvl]:
                self.plan.position_tables(lvl)
        self._rehome_inputs()
        self.stream = torch.cuda.Stream(device=g.device)
        self.stream.wait_stream(torch.cuda.current_stream(g.device))
        with torch.cuda.stream(self.stream), torch.no_grad():
            for _ in range(max(int(warmup), 3 if self.refresh_weights else 1)):
                if self.refresh_weights:
                    # (three epochs of eager refreshes: weights of other models age out of the backend's tables and the tables are rebuilt NOW --
                    # a rebuild is a host-to-device copy, which a capture cannot hold)
                    self.be.invalidate_weights()
                _drop_outputs(g)
                model(g)
        self.stream.synchronize()
        if self.refresh_weights:
            self.be.invalidate_weights()                     # every per-weight cache is stale now: the recorded forward starts by refreshing them
        self.weights_stamp = self._stamp()
        self.config_stamp = self._config_stamp()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream), torch.no_grad():
            _drop_outputs(g)
            model(g)
            self.outputs: Dict[Tuple[str, str], torch.Tensor] = {(lvl, k): g.nodes[lvl].data[k] for lvl in ("n2", "n3", "n4", "n4_improper")
                                                                for k in ("k", "eq") if k in g.nodes[lvl].data}
            # every output behind one another in ONE buffer (a node of the graph): one copy back to the host instead of six
            self.out_flat = torch.cat([t.reshape(-1).float() for t in self.outputs.values()]) if self.outputs else None
        self._pinned = _pinned_by_graph(self.be)
        torch.cuda.current_stream(g.device).wait_stream(self.stream)
        self.out_host = torch.empty(self.out_flat.shape, dtype=torch.float32).pin_memory() if self.out_flat is not None else None

    def _stamp(self):
        ps = self.__dict__.get("_params")
        if ps is None:                                       # (walking the module tree costs 0.25 ms of a 2.7 ms predict: once)
            ps = self._params = list(self.model.parameters())
        return (self.be._wepoch, sum(p._version for p in ps))

    def _config_stamp(self):
        """what is baked into the recorded launches besides the weights' values: the backend settings a product's plan and arithmetic
        depend on, and WHICH tensors the parameters are (model.to(), load_state_dict(assign=True), p.data = ... replace them: the graph
        would go on reading the old storage, which its pinned caches keep alive)"""
        be = self.be
        ps = self.__dict__.get("_params")
        if ps is None:
            ps = self._params = list(self.model.parameters())
        return (be.gemm_precision_name, getattr(be, "gemm_precision_bwd", None), be.inference_pairs, be.training_pairs, be._tails, be.plan_override,
                be.splitk_reduce, be.weight_pairs_min_rows, be.pairs_min_rows, tuple(p.data_ptr() for p in ps))

    def valid(self) -> bool:
        """the per-weight caches (maxima, pair splits) were filled OUTSIDE the graph: a graph captured before the weights changed must not be
        replayed -- unless it refreshes them itself (refresh_weights).  Settings and parameter storage are checked in either mode."""
        if self._config_stamp() != self.config_stamp:
            return False
        return self.refresh_weights or self._stamp() == self.weights_stamp

    def replay(self):
        self.graph.replay()
        if self.refresh_weights:
            _touch_weight_caches(self.be)
        return self.g

    # ---- another graph of the same shape signature into the captured tensors
    @staticmethod
    def signature(g) -> tuple:
        """everything the recorded kernels' arguments and grids depend on: node counts, edge count, the input features' shapes"""
        return (tuple(int(g.num_nodes(nt)) for nt in ("n1", "n2", "n3", "n4", "n4_improper", "g")), int(g.num_edges()),
                tuple(sorted((k, tuple(v.shape), str(v.dtype)) for k, v in g._data["n1"].items() if torch.is_tensor(v) and k != "h")))

    @staticmethod
    def _plan_tensors(plan):
        out = {}
        for name, val in vars(plan).items():
            if torch.is_tensor(val):
                out[name] = val
            elif isinstance(val, dict):
                for k, v in val.items():
                    if torch.is_tensor(v):
                        out[f"{name}.{k}"] = v
                    elif isinstance(v, tuple):
                        for i, t in enumerate(v):
                            if torch.is_tensor(t):
                                out[f"{name}.{k}.{i}"] = t
        return out

    # ---- the inputs of the recorded graph live in ONE device buffer (every plan table and input feature is a view of it), mirrored by
    # one pinned host buffer: loading another graph of the same signature is a few dozen small host copies and ONE transfer
    def _input_slots(self, plan, g):
        slots = dict(self._plan_tensors(plan))
        for k, t in g._data["n1"].items():
            if torch.is_tensor(t) and k != "h":
                slots["n1." + k] = t
        return slots

    def _rehome_inputs(self) -> None:
        slots = self._input_slots(self.plan, self.g)
        self._layout, off = {}, 0
        for name, t in slots.items():
            nbytes = t.numel() * t.element_size()
            self._layout[name] = (off, nbytes, t.dtype, tuple(t.shape))
            off += (nbytes + 15) // 16 * 16
        self._in_dev = torch.zeros(max(off, 16), dtype=torch.uint8, device=self.g.device)
        self._in_host = torch.zeros(max(off, 16), dtype=torch.uint8).pin_memory()
        self._host_views = {name: self._in_host[o:o + nbytes].view(dtype).view(shape) for name, (o, nbytes, dtype, shape) in self._layout.items()}
        views = {}
        for name, (o, nbytes, dtype, shape) in self._layout.items():
            v = self._in_dev[o:o + nbytes].view(dtype).view(shape)
            v.copy_(slots[name])
            views[name] = v
        # the plan's attributes and the graph's features now point into the buffer
        for name, v in views.items():
            parts = name.split(".")
            if parts[0] == "n1":
                self.g._data["n1"][name[3:]] = v
            elif len(parts) == 1:
                setattr(self.plan, name, v)
            elif len(parts) == 2:
                getattr(self.plan, parts[0])[parts[1]] = v
            else:
                d = getattr(self.plan, parts[0])
                tup = list(d[parts[1]])
                tup[int(parts[2])] = v
                d[parts[1]] = tuple(tup)

    def load(self, g_host) -> None:
        """g_host: a MolBatch on the CPU with this graph's signature.  Its input features and its index structures (plan, position
        tables: built on the host) are laid out in the pinned mirror of the input buffer and sent in ONE transfer; nothing is allocated
        on the device."""
        plan = g_host.plan()
        for lvl in self.plan.__dict__.get("_pos_tables", {}):
            plan.position_tables(lvl)
        src = self._input_slots(plan, g_host)
        if set(src) != set(self._layout):
            raise ValueError("load: the graph has other input tables than the captured one")
        for name, (o, nbytes, dtype, shape) in self._layout.items():
            t = src[name]
            if tuple(t.shape) != shape or t.dtype != dtype:
                raise ValueError(f"load: {name} is {t.dtype} {tuple(t.shape)}, the captured graph holds {dtype} {shape}")
            self._host_views[name].copy_(t)
        self._in_dev.copy_(self._in_host, non_blocking=True)

    def read_outputs(self, g_host) -> None:
        """k / eq of every level from the graph's output buffer (one transfer) into g_host's node data"""
        if self.out_flat is None:
            return
        self.out_host.copy_(self.out_flat, non_blocking=True)
        torch.cuda.current_stream(self.out_flat.device).synchronize()
        off = 0
        for (lvl, k), t in self.outputs.items():
            n = t.numel()
            g_host.nodes[lvl].data[k] = self.out_host[off:off + n].view(t.shape).clone()
            off += n


class ForwardCache:
    """captured forwards by shape signature (`Grappa.predict`): a signature is captured the SECOND time it is seen (a one-off molecule
    never pays for a capture), at most `max_entries` graphs are kept (least recently used out)."""

    def __init__(self, model, device, max_entries: int = 16, max_seen: int = 512):
        self.model, self.device, self.max_entries, self.max_seen = model, torch.device(device), int(max_entries), int(max_seen)
        self.seen: Dict[tuple, int] = {}
        self.entries: "Dict[tuple, CapturedForward]" = {}
        self.refresh_weights = False                         # set once the weights were seen to change between two calls

    def __call__(self, g_host):
        """-> the parametrised graph on the CPU (k / eq written into g_host's tuple levels), or None: run the eager path"""
        sig = CapturedForward.signature(g_host)
        ent = self.entries.get(sig)
        if ent is not None and not ent.valid():
            if ent._config_stamp() != ent.config_stamp:
                # a backend setting or the parameters' storage changed: what the graphs recorded is no longer what an eager call would launch.
                # Drop them; the signatures are recorded again (in the current mode) when they come back
                self.entries.clear()
            else:
                # the weights' VALUES changed: every graph recorded so far reads stale per-weight caches.  From now on the graphs refresh them themselves
                self.entries.clear()
                self.refresh_weights = True
            ent = None
        if ent is None:
            n = self.seen.pop(sig, 0) + 1
            self.seen[sig] = n                               # (most recently seen last)
            if len(self.seen) > self.max_seen:               # a long-running service sees arbitrarily many signatures: forget the oldest
                self.seen.pop(next(iter(self.seen)))
            if n < 2:
                return None
            if len(self.entries) >= self.max_entries:
                self.entries.pop(next(iter(self.entries)))
            ent = CapturedForward(self.model, g_host.to(self.device), refresh_weights=self.refresh_weights)
            self.entries[sig] = ent
            ent.replay()                                     # (recording executes nothing: the outputs are filled by the first replay)
        else:
            self.entries[sig] = self.entries.pop(sig)        # most recently used last
            ent.load(g_host)
            ent.replay()
        ent.read_outputs(g_host)
        return g_host
