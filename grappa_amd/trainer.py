"""`Trainer`: the reference's training loop (training/lightning_model.py `LitModel` + data/GraphDataLoader.py sampling) on the
MI355X path, without Lightning: dataset resident in HBM (`DeviceDataset`), batches assembled on the device, one train step =
GrappaModel -> Energy -> MolwiseLoss -> backward -> (bucketed all-reduce) -> fused clip + Adam, per-dataset RMSEs from
`FastEvaluator`, learning-rate warm-up / restarts / decay from `TrainSchedule`.  Nothing in the loop copies a tensor to the host
except the scalar loss that is logged once per epoch and the pooled metrics.

Sampling follows GraphDataLoader (data/GraphDataLoader.py:100-138): plain shuffling (a `torch.randperm` per epoch) or, with
per-dataset `weights` / `balance_factor`, weighted sampling with replacement.  Under `torch.distributed` every rank draws the SAME
global batch (same seed) and keeps its size-balanced share of it (`dist.shard_indices`).
"""
from __future__ import annotations

import os
import time
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch
import torch.distributed as tdist

from .device_dataset import DeviceDataset, ShapeBuckets
from .dist import BucketedGradReducer, shard_indices
from .energy import Energy
from .evaluation import FastEvaluator
from .loss import MolwiseLoss
from .optim import FlatParams, FusedAdam
from .schedule import TrainSchedule


def epoch_batches(names: List[str], batch_size: int, shuffle: bool = True, weights: Dict[str, float] = {}, balance_factor: float = 0.,
                  generator: Optional[torch.Generator] = None, min_last: int = 1, sizes: Optional[Sequence[int]] = None,
                  size_window: int = 0) -> List[np.ndarray]:
    """molecule ids of every batch of one epoch (the last batch may be smaller), GraphDataLoader semantics.  A trailing batch of
    fewer than `min_last` molecules is appended to the batch before it (data parallel: every rank needs at least one molecule).
    size_window = W >= 2 with `sizes` (atoms per molecule): the size-aware sampler of SURVEY 8(f) N1.  Batches are ragged
    concatenations -- nothing is padded -- so what sizes can buy is EQUAL WORK per step, not less padding: the draws of every W
    consecutive batches are sorted by size and dealt to those W batches in snake order, which keeps the epoch's draws (and every
    window's) exactly as sampled and evens out the atoms per batch (step time, workspace high-water mark, rank balance)."""
    n = len(names)
    assert 0 <= balance_factor <= 1, f"balance_factor must be between 0 and 1, but got {balance_factor}"
    if shuffle and (len(weights) or balance_factor > 0):
        w = np.array([weights.get(x, 1.0) for x in names], dtype=np.float64)
        if balance_factor > 0:
            occ = {x: names.count(x) / n for x in set(names)}
            balanced = 1.0 / float(len(occ))
            ratio = {x: float((1.0 - balance_factor) * balanced + balance_factor * occ[x]) for x in occ}
            w = w * np.array([1.0 / ratio[x] for x in names])
        order = torch.multinomial(torch.as_tensor(w, dtype=torch.double), n, replacement=True, generator=generator).numpy()
    elif len(weights) > 0:
        raise ValueError("Weights are only supported with shuffle=True")
    elif shuffle:
        order = torch.randperm(n, generator=generator).numpy()
    else:
        order = np.arange(n)
    batches = [order[i:i + batch_size] for i in range(0, n, batch_size)]
    if size_window >= 2 and sizes is not None and len(batches) > 1:
        sz = np.asarray(sizes)
        for w0 in range(0, len(batches), size_window):
            group = batches[w0:w0 + size_window]
            full = [b for b in group if len(b) == batch_size]          # a short last batch keeps its draws
            if len(full) < 2:
                continue
            pool = np.concatenate(full)
            pool = pool[np.argsort(-sz[pool], kind="stable")]
            lanes = [[] for _ in full]
            for r in range(0, len(pool), len(full)):                     # snake deal: 0..W-1, W-1..0, ...
                chunk = pool[r:r + len(full)]
                idx = range(len(chunk)) if (r // len(full)) % 2 == 0 else range(len(full) - 1, len(full) - 1 - len(chunk), -1)
                for lane, m in zip(idx, chunk):
                    lanes[lane].append(m)
            it = iter(lanes)
            for i, b in enumerate(group):
                if len(b) == batch_size:
                    batches[w0 + i] = np.asarray(next(it), dtype=order.dtype)
    if len(batches) > 1 and len(batches[-1]) < min_last:
        tail = batches.pop()
        batches[-1] = np.concatenate([batches[-1], tail])
    if batches and len(batches[-1]) < min_last:
        raise ValueError(f"{n} molecules cannot be dealt to {min_last} ranks")
    return batches


class Trainer:
    def __init__(self, model, train_set: DeviceDataset, val_set: Optional[DeviceDataset] = None, batch_size: int = 32,
                 conf_strategy: Union[str, int] = 32, val_batch_size: int = 32, val_conf_strategy: Union[str, int] = "max",
                 lr: float = 1.5e-5, weight_decay: float = 0., gradient_clip_val: Optional[float] = 10.0,
                 proper_regularisation: float = 1e-3, improper_regularisation: float = 0., param_weights_by_dataset: Dict[str, float] = {},
                 weights: Dict[str, float] = {}, balance_factor: float = 0., seed: int = 0, size_window: int = 0, recorded: bool = False,
                 shape_buckets: int = 4, max_recorded_steps: int = 16, pipelined: bool = True, **schedule_kwargs):
        """recorded: every train step is the replay of a hipGraph (capture.CapturedTrainStep) -- the batches of an epoch are padded to a handful
        of shapes (`shape_buckets`, device_dataset.ShapeBuckets: a padding molecule with all-dummy conformations behind the real ones, which the
        loss skips), one graph is recorded per shape the first time it occurs and every later batch of that shape is copied into the graph's
        inputs.  The step is then bound by the GPU instead of the host's ~640 launches (batch 32: 2x).  One GPU only; a batch that fits no
        bucket runs eagerly (counted in `recorded_stats`).  pipelined: the next batch is prepared (padding molecule, device collate, position tables:
        ~3 ms of host work, on a side stream) by a worker thread while this thread launches the current batch's graph.
        Without a GPU the same epochs run eagerly.  A run resumed from a checkpoint records its graphs anew: with dropout on it continues with
        other masks than the uninterrupted run (a recorded step's dropout seeds are constants of its graph, salted by a device word), i.e. it
        is equivalent but not bit-identical; the eager trainer's resume is bit for bit."""
        self.model, self.train_set, self.val_set = model, train_set, val_set
        self.recorded, self.shape_buckets, self.max_recorded_steps = bool(recorded), int(shape_buckets), int(max_recorded_steps)
        self._buckets: Optional[ShapeBuckets] = None
        self._steps: Dict[tuple, object] = {}
        self._prep_stream, self._load_done = None, None
        self._unrecordable: Dict[tuple, str] = {}          # shape signatures whose recording failed (reason): served eagerly
        self._val_buckets: Optional[ShapeBuckets] = None
        self._eval_steps: Dict[tuple, object] = {}
        self.pipelined, self._prep_pool = bool(pipelined), None
        self.recorded_stats = {"replayed": 0, "eager": 0, "graphs_recorded": 0, "padding_rows": 0, "real_rows": 0}
        self._graphs_off = False           # set when the recorded mode turned out not to be usable for this run (train_epoch)
        self.batch_size, self.conf_strategy = batch_size, conf_strategy
        self.val_batch_size, self.val_conf_strategy = val_batch_size, val_conf_strategy
        self.weights, self.balance_factor = dict(weights), balance_factor
        self.size_window = int(size_window)            # >= 2: batches of equal work (epoch_batches)
        self.schedule = TrainSchedule(lr=lr, **schedule_kwargs)
        self.loss_fn = MolwiseLoss(proper_regularisation=proper_regularisation, improper_regularisation=improper_regularisation,
                                   param_weights_by_dataset=param_weights_by_dataset, **self.schedule.initial_loss_weights())
        self.energy = Energy()
        self.flat = FlatParams(model)
        self.opt = FusedAdam(self.flat, lr=lr, weight_decay=weight_decay, max_grad_norm=gradient_clip_val)
        self.reducer = BucketedGradReducer(model, self.flat)
        self.evaluator = FastEvaluator()
        self.gen = torch.Generator().manual_seed(seed)
        self.world = tdist.get_world_size() if tdist.is_available() and tdist.is_initialized() else 1
        self.rank = tdist.get_rank() if self.world > 1 else 0
        self.history: List[Dict] = []
        self.next_epoch = 0

    # ------------------------------------------------------------------------------------------------------------------
    def _my_share(self, ids: np.ndarray) -> np.ndarray:
        if self.world == 1:
            return ids
        sizes = self.train_set.count["n1"][ids]
        return ids[shard_indices(sizes.tolist(), self.world, self.rank)]

    def train_step(self, ids: np.ndarray) -> torch.Tensor:
        self.opt.lr = self.schedule.next_lr()
        mine = self._my_share(np.asarray(ids))
        g, names = self.train_set.collate(mine, self.conf_strategy)
        self.loss_fn.global_batch_size = len(ids) if self.world > 1 else None
        self.opt.zero_grad()
        loss = self.loss_fn(self.energy(self.model(g)), list(names))
        loss.backward()
        self.reducer.finish()
        self.opt.step()
        return loss.detach()

    # ---- recorded steps -------------------------------------------------------------------------------------------------------------
    def _step_stamp(self) -> tuple:
        """what a recorded step has baked in besides the batch's shape: the loss weights the schedule moves, the optimiser's constants"""
        lf, o = self.loss_fn, self.opt
        from . import ops
        from .backend import get_backend
        be = get_backend()
        pw = getattr(self.model, "parameter_writer", None)
        #  ... and the backend settings baked into the recorded launches: arithmetic, operand formats, plan options, head streams,
        # activation storage, fused layer switches -- a setting changed in the middle of a run means new graphs, not stale ones
        backend_cfg = tuple(getattr(be, k, None) for k in ("gemm_precision_name", "gemm_precision_bwd_name", "inference_pairs", "training_pairs", "backward_pairs",
                                                            "_tails", "plan_override", "splitk_reduce", "weight_pairs_min_rows", "pairs_min_rows",
                                                            "fused_writer_layer", "fused_writer_layer_bwd", "group_launches", "wgrads_aside"))
        return (float(lf.gradient_weight), float(lf.energy_weight), float(lf.param_weight), float(lf.tuplewise_weight), float(lf.proper_regularisation),
                float(lf.improper_regularisation), tuple(sorted(lf.weights.items())), tuple(sorted(lf.param_weights_by_dataset.items())),
                tuple(o.betas), float(o.eps), float(o.weight_decay), o.max_grad_norm, bool(self.model.training), backend_cfg,
                str(ops.act_dtype()), getattr(pw, "head_streams", None), getattr(pw, "merged_heads", None))

    def calibrate_buckets(self, batches: Sequence[np.ndarray]) -> ShapeBuckets:
        self._buckets = ShapeBuckets(self.train_set, batches, n_buckets=self.shape_buckets)
        self.train_set.enable_padding(self._buckets.max_pad)
        return self._buckets

    def _prepare_padded(self, ids: np.ndarray, caps, tot):
        """the padded batch of `ids`, assembled on a SIDE stream: padding molecule, device collate, the parameter-loss weights, the position
        tables -- ~150 small copies and kernels that would otherwise sit on the main stream between two replays (1.5-3 ms of an 8 ms step);
        here they run while the previous step's graph executes.  -> (graph, names, event recorded behind the last of them)"""
        dev = self.train_set.device
        if self._prep_stream is None:
            self._prep_stream = torch.cuda.Stream(device=dev)
            self._prep_stream.wait_stream(torch.cuda.current_stream(dev))
        side = self._prep_stream
        if self._load_done is not None:
            side.wait_event(self._load_done)         # the previous batch's tensors (this stream's allocations) were read by the main stream's copy-in
        with torch.cuda.stream(side):
            g, names = self.train_set.collate(ids, self.conf_strategy, pad_to=caps)
            plan = g.plan()
            # the per-molecule weights of the parameter loss are an input of the graph (MolwiseLoss reads plan.param_weight_rows)
            plan.param_weight_rows = self.loss_fn.param_weights_of(list(names), plan.B).pin_memory().to(plan.device, non_blocking=True)
            for lvl in ("n2", "n3", "n4", "n4_improper"):
                if plan.T[lvl]:
                    plan.position_tables(lvl)
            ready = torch.cuda.Event()
            ready.record(side)
        return g, names, ready

    def _prepare(self, ids: np.ndarray):
        """what a recorded step needs of the batch `ids`: None if no bucket takes it (it runs eagerly), else (graph, names, event, totals, caps).
        Called on the trainer's thread, or -- pipelined epochs -- on its worker thread while the trainer's thread launches the previous graph"""
        ids = self._my_share(np.asarray(ids))              # data parallelism: this rank's molecules of the batch (the buckets were cut on shards)
        tot = self.train_set.totals(ids)
        caps = self._buckets.choose(tot) if self._buckets is not None else None
        if caps is None or any(caps[k] - tot[k] > self.train_set.pad_caps[k] for k in tot):
            return None
        if self.train_set.device.type == "cuda" and self.train_set.device.index is not None:
            torch.cuda.set_device(self.train_set.device)       # (a worker thread starts on device 0)
        return self._prepare_padded(ids, caps, tot) + (tot, caps)

    def train_step_recorded(self, ids: np.ndarray, prepared="unset", after_load=None) -> torch.Tensor:
        """prepared: the result of `_prepare(ids)` if the caller made it ahead; after_load: called once this batch is in the graph's inputs and before
        the graph is launched (pipelined epochs start the next batch's preparation there: launching a hipGraph of ~650 kernel nodes keeps the
        launching thread in the runtime for ~5 ms -- with the interpreter lock released)"""
        from .capture import CapturedTrainStep, train_signature
        ids = np.asarray(ids)
        if isinstance(prepared, str):
            prepared = self._prepare(ids)
        if prepared is None:
            self.recorded_stats["eager"] += 1
            if self._prep_stream is not None:
                torch.cuda.current_stream(self.train_set.device).wait_stream(self._prep_stream)
                self._prep_stream.wait_stream(torch.cuda.current_stream(self.train_set.device))     # (the eager collate rewrites nothing of the side stream's, but keep the order plain)
            loss = self.train_step(ids)
            if after_load is not None:
                after_load()                    # (after the eager collate: the conformation selection draws from the same host generator, batch by batch)
            return loss
        lr = self.schedule.next_lr()
        g, names, ready, tot, caps = prepared
        main = torch.cuda.current_stream(self.train_set.device)
        main.wait_event(ready)
        # data parallelism: the loss's 1 / B_global is a constant of the recorded step, so it is part of the key; the step is
        # two graphs with the eager all-reduce between them (capture.CapturedTrainStep reducer=)
        self.loss_fn.global_batch_size = len(ids) if self.world > 1 else None
        reducer = self.reducer if self.world > 1 else None
        key = (train_signature(g), self._step_stamp(), self.loss_fn.global_batch_size)
        step = self._steps.pop(key, None) if key not in self._unrecordable else None
        if step is None and key not in self._unrecordable:
            if len(self._steps) >= self.max_recorded_steps:
                old = self._steps.pop(next(iter(self._steps)))           # least recently used out
                self._release(old)
            try:
                step = CapturedTrainStep(self.model, self.energy, self.loss_fn, self.opt, g, preserve_state=True, static_inputs=True, reducer=reducer)
                self.recorded_stats["graphs_recorded"] += 1
            except Exception as e:  # noqa: BLE001  (a recording that fails -- memory, a call a capture cannot hold -- must not end the run)
                import warnings
                self._unrecordable[key] = repr(e)
                warnings.warn(f"Trainer(recorded=True): recording a train step failed ({e!r}); batches of this shape run eagerly")
                step = None
        elif step is not None:
            try:
                step.load(g)
            except Exception as e:  # noqa: BLE001  (an input-table or shape mismatch of THIS batch must not end the run: it runs eagerly)
                import warnings
                warnings.warn(f"Trainer(recorded=True): a batch did not fit its recorded step ({e!r}); it runs eagerly")
                self._steps[key] = step
                step = None
        if step is None:
            # the padded batch itself, eagerly: the loss skips its padding molecule (parameters and optimiser state are as before the attempt)
            self.recorded_stats["eager"] += 1
            self.opt.lr = lr
            self.opt.zero_grad()
            loss = self.loss_fn(self.energy(self.model(g)), list(names))
            loss.backward()
            if reducer is not None:
                reducer.finish()
            self.opt.step()
            self._load_done = torch.cuda.Event()
            self._load_done.record(main)
            if after_load is not None:
                after_load()
            return loss.detach()
        self._load_done = torch.cuda.Event()
        self._load_done.record(main)
        if after_load is not None:
            after_load()
        self._steps[key] = step                                          # most recently used last
        self.opt.lr = lr
        self.recorded_stats["replayed"] += 1
        self.recorded_stats["real_rows"] += sum(tot.values())
        self.recorded_stats["padding_rows"] += sum(caps[k] - tot[k] for k in tot)
        return step()

    def _release(self, step) -> None:
        """an evicted recorded step: the backend's per-stream workspaces it pinned are dropped with it (one entry per capture stream
        was never pruned), its memory pool goes back to the allocator with the graph object"""
        from .backend import get_backend
        be = get_backend()
        st = getattr(step, "stream", None)
        if st is not None:
            for k in [k for k in list(be._ws) if isinstance(k, tuple) and len(k) >= 2 and k[1] == st.cuda_stream]:
                be._ws.pop(k, None)
            for k in [k for k in list(be._side_streams) if isinstance(k, tuple) and len(k) >= 2 and k[1] == st.cuda_stream]:
                be._side_streams.pop(k, None)

    def train_epoch(self, epoch: int) -> float:
        self.model.train()
        self._unrecordable.clear()                  # (a shape that failed to record once -- e.g. out of memory next to other graphs -- gets another try per epoch)
        self.schedule.on_train_epoch_start(epoch, self.loss_fn, self.opt)
        total, count = None, 0
        batches = epoch_batches(self.train_set.names, self.batch_size, True, self.weights, self.balance_factor, self.gen, min_last=self.world,
                                sizes=self.train_set.count["n1"] if self.size_window >= 2 else None, size_window=self.size_window)
        use_graphs = self.recorded and torch.cuda.is_available() and not self._graphs_off
        if use_graphs and self._buckets is None:
            # (what the docstring promises -- a dataset the padded batches cannot serve, or a calibration that finds no caps, means the
            #  eager step for the whole run, with one warning, not an exception in the middle of fit())
            try:
                if not getattr(self.train_set, "bonds_are_n2", True):
                    raise ValueError("the dataset's bonds are not its n2 tuples: batches of a fixed shape cannot be padded")
                self.calibrate_buckets([self._my_share(np.asarray(b)) for b in batches])      # data parallel: this rank's shards are what it records
            except Exception as e:  # noqa: BLE001
                import warnings
                warnings.warn(f"Trainer(recorded=True): {e!r}; every step runs eagerly")
                self._graphs_off, use_graphs = True, False
        nxt = None
        if use_graphs and self.pipelined and len(batches) > 1:
            if self._prep_pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._prep_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="grappa-prep")
            nxt = self._prep_pool.submit(self._prepare, np.asarray(batches[0]))
        for bi, ids in enumerate(batches):
            if nxt is not None:
                # the batch was prepared while the previous graph was being launched; the next one is handed to the worker as soon as this one
                # sits in its graph's inputs
                try:
                    prepared, box = nxt.result(), []
                except Exception as e:  # noqa: BLE001  (the worker's preparation failed: this batch runs eagerly, the run goes on)
                    import warnings
                    warnings.warn(f"Trainer(recorded=True): preparing a padded batch failed ({e!r}); it runs eagerly")
                    prepared, box = None, []
                follow = (lambda j=bi + 1: box.append(self._prep_pool.submit(self._prepare, np.asarray(batches[j])))) if bi + 1 < len(batches) else None
                loss = self.train_step_recorded(ids, prepared, follow)
                nxt = box[0] if box else None
            else:
                loss = self.train_step_recorded(ids) if use_graphs else self.train_step(ids)
            if use_graphs:
                loss = loss.clone()                    # (a recorded step returns the graph's own loss tensor: the next replay overwrites it)
            total = loss * len(ids) if total is None else total + loss * len(ids)      # stays on the device
            count += len(ids)
        if self.world > 1:                 # a rank's loss is its share of sum_m l_m / B_global: the batch loss is the sum over ranks
            tdist.all_reduce(total, op=tdist.ReduceOp.SUM)
        return float(total) / max(count, 1)                                            # the epoch's only host sync

    @torch.no_grad()
    def validate(self, epoch: int):
        if self.val_set is None:
            return None
        self.model.eval()
        # data parallel: the validation batches are dealt to the ranks round-robin and the per-dataset squared-error sums added over the
        # ranks before they are pooled (every rank ends with the same metrics, hence the same schedule decisions)
        if self.world > 1:
            self.evaluator.register(sorted(set(self.val_set.names)), self.flat.data.device)
        batches = epoch_batches(self.val_set.names, self.val_batch_size, shuffle=False)
        use_graphs = self.recorded and self.world == 1 and torch.cuda.is_available() and self.val_set.bonds_are_n2
        if use_graphs and self._val_buckets is None:
            # the validation batches are the same every epoch: their shapes (bucket x conformations) are recorded once and replayed from then on
            self._val_buckets = ShapeBuckets(self.val_set, batches, n_buckets=self.shape_buckets)
            self.val_set.enable_padding(self._val_buckets.max_pad)
        for b, ids in enumerate(batches):
            if b % self.world != self.rank:
                continue
            if use_graphs and self._validate_recorded(np.asarray(ids)):
                continue
            g, names = self.val_set.collate(ids, self.val_conf_strategy)
            self.evaluator.step(self.energy(self.model(g)), list(names))
        if self.world > 1:
            self.evaluator.all_reduce()
        metrics = self.evaluator.pool()
        es = self.schedule.on_validation_epoch_end(epoch, metrics)
        return metrics, es

    def _validate_recorded(self, ids: np.ndarray) -> bool:
        """one validation batch through a recorded forward + energy pass (capture.CapturedEvalStep); False: no bucket takes it or its recording
        failed -- the caller runs it eagerly"""
        from .capture import CapturedEvalStep, train_signature
        tot = self.val_set.totals(ids)
        caps = self._val_buckets.choose(tot)
        if caps is None or any(caps[k] - tot[k] > self.val_set.pad_caps[k] for k in tot):
            return False
        g, names = self.val_set.collate(ids, self.val_conf_strategy, pad_to=caps)
        key = (train_signature(g), "eval")
        if key in self._unrecordable:
            return False
        step = self._eval_steps.pop(key, None)
        if step is None:
            if len(self._eval_steps) >= 2 * self.max_recorded_steps:
                self._eval_steps.pop(next(iter(self._eval_steps)))
            try:
                step = CapturedEvalStep(self.model, self.energy, g)
                self.recorded_stats["eval_graphs_recorded"] = self.recorded_stats.get("eval_graphs_recorded", 0) + 1
            except Exception as e:  # noqa: BLE001
                import warnings
                self._unrecordable[key] = repr(e)
                warnings.warn(f"Trainer(recorded=True): recording a validation pass failed ({e!r}); batches of this shape run eagerly")
                return False
        else:
            step.load(g)
        self._eval_steps[key] = step
        self.evaluator.step(step(), list(names))
        self.recorded_stats["eval_replayed"] = self.recorded_stats.get("eval_replayed", 0) + 1
        return True

    def fit(self, max_epochs: int, log=None, checkpoint: Optional[str] = None, checkpoint_every: int = 1) -> List[Dict]:
        """epochs [next_epoch, max_epochs): a fresh trainer starts at 0, one that has loaded a checkpoint where that run stopped.
        checkpoint: path of the resumable state written every `checkpoint_every` epochs (the reference's `last.ckpt`)"""
        for epoch in range(self.next_epoch, max_epochs):
            rec = {"epoch": epoch, "train_loss": self.train_epoch(epoch), "lr": self.schedule.lr}
            val = self.validate(epoch)
            if val is not None:
                rec["val_metrics"], rec["early_stopping_loss"] = val
            self.history.append(rec)
            self.next_epoch = epoch + 1
            if log is not None:
                log(rec)
            if checkpoint is not None and (self.next_epoch % max(int(checkpoint_every), 1) == 0 or self.schedule.should_stop):
                self.save_checkpoint(checkpoint)
            if self.schedule.should_stop:
                break
        return self.history

    # ---- resumable state (reference: Lightning's last.ckpt + resume_trainrun.py; here one file, written by rank 0)
    _SCHEDULE_STATE = ("lr", "warmup_step", "param_weight", "tuplewise_weight", "best_early_stopping_loss", "epochs_without_improvement", "should_stop")

    def checkpoint_dict(self) -> Dict:
        """everything the next epoch depends on: parameters, Adam moments and step count, the schedule's counters and the loss weights
        it has set, the samplers' generators (batch order; dropout seeds), elapsed time, history.  A run resumed from it repeats the
        uninterrupted run bit for bit (tests/test_trainer.py)."""
        from . import ops
        self.schedule.elapsed_time += time.time() - self.schedule.time_start
        self.schedule.time_start = time.time()
        return {"format": "grappa_amd.trainer/1", "next_epoch": self.next_epoch, "model": self.model_dict(),
                "optimizer": {k: (v.detach().cpu().clone() if torch.is_tensor(v) else v) for k, v in self.opt.state_dict().items()},
                "schedule": {k: getattr(self.schedule, k) for k in self._SCHEDULE_STATE}, "elapsed_time": self.schedule.elapsed_time,
                "loss_weights": {k: getattr(self.loss_fn, k) for k in ("gradient_weight", "energy_weight", "param_weight", "tuplewise_weight")},
                "sampler_generator": self.gen.get_state(), "dropout_seed": dict(ops._SEED), "torch_rng": torch.get_rng_state(),
                "history": list(self.history)}

    @staticmethod
    def _plain(o):
        """numpy scalars -> Python numbers, recursively: the file then holds tensors and plain containers only and loads with weights_only=True"""
        import numpy as np
        if isinstance(o, dict):
            return {k: Trainer._plain(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return type(o)(Trainer._plain(v) for v in o)
        if isinstance(o, np.generic):
            return o.item()
        return o

    def save_checkpoint(self, path: str) -> None:
        if self.rank == 0:
            tmp = f"{path}.tmp"
            torch.save(self._plain(self.checkpoint_dict()), tmp)
            os.replace(tmp, path)              # a run killed while writing leaves the previous checkpoint intact

    def load_checkpoint(self, path: str, trusted=None) -> int:
        """-> the epoch the run continues with (`fit(max_epochs)` picks it up).  Every rank loads the same file."""
        from . import ops
        from .loading import _torch_load
        ck = _torch_load(path, trusted)          # tensors and plain containers only (a tampered file cannot run code) unless the caller vouches
        if ck.get("format") != "grappa_amd.trainer/1":
            raise ValueError(f"{path} is not a trainer checkpoint of this engine (format {ck.get('format')!r})")
        self.model.load_state_dict(ck["model"]["state_dict"])
        self.flat.invalidate()                  # the parameters changed under the flat buffer's feet: cached weight maxima / planes are stale
        self.opt.load_state_dict({k: (v.to(self.flat.data.device) if torch.is_tensor(v) else v) for k, v in ck["optimizer"].items()})
        for k, v in ck["schedule"].items():
            setattr(self.schedule, k, v)
        self.schedule.elapsed_time, self.schedule.time_start = float(ck["elapsed_time"]), time.time()
        for k, v in ck["loss_weights"].items():
            setattr(self.loss_fn, k, v)
        self.gen.set_state(ck["sampler_generator"])
        ops._SEED.update(ck["dropout_seed"])
        torch.set_rng_state(ck["torch_rng"])
        self.history = list(ck["history"])
        self.next_epoch = int(ck["next_epoch"])
        return self.next_epoch

    # ---- export in the reference's container format (utils/loading_utils.py:64-73, training/export_model.py:84-97)
    def model_dict(self) -> Dict:
        return {"state_dict": {k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()},
                "config": {"model_config": dict(self.model.model_config)}, "split_names": None}

    def save(self, path: str) -> None:
        torch.save(self.model_dict(), path)
