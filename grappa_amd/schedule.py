"""Epoch / step schedule of a Grappa training run: learning-rate warm-up after every optimiser
restart, epoch-dependent loss weights, and the early-stopping-metric learning-rate decay.

Mirror of the bookkeeping in the reference's `LitModel` (training/lightning_model.py): `get_lr` :123-141, `set_lr` :144-151,
`on_train_epoch_start` :181-201, `on_validation_epoch_end` :246-293 -- without Lightning: the caller owns the loop and calls
    sched.on_train_epoch_start(epoch, loss_fn, optimizer)   once per epoch
    optimizer.lr = sched.next_lr()                          once per training step
    sched.on_validation_epoch_end(epoch, metrics)           with FastEvaluator.pool() of the validation set
The optimiser is `grappa_amd.optim.FusedAdam` (a restart zeroes its moment buffers and step count, which is what re-creating
`torch.optim.Adam` does in the reference)."""
import time
from typing import Dict, List, Optional

from .evaluation import early_stopping_loss


class TrainSchedule:
    def __init__(self, lr: float = 1e-4, start_qm_epochs: int = 1, add_restarts: List[int] = (), warmup_steps: int = int(2e2),
                 energy_weight: float = 1., gradient_weight: float = 1e-1, tuplewise_weight: float = 0., param_weight: float = 1e-4,
                 early_stopping_energy_weight: float = 2., patience: int = 30, lr_decay: float = 0.8, time_limit: Optional[float] = None,
                 finish_criterion: Dict[float, float] = {}, param_loss_epochs: Optional[int] = None):
        self.lr = lr
        self.start_qm_epochs = start_qm_epochs
        self.restarts = sorted(set([start_qm_epochs] + list(add_restarts)))
        self.warmup_steps = warmup_steps
        self.warmup_step = None
        self.patience, self.lr_decay = patience, lr_decay
        self.energy_weight, self.gradient_weight = energy_weight, gradient_weight
        self.tuplewise_weight, self.param_weight = tuplewise_weight, param_weight
        self.param_loss_epochs = param_loss_epochs
        if param_loss_epochs is not None:
            self.restarts = sorted(set(self.restarts + [param_loss_epochs]))
        self.early_stopping_energy_weight = early_stopping_energy_weight
        self.finish_criterion = dict(finish_criterion)
        self.time_limit = time_limit
        self.elapsed_time = 0.
        self.time_start = time.time()
        self.best_early_stopping_loss = float("inf")
        self.epochs_without_improvement = 0
        self.should_stop = False

    def initial_loss_weights(self) -> Dict[str, float]:
        """before `start_qm_epochs` only the classical-parameter loss is trained (lightning_model.py:60)"""
        if self.start_qm_epochs > 0:
            return dict(gradient_weight=0, energy_weight=0, param_weight=1e-3, tuplewise_weight=self.tuplewise_weight)
        return dict(gradient_weight=self.gradient_weight, energy_weight=self.energy_weight, param_weight=self.param_weight,
                    tuplewise_weight=self.tuplewise_weight)

    def next_lr(self) -> float:
        """linear warm-up from 0 over `warmup_steps` steps after a restart, then `lr` (get_lr :123-141)"""
        if self.warmup_step is not None:
            if self.warmup_step >= self.warmup_steps:
                self.warmup_step = None
                return self.lr
            lr = float(self.warmup_step) / self.warmup_steps * self.lr
            self.warmup_step += 1
            return lr
        return self.lr

    def on_train_epoch_start(self, epoch: int, loss_fn=None, optimizer=None) -> bool:
        """-> True if the optimiser was restarted this epoch"""
        restarted = epoch in self.restarts
        if restarted:
            if optimizer is not None:
                optimizer.reset_state()
                optimizer.lr = self.lr
            self.warmup_step = 0
        if self.param_loss_epochs is not None and epoch >= self.param_loss_epochs:
            self.param_weight = 0.
            self.tuplewise_weight = 0.
        if epoch >= self.start_qm_epochs and loss_fn is not None:
            loss_fn.gradient_weight = float(self.gradient_weight)
            loss_fn.energy_weight = float(self.energy_weight)
            loss_fn.param_weight = float(self.param_weight)
            loss_fn.tuplewise_weight = float(self.tuplewise_weight)
        return restarted

    def on_validation_epoch_end(self, epoch: int, metrics) -> Optional[float]:
        """metrics = FastEvaluator.pool() of the validation loader -> the early-stopping loss (None before the QM epochs)"""
        es = None
        if epoch > self.start_qm_epochs:
            es = early_stopping_loss(metrics, self.early_stopping_energy_weight)
            elapsed = (time.time() - self.time_start + self.elapsed_time) / 3600.
            relevant = {k: v for k, v in self.finish_criterion.items() if k < elapsed}
            if es > (min(relevant.values()) if relevant else float("inf")):
                self.should_stop = True
            if self.patience > 0:
                if es < self.best_early_stopping_loss:
                    self.best_early_stopping_loss = float(es)
                    self.epochs_without_improvement = 0
                else:
                    self.epochs_without_improvement += 1
                if self.epochs_without_improvement > self.patience:
                    self.lr *= self.lr_decay
                    self.epochs_without_improvement = 0
                    self.best_early_stopping_loss = float(es)
        if self.time_limit is not None and time.time() - self.time_start + self.elapsed_time > self.time_limit * 3600.:
            self.should_stop = True
        return es
