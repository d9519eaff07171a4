"""Flat parameter/gradient buffers + fused Adam with global-norm clipping.

The reference trains with torch.optim.Adam(lr, weight_decay) under Lightning's gradient_clip_val=10
(training/lightning_model.py:297-299, lightning_trainer.py:92).  Here all parameters (40.8 M for the
production model) live in ONE fp32 buffer and all gradients in another: the HIP kernels accumulate
into views of the gradient buffer, data parallelism is one all-reduce of it, and the optimiser step is
two kernels (sum of squares, Adam with the clip factor computed on the device -- no host sync).
"""
from typing import Iterable, List, Optional

import torch

from .backend import get_backend


class FlatParams:
    """Re-points every distinct parameter of a module into one contiguous buffer (and its .grad likewise).
    Invariant: `p.grad` of every parameter is a view of `self.grad`.  `FlatParams.zero_grad()` / `FusedAdam.zero_grad()` keep it;
    `module.zero_grad()` (set_to_none) breaks it and the next backward pass restores it (ops._pgrad); `torch.autograd.grad` on these
    parameters is not supported (the kernels write parameter gradients themselves, autograd sees None)."""

    def __init__(self, module: torch.nn.Module):
        params, seen = [], set()
        for p in module.parameters():
            if id(p) not in seen and p.requires_grad:
                seen.add(id(p))
                params.append(p)
        if not params:
            raise ValueError("module has no trainable parameters")
        dev = params[0].device
        sizes = [((p.numel() + 3) // 4) * 4 for p in params]        # keep every view 16-byte aligned
        total = sum(sizes)
        self.data = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        for p, sz in zip(params, sizes):
            n = p.numel()
            self.data[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.data[off:off + n].view(p.shape)
            p.grad = self.grad[off:off + n].view(p.shape)
            p._grappa_flat = (self.grad, off)          # ops._pgrad keeps the gradient inside the flat buffer
            off += sz
        self.params: List[torch.nn.Parameter] = params
        self.numel = total
        self._offsets = {id(p): (o, o + sz) for p, o, sz in zip(params, [sum(sizes[:i]) for i in range(len(sizes))], sizes)}

    def range_of(self, module: torch.nn.Module):
        """[start, end) of the contiguous slice of the flat buffers that holds `module`'s parameters (registration order keeps a
        sub-module's parameters together); raises if they are not contiguous."""
        spans = sorted(self._offsets[id(p)] for p in module.parameters() if id(p) in self._offsets)
        if not spans:
            raise ValueError("module has no parameters in this buffer")
        for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
            if a1 != b0:
                raise ValueError("the module's parameters are not contiguous in the flat buffer")
        return spans[0][0], spans[-1][1]

    def invalidate(self) -> None:
        """call after writing `self.data` directly (broadcast of the parameters, EMA, a checkpoint copied into the flat buffer): such
        writes do not move the parameters' version counters, so the backend's per-weight caches must be told"""
        be = get_backend()
        if hasattr(be, "invalidate_weights"):
            be.invalidate_weights()

    def zero_grad(self):
        be = get_backend()
        if hasattr(be, "drop_deferred"):
            be.drop_deferred()      # products a failed backward pass left queued must not be added to the fresh buffer
        self.grad.zero_()


class FusedAdam:
    def __init__(self, flat: FlatParams, lr: float = 1.5e-5, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 max_grad_norm: Optional[float] = 10.0):
        self.flat = flat
        self._lr_t = self._step_t = None           # device-side learning rate / step count (enable_dynamic: captured steps)
        self.lr, self.betas, self.eps, self.weight_decay, self.max_grad_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.m = torch.zeros_like(flat.data)
        self.v = torch.zeros_like(flat.data)
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=flat.data.device)
        self.step_count = 0

    def zero_grad(self):
        self.flat.zero_grad()

    # ---- device-side scalars: a train step captured in a hipGraph (capture.CapturedTrainStep) freezes every kernel argument, so the
    # learning rate and the step count of the bias corrections are read from device memory there (grappa_adam_step_dyn_f32)
    @property
    def lr(self) -> float:
        return self._lr

    @lr.setter
    def lr(self, value: float) -> None:
        self._lr = float(value)
        if self._lr_t is not None:
            self._lr_t.fill_(self._lr)

    @property
    def step_count(self) -> int:
        return self._step_count

    @step_count.setter
    def step_count(self, value: int) -> None:
        self._step_count = int(value)

    def enable_dynamic(self) -> None:
        if self._lr_t is None:
            dev = self.flat.data.device
            self._lr_t = torch.full((1,), self._lr, dtype=torch.float32, device=dev)
            self._step_t = torch.full((1,), self._step_count, dtype=torch.int32, device=dev)

    def sync_dynamic(self) -> None:
        """host values -> device scalars (after reset_state / load_state_dict)"""
        if self._lr_t is not None:
            self._lr_t.fill_(self._lr)
            self._step_t.fill_(self._step_count)

    def step(self, grad_scale: float = 1.0):
        be = get_backend()
        if hasattr(be, "flush_wgrads"):
            be.flush_wgrads()                      # (normally empty: the backward pass's end-of-pass callback launched them)
        self.step_count += 1
        sumsq = None
        if self.max_grad_norm is not None:
            be.sumsq(self.flat.grad, self.sumsq, accumulate=False)
            sumsq = self.sumsq
        if self._lr_t is not None:
            self._step_t.add_(1)                   # (a kernel of the captured graph: the count advances with every replay)
            be.adam_step_dyn(self.flat.data, self.flat.grad, self.m, self.v, self._lr_t, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                             self._step_t, grad_scale, sumsq, self.max_grad_norm if self.max_grad_norm is not None else 0.0)
        else:
            be.adam_step(self.flat.data, self.flat.grad, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                         self.step_count, grad_scale, sumsq, self.max_grad_norm if self.max_grad_norm is not None else 0.0)
        if hasattr(be, "invalidate_weight_planes"):
            be.invalidate_weight_planes()          # the kernel wrote the parameters through raw pointers: cached bf16 planes are stale

    def reset_state(self):
        """what re-creating torch.optim.Adam does at a restart epoch (training/lightning_model.py:144-151)"""
        self.m.zero_()
        self.v.zero_()
        self.step_count = 0
        self.sync_dynamic()

    def grad_norm(self) -> torch.Tensor:
        return torch.sqrt(self.sumsq[0])

    def state_dict(self):
        return {"m": self.m, "v": self.v, "step": self.step_count, "lr": self.lr}

    def load_state_dict(self, sd):
        self.m.copy_(sd["m"]); self.v.copy_(sd["v"]); self.step_count = int(sd["step"]); self.lr = float(sd.get("lr", self.lr))
        self.sync_dynamic()
