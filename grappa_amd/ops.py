"""Block-level autograd nodes of the Grappa hot path.

Each `torch.autograd.Function` below is one block of the reference's model with a hand-sequenced
forward AND backward made of C-ABI kernel calls (grappa_amd.backend): nothing inside a block is
differentiated by torch.  Gradient fan-in inside a block (residual branches) is folded into GEMM
epilogues (`res=`), parameter gradients are accumulated by the kernels straight into `param.grad`
(flat-buffer friendly: one RCCL all-reduce / one fused Adam over all of them), and dropout masks are
regenerated from a counter-based hash instead of being stored.

Reference blocks restated (file:line under /root/reference/src/grappa/):
  LinearFn            models/graph_attention.py:98-101, :125-127 (pre_dense / post_dense)
  AttBlockFn          models/graph_attention.py:276-310  (ResidualAttentionBlock + DGL DotGatConv)
  ConvBlockFn         models/graph_attention.py:383-415  (ResidualConvBlock + DGL SAGEConv 'mean')
  ProjGatherFn        models/interaction_parameters.py:155-180 (RepProjector) + perm_equiv_transformer.py:134-141
  TransformerLayerFn  models/network_utils.py:112-133 (DottedAttWithMLP) incl. :44-54 (FeedForwardLayer)
  SymmetriserFn       models/perm_equiv_transformer.py:239-276
  ParamOutFn          models/interaction_parameters.py:252-266, :347-362, :536-560
  MMEnergyFn          models/energy.py:99-145 + models/internal_coordinates.py:15-125
  MolwiseLossFn       training/loss.py:45-167
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import torch
from torch.autograd import Function

from .backend import get_backend
from .constants import TUPLE_LEVELS

ELU = 1
_SEED = {"base": 0x5DEECE66D, "counter": 0}


# the first transformer layer of a writer on (atom, position) rows instead of tokens where those are far fewer (ProjFirstLayerFn);
# GRAPPA_FIRST_LAYER_ROWS=0: always the token formulation (ProjGatherFn + TransformerLayerFn)
FIRST_LAYER_ON_ATOM_ROWS = os.environ.get("GRAPPA_FIRST_LAYER_ROWS", "1") not in ("0", "")


def manual_seed(seed: int) -> None:
    """seed of the counter-based dropout masks (independent of torch's generators)."""
    _SEED["base"] = int(seed) & (2 ** 63 - 1)
    _SEED["counter"] = 0


def next_seed() -> int:
    _SEED["counter"] += 1
    return (_SEED["base"] * 6364136223846793005 + _SEED["counter"] * 1442695040888963407) & (2 ** 63 - 1)


F32 = torch.float32
_ACT = {"dtype": torch.bfloat16 if os.environ.get("GRAPPA_ACT_DTYPE", "f32") == "bf16" else None}


def set_activation_dtype(name) -> None:
    """"f32" (default) or "bf16": the bf16 STORAGE configuration (BASELINE configs[2], "bf16, MFMA dense heads"): every activation
    and activation gradient of the GNN blocks and the writer heads lives in HBM as bf16 (half the bytes of every HBM-bound kernel,
    dense products straight from bf16 operands by LDS-DMA); LayerNorm statistics, softmax, accumulation, the atom embedding `h`, the
    parameters k / eq, energies, forces, the loss, weights and weight gradients stay fp32.  Pair it with
    `backend.set_gemm_precision("bf16")` so that the few products that still read fp32 operands round them the same way."""
    if name not in ("f32", "bf16", None):
        raise ValueError(f"activation dtype {name!r}: expected 'f32' or 'bf16'")
    _ACT["dtype"] = torch.bfloat16 if name == "bf16" else None


def act_dtype():
    """element type the two entry points of the activation chain (pre_dense, the rep projectors) produce; None = float32"""
    return _ACT["dtype"]


def _new(shape, like: torch.Tensor, dtype=None) -> torch.Tensor:
    """activation buffer in the element type of `like` (float32, or bfloat16 in the bf16 storage configuration) unless `dtype` says
    otherwise: statistics, softmax weights, parameters' gradients and everything the user reads stay float32"""
    return torch.empty(shape, dtype=like.dtype if dtype is None else dtype, device=like.device)


def _zeros(shape, like: torch.Tensor, dtype=None) -> torch.Tensor:
    return torch.zeros(shape, dtype=like.dtype if dtype is None else dtype, device=like.device)


def _pgrad(p: torch.Tensor) -> torch.Tensor:
    """gradient buffer of a parameter (kernels accumulate into it).  A parameter that lives in a `FlatParams` buffer keeps its
    gradient INSIDE the flat gradient buffer (the all-reduce and the fused Adam read that buffer, not `p.grad`): if something
    detached it -- `module.zero_grad()` / `optimizer.zero_grad(set_to_none=True)` set `p.grad = None` -- the view is restored and
    its slice zeroed (which is what the caller asked for); a foreign tensor in `p.grad` is an error."""
    home = getattr(p, "_grappa_flat", None)
    if home is not None:
        flat_grad, lo = home
        view = flat_grad[lo:lo + p.numel()].view(p.shape)
        if p.grad is None:
            view.zero_()
            p.grad = view
        elif p.grad.data_ptr() != view.data_ptr():
            raise RuntimeError("a parameter of a FlatParams buffer has a .grad outside of the flat gradient buffer; use "
                               "FlatParams.zero_grad() / FusedAdam.zero_grad() (torch.autograd.grad and optimizers that replace "
                               ".grad are not supported with flat buffers)")
        return p.grad
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


def _wants_amax(be, backward: bool):
    """True where the products of that pass run on fp16 pieces and read row maxima; otherwise the epilogue need not write them.  (True
    still makes `gemm` return the pair (record of A, record of the output), which is what the callers unpack.)"""
    f = getattr(be, "wants_amax", None)
    return True if f is None else (True if f(backward) else "pair")


def _wgrads_aside(be, all_streams: bool = False) -> None:
    """end of a GNN block's backward pass (or of the writer heads'): the weight gradients queued so far go out on a side stream beside
    the small products that follow (backend.launch_wgrads_aside)"""
    f = getattr(be, "launch_wgrads_aside", None)
    if f is not None:
        f(all_streams)


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


def _rows(t: torch.Tensor) -> torch.Tensor:
    """a (rows, cols) table the kernels can address as it stands (unit inner stride, rows that do not overlap), else a contiguous copy"""
    return t if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1] else t.contiguous()


def _padded_cols(rows: int, cols: int, like: torch.Tensor) -> torch.Tensor:
    """zeroed (rows, cols) view of a table whose rows are a multiple of four floats long: a weight-gradient product reads its operands
    across their rows in 16-byte pieces only from such tables (gemm_f32.hip grouped products; an odd leading dimension -- the 511
    columns of a writer's projection -- sends it to the scalar-load kernel, 95 instead of 19 us at C2)"""
    return _zeros((rows, (cols + 3) // 4 * 4), like)[:, :cols]


# ------------------------------------------------------------------------------------------------
# shared sub-sequences (plain functions on the backend; used inside the Functions)
def _linear_bwd_params(be, dz, x, w, b, xs=None, zs=None):
    """dW += dz^T x ; db += colsum(dz).  xs / zs: the backend's records of the largest magnitudes of x (what the forward product
    returned) and of dz (what the kernel that produced dz returned), or None; returns the record for dz, to be handed to the
    input-gradient product that follows."""
    N, K = w.shape                                  # (dz / x may be None: the record carries the tensor in the pair format)
    M = (dz if dz is not None else zs.pairs).shape[0]
    if M == 0:
        return zs
    want_b = b is not None and b.requires_grad
    if w.requires_grad:
        # the bias gradient colsum(dz) is produced by the same kernel from the dz tiles it stages anyway; on the GPU the products of
        # one backward pass are queued and launched as a group (backend.gemm_wgrad)
        if hasattr(be, "gemm_wgrad"):
            return be.gemm_wgrad(dz, x, _pgrad(w), _pgrad(b) if want_b else None, x_scales=xs, dz_scales=zs)
        return be.gemm(dz, x, _pgrad(w), M=N, N=K, K=M, a_kcontig=False, b_kcontig=False, accumulate=True,
                       a_colsum=_pgrad(b) if want_b else None, b_scales=xs, a_scales=zs)
    if want_b:
        be.colsum(dz, _pgrad(b), accumulate=True)
    return zs


def _pairs(be, x, lean) -> bool:
    """HIP backend: may the producer of the rows `x` (or of rows of its shape) hand them on in the pair format?  lean: 1 = inference
    (no gradient will be asked for), 2 = training with the pair format as the storage format of the products' operands (round 4:
    LayerNorm, the tuple attention and the dropout backward write pairs ONLY; the weight-gradient products read them through C ABI 8)"""
    return bool(lean) and hasattr(be, "pairs_ok") and x.dim() == 2 and be.pairs_ok(x, x.shape[1], training=lean == 2)


def _bwd_pairs(be, M, width, *weights) -> bool:
    """backward pass, training with pairs: may a row-wise producer of (M, width) gradient rows write them in the pair format only?  (Every
    weight behind them must take a gradient: the bias gradient of a frozen weight would need the fp32 rows.)"""
    f = getattr(be, "training_pairs_ok", None)
    return f is not None and getattr(be, "backward_pairs", True) and f(M, width) and all(w is None or w.requires_grad for w in weights)


_INFERENCE = {"on": False}


def mark_mode() -> None:
    """called by the modules right before they apply one of the functions below: are gradients disabled (torch.no_grad()), so that no
    node will run backward?  (Inside Function.forward the grad mode is always off, and ctx.needs_input_grad is True for every parameter
    whatever the mode -- the function cannot see it by itself.)"""
    _INFERENCE["on"] = not torch.is_grad_enabled()


def _infer(ctx) -> int:
    """-> 1: inference (no node will run backward); 2: training on a backend whose products read the pair format (HipBackend.training_pairs);
    0: neither.  (Second condition of 1: whatever the flag says, an input activation that asks for a gradient means a backward pass will come.)"""
    if _INFERENCE["on"] and not ctx.needs_input_grad[0]:
        return 1
    return 2 if getattr(get_backend(), "training_pairs", False) else 0


def _ln_fwd(be, x, w, b, infer=False, need_y=True):
    """-> (y, mean, rstd, record).  need_y=False (inference, pair format): y is NOT written (returned None) -- the product behind reads the
    pairs, and whoever adds these rows as a residual has its epilogue recompute them from x (backend.gemm res_ln)"""
    if w is None:                                   # layer_norm=False: the block works on its input as it is
        return x, None, None, None
    M = x.shape[0]
    mean, rstd = _new((M,), x, F32), _new((M,), x, F32)
    if _pairs(be, x, infer):                        # the pair format (the record's .pairs): the product behind reads operands split once
        y = _new(x.shape, x) if need_y else None
        return y, mean, rstd, be.layernorm_fwd(x, w, b, y, mean, rstd, pairs=True)
    y = _new(x.shape, x)
    sy = be.layernorm_fwd(x, w, b, y, mean, rstd)          # (HIP backend, fp16-split products: the row maxima of y; else None)
    return y, mean, rstd, sy


def _ln_bwd(be, dy, x, mean, rstd, w, b, drop=None):
    """-> (dx, the backend's record of dx's row maxima or None).  drop = (p, seed) of the dropout whose backward reads dx next: where the
    backend can, the same launch writes that too and dx carries it as dx._grappa_masked = (dz, record of dz, p, seed) -- `_masked_grad`
    picks it up, in this Function's backward or in the one of the layer in front"""
    if w is None:                                   # layer_norm=False
        return dy, None
    dx = _new(x.shape, x)
    if x.shape[0] == 0:
        return dx, None
    # a frozen affine pair still needs somewhere to put the reductions the kernel produces
    dw = _pgrad(w) if w.requires_grad else torch.zeros_like(w)
    db = _pgrad(b) if b.requires_grad else torch.zeros_like(b)
    if drop is not None and drop[0] > 0 and getattr(be, "drop_fusable", None) is not None and be.drop_fusable(x):
        sdx, dz, sz = be.layernorm_bwd(dy, x, mean, rstd, w, dx, dw, db, accumulate=True, drop=drop)
        dx._grappa_masked = (dz, sz, float(drop[0]), int(drop[1]), dx._version, dx.data_ptr())
        return dx, sdx
    sdx = be.layernorm_bwd(dy, x, mean, rstd, w, dx, dw, db, accumulate=True)
    return dx, sdx


def _masked_grad(dy, drop_p, seed):
    """the dropout backward of dy if the kernel that produced dy wrote it already (see _ln_bwd) -> (dz, record) or None"""
    m = getattr(dy, "_grappa_masked", None)
    if m is None or m[2] != float(drop_p) or m[3] != int(seed) or m[0].shape != dy.shape:
        return None
    # the tensor must still hold what the kernel wrote: autograd accumulates a second consumer's gradient IN PLACE into the same object (its
    # version counter moves) -- the precomputed dz would then be the backward of a part of the gradient only
    if m[4] != dy._version or m[5] != dy.data_ptr():
        return None
    return m[0], m[1]


def _ff_fwd(be, x, norm_w, norm_b, w1, b1, w2, b2, act2, drop_p, seed, skip, infer=False):
    """FeedForward: xn = LN(x); u = ELU(xn W1^T + b1); y = act2(u W2^T + b2); out = drop(y) (+ xn)."""
    M = x.shape[0]
    Hd, Nout = w1.shape[0], w2.shape[0]
    narrow = Nout <= 32                       # the output maps' last product (2 .. 12 columns): fp32 kernels, fp32 operands and result
    # inference through the pair format: the normalised rows are written as pairs only when the product that adds them back (skip) can
    # recompute them in its epilogue -- or nobody adds them back
    lean = _pairs(be, x, infer) and norm_w is not None and Hd > 32 and M > 32 and (not skip or not narrow)
    xn, mean, rstd, sxn = _ln_fwd(be, x, norm_w, norm_b, infer, need_y=not lean)
    u = _new((M, Hd), x, F32 if narrow else None)
    # u feeds the next product: its row maxima come out of this product's epilogue -- where a product will read them
    sxn, su = be.gemm(xn, w1, u, M=M, N=Hd, K=x.shape[1], bias=b1, act=ELU, a_scales=sxn, out_amax=_wants_amax(be, False))
    out = _new((M, Nout), x, F32 if narrow else None)
    res = xn if skip else None
    pre = None
    ln_res = (mean, rstd, norm_w, norm_b) if (skip and xn is None) else None      # (lean) the residual = LayerNorm(x), recomputed by the epilogue from x
    if act2:
        pre = _new((M, Nout), x)
        kw = dict(res=x, res_ln=ln_res) if ln_res else dict(res=res)
        su = be.gemm(u, w2, pre, M=M, N=Nout, K=Hd, bias=b2, act=ELU, drop_p=drop_p, drop_seed=seed, out2=out, a_scales=su, **kw)
    elif skip and xn is None:                  # (lean) the residual = LayerNorm(x), recomputed by the epilogue from x
        su = be.gemm(u, w2, out, M=M, N=Nout, K=Hd, bias=b2, drop_p=drop_p, drop_seed=seed, res=x, res_ln=(mean, rstd, norm_w, norm_b), a_scales=su)
    else:
        su = be.gemm(u, w2, out, M=M, N=Nout, K=Hd, bias=b2, drop_p=drop_p, drop_seed=seed, res=res, a_scales=su)
    return out, (x, mean, rstd, xn, u, pre, sxn, su)


def _ff_bwd(be, saved, dout, norm_w, norm_b, w1, b1, w2, b2, act2, drop_p, seed, skip, sdout=None, then_drop=None):
    """-> (dx, record of dx's row maxima or None); sdout: the record for dout when the kernel that produced it wrote one; then_drop: (p, seed)
    of the dropout whose backward reads dx next (_ln_bwd)"""
    x, mean, rstd, xn, u, pre, sxn, su = saved          # (xn is None where the forward pass wrote the normalised rows as pairs only: sxn.pairs)
    M = x.shape[0]
    if not dout.is_contiguous():
        dout, sdout = dout.contiguous(), None
    ready = _masked_grad(dout, drop_p, seed) if (drop_p > 0 and not act2) else None
    if ready is not None:
        dz2, sz = ready
    elif act2 or drop_p > 0:
        if dout.dtype == F32 and _bwd_pairs(be, M, dout.shape[1], w2):
            dz2, sz = None, be.act_dropout_bwd(dout, pre if act2 else None, drop_p, seed, None, pairs=True)      # pairs only: both products behind read them
        else:
            dz2 = _new(dout.shape, dout)
            sz = be.act_dropout_bwd(dout, pre if act2 else None, drop_p, seed, dz2)
    else:
        dz2, sz = dout, sdout
    sz = _linear_bwd_params(be, dz2, u, w2, b2, su, sz)
    dz1 = _new(u.shape, x)                                         # (u is fp32 in front of a narrow output product, x never)
    _, sz = be.gemm(dz2, w2, dz1, M=M, N=u.shape[1], K=w2.shape[0], b_kcontig=False, aux=u, a_scales=sz, out_amax=_wants_amax(be, True))   # fused ELU'(u)
    sz = _linear_bwd_params(be, dz1, xn, w1, b1, sxn, sz)
    dxn = _new(x.shape, x)
    be.gemm(dz1, w1, dxn, M=M, N=x.shape[1], K=u.shape[1], b_kcontig=False, res=dout if skip else None, a_scales=sz)
    return _ln_bwd(be, dxn, x, mean, rstd, norm_w, norm_b, drop=then_drop)


def _drop_bwd(be, dy, drop_p, seed, w):
    """dz = dropout mask of the forward pass applied to dy (no activation) -> (dz or None, record): in the pair format only where both
    products behind it read pairs; dy itself without dropout"""
    if drop_p <= 0:
        return dy, None
    ready = _masked_grad(dy, drop_p, seed)
    if ready is not None:
        return ready
    if dy.dtype == F32 and _bwd_pairs(be, dy.shape[0], dy.shape[1], w):
        return None, be.act_dropout_bwd(dy, None, drop_p, seed, None, pairs=True)
    dz = _new(dy.shape, dy)
    return dz, be.act_dropout_bwd(dy, None, drop_p, seed, dz)


# ------------------------------------------------------------------------------------------------
class LinearFn(Function):
    """y = drop(act(x W^T + b))   (pre_dense / post_dense)."""

    @staticmethod
    def forward(ctx, x, w, b, act, drop_p, seed, out_dtype=None):
        be = get_backend()
        x = _rows(x)
        M, K = x.shape
        N = w.shape[0]
        y = _new((M, N), x, out_dtype)
        pre = None
        if act and drop_p > 0:
            pre = _new((M, N), x, out_dtype)
            ctx.sx = be.gemm(x, w, pre, M=M, N=N, K=K, bias=b, act=act, drop_p=drop_p, drop_seed=seed, out2=y)
        else:
            ctx.sx = be.gemm(x, w, y, M=M, N=N, K=K, bias=b, act=act, drop_p=drop_p, drop_seed=seed)
        ctx.save_for_backward(x, w, b, pre if pre is not None else (y if act else None))
        ctx.cfg = (act, drop_p, seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        be = get_backend()
        x, w, b, ysaved = ctx.saved_tensors
        act, drop_p, seed = ctx.cfg
        dy = _c(dy)
        if act or drop_p > 0:
            dz = _new(dy.shape, dy)
            sz = be.act_dropout_bwd(dy, ysaved if act else None, drop_p, seed, dz)
        else:
            dz, sz = dy, None
        sz = _linear_bwd_params(be, dz, x, w, b, ctx.sx, sz)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _new(x.shape, x)
            be.gemm(dz, w, dx, M=x.shape[0], N=x.shape[1], K=w.shape[0], b_kcontig=False, a_scales=sz)
        return dx, None, None, None, None, None, None


class AttBlockFn(Function):
    @staticmethod
    def forward(ctx, h, plan, heads, drop_p, seed1, seed2, ln_w, ln_b, w_fc, w_r, b_r, ln2_w, ln2_b, w1, b1, w2, b2):
        be = get_backend()
        h = _c(h)
        N, Fd = h.shape
        infer = _infer(ctx)
        lean = _pairs(be, h, infer) and ln_w is not None and N > 32 and w_fc.shape[0] > 32 and Fd > 32      # h1 as pairs only (see _ff_fwd)
        h1, mean1, rstd1, sh1 = _ln_fwd(be, h, ln_w, ln_b, infer, need_y=not lean)
        ft = _new((N, w_fc.shape[0]), h)
        sh1 = be.gemm(h1, w_fc, ft, M=N, N=w_fc.shape[0], K=Fd, a_scales=sh1)
        m = _new(ft.shape, h)
        alpha = _new((plan.E, heads), h, F32)
        be.gat_fwd(plan, ft, heads, ft.shape[1] // heads, m, alpha)
        h3 = _new((N, Fd), h)
        if h1 is None:
            sm = be.gemm(m, w_r, h3, M=N, N=Fd, K=m.shape[1], bias=b_r, drop_p=drop_p, drop_seed=seed1, res=h, res_ln=(mean1, rstd1, ln_w, ln_b))
        else:
            sm = be.gemm(m, w_r, h3, M=N, N=Fd, K=m.shape[1], bias=b_r, drop_p=drop_p, drop_seed=seed1, res=h1)
        if w1 is None:                              # self_interaction=False: attention + head reducer + skip only
            out, ff_saved = h3, None
        else:
            out, ff_saved = _ff_fwd(be, h3, ln2_w, ln2_b, w1, b1, w2, b2, True, drop_p, seed2, True, infer)
        ctx.plan, ctx.cfg, ctx.scales = plan, (heads, drop_p, seed1, seed2), (sh1, sm)
        ctx.ff_saved = ff_saved
        ctx.save_for_backward(h, mean1, rstd1, h1, ft, m, alpha, ln_w, ln_b, w_fc, w_r, b_r, ln2_w, ln2_b, w1, b1, w2, b2)
        return out

    @staticmethod
    def backward(ctx, dout):
        be = get_backend()
        h, mean1, rstd1, h1, ft, m, alpha, ln_w, ln_b, w_fc, w_r, b_r, ln2_w, ln2_b, w1, b1, w2, b2 = ctx.saved_tensors
        heads, drop_p, seed1, seed2 = ctx.cfg
        plan = ctx.plan
        N, Fd = h.shape
        if ctx.ff_saved is None:
            dh3, sz = _c(dout), None
        else:
            dh3, sz = _ff_bwd(be, ctx.ff_saved, dout, ln2_w, ln2_b, w1, b1, w2, b2, True, drop_p, seed2, True)
        ctx.ff_saved = None
        if drop_p > 0 and dh3.dtype == F32 and _bwd_pairs(be, N, Fd, w_r):
            dzr, sz = None, be.act_dropout_bwd(dh3, None, drop_p, seed1, None, pairs=True)
        elif drop_p > 0:
            dzr = _new(dh3.shape, dh3)
            sz = be.act_dropout_bwd(dh3, None, drop_p, seed1, dzr)
        else:
            dzr = dh3
        sh1, sm = ctx.scales
        sz = _linear_bwd_params(be, dzr, m, w_r, b_r, sm, sz)
        dm = _new(m.shape, m)
        be.gemm(dzr, w_r, dm, M=N, N=m.shape[1], K=Fd, b_kcontig=False, a_scales=sz)
        dft = _new(ft.shape, ft)
        be.gat_bwd(plan, ft, m, alpha, dm, heads, ft.shape[1] // heads, dft)
        sz = _linear_bwd_params(be, dft, h1, w_fc, None, sh1)
        dh1 = _new(h.shape, h)
        be.gemm(dft, w_fc, dh1, M=N, N=Fd, K=ft.shape[1], b_kcontig=False, res=dh3, a_scales=sz)        # + residual branch
        dh, _ = _ln_bwd(be, dh1, h, mean1, rstd1, ln_w, ln_b)
        _wgrads_aside(be)
        return (dh,) + (None,) * 16


class ConvBlockFn(Function):
    """LN -> SAGE(mean): ELU(W_self h + W_neigh mean_N(h) + b) -> drop -> +skip -> LN -> ELU(W h + b) -> drop -> +skip."""

    @staticmethod
    def forward(ctx, h, plan, drop_p, seed1, seed2, ln_w, ln_b, w_self, w_neigh, bias, ln2_w, ln2_b, w, b):
        be = get_backend()
        h = _c(h)
        N, Fd = h.shape
        h1, mean1, rstd1, sh1 = _ln_fwd(be, h, ln_w, ln_b)
        mn = _new((N, Fd), h)
        be.neighbor_mean(plan, h1, mn, False)
        t = _new((N, Fd), h, F32)                   # the pre-activation addend of the next product: fp32 in every configuration
        smn = be.gemm(mn, w_neigh, t, M=N, N=Fd, K=Fd)
        y1 = _new((N, Fd), h)          # ELU output before dropout
        h3 = _new((N, Fd), h)
        sh1 = be.gemm(h1, w_self, y1, M=N, N=Fd, K=Fd, bias=bias, pre=t, act=ELU, drop_p=drop_p, drop_seed=seed1, res=h1, out2=h3, a_scales=sh1)
        if w is None:                               # self_interaction=False: the block ends behind the SAGE step
            h4 = mean2 = rstd2 = y2 = sh4 = None
            out = h3
        else:
            h4, mean2, rstd2, sh4 = _ln_fwd(be, h3, ln2_w, ln2_b)
            y2 = _new((N, Fd), h)
            out = _new((N, Fd), h)
            sh4 = be.gemm(h4, w, y2, M=N, N=Fd, K=Fd, bias=b, act=ELU, drop_p=drop_p, drop_seed=seed2, res=h4, out2=out, a_scales=sh4)
        ctx.plan, ctx.cfg, ctx.scales = plan, (drop_p, seed1, seed2), (smn, sh1, sh4)
        ctx.save_for_backward(h, mean1, rstd1, h1, mn, y1, h3, mean2, rstd2, h4, y2, ln_w, ln_b, w_self, w_neigh, bias, ln2_w, ln2_b, w, b)
        return out

    @staticmethod
    def backward(ctx, dout):
        be = get_backend()
        (h, mean1, rstd1, h1, mn, y1, h3, mean2, rstd2, h4, y2, ln_w, ln_b, w_self, w_neigh, bias, ln2_w, ln2_b, w, b) = ctx.saved_tensors
        drop_p, seed1, seed2 = ctx.cfg
        plan = ctx.plan
        N, Fd = h.shape
        dout = _c(dout)
        smn, sh1, sh4 = ctx.scales
        if w is None:
            dh3 = dout
        else:
            dz2 = _new(dout.shape, dout)
            sz = be.act_dropout_bwd(dout, y2, drop_p, seed2, dz2)
            sz = _linear_bwd_params(be, dz2, h4, w, b, sh4, sz)
            dh4 = _new(h4.shape, h4)
            be.gemm(dz2, w, dh4, M=N, N=Fd, K=Fd, b_kcontig=False, res=dout, a_scales=sz)
            dh3, _ = _ln_bwd(be, dh4, h3, mean2, rstd2, ln2_w, ln2_b)
        dz1 = _new(dh3.shape, dh3)
        sz = be.act_dropout_bwd(dh3, y1, drop_p, seed1, dz1)
        sz = _linear_bwd_params(be, dz1, h1, w_self, bias, sh1, sz)
        sz = _linear_bwd_params(be, dz1, mn, w_neigh, None, smn, sz)
        dmn = _new(mn.shape, mn)
        be.gemm(dz1, w_neigh, dmn, M=N, N=Fd, K=Fd, b_kcontig=False, a_scales=sz)
        dh1 = _new(h1.shape, h1)
        if h1.dtype == F32:
            be.neighbor_mean(plan, dmn, dh1, True)                     # transpose of the mean aggregation
            be.gemm(dz1, w_self, dh1, M=N, N=Fd, K=Fd, b_kcontig=False, res=dh3, accumulate=True, a_scales=sz)
        else:
            # bf16 storage: a bf16 output cannot be accumulated into; the aggregated gradient enters the product as its fp32 addend
            agg = _new(h1.shape, h1, F32)
            be.neighbor_mean(plan, dmn, agg, True)
            be.gemm(dz1, w_self, dh1, M=N, N=Fd, K=Fd, b_kcontig=False, pre=agg, res=dh3, a_scales=sz)
        dh, _ = _ln_bwd(be, dh1, h, mean1, rstd1, ln_w, ln_b)
        _wgrads_aside(be)
        return (dh,) + (None,) * 13


class SplitHeadsFn(Function):
    """h -> one alias of the atom embedding per writer head.  The four heads read the same h (reference models/interaction_parameters.py:
    125-135) and, with GRAPPA_HEAD_STREAMS > 1, run forward and backward on HIP streams of their own.  Each head's backward pass delivers
    ITS gradient of h to this node, which adds them with the library's own kernel on the caller's stream: autograd orders that stream
    behind every head's stream first, so this node is also the join of the streams, and no torch kernel accumulates gradients on a side
    stream (multi-queue)."""

    @staticmethod
    def forward(ctx, h, n):
        ctx.set_materialize_grads(False)
        return tuple(h.view_as(h) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        be = get_backend()
        if getattr(be, "gnn_tails", False):
            be.set_tail_launches(True)               # the GNN's backward pass has the chip to itself (model.GrappaModel.forward switches back)
        _wgrads_aside(be, all_streams=True)          # every head is done: what they left queued runs beside the GNN's backward pass
        gs = [_c(g) for g in gs if g is not None]
        if not gs:
            return None, None
        acc = gs[0]
        for g in gs[1:]:
            out = acc if acc is not gs[0] else torch.empty_like(acc)
            be.add(acc.reshape(-1), g.reshape(-1), out.reshape(-1))
            acc = out
        return acc, None


def _attention(be, qkv, s, T, nheads, like, infer):
    """-> (att or None, record of att's row maxima): the s <= 4 tokens of every tuple attend to each other.  Inference on the HIP
    backend: the output is written in the pair format only (the record's .pairs) -- nothing but the out-projection reads it."""
    M, Fd = qkv.shape[0], qkv.shape[1] // 3
    if not T:
        return _new((M, Fd), like), None
    if infer and hasattr(be, "pairs_ok") and be.pairs_ok(qkv, Fd, training=infer == 2) and 32 < Fd <= 512 and M > 32:        # (the out-projection must be able to read pairs: M, N > 32)
        return None, be.seqattn_fwd(qkv, s, T, nheads, None, pairs=True)
    att = _new((M, Fd), like)
    return att, be.seqattn_fwd(qkv, s, T, nheads, att)


class ProjGatherFn(Function):
    """a = ELU(h W^T + b) (N, Wp); x[pos*T+t] = [a[idx[t,pos]], pe[pos]] (s*T, Wp + has_pe)."""

    @staticmethod
    def forward(ctx, h, w, b, idx32, inv_ptr, inv_rows, s, pe, out_dtype=None):
        be = get_backend()
        h = _c(h)
        N, R = h.shape
        Wp = w.shape[0]
        Fd = Wp + (1 if pe is not None else 0)
        T = idx32.shape[0]
        a = _zeros((N, Fd), h, out_dtype)
        ctx.sh = be.gemm(h, w, a[:, :Wp], M=N, N=Wp, K=R, bias=b, act=ELU)
        x = _new((s * T, Fd), a)
        if T:
            be.tuple_gather_fwd(a, idx32, s, pe, x)
        ctx.save_for_backward(h, w, b, a, inv_ptr, inv_rows)
        ctx.cfg = (s, T, Wp, pe is not None)
        return x

    @staticmethod
    def backward(ctx, dx):
        be = get_backend()
        h, w, b, a, inv_ptr, inv_rows = ctx.saved_tensors
        s, T, Wp, has_pe = ctx.cfg
        N, R = h.shape
        if T == 0:
            return (torch.zeros_like(h),) + (None,) * 8
        dx = _c(dx)
        da = _new(a.shape, a)
        be.tuple_gather_bwd(inv_ptr, inv_rows, dx, da, has_pe, False)
        dz = _padded_cols(N, Wp, a)
        sz = be.act_dropout_bwd(da[:, :Wp], a[:, :Wp], 0.0, 0, dz)
        sz = _linear_bwd_params(be, dz, h, w, b, ctx.sh, sz)
        dh = _new(h.shape, h)
        be.gemm(dz, w, dh, M=N, N=R, K=Wp, b_kcontig=False, a_scales=sz)
        return (dh,) + (None,) * 8


class TransformerLayerFn(Function):
    """x1 = LN(x); a = MHA(x1); x2 = drop(a Wo^T + bo) + x1; out = FF(x2) (skip on the normed input)."""

    @staticmethod
    def forward(ctx, x, s, T, nheads, drop_p, seed1, seed2, n1_w, n1_b, w_in, b_in, w_o, b_o, nf_w, nf_b, w1, b1, w2, b2):
        be = get_backend()
        ctx.up_drop = getattr(x, "_grappa_drop", None)       # (p, seed) of the dropout that produced x: the layer in front (see below)
        x = _c(x)
        M, Fd = x.shape
        infer = _infer(ctx)
        fused = getattr(be, "writer_layer_ok", None)
        if fused is not None and T and fused(x, s, nheads, n1_w, n1_b, w_in, b_in, w_o, b_o, nf_w, nf_b, w1, b1, w2, b2):
            # the whole layer as ONE launch (C ABI 11 grappa_writer_head_fwd, csrc/writer_layer.hip); training: the tensors the backward pass
            # below reads are by-products of that launch
            out = _new((M, Fd), x)
            sv = None
            if infer != 1:
                # x2 is read by the LayerNorm backward only: with the fused backward kernel it is kept in that kernel's tile order
                tiled = bool(getattr(be, "fused_writer_layer_bwd", False))
                rows2 = be.lib.grappa_writer_head_tiles(s, T) * 64 if tiled else M
                sv = dict(mean1=_new((M,), x, F32), rstd1=_new((M,), x, F32), meanf=_new((M,), x, F32), rstdf=_new((M,), x, F32),
                          x1=_new((M, Fd), x), qkv=_new((M, 3 * Fd), x), att=_new((M, Fd), x), x2=_new((rows2, Fd), x), x3=_new((M, Fd), x),
                          u=_new((M, Fd), x), x2_tiled=tiled)
            be.writer_layer_fwd(x, s, T, nheads, drop_p, seed1, seed2, n1_w, n1_b, w_in, b_in, w_o, b_o, nf_w, nf_b, w1, b1, w2, b2, out, save=sv)
            ctx.cfg, ctx.scales = (s, T, nheads, drop_p, seed1, seed2), (None, None)
            ctx.fused_saved = sv if (sv is not None and sv.get("x2_tiled")) else None        # (tiled x2 <=> the fused backward kernel runs this layer's backward)
            if sv is not None:
                ctx.ff_saved = (sv["x2"], sv["meanf"], sv["rstdf"], sv["x3"], sv["u"], None, None, None)
                ctx.save_for_backward(x, sv["mean1"], sv["rstd1"], sv["x1"], sv["qkv"], sv["att"], n1_w, n1_b, w_in, b_in, w_o, b_o, nf_w, nf_b, w1, b1, w2, b2)
            else:
                ctx.ff_saved = None
            if drop_p > 0:
                out._grappa_drop = (drop_p, seed2)
            return out
        lean = _pairs(be, x, infer) and n1_w is not None and M > 32 and Fd > 32      # x1 as pairs only; the out-projection recomputes it as its residual
        x1, mean1, rstd1, sx1 = _ln_fwd(be, x, n1_w, n1_b, infer, need_y=not lean)
        qkv = _new((M, 3 * Fd), x)
        sx1 = be.gemm(x1, w_in, qkv, M=M, N=3 * Fd, K=Fd, bias=b_in, a_scales=sx1)
        att, satt = _attention(be, qkv, s, T, nheads, x, infer)
        x2 = _new((M, Fd), x)
        if x1 is None:
            satt = be.gemm(att, w_o, x2, M=M, N=Fd, K=Fd, bias=b_o, drop_p=drop_p, drop_seed=seed1, res=x, res_ln=(mean1, rstd1, n1_w, n1_b), a_scales=satt)
        else:
            satt = be.gemm(att, w_o, x2, M=M, N=Fd, K=Fd, bias=b_o, drop_p=drop_p, drop_seed=seed1, res=x1, a_scales=satt)
        out, ff_saved = _ff_fwd(be, x2, nf_w, nf_b, w1, b1, w2, b2, False, drop_p, seed2, True, infer)
        ctx.cfg, ctx.scales = (s, T, nheads, drop_p, seed1, seed2), (sx1, satt)
        ctx.ff_saved = ff_saved
        ctx.save_for_backward(x, mean1, rstd1, x1, qkv, att, n1_w, n1_b, w_in, b_in, w_o, b_o, nf_w, nf_b, w1, b1, w2, b2)
        if drop_p > 0:
            # the layer behind reads this: its LayerNorm backward then writes the backward of THIS layer's last dropout with its result
            out._grappa_drop = (drop_p, seed2)
        return out

    @staticmethod
    def backward(ctx, dout):
        be = get_backend()
        x, mean1, rstd1, x1, qkv, att, n1_w, n1_b, w_in, b_in, w_o, b_o, nf_w, nf_b, w1, b1, w2, b2 = ctx.saved_tensors
        s, T, nheads, drop_p, seed1, seed2 = ctx.cfg
        M, Fd = x.shape
        if M == 0:
            return (torch.zeros_like(x),) + (None,) * 18
        sv = getattr(ctx, "fused_saved", None)
        if sv is not None:
            # the input-gradient chain as ONE launch (grappa_writer_head_bwd); the four weight gradients = the grouped products of the pass over
            # the operands it leaves behind
            ctx.fused_saved = ctx.ff_saved = None
            dx, dz2, dz1, dzo, dqkv = be.writer_layer_bwd(_c(dout), x, s, T, nheads, drop_p, seed1, seed2, sv, n1_w, n1_b, w_in, w_o, nf_w, nf_b, w1, w2)
            _linear_bwd_params(be, dz2, sv["u"], w2, b2, None, None)
            _linear_bwd_params(be, dz1, sv["x3"], w1, b1, None, None)
            _linear_bwd_params(be, dzo, att, w_o, b_o, None, None)
            _linear_bwd_params(be, dqkv, x1, w_in, b_in, None, None)
            return (dx,) + (None,) * 18
        dx2, sz = _ff_bwd(be, ctx.ff_saved, dout, nf_w, nf_b, w1, b1, w2, b2, False, drop_p, seed2, True, then_drop=(drop_p, seed1))
        ctx.ff_saved = None
        dzo, sz = _drop_bwd(be, dx2, drop_p, seed1, w_o)
        sx1, satt = ctx.scales
        sz = _linear_bwd_params(be, dzo, att, w_o, b_o, satt, sz)           # (att is None where the attention wrote pairs only: satt.pairs)
        datt = _new((M, Fd), x)
        be.gemm(dzo, w_o, datt, M=M, N=Fd, K=Fd, b_kcontig=False, a_scales=sz)
        dqkv = _new(qkv.shape, qkv)
        sz = be.seqattn_bwd(qkv, datt, s, T, nheads, dqkv)
        sz = _linear_bwd_params(be, dqkv, x1, w_in, b_in, sx1, sz)
        dx1 = _new(x.shape, x)
        be.gemm(dqkv, w_in, dx1, M=M, N=Fd, K=3 * Fd, b_kcontig=False, res=dx2, a_scales=sz)
        dx, _ = _ln_bwd(be, dx1, x, mean1, rstd1, n1_w, n1_b, drop=ctx.up_drop)
        return (dx,) + (None,) * 18


class ProjFirstLayerFn(Function):
    """ProjGatherFn followed by the first TransformerLayerFn, with that layer's LayerNorm and QKV product done once per
    (atom, position) row instead of once per token: a token x[pos*T + t] = [a[idx[t, pos]], pe[pos]] depends on its tuple only through
    (idx[t, pos], pos), and so do LN(x) and LN(x) W_in^T + b_in -- s*N table rows instead of s*T tokens (propers: 32,932 instead of
    83,328 on the C2 batch).  The normed rows and their q, k, v are gathered to the tokens behind the product; backward sums the
    token gradients into the table rows first (inverse incidence of the table) and runs the dgrad / wgrad products and the
    LayerNorm backward on the table.  Same arithmetic per row as the two functions it replaces (reference
    models/interaction_parameters.py:155-180, perm_equiv_transformer.py:127-151, network_utils.py:112-133).
    tabs = batch.BatchPlan.position_tables(level)."""

    @staticmethod
    def forward(ctx, h, w, b, tabs, s, T, pe, out_dtype, nheads, drop_p, seed1, seed2, n1_w, n1_b, w_in, b_in, w_o, b_o, nf_w, nf_b, w1, b1, w2, b2):
        be = get_backend()
        idx_id, invid_ptr, invid_rows, idx_tab, invtab_ptr, invtab_rows = tabs
        h = _c(h)
        N, R = h.shape
        Wp = w.shape[0]
        Fd = Wp + (1 if pe is not None else 0)
        a = _zeros((N, Fd), h, out_dtype)
        sh = be.gemm(h, w, a[:, :Wp], M=N, N=Wp, K=R, bias=b, act=ELU)
        tab = _new((s * N, Fd), a)
        be.tuple_gather_fwd(a, idx_id, s, pe, tab)                 # tab[pos*N + n] = [a[n], pe[pos]]
        infer = _infer(ctx)
        # (training with pairs: the table's normalised rows are gathered to the tokens as the out-projection's residual, so they stay fp32)
        x1_tab, mean1, rstd1, sx1 = _ln_fwd(be, tab, n1_w, n1_b, infer if infer == 1 else 0)
        qkv_tab = _new((s * N, 3 * Fd), tab)
        sx1 = be.gemm(x1_tab, w_in, qkv_tab, M=s * N, N=3 * Fd, K=Fd, bias=b_in, a_scales=sx1)
        M = s * T
        if hasattr(be, "credit"):                 # profiling: the per-token count of this product (and of its two backward products below)
            be.credit("gemm_saved", 2.0 * (M - s * N) * 3 * Fd * Fd)
        fused = getattr(be, "writer_layer_ok", None)
        if fused is not None and fused(tab, s, nheads, n1_w, n1_b, w_in, b_in, w_o, b_o, nf_w, nf_b, w1, b1, w2, b2) and getattr(be, "fused_first_layer", True):
            # everything behind the table-level LayerNorm and q | k | v product as ONE launch: the fused writer layer in its GATHER mode takes x1 and
            # q | k | v of a token from the table rows (C ABI 11 grappa_writer_layer_desc.gather_idx) -- no token-level copy of either exists
            out = _new((M, Fd), tab)
            sv = None
            if infer != 1:
                tiled = bool(getattr(be, "fused_writer_layer_bwd", False))
                rows2 = be.lib.grappa_writer_head_tiles(s, T) * 64 if tiled else M
                sv = dict(meanf=_new((M,), tab, F32), rstdf=_new((M,), tab, F32), att=_new((M, Fd), tab), x2=_new((rows2, Fd), tab), x3=_new((M, Fd), tab),
                          u=_new((M, Fd), tab), x2_tiled=tiled)
            be.writer_layer_fwd(None, s, T, nheads, drop_p, seed1, seed2, n1_w, n1_b, w_in, b_in, w_o, b_o, nf_w, nf_b, w1, b1, w2, b2, out, save=sv,
                                gather=(idx_tab, x1_tab, qkv_tab))
            ctx.cfg, ctx.scales = (s, T, N, Wp, pe is not None, nheads, drop_p, seed1, seed2), (sh, sx1, None)
            ctx.fused_saved = sv if (sv is not None and sv["x2_tiled"]) else None
            ctx.fused_gather = True
            if sv is not None:
                ctx.ff_saved = (sv["x2"], sv["meanf"], sv["rstdf"], sv["x3"], sv["u"], None, None, None)
                ctx.save_for_backward(h, a, tab, mean1, rstd1, x1_tab, qkv_tab, sv["att"], invid_ptr, invid_rows, invtab_ptr, invtab_rows,
                                      w, b, n1_w, n1_b, w_in, b_in, w_o, b_o, nf_w, nf_b, w1, b1, w2, b2, idx_tab)
            if drop_p > 0:
                out._grappa_drop = (drop_p, seed2)
            return out
        ctx.fused_gather = False
        x1, qkv = _new((M, Fd), tab), _new((M, 3 * Fd), tab)
        be.tuple_gather_fwd(x1_tab, idx_tab, s, None, x1)          # x1[pos*T + t] = x1_tab[pos*N + idx[t, pos]]
        be.tuple_gather_fwd(qkv_tab, idx_tab, s, None, qkv)
        del qkv_tab
        att, satt = _attention(be, qkv, s, T, nheads, tab, infer)
        x2 = _new((M, Fd), tab)
        satt = be.gemm(att, w_o, x2, M=M, N=Fd, K=Fd, bias=b_o, drop_p=drop_p, drop_seed=seed1, res=x1, a_scales=satt)
        del x1
        out, ff_saved = _ff_fwd(be, x2, nf_w, nf_b, w1, b1, w2, b2, False, drop_p, seed2, True, infer)
        ctx.cfg, ctx.scales = (s, T, N, Wp, pe is not None, nheads, drop_p, seed1, seed2), (sh, sx1, satt)
        ctx.ff_saved = ff_saved
        ctx.save_for_backward(h, a, tab, mean1, rstd1, x1_tab, qkv, att, invid_ptr, invid_rows, invtab_ptr, invtab_rows,
                              w, b, n1_w, n1_b, w_in, b_in, w_o, b_o, nf_w, nf_b, w1, b1, w2, b2)
        if drop_p > 0:
            out._grappa_drop = (drop_p, seed2)            # (TransformerLayerFn.forward)
        return out

    @staticmethod
    def backward(ctx, dout):
        be = get_backend()
        idx_tab = None
        if getattr(ctx, "fused_gather", False):
            (h, a, tab, mean1, rstd1, x1_tab, qkv, att, invid_ptr, invid_rows, invtab_ptr, invtab_rows,
             w, b, n1_w, n1_b, w_in, b_in, w_o, b_o, nf_w, nf_b, w1, b1, w2, b2, idx_tab) = ctx.saved_tensors      # (qkv: the TABLE here)
        else:
            (h, a, tab, mean1, rstd1, x1_tab, qkv, att, invid_ptr, invid_rows, invtab_ptr, invtab_rows,
             w, b, n1_w, n1_b, w_in, b_in, w_o, b_o, nf_w, nf_b, w1, b1, w2, b2) = ctx.saved_tensors
        s, T, N, Wp, has_pe, nheads, drop_p, seed1, seed2 = ctx.cfg
        sh, sx1, satt = ctx.scales
        M, Fd = s * T, qkv.shape[1] // 3
        R = h.shape[1]
        sv = getattr(ctx, "fused_saved", None)
        if idx_tab is not None and sv is not None:
            # the input-gradient chain down to the attention as ONE launch (grappa_writer_head_bwd in its gather mode): the token-level gradients of
            # q | k | v and of the skip branch come back, the table-level part below is unchanged
            ctx.fused_saved = ctx.ff_saved = None
            dx2, dz2, dz1, dzo, dqkv = be.writer_layer_bwd(_c(dout), None, s, T, nheads, drop_p, seed1, seed2, sv, n1_w, n1_b, w_in, w_o, nf_w, nf_b, w1, w2,
                                                           gather=(idx_tab, qkv))
            _linear_bwd_params(be, dz2, sv["u"], w2, b2, None, None)
            _linear_bwd_params(be, dz1, sv["x3"], w1, b1, None, None)
            _linear_bwd_params(be, dzo, att, w_o, b_o, None, None)
        else:
            if idx_tab is not None:                 # fused forward, unfused backward (A/B switch): the token-level q | k | v the unfused attention backward reads
                qkv_tok = _new((M, 3 * Fd), tab)
                be.tuple_gather_fwd(qkv, idx_tab, s, None, qkv_tok)
                qkv = qkv_tok
            dx2, sz = _ff_bwd(be, ctx.ff_saved, dout, nf_w, nf_b, w1, b1, w2, b2, False, drop_p, seed2, True, then_drop=(drop_p, seed1))
            ctx.ff_saved = None
            dzo, sz = _drop_bwd(be, dx2, drop_p, seed1, w_o)
            sz = _linear_bwd_params(be, dzo, att, w_o, b_o, satt, sz)
            datt = _new((M, Fd), qkv)
            be.gemm(dzo, w_o, datt, M=M, N=Fd, K=Fd, b_kcontig=False, a_scales=sz)
            dqkv = _new(qkv.shape, qkv)
            be.seqattn_bwd(qkv, datt, s, T, nheads, dqkv)                      # (its rows are summed into the table before any product reads them)
        # token gradients -> table rows (pos*N + n): the q, k, v gradients and the skip branch of x1
        dqkv_tab, dres_tab = _new((s * N, 3 * Fd), qkv), _new((s * N, Fd), qkv)
        be.tuple_gather_bwd(invtab_ptr, invtab_rows, dqkv, dqkv_tab, False, False)
        be.tuple_gather_bwd(invtab_ptr, invtab_rows, dx2, dres_tab, False, False)
        del dqkv
        sz = _linear_bwd_params(be, dqkv_tab, x1_tab, w_in, b_in, sx1, None)
        if hasattr(be, "credit"):
            be.credit("gemm_saved", 2.0 * 2.0 * (M - s * N) * 3 * Fd * Fd)
        dx1_tab = _new(x1_tab.shape, x1_tab)
        be.gemm(dqkv_tab, w_in, dx1_tab, M=s * N, N=Fd, K=3 * Fd, b_kcontig=False, res=dres_tab, a_scales=sz)
        dtab, _ = _ln_bwd(be, dx1_tab, tab, mean1, rstd1, n1_w, n1_b)
        da = _new(a.shape, a)
        be.tuple_gather_bwd(invid_ptr, invid_rows, dtab, da, has_pe, False)      # sum over the positions of an atom
        dz = _padded_cols(N, Wp, a)
        sz = be.act_dropout_bwd(da[:, :Wp], a[:, :Wp], 0.0, 0, dz)
        sz = _linear_bwd_params(be, dz, h, w, b, sh, sz)
        dh = _new(h.shape, h)
        be.gemm(dz, w, dh, M=N, N=R, K=Wp, b_kcontig=False, a_scales=sz)
        return (dh,) + (None,) * 23


class SymmetriserFn(Function):
    """o[p*T+t] = mlp(concat_j x[perm_p[j]*T + t]) for every permutation p (the sum over p is taken by ParamOutFn).
    layers: list of (norm_w, norm_b, w1, b1, w2, b2, skip)."""

    @staticmethod
    def forward(ctx, x, s, T, perms, n_layers, *params):
        be = get_backend()
        x = _c(x)
        Fd = x.shape[1]
        P = len(perms)
        layers = [params[6 * i:6 * i + 6] for i in range(n_layers)]
        z = _new((P * T, s * Fd), x)
        if T:
            be.perm_concat_fwd(x, s, T, perms, z)
        saved = []
        cur = z
        for i, (nw, nb, w1, b1, w2, b2) in enumerate(layers):
            skip = (i != 0) and (i != n_layers - 1)
            cur, sv = _ff_fwd(be, cur, nw, nb, w1, b1, w2, b2, False, 0.0, 0, skip, _infer(ctx))
            saved.append(sv)
        ctx.cfg = (s, T, perms, n_layers, tuple(x.shape), x.dtype)
        ctx.saved_layers = saved
        ctx.layers = layers
        return cur

    @staticmethod
    def backward(ctx, dout):
        be = get_backend()
        s, T, perms, n_layers, xshape, xdtype = ctx.cfg
        nret = 5 + 6 * n_layers
        if T == 0:
            return (torch.zeros(xshape, dtype=xdtype, device=dout.device),) + (None,) * (nret - 1)
        g, sg = _c(dout), None
        for i in reversed(range(n_layers)):
            nw, nb, w1, b1, w2, b2 = ctx.layers[i]
            skip = (i != 0) and (i != n_layers - 1)
            g, sg = _ff_bwd(be, ctx.saved_layers[i], g, nw, nb, w1, b1, w2, b2, False, 0.0, 0, skip, sg)
        ctx.saved_layers = None
        dx = torch.empty(xshape, dtype=xdtype, device=dout.device)
        be.perm_concat_bwd(g, s, T, perms, dx)
        return (dx,) + (None,) * (nret - 1)


class ParamOutFn(Function):
    """(k, eq) = output map of sum_p o[p*T+t]; kind 0 bond, 1 angle, 2 torsion (eq is None)."""

    @staticmethod
    def forward(ctx, o, kind, T, P, n_per, gated, cutoff, consts):
        be = get_backend()
        o = _c(o)
        if kind == 2:
            k = _new((T, n_per), o)
            eq = None
        else:
            k, eq = _new((T,), o), _new((T,), o)
        if T:
            be.param_out_fwd(kind, o, T, P, n_per, gated, cutoff, consts, k, eq)
        ctx.save_for_backward(o, consts)
        ctx.cfg = (kind, T, P, n_per, gated, cutoff)
        if kind == 2:
            return k
        return k, eq

    @staticmethod
    def backward(ctx, dk, deq=None):
        be = get_backend()
        o, consts = ctx.saved_tensors
        kind, T, P, n_per, gated, cutoff = ctx.cfg
        d_o = _new(o.shape, o)
        dk = _c(dk) if dk is not None else None
        deq = _c(deq) if deq is not None else None
        if T:
            be.param_out_bwd(kind, o, T, P, n_per, gated, cutoff, consts, dk, deq, d_o)
        d_consts = None
        if ctx.needs_input_grad[7]:                 # learnable_statistics=True: the statistics are parameters
            d_consts = torch.zeros_like(consts)
            if T:
                be.param_out_bwd_stats(kind, o, T, P, n_per, gated, cutoff, consts, dk, deq, d_consts)
        return (d_o,) + (None,) * 6 + (d_consts,)


class MMEnergyFn(Function):
    """(E (B,C), E_terms (4,B,C), G (N,C,3)) from xyz and the force-field parameters; backward -> dL/dk, dL/deq."""

    @staticmethod
    def forward(ctx, xyz, plan, n_per, offset_torsion, want_gradient, tuple_out, k2, eq2, k3, eq3, k4, k4i):
        be = get_backend()
        xyz = _c(xyz)
        ks = [_c(k2), _c(k3), _c(k4), _c(k4i)]
        eqs = [_c(eq2), _c(eq3), None, None]
        B, Cc = plan.B, xyz.shape[1]
        energy = _new((B, Cc), xyz)
        terms = _new((4, B, Cc), xyz)
        te = tx = None
        if tuple_out is not None:
            te, tx = tuple_out
        be.mm_energy_fwd(plan, xyz, ks, eqs, n_per, offset_torsion, energy, terms, te, tx)
        grad = _new(xyz.shape, xyz)
        if want_gradient:
            be.mm_gradient_fwd(plan, xyz, ks, eqs, n_per, grad)
        else:
            grad.zero_()
        ctx.plan, ctx.cfg = plan, (n_per, offset_torsion, want_gradient)
        ctx.save_for_backward(xyz, *ks, eqs[0], eqs[1])
        ctx.mark_non_differentiable(terms)
        return energy, terms, grad

    @staticmethod
    def backward(ctx, gE, gterms, gG):
        be = get_backend()
        xyz, k2, k3, k4, k4i, eq2, eq3 = ctx.saved_tensors
        n_per, offset_torsion, want_gradient = ctx.cfg
        plan = ctx.plan
        ks, eqs = [k2, k3, k4, k4i], [eq2, eq3, None, None]
        gks = [torch.zeros_like(k) for k in ks]
        geqs = [torch.zeros_like(eq2), torch.zeros_like(eq3), None, None]
        be.mm_bwd(plan, xyz, ks, eqs, n_per, offset_torsion, _c(gE) if gE is not None else None,
                  _c(gG) if (gG is not None and want_gradient) else None, gks, geqs)
        return None, None, None, None, None, None, gks[0], geqs[0], gks[1], geqs[1], gks[2], gks[3]


class MolwiseLossFn(Function):
    """scalar loss = sum_m l_m * inv_B; the gradients are produced by the same kernels in the forward pass."""

    @staticmethod
    def forward(ctx, plan, cfg, energy, gradient, k2, eq2, k3, eq3, k4, k4i):
        be = get_backend()
        dev = plan.device
        B = plan.B
        inv_B = cfg["inv_B"]
        loss_mol = torch.zeros((B,), dtype=torch.float32, device=dev)
        wE, wG = cfg["energy_weight"], cfg["gradient_weight"]
        gE = torch.zeros_like(energy) if (energy is not None and wE != 0) else None
        gG = torch.zeros_like(gradient) if (gradient is not None and wG != 0) else None
        if wE != 0 or wG != 0:
            be.loss_ef(plan, _c(energy) if wE != 0 else None, cfg["energy_ref"] if wE != 0 else None, cfg["is_dummy"],
                       _c(gradient) if wG != 0 else None, cfg["gradient_ref"] if wG != 0 else None, wE, wG, inv_B, loss_mol, gE, gG)
        params = [k2, eq2, k3, eq3, k4, k4i]
        gps = [None] * 6
        if cfg["param_active"]:
            params_c = [(_c(p) if (p is not None and cfg["param_used"][i]) else None) for i, p in enumerate(params)]
            gps = [torch.zeros_like(p) if p is not None else None for p in params_c]
            be.loss_param(plan, params_c, cfg["refs"], cfg["fac"], cfg["reg"], cfg["pw"], inv_B, loss_mol, gps)
        ctx.grads = (gE, gG, gps)
        ctx.loss_mol = loss_mol
        ctx.mark_non_differentiable(loss_mol)
        return loss_mol.sum() * inv_B, loss_mol

    @staticmethod
    def backward(ctx, gl, _gmol):
        gE, gG, gps = ctx.grads
        one = (gl.numel() == 1 and float(gl) == 1.0) if gl.device.type == "cpu" else False

        def sc(t):
            if t is None:
                return None
            return t if one else t * gl

        return (None, None, sc(gE), sc(gG)) + tuple(sc(g) for g in gps)


# ------------------------------------------------------------------------------------------------
# The writer heads layer-locked (round 4).  The four heads (bond / angle / proper / improper) have the same architecture and read the
# same atom embedding; head by head, each of their products leaves most of the chip idle at small batches (batch 32: 44 .. 176
# workgroups of one tile time each on 256 CUs) and costs a launch.  Here one autograd node runs ONE transformer layer (or the
# symmetrisers) of all heads: row-wise kernels head by head, every product of the layer as ONE grouped launch (backend.gemm_group).
# Same arithmetic per head as TransformerLayerFn / SymmetriserFn above (same kernels, same K cuts: the same bits).
def _gemm_group(be, calls):
    f = getattr(be, "gemm_group", None)
    if f is None or len(calls) < 2:
        return [be.gemm(*a, **k) for a, k in calls]
    return f(calls)


def _ln_fwd_multi(be, xs, ws, bs, infer, need_y):
    """_ln_fwd for several heads: one batched launch where the backend has it and no head writes the pair format"""
    f = getattr(be, "layernorm_fwd_batched", None)
    if f is not None and len(xs) >= 2 and all(w is not None for w in ws) and not any(_pairs(be, x, infer) for x in xs):
        r = f(xs, ws, bs)
        if r is not None:
            return r
    return [_ln_fwd(be, x, w, b, infer, need_y=ny) for x, w, b, ny in zip(xs, ws, bs, need_y)]


def _ln_bwd_multi(be, dys, xs, means, rstds, ws, bs):
    f = getattr(be, "layernorm_bwd_batched", None)
    if f is not None and len(xs) >= 2 and all(w is not None and w.requires_grad and b.requires_grad for w, b in zip(ws, bs)):
        r = f([(dy, x, m, r_, w, _pgrad(w), _pgrad(b)) for dy, x, m, r_, w, b in zip(dys, xs, means, rstds, ws, bs)])
        if r is not None:
            return r
    return [_ln_bwd(be, dy, x, m, r_, w, b) for dy, x, m, r_, w, b in zip(dys, xs, means, rstds, ws, bs)]


def _drop_bwd_multi(be, dys, drops, ws, ys=None):
    """dropout (and ELU') backward of several heads -> [(dz or None, record)]; ys: the saved ELU outputs (act2) or None"""
    n = len(dys)
    ys = ys or [None] * n
    f = getattr(be, "act_dropout_bwd_batched", None)
    plain = all((p > 0 or y is not None) for (p, _s), y in zip(drops, ys)) and not any(dy.dtype == F32 and _bwd_pairs(be, dy.shape[0], dy.shape[1], w) for dy, w in zip(dys, ws))
    if f is not None and n >= 2 and plain:
        r = f([(dy, y, p, sd) for dy, y, (p, sd) in zip(dys, ys, drops)])
        if r is not None:
            return r
    out = []
    for dy, (p, sd), w, y in zip(dys, drops, ws, ys):
        if p <= 0 and y is None:
            out.append((dy, None))
        elif dy.dtype == F32 and _bwd_pairs(be, dy.shape[0], dy.shape[1], w):
            out.append((None, be.act_dropout_bwd(dy, y, p, sd, None, pairs=True)))
        else:
            dz = _new(dy.shape, dy)
            out.append((dz, be.act_dropout_bwd(dy, y, p, sd, dz)))
    return out


def _attention_multi(be, qkvs, cfgs, likes, infer):
    """_attention for several heads (cfgs: (s, T, nheads)): one batched launch unless a head writes the pair format"""
    f = getattr(be, "seqattn_batched", None)
    def wants_pairs(qkv):
        Fd = qkv.shape[1] // 3
        return infer and hasattr(be, "pairs_ok") and be.pairs_ok(qkv, Fd, training=infer == 2) and 32 < Fd <= 512 and qkv.shape[0] > 32
    if f is not None and len(qkvs) >= 2 and not any(wants_pairs(q) for q in qkvs):
        r = f([(q, s, T, nh) for q, (s, T, nh) in zip(qkvs, cfgs)], False)
        if r is not None:
            return r
    return [_attention(be, q, s, T, nh, like, infer) for q, (s, T, nh), like in zip(qkvs, cfgs, likes)]


def _attention_bwd_multi(be, qkvs, datts, cfgs):
    f = getattr(be, "seqattn_batched", None)
    if f is not None and len(qkvs) >= 2:
        r = f([(q, d, s, T, nh) for q, d, (s, T, nh) in zip(qkvs, datts, cfgs)], True)
        if r is not None:
            return r
    out = []
    for q, d, (s, T, nh) in zip(qkvs, datts, cfgs):
        dqkv = _new(q.shape, q)
        out.append((dqkv, be.seqattn_bwd(q, d, s, T, nh, dqkv)))
    return out


def _ff_fwd_multi(be, xs, params, act2, drops, skip, infer):
    """_ff_fwd for several heads at once: xs [x], params [(norm_w, norm_b, w1, b1, w2, b2)], drops [(drop_p, seed)], skip [bool] -> ([out], [saved])"""
    n = len(xs)
    pre1, calls1 = [], []
    leans = []
    for x, (norm_w, norm_b, w1, b1, w2, b2), sk in zip(xs, params, skip):
        leans.append(_pairs(be, x, infer) and norm_w is not None and w1.shape[0] > 32 and x.shape[0] > 32 and (not sk or not w2.shape[0] <= 32))
    lns = _ln_fwd_multi(be, xs, [p[0] for p in params], [p[1] for p in params], infer, [not l for l in leans])
    for x, (norm_w, norm_b, w1, b1, w2, b2), sk, (xn, mean, rstd, sxn) in zip(xs, params, skip, lns):
        M, Hd, Nout = x.shape[0], w1.shape[0], w2.shape[0]
        narrow = Nout <= 32
        u = _new((M, Hd), x, F32 if narrow else None)
        pre1.append((xn, mean, rstd, u, narrow))
        calls1.append(((xn, w1, u), dict(M=M, N=Hd, K=x.shape[1], bias=b1, act=ELU, a_scales=sxn, out_amax=_wants_amax(be, False))))
    r1 = _gemm_group(be, calls1)
    outs, saved, calls2 = [], [], []
    for i in range(n):
        x, (norm_w, norm_b, w1, b1, w2, b2), (drop_p, seed) = xs[i], params[i], drops[i]
        xn, mean, rstd, u, narrow = pre1[i]
        sxn, su = r1[i]
        M, Hd, Nout = x.shape[0], w1.shape[0], w2.shape[0]
        out = _new((M, Nout), x, F32 if narrow else None)
        ln_res = (mean, rstd, norm_w, norm_b) if (skip[i] and xn is None) else None
        pre = None
        if act2:
            pre = _new((M, Nout), x)
            kw = dict(res=x, res_ln=ln_res) if ln_res else dict(res=xn if skip[i] else None)
            calls2.append(((u, w2, pre), dict(M=M, N=Nout, K=Hd, bias=b2, act=ELU, drop_p=drop_p, drop_seed=seed, out2=out, a_scales=su, **kw)))
        elif ln_res:
            calls2.append(((u, w2, out), dict(M=M, N=Nout, K=Hd, bias=b2, drop_p=drop_p, drop_seed=seed, res=x, res_ln=ln_res, a_scales=su)))
        else:
            calls2.append(((u, w2, out), dict(M=M, N=Nout, K=Hd, bias=b2, drop_p=drop_p, drop_seed=seed, res=xn if skip[i] else None, a_scales=su)))
        outs.append(out)
        saved.append([x, mean, rstd, xn, u, pre, sxn, None])
    r2 = _gemm_group(be, calls2)
    for sv, su in zip(saved, r2):
        sv[7] = su
    return outs, [tuple(sv) for sv in saved]


def _ff_bwd_multi(be, saved, douts, params, act2, drops, skip, sdouts=None):
    """_ff_bwd for several heads at once -> ([dx], [record of dx or None])"""
    n = len(saved)
    sdouts = sdouts or [None] * n
    dz2s, szs, calls = [], [], []
    douts, sdouts = list(douts), list(sdouts)
    for i in range(n):
        if not douts[i].is_contiguous():
            douts[i], sdouts[i] = douts[i].contiguous(), None
    dzs = _drop_bwd_multi(be, douts, drops, [p[4] for p in params], [sv[5] for sv in saved] if act2 else None)
    for i in range(n):
        x, mean, rstd, xn, u, pre, sxn, su = saved[i]
        norm_w, norm_b, w1, b1, w2, b2 = params[i]
        M = x.shape[0]
        dz2, sz = dzs[i]
        if dz2 is douts[i]:                      # (no dropout, no activation: the incoming gradient itself, with its producer's record)
            sz = sdouts[i]
        sz = _linear_bwd_params(be, dz2, u, w2, b2, su, sz)
        dz1 = _new(u.shape, x)
        dz2s.append(dz1)
        calls.append(((dz2, w2, dz1), dict(M=M, N=u.shape[1], K=w2.shape[0], b_kcontig=False, aux=u, a_scales=sz, out_amax=_wants_amax(be, True))))
    r = _gemm_group(be, calls)
    calls, dxns = [], []
    for i in range(n):
        x, mean, rstd, xn, u, pre, sxn, su = saved[i]
        norm_w, norm_b, w1, b1, w2, b2 = params[i]
        dz1 = dz2s[i]
        sz = _linear_bwd_params(be, dz1, xn, w1, b1, sxn, r[i][1])
        dxn = _new(x.shape, x)
        dxns.append(dxn)
        calls.append(((dz1, w1, dxn), dict(M=x.shape[0], N=x.shape[1], K=u.shape[1], b_kcontig=False, res=douts[i] if skip[i] else None, a_scales=sz)))
    _gemm_group(be, calls)
    out = _ln_bwd_multi(be, dxns, [sv[0] for sv in saved], [sv[1] for sv in saved], [sv[2] for sv in saved], [p[0] for p in params], [p[1] for p in params])
    return [o[0] for o in out], [o[1] for o in out]


_NP_LAYER = 13        # tensors per head of MultiTransformerLayerFn: x, n1_w, n1_b, w_in, b_in, w_o, b_o, nf_w, nf_b, w1, b1, w2, b2


class MultiTransformerLayerFn(Function):
    """TransformerLayerFn for several heads at once.  cfgs: per head (s, T, nheads, drop_p, seed1, seed2); tensors: _NP_LAYER per head."""

    @staticmethod
    def forward(ctx, cfgs, *t):
        be = get_backend()
        n = len(cfgs)
        heads = [t[_NP_LAYER * i:_NP_LAYER * (i + 1)] for i in range(n)]
        infer = 1 if (_INFERENCE["on"] and not any(ctx.needs_input_grad[1 + _NP_LAYER * i] for i in range(n))) else \
            (2 if getattr(be, "training_pairs", False) else 0)
        xs = [_c(h[0]) for h in heads]
        st, calls = [], []
        leans = [_pairs(be, x, infer) and h[1] is not None and x.shape[0] > 32 and x.shape[1] > 32 for x, h in zip(xs, heads)]
        lns = _ln_fwd_multi(be, xs, [h[1] for h in heads], [h[2] for h in heads], infer, [not l for l in leans])
        for x, (_, n1_w, n1_b, w_in, b_in, *_r), (x1, mean1, rstd1, sx1) in zip(xs, heads, lns):
            M, Fd = x.shape
            qkv = _new((M, 3 * Fd), x)
            st.append([x1, mean1, rstd1, qkv])
            calls.append(((x1, w_in, qkv), dict(M=M, N=3 * Fd, K=Fd, bias=b_in, a_scales=sx1)))
        sx1s = _gemm_group(be, calls)
        calls, x2s, atts = [], [], []
        att_r = _attention_multi(be, [q[3] for q in st], [(c[0], c[1], c[2]) for c in cfgs], xs, infer)
        for i, (x, hp) in enumerate(zip(xs, heads)):
            s, T, nheads, drop_p, seed1, seed2 = cfgs[i]
            _, n1_w, n1_b, w_in, b_in, w_o, b_o = hp[:7]
            x1, mean1, rstd1, qkv = st[i]
            M, Fd = x.shape
            att, satt = att_r[i]
            x2 = _new((M, Fd), x)
            kw = dict(res=x, res_ln=(mean1, rstd1, n1_w, n1_b)) if x1 is None else dict(res=x1)
            calls.append(((att, w_o, x2), dict(M=M, N=Fd, K=Fd, bias=b_o, drop_p=drop_p, drop_seed=seed1, a_scales=satt, **kw)))
            x2s.append(x2)
            atts.append(att)
        satts = _gemm_group(be, calls)
        outs, ff_saved = _ff_fwd_multi(be, x2s, [hp[7:13] for hp in heads], False, [(c[3], c[5]) for c in cfgs], [True] * n, infer)
        ctx.cfgs, ctx.scales, ctx.ff_saved, ctx.n = cfgs, list(zip(sx1s, satts)), ff_saved, n
        flat = []
        for i in range(n):
            flat += [xs[i], st[i][1], st[i][2], st[i][0], st[i][3], atts[i]] + list(heads[i][1:])
        ctx.save_for_backward(*flat)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        be = get_backend()
        n, cfgs = ctx.n, ctx.cfgs
        sv = ctx.saved_tensors
        per = 6 + (_NP_LAYER - 1)
        H = [sv[per * i:per * (i + 1)] for i in range(n)]
        params = [h[6 + 6:6 + 12] for h in H]                       # nf_w, nf_b, w1, b1, w2, b2
        dx2s, szs = _ff_bwd_multi(be, ctx.ff_saved, douts, params, False, [(c[3], c[5]) for c in cfgs], [True] * n)
        ctx.ff_saved = None
        calls, datts = [], []
        dzos = _drop_bwd_multi(be, dx2s, [(c[3], c[4]) for c in cfgs], [H[i][10] for i in range(n)])
        for i in range(n):
            x, mean1, rstd1, x1, qkv, att, n1_w, n1_b, w_in, b_in, w_o, b_o = H[i][:12]
            s, T, nheads, drop_p, seed1, seed2 = cfgs[i]
            M, Fd = x.shape
            dzo, sz = dzos[i]
            sz = sz if drop_p > 0 else szs[i]
            sx1, satt = ctx.scales[i]
            sz = _linear_bwd_params(be, dzo, att, w_o, b_o, satt, sz)
            datt = _new((M, Fd), x)
            datts.append(datt)
            calls.append(((dzo, w_o, datt), dict(M=M, N=Fd, K=Fd, b_kcontig=False, a_scales=sz)))
        _gemm_group(be, calls)
        calls, dx1s = [], []
        dq = _attention_bwd_multi(be, [H[i][4] for i in range(n)], datts, [(c[0], c[1], c[2]) for c in cfgs])
        for i in range(n):
            x, mean1, rstd1, x1, qkv, att, n1_w, n1_b, w_in, b_in, w_o, b_o = H[i][:12]
            s, T, nheads, drop_p, seed1, seed2 = cfgs[i]
            M, Fd = x.shape
            dqkv, sz = dq[i]
            sz = _linear_bwd_params(be, dqkv, x1, w_in, b_in, ctx.scales[i][0], sz)
            dx1 = _new(x.shape, x)
            dx1s.append(dx1)
            calls.append(((dqkv, w_in, dx1), dict(M=M, N=Fd, K=3 * Fd, b_kcontig=False, res=dx2s[i], a_scales=sz)))
        _gemm_group(be, calls)
        grads = [None]
        dxs = _ln_bwd_multi(be, dx1s, [H[i][0] for i in range(n)], [H[i][1] for i in range(n)], [H[i][2] for i in range(n)],
                            [H[i][6] for i in range(n)], [H[i][7] for i in range(n)])
        for i in range(n):
            grads += [dxs[i][0]] + [None] * (_NP_LAYER - 1)
        return tuple(grads)


class MultiSymmetriserFn(Function):
    """SymmetriserFn for several heads at once.  cfgs: per head (s, T, perms, n_layers); tensors: per head x followed by 6 per layer."""

    @staticmethod
    def forward(ctx, cfgs, *t):
        be = get_backend()
        n = len(cfgs)
        heads, pos, xpos = [], 0, []
        for (s, T, perms, nl) in cfgs:
            xpos.append(pos)
            heads.append((t[pos], [t[pos + 1 + 6 * i:pos + 7 + 6 * i] for i in range(nl)]))
            pos += 1 + 6 * nl
        infer = 1 if (_INFERENCE["on"] and not any(ctx.needs_input_grad[1 + p] for p in xpos)) else (2 if getattr(be, "training_pairs", False) else 0)
        cur, saved = [], [[] for _ in range(n)]
        for (s, T, perms, nl), (x, layers) in zip(cfgs, heads):
            x = _c(x)
            z = _new((len(perms) * T, s * x.shape[1]), x)
            be.perm_concat_fwd(x, s, T, perms, z)
            cur.append(z)
        for li in range(max(c[3] for c in cfgs)):
            act = [i for i in range(n) if li < cfgs[i][3]]
            outs, sv = _ff_fwd_multi(be, [cur[i] for i in act], [heads[i][1][li] for i in act], False, [(0.0, 0)] * len(act),
                                     [(li != 0) and (li != cfgs[i][3] - 1) for i in act], infer)
            for j, i in enumerate(act):
                cur[i] = outs[j]
                saved[i].append(sv[j])
        ctx.cfgs, ctx.saved_layers, ctx.layers = cfgs, saved, [h[1] for h in heads]
        ctx.xmeta = [(tuple(h[0].shape), h[0].dtype) for h in heads]
        return tuple(cur)

    @staticmethod
    def backward(ctx, *douts):
        be = get_backend()
        cfgs, n = ctx.cfgs, len(ctx.cfgs)
        g, sg = [_c(d) for d in douts], [None] * n
        for li in reversed(range(max(c[3] for c in cfgs))):
            act = [i for i in range(n) if li < cfgs[i][3]]
            dxs, sdx = _ff_bwd_multi(be, [ctx.saved_layers[i][li] for i in act], [g[i] for i in act], [ctx.layers[i][li] for i in act], False,
                                     [(0.0, 0)] * len(act), [(li != 0) and (li != cfgs[i][3] - 1) for i in act], [sg[i] for i in act])
            for j, i in enumerate(act):
                g[i], sg[i] = dxs[j], sdx[j]
        ctx.saved_layers = None
        grads = [None]
        for i, (s, T, perms, nl) in enumerate(cfgs):
            shape, dtype = ctx.xmeta[i]
            dx = torch.empty(shape, dtype=dtype, device=g[i].device)
            be.perm_concat_bwd(g[i], s, T, perms, dx)
            grads += [dx] + [None] * (6 * nl)
        return tuple(grads)
