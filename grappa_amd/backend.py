"""Tensor-level marshalling onto the C ABI (include/grappa_hip.h).

`HipBackend` is the product's only compute backend: every method checks its tensor arguments on
the host (device, dtype, contiguity, shapes the kernels assume) and enqueues HIP kernels of
libgrappa_hip.so on torch's current stream.  There is deliberately NO CPU implementation here;
`get_backend()` raises when the library or a GPU is missing.  Tests on CPU-only machines install
a test-only backend through `set_backend()` (tests/conftest.py) to exercise the host logic.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import List, Optional, Sequence

import torch

from . import _lib

_BACKEND = None


def set_backend(b) -> None:
    global _BACKEND
    _BACKEND = b


def get_backend():
    global _BACKEND
    if _BACKEND is None:
        _BACKEND = HipBackend()
    return _BACKEND


class GrappaHipError(RuntimeError):
    pass


_ERR = {-1: "GRAPPA_ERR_ARG (unsupported shape / null pointer)", -2: "GRAPPA_ERR_LAUNCH", -3: "GRAPPA_ERR_WORKSPACE"}


def _chk(rc: int, what: str) -> None:
    if rc != 0:
        raise GrappaHipError(f"{what} failed: {_ERR.get(rc, rc)}")


def _loss_mols(plan) -> int:
    """molecules the loss runs over: all of the batch, or its leading `n_real_mols` when the batch ends in a padding molecule
    (DeviceDataset.collate(pad_to=...): the kernels are one workgroup per molecule, the gradient arrays are zero-initialised, so rows of the
    padding molecule get exactly 0 from the loss)"""
    n = getattr(plan, "n_real_mols", None)
    return plan.B if n is None else int(n)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


_ACT_DTYPES = (torch.float32, torch.bfloat16)


def _sfx(t: torch.Tensor) -> str:
    """suffix of the C entry point for an activation tensor's element type"""
    return "bf16" if t.dtype == torch.bfloat16 else "f32"


def _same_dtype(*ts) -> torch.dtype:
    dts = {t.dtype for t in ts if t is not None}
    if len(dts) != 1 or next(iter(dts)) not in _ACT_DTYPES:
        raise ValueError(f"activation tensors of one kernel must share one element type (float32 or bfloat16), got {sorted(map(str, dts))}")
    return next(iter(dts))


def _f32_2d(t: torch.Tensor, name: str, dev, dtype=torch.float32) -> int:
    """validate a (rows, cols) view of element type `dtype` (None: float32 or bfloat16) with unit inner stride; return its leading dimension."""
    ok = t.dtype in _ACT_DTYPES if dtype is None else t.dtype == dtype
    if not ok or t.device != dev or t.dim() != 2:
        raise ValueError(f"{name}: expected a 2-d {dtype or 'float32/bfloat16'} tensor on {dev}, got {t.dtype} {tuple(t.shape)} on {t.device}")
    if t.shape[1] > 1 and t.stride(1) != 1:
        raise ValueError(f"{name}: inner stride must be 1")
    if t.shape[0] > 1:
        if t.stride(0) < t.shape[1]:
            raise ValueError(f"{name}: overlapping rows (stride {t.stride(0)} < {t.shape[1]} columns)")
        return t.stride(0)
    return max(t.shape[1], 1)


def _flat(t: torch.Tensor, name: str, dev, dtype=torch.float32) -> None:
    if t.dtype != dtype or t.device != dev or not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous {dtype} tensor on {dev}")


DEFAULT_GEMM_PRECISION = "f32_f16x3"


_AMAX_LOG = None      # tools/amax_passes.py sets a list here


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
if _raw_stream is None or not hasattr(torch._C, "_cuda_getDevice"):          # (another torch build: the slow, public way)
    def _raw_stream(_device_index):      # noqa: F811
        return torch.cuda.current_stream().cuda_stream


class Amax:
    """largest magnitudes of a 2-d fp32 tensor as int32 tensors of fp32 bit patterns: per row (activations: written by the kernel
    that produced the tensor, or by one pass of grappa_amax_f32), per column (weights), and of the whole tensor (`tmax`, one value:
    max over the rows, for the weight-gradient products whose reduction runs over the rows)"""
    __slots__ = ("row", "col", "tmax", "pairs", "parts", "nseg")

    def __init__(self, row=None, col=None, tmax=None, pairs=None, parts=None, nseg=0):
        self.row, self.col, self.tmax = row, col, tmax
        # the row maxima as the per-segment partials a product's epilogue wrote (C ABI 8 out_amax_parts: nseg arrays of `rows` values): the
        # products that read the tensor take the maximum over the segments themselves, nothing launches a combine
        self.parts, self.nseg = parts, nseg
        # the tensor itself in the PAIR format (include/grappa_hip.h ABI 5; (rows, 2 * round_up(cols, 32)) float16), written by its
        # producer beside -- or instead of -- the fp32 tensor: a product that gets this record as a_scales reads the pairs
        self.pairs = pairs


class _PendingGemm:
    """a product whose descriptor is built but not launched (HipBackend.gemm_group)"""
    __slots__ = ("d", "shape", "flops", "nbytes", "ret", "dev", "keep", "groupable", "key")

    def __init__(self, d, shape, flops, nbytes, ret, dev, keep, groupable):
        self.d, self.shape, self.flops, self.nbytes, self.ret, self.dev, self.keep, self.groupable = d, shape, flops, nbytes, ret, dev, keep, groupable
        self.key = (int(d.b_kcontig), int(d.a_planes), int(d.b_planes), int(d.precision))


class HipBackend:
    name = "hip"

    def __init__(self):
        if not torch.cuda.is_available():
            raise RuntimeError("grappa_amd needs an AMD GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self.lib = _lib.load()
        self._ws = {}
        self._ws_need = {}             # (M, N, K) -> workspace bytes of a product of that shape (any tail / reduction setting)
        # arithmetic of the dense products (include/grappa_hip.h GRAPPA_GEMM_*): "f32_f16x3" = fp32 operands, every row scaled by a
        # power of two, split into two fp16 pieces (3 partial products on the fp16 matrix cores, fp32 accumulation); "f32_bf16x6" =
        # three bf16 pieces, 6 products (the default until round 2).  Both are at least as close to the exact product as the native
        # fp32 MFMA (tests/test_gpu_ops.py::test_gemm_precision_modes) at ~2.3x / ~1.7x its speed; "f32" = native fp32 MFMA
        self.set_gemm_precision(os.environ.get("GRAPPA_GEMM_PRECISION", DEFAULT_GEMM_PRECISION))
        # optional, OFF by default: a separate arithmetic for the products of the BACKWARD pass (dgrad / wgrad layouts), e.g.
        # "bf16x3" (two bf16 pieces per operand, 2^-16 per product).  The forward pass -- parameters, energies, forces, loss -- is
        # untouched by it; the reference itself trains with torch.set_float32_matmul_precision('medium') (training/trainrun.py:3).
        self.set_gemm_precision_bwd(os.environ.get("GRAPPA_GEMM_PRECISION_BWD") or None)
        self._prof = None      # list of (kernel family, algorithmic flops, algorithmic bytes, start event, end event) when profiling
        # opt-in (GRAPPA_WEIGHT_PLANES=1): forward and dgrad products read the weight matrix from its bf16 planes (split once per
        # optimiser step, both orientations) through LDS-DMA instead of re-splitting it in every workgroup of every launch
        # (csrc/gemm_planes.hip gemm_wplanes_kernel).  Results are bit-identical to the default; measured speed is the same within
        # +-3 %, which is why it is not the default.
        # weight-gradient products of a backward pass are queued and launched together (grappa_gemm_f32_grouped): alone, each must
        # cut its K (= tokens) 10 - 32 ways to fill the chip and pays for that many partial tiles per output tile
        self.defer_wgrads = os.environ.get("GRAPPA_DEFER_WGRADS", "1") not in ("0", "")
        self._wq = {}                  # (autograd graph task id, stream handle) -> (stream, [(dz, x, dW, db, maxima) kept alive until the launch])
        self.defer_ln = os.environ.get("GRAPPA_DEFER_LN_REDUCTIONS", "1") not in ("0", "")      # tuning: 0 = reduce every LayerNorm's parameter gradients at once
        self._lnq = []                 # deferred LayerNorm parameter gradients: (partials, rows, W, dgamma ptr, dbeta ptr, dgamma, dbeta, stream, task id)
        self._tasks = set()            # autograd graph task ids (backward passes) that have an end-of-pass callback registered and not yet run
        self.wgrad_queue_bytes = int(float(os.environ.get("GRAPPA_WGRAD_QUEUE_GB", "12")) * 2 ** 30)
        # inference (no gradient asked for): LayerNorm and the tuple attention write the A operand of the product behind them in the pair
        # format and the product reads operands split once (csrc/gemm_pairs.hip) -- same bits as the fp32-operand product of the same K cuts
        self.inference_pairs = os.environ.get("GRAPPA_INFERENCE_PAIRS", "1") not in ("0", "")
        # training (round 4): the same producers write pairs ONLY, the dropout backward too, and the weight-gradient products read pairs
        # (C ABI 8: token rows moved onto the tensor's scale inside the kernel) -- no fp32 copy of a normalised activation exists any more
        self.training_pairs = os.environ.get("GRAPPA_TRAINING_PAIRS", "1") not in ("0", "")
        # the same product of the four writer heads as ONE launch (gemm_group); GRAPPA_GROUP_LAUNCHES=0: one by one
        self.group_launches = os.environ.get("GRAPPA_GROUP_LAUNCHES", "1") not in ("0", "")
        # opt-in (GRAPPA_AMAX_PARTS=1): the row maxima of a product's output stay the per-segment partials its epilogue writes and the
        # products that read the tensor combine them (C ABI 8 out_amax_parts / a_amax_nseg: 62 combine launches per C2 step gone).  Measured
        # SLOWER: 37.4 against 35.9 ms per C2 step -- every workgroup of the consumer re-reads 16 partials per row in its prologue and
        # epilogue, which costs more than the one small launch it replaces (profiles/r4_amax_parts_ab.txt) -- hence not the default
        self.amax_parts = os.environ.get("GRAPPA_AMAX_PARTS", "0") not in ("0", "")
        # the dropout backward of a layer's two dropouts written by the LayerNorm backward that produces their input (C ABI 9): 30 of the
        # 40 act_dropout_bwd launches of a C2 step gone with the read of the gradient they made.  GRAPPA_FUSE_LN_DROP=0: launches of their own
        self.fuse_ln_drop = os.environ.get("GRAPPA_FUSE_LN_DROP", "1") not in ("0", "")
        self.pairs_min_rows = int(os.environ.get("GRAPPA_PAIRS_MIN_ROWS", "12288"))
        # the dropout backward writing ITS rows (gradients) as pairs too: the input-gradient products behind gain (pair kernel), the weight-
        # gradient products lose a little (a pair-format A operand costs 3 - 5 %, a pair-format B operand gains 9 %: tools/wgrad_pairs_bench.py)
        # Measured on the C2 step: 34.9 ms with the forward producers' pairs only, 35.0 - 35.1 with the backward producers' too, 35.35 without
        # pairs (profiles/r4_backward_pairs_ab.txt) -- hence off by default
        self.backward_pairs = os.environ.get("GRAPPA_BACKWARD_PAIRS", "0") not in ("0", "")
        # round 5: forward / input-gradient products whose A operand reaches them as fp32 rows (the second product of a feed-forward, every
        # input-gradient product) read the WEIGHT from its pairs and split A's fragments in registers (csrc/gemm_wpairs_il.hip: the pinned
        # pipeline, two workgroups per CU) where that beats the fp32-operand kernel: tables of at least `wpairs_min_rows` rows (on the C2 shapes
        # 1.08 - 1.12 x at >= 28 k rows, slower than the 512-thread kernel on one round of tiles: profiles/r5_pairs_lab_v4.txt)
        self.weight_pairs_min_rows = int(os.environ.get("GRAPPA_WPAIRS_MIN_ROWS", "24000"))
        self._tails = None             # tail launches of the products: True / False, None = the library's default (grappa_gemm_desc.plan_tail)
        self._salt = None
        self._salt_ptr = None          # address of the dropout salt word while enabled (enable_dropout_salt)
        self.plan_override = None      # tuning / tests: (cfg or -1, nsplit or 0, tail: -1 model, 0 never, 1 forced) applied to every product
        self.splitk_reduce = 0         # tests: 0 library default, 1 a reduction launch, 2 inside the product's launch
        self._tails_pinned = False
        if os.environ.get("GRAPPA_PLAN_TAILS", "") != "":
            self.pin_tail_launches(os.environ["GRAPPA_PLAN_TAILS"] != "0")
        self.gnn_tails = os.environ.get("GRAPPA_GNN_TAILS", "0") not in ("0", "")      # tuning: tail launches for the GNN's products while the heads run without
        self.wgrads_aside = os.environ.get("GRAPPA_WGRADS_ASIDE", "1") not in ("0", "")      # tuning: 0 = every queued product waits for the end of the pass
        self._side_streams = {}        # (device, caller's stream handle) -> the side stream of launch_wgrads_aside
        self._aside = []               # (side stream, items kept alive) since the last flush
        self.weight_planes = os.environ.get("GRAPPA_WEIGHT_PLANES", "0") not in ("0", "")
        # round 6: a transformer layer of a writer head as ONE kernel (csrc/writer_layer.hip, grappa_writer_head_fwd) where its shape allows:
        # bf16 storage configuration, 512 features, 8 heads.  GRAPPA_FUSED_WRITER_LAYER=0: the unfused sequence (A/B, tests)
        self.fused_writer_layer = os.environ.get("GRAPPA_FUSED_WRITER_LAYER", "1") not in ("0", "")
        # ... and its backward pass: the input-gradient chain as one kernel (grappa_writer_head_bwd); 0: the unfused backward over the tensors
        # the fused forward saved
        self.fused_writer_layer_bwd = os.environ.get("GRAPPA_FUSED_WRITER_LAYER_BWD", "1") not in ("0", "")
        # ... and the first layer of the angle / proper heads (ops.ProjFirstLayerFn: LayerNorm + q | k | v on (atom, position) rows) through the same
        # kernels in their gather mode; 0: the unfused sequence behind the table-level products
        self.fused_first_layer = os.environ.get("GRAPPA_FUSED_FIRST_LAYER", "1") not in ("0", "")
        self._wplanes = {}     # (data_ptr, rows, cols, transposed) -> (version key, planes tensor)
        self._wpairs = {}      # (data_ptr, rows, cols, "pairs" | "pairsT") -> [version key, pairs, weakref of the weight, epoch of last use, transposed, maxima record]
        self._wptable = None   # (device table of grappa_split_pairs_item, count, tiles, records kept alive)
        self._wepoch = 0
        # precision "f32_f16x3": largest |element| per row / column of every operand (grappa_amax_f32).  Weights: cached until the
        # optimiser step; activations: an `Amax` record travels with the tensor through ops.py (gemm returns it, the backward
        # products receive it), anything missing is computed by one pass over the tensor
        self._wamax = {}       # (data_ptr, rows, cols, ld) -> [version key, weight kept alive, Amax, epoch of last use, batchable]
        self._wtable = None    # (device table of grappa_amax_item, count) of the batchable entries
        # weight-gradient products reduce over the tokens, so each operand gets ONE scale (its largest magnitude): columns more than
        # 2^16 below it lose relative precision gradually.  GRAPPA_WGRAD_COLUMN_MAXIMA=1 gives every column its own scale instead,
        # at the price of one extra pass over both operands of every weight-gradient product (rigorous, ~15 % slower steps)
        self.wgrad_column_maxima = os.environ.get("GRAPPA_WGRAD_COLUMN_MAXIMA", "0") not in ("0", "")

    def set_gemm_precision(self, name: str) -> None:
        if name not in _lib.GEMM_PRECISIONS:
            raise ValueError(f"gemm precision {name!r}: expected one of {sorted(_lib.GEMM_PRECISIONS)}")
        self.gemm_precision_name = name
        self.gemm_precision = _lib.GEMM_PRECISIONS[name]

    def set_gemm_precision_bwd(self, name) -> None:
        if name is not None and name not in _lib.GEMM_PRECISIONS:
            raise ValueError(f"gemm precision {name!r}: expected one of {sorted(_lib.GEMM_PRECISIONS)}")
        self.gemm_precision_bwd_name = name
        self.gemm_precision_bwd = None if name is None else _lib.GEMM_PRECISIONS[name]

    # ------------------------------------------------------------------ weight planes
    def invalidate_weight_planes(self) -> None:
        """the parameters were changed behind torch's back (fused Adam writes the flat buffer through the C ABI)"""
        self._wepoch += 1

    def invalidate_weights(self) -> None:
        """PUBLIC: call after writing parameters through anything torch's version counters do not see -- the flat buffer
        (`FlatParams.data[...]`, a broadcast or an EMA into it), `p.data.*`, a raw-pointer writer.  Everything cached per weight (row /
        column maxima = the scales of the fp16-split products, bf16 planes, fp16 pairs) is rebuilt at its next use.  In-place torch ops on
        the parameter itself (`p.copy_`, `load_state_dict`) and `FusedAdam.step` need no call."""
        self._wepoch += 1

    def _planes_of_weight(self, w: torch.Tensor, transposed: bool) -> torch.Tensor:
        """bf16 planes (3, rows_pad, cols_pad) of W (rows x cols) or of W^T; zero padded to multiples of 32, cached per weight and
        refreshed when the weight changed (torch's version counter for in-place torch ops, the epoch for the fused Adam)"""
        R, Cc = w.shape
        key = (w.data_ptr(), R, Cc, transposed)
        ver = (w._version, self._wepoch)
        hit = self._wplanes.get(key)
        # an entry belongs to ONE tensor object (weak reference): a weight freed and another allocated at the same address with the same
        # shape and version count must not be served the old one's planes (two models loaded one after the other, no optimiser step between)
        if hit is not None and hit[2]() is not w:
            hit = None
        if hit is not None and hit[0] == ver:
            return hit[1]
        rows, cols = (Cc, R) if transposed else (R, Cc)
        if hit is not None:
            planes = hit[1]
        else:
            planes = torch.zeros((3, (rows + 31) // 32 * 32, (cols + 31) // 32 * 32), dtype=torch.bfloat16, device=w.device)
            for k in [k for k, e in self._wplanes.items() if e[2]() is None]:      # entries of weights that no longer exist
                del self._wplanes[k]
        _chk(self.lib.grappa_split_planes_f32(self._stream(), R, Cc, w.data_ptr(), _f32_2d(w, "W", w.device), planes.data_ptr(), planes.stride(1),
                                              planes.stride(0), int(transposed)), "grappa_split_planes_f32")
        self._wplanes[key] = (ver, planes, weakref.ref(w))
        return planes

    def _pairs_of_weight(self, w: torch.Tensor, transposed: bool = False) -> torch.Tensor:
        """W (out features x in features) in the pair format, rows scaled by their own maxima -- the B operand of the forward products -- or
        (transposed) W^T, rows = in features scaled by W's column maxima: the B operand of the input-gradient products.  Cached per weight
        and orientation; after an optimiser step the first stale entry refreshes EVERY registered one in one launch
        (grappa_split_pairs_f32_batched), like the weights' maxima."""
        R, Cc = w.shape
        key = (w.data_ptr(), R, Cc, "pairsT" if transposed else "pairs")
        ver = (w._version, self._wepoch)
        hit = self._wpairs.get(key)
        if hit is not None and hit[2]() is not w:
            hit = None
        if hit is not None:
            hit[3] = self._wepoch
            if hit[0] == ver:
                return hit[1]
            self._refresh_weight_pairs()
            if hit[0] == ver:
                return hit[1]
        am = self._amax_of_weight(w)
        rows, cols = (Cc, R) if transposed else (R, Cc)
        pairs = torch.zeros((rows, 2 * ((cols + 31) // 32 * 32)), dtype=torch.float16, device=w.device)
        _chk(self.lib.grappa_split_pairs_f32(self._stream(), R, Cc, w.data_ptr(), _f32_2d(w, "W", w.device), (am.col if transposed else am.row).data_ptr(),
                                             pairs.data_ptr(), pairs.stride(0), int(transposed)), "grappa_split_pairs_f32")
        for k in [k for k, e in self._wpairs.items() if e[2]() is None]:      # entries of weights that no longer exist
            del self._wpairs[k]
        self._wpairs[key] = [ver, pairs, weakref.ref(w), self._wepoch, transposed, am]
        self._wptable = None
        return pairs

    def _refresh_weight_pairs(self) -> None:
        import numpy as np
        # (never while a hipGraph is being recorded: a changed table is a host-to-device copy, which a capture cannot hold; what would have
        # aged out is refreshed once more instead)
        ageing = not torch.cuda.is_current_stream_capturing()
        for k in [k for k, e in self._wpairs.items() if e[2]() is None or (ageing and e[3] < self._wepoch - 1)]:      # dead, or unused since the step before last
            del self._wpairs[k]
            self._wptable = None
        live = [e for e in self._wpairs.values()]
        if not live:
            return
        for e in live:                                  # the maxima first (one batched launch of their own when stale)
            e[5] = self._amax_of_weight(e[2]())
        if self._wptable is None:
            dt = np.dtype([("x", "<u8"), ("amax", "<u8"), ("pairs", "<u8"), ("R", "<i4"), ("C", "<i4"), ("ldx", "<i4"), ("ldp", "<i4"),
                           ("transpose", "<i4"), ("tile_begin", "<i4")])
            tab = np.zeros(len(live), dtype=dt)
            tiles = 0
            for i, e in enumerate(live):
                w, am = e[2](), e[5]
                tab[i] = (w.data_ptr(), (am.col if e[4] else am.row).data_ptr(), e[1].data_ptr(), w.shape[0], w.shape[1], w.stride(0), e[1].stride(0),
                          int(e[4]), tiles)
                tiles += ((w.shape[0] + 31) // 32) * ((w.shape[1] + 31) // 32)
            self._wptable = (torch.from_numpy(tab.view(np.uint8).copy()).to(live[0][1].device), len(live), tiles, [e[5] for e in live])
        tab, n, tiles, _keep = self._wptable
        _chk(self.lib.grappa_split_pairs_f32_batched(self._stream(), n, tiles, tab.data_ptr()), "grappa_split_pairs_f32_batched")
        for e in live:
            e[0] = (e[2]()._version, self._wepoch)

    def to_pairs(self, x: torch.Tensor, have: "Optional[Amax]" = None) -> "Amax":
        """an fp32 (rows, cols) tensor in the pair format by a pass of its own (producers that hold whole rows write it themselves:
        layernorm_fwd(pairs=True), seqattn_fwd(pairs=True), act_dropout_bwd(pairs=True)): -> record with .row and .pairs"""
        R, Cc = x.shape
        am = self.amax(x, have, rows=True)
        pr = torch.zeros((R, 2 * ((Cc + 31) // 32 * 32)), dtype=torch.float16, device=x.device)
        _chk(self.lib.grappa_split_pairs_f32(self._stream(), R, Cc, x.data_ptr(), _f32_2d(x, "x", x.device), am.row.data_ptr(), pr.data_ptr(),
                                             pr.stride(0), 0), "grappa_split_pairs_f32")
        return Amax(row=am.row, pairs=pr)

    def pairs_ok(self, x: torch.Tensor, width: int, training: bool = False) -> bool:
        """can a producer of `x` (rows of `width` columns) hand the following forward product its operand in the pair format?"""
        if training:
            return x.dtype == torch.float32 and self.training_pairs_ok(x.shape[0], width)
        return (self.inference_pairs and x.dtype == torch.float32 and x.shape[0] > 32 and width % 32 == 0 and self.gemm_precision_name == "f32_f16x3")

    def training_pairs_ok(self, rows: int, width: int) -> bool:
        """training with the pair format as the storage format of the products' operands: rows of this shape as pairs ONLY?  Tables with
        fewer rows than `pairs_min_rows` stay fp32 (the GNN's 8,233 atom rows at C2: the two-workgroup pair kernel gains nothing there)"""
        return (self.training_pairs and rows >= self.pairs_min_rows and rows > 32 and width % 32 == 0 and 32 < width <= 2048 and
                self.gemm_precision_name == "f32_f16x3" and self.gemm_precision_bwd is None and not self.wgrad_column_maxima and self.defer_wgrads)

    # ------------------------------------------------------------------ row / column maxima (scales of the fp16-split products)
    def _amax_launch(self, t: torch.Tensor, rows: bool, cols: bool, row_out=None, col_out=None):
        R, Cc = t.shape
        if _AMAX_LOG is not None:                  # tools: which tensors still need a pass of their own
            import traceback
            fr = [f for f in traceback.extract_stack(limit=8) if f.filename.endswith("ops.py")]
            _AMAX_LOG.append((R, Cc, rows, cols, fr[-1].lineno if fr else 0))
        dev = t.device
        ld = _f32_2d(t, "amax operand", dev)
        row = (row_out if row_out is not None else torch.empty(R, dtype=torch.int32, device=dev)) if rows else None
        col = (col_out if col_out is not None else torch.empty(Cc, dtype=torch.int32, device=dev)) if cols else None
        need = self.lib.grappa_amax_f32_workspace_bytes(R, Cc) if cols else 0
        ws = self._workspace_amax(need, dev) if need else None
        self._timed("amax", 0.0, 4.0 * R * Cc,
                    lambda: _chk(self.lib.grappa_amax_f32(self._stream(), R, Cc, t.data_ptr(), ld, _ptr(row), _ptr(col), _ptr(ws),
                                                          ws.numel() if ws is not None else 0), "grappa_amax_f32"))
        return row, col

    def _workspace_amax(self, nbytes: int, dev) -> torch.Tensor:
        key = (dev, self._stream(), "amax")      # not the products' workspace: a queued group may hold that
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=dev)
            self._ws[key] = ws
        return ws

    def amax(self, t: torch.Tensor, have: "Optional[Amax]" = None, rows: bool = False, cols: bool = False, tmax: bool = False) -> "Amax":
        """`have` completed by the maxima asked for (one pass over t for missing row / column maxima; the whole-tensor maximum is
        a reduction of the row maxima)"""
        am = have if have is not None else Amax()
        if rows and am.row is None and am.parts is not None:           # a consumer that wants one array: combine the partials (one small launch)
            am.row = torch.empty(t.shape[0], dtype=torch.int32, device=t.device)
            _chk(self.lib.grappa_amax_combine(self._stream(), t.shape[0], am.nseg, am.parts.data_ptr(), am.row.data_ptr()), "grappa_amax_combine")
        need_r, need_c = (rows or tmax) and am.row is None and am.parts is None, cols and am.col is None
        if need_r or need_c:
            r, c = self._amax_launch(t, need_r, need_c)
            am.row = r if need_r else am.row
            am.col = c if need_c else am.col
        if tmax and am.tmax is None:
            self._tmax_of([am], t.device)
        return am

    def _tmax_of(self, records, dev) -> None:
        """whole-tensor maxima of the records that lack one, from their row maxima: one launch per 32 records"""
        todo = [r for r in records if r.tmax is None]
        if not todo:
            return
        n = len(todo)
        out = torch.empty(n, dtype=torch.int32, device=dev)
        src = [r.row if r.row is not None else r.parts for r in todo]       # (the maximum over all partials is the tensor's too)
        ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in src])
        lens = (C.c_int * n)(*[t.numel() for t in src])
        self._timed("amax", 0.0, 4.0 * sum(t.numel() for t in src),
                    lambda: _chk(self.lib.grappa_amax_reduce(self._stream(), n, ptrs, lens, out.data_ptr()), "grappa_amax_reduce"))
        for i, r in enumerate(todo):
            r.tmax = out[i:i + 1]

    def wants_amax(self, backward: bool = False) -> bool:
        """do the dense products of this pass run on fp16 pieces, i.e. should producers write the row maxima of their outputs"""
        prec = self.gemm_precision_bwd if (backward and self.gemm_precision_bwd is not None) else self.gemm_precision
        return prec == _lib.GEMM_PRECISIONS["f32_f16x3"]

    def _new_row_amax(self, t: torch.Tensor, backward: bool, want) -> "Optional[torch.Tensor]":
        if want is False or t.dtype != torch.float32 or t.shape[0] == 0 or (want is None and not self.wants_amax(backward)):
            return None
        return torch.empty(t.shape[0], dtype=torch.int32, device=t.device)

    def _amax_of_weight(self, w: torch.Tensor) -> "Amax":
        """row and column maxima of a weight matrix, refreshed when the weight changed (as _planes_of_weight).  Entries keep their
        weight alive, so its address cannot be handed to another tensor while the entry exists.  After an optimiser step the first
        stale weight refreshes EVERY registered weight in one launch (grappa_amax_f32_batched: one workgroup per weight).
        That refresh rewrites all maxima arrays (zero, then atomic max) on the CALLING stream: it is safe because parameters change between
        steps, so the first stale use is the GNN's on the main stream, before the writer heads fork.  Never hand this function a tensor
        that is rewritten inside a step on a head's stream (an experiment that did -- a per-head zero-extended copy of the projection
        weight -- raced with the other heads' products and trained on NaN: DESIGN.md section 6, "rejected this round")."""
        R, Cc = w.shape
        key = (w.data_ptr(), R, Cc, w.stride(0))
        ver = (w._version, self._wepoch)
        hit = self._wamax.get(key)
        if hit is not None:
            hit[3] = self._wepoch
            if hit[0] == ver:
                return hit[2]
            if hit[4]:                                     # registered for the batched refresh
                self._refresh_weight_amax()
                if hit[0] == ver:
                    return hit[2]
            elif hit[1] is w:
                # a weight the batched kernel cannot take (odd width): a pass of its own INTO the arrays it already has -- the same
                # addresses every step (a recorded hipGraph holds them) and no change to the batched kernel's table
                self._amax_launch(w, True, True, hit[2].row, hit[2].col)
                hit[2].tmax = None
                hit[0] = ver
                return hit[2]
        am = self.amax(w, None, rows=True, cols=True)      # a new weight: a pass of its own
        batchable = Cc % 4 == 0 and Cc <= 2048 and w.stride(0) % 4 == 0 and w.data_ptr() % 16 == 0 and w.stride(1) == 1
        self._wamax[key] = [ver, w, am, self._wepoch, batchable]
        self._wtable = None
        return am

    def _refresh_weight_amax(self) -> None:
        import numpy as np
        # weights not used since the step before last leave the table (dead models of a test session, frozen heads, ...) -- not while a
        # hipGraph is being recorded (see _refresh_weight_pairs)
        if not torch.cuda.is_current_stream_capturing():
            for k in [k for k, e in self._wamax.items() if e[3] < self._wepoch - 1]:
                del self._wamax[k]
                self._wtable = None
        live = [e for e in self._wamax.values() if e[4]]
        if not live:
            return
        if self._wtable is None:
            dt = np.dtype([("x", "<u8"), ("R", "<i4"), ("C", "<i4"), ("ld", "<i4"), ("pad", "<i4"), ("row", "<u8"), ("col", "<u8")])
            tab = np.zeros(len(live), dtype=dt)
            for i, e in enumerate(live):
                w, am = e[1], e[2]
                tab[i] = (w.data_ptr(), w.shape[0], w.shape[1], w.stride(0), 0, am.row.data_ptr(), am.col.data_ptr())
            self._wtable = (torch.from_numpy(tab.view(np.uint8).copy()).to(live[0][1].device), len(live))
        tab, n = self._wtable
        self._timed("amax", 0.0, 4.0 * sum(e[1].numel() for e in live),
                    lambda: _chk(self.lib.grappa_amax_f32_batched(self._stream(), n, tab.data_ptr()), "grappa_amax_f32_batched"))
        for e in live:
            e[0] = (e[1]._version, self._wepoch)

    # ------------------------------------------------------------------ in-process kernel timing (bench.py roofline)
    def start_profile(self) -> None:
        self._prof = []

    def stop_profile(self):
        """-> {family: (launches, total_ms, algorithmic_flops, algorithmic_bytes)}; events are recorded on the stream
        the kernels run on (torch's current stream)."""
        torch.cuda.synchronize()
        out = {}
        self.last_profile_details = details = []      # (family, detail, ms, flops, bytes) per launch: tools/step_gemm_table.py
        for name, fl, by, e0, e1, detail in self._prof or []:
            n, ms, f, b = out.get(name, (0, 0.0, 0.0, 0.0))
            t = e0.elapsed_time(e1) if e0 is not None else 0.0
            out[name] = (n + 1, ms + t, f + fl, b + by)
            if detail is not None:
                details.append((name, detail, t, fl, by))
        self._prof = None
        return out

    def credit(self, name, flops) -> None:
        """profiling only: algorithmic work (SURVEY 8(d) counts per token) that a formulation did not have to launch -- e.g. the first
        writer layer's products on (atom, position) rows instead of tokens -- recorded under its own family, without time"""
        if self._prof is not None:
            self._prof.append((name, float(flops), 0.0, None, None, None))

    def _timed(self, name, flops, nbytes, launch, detail=None) -> None:
        if self._prof is None:
            launch()
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        self._prof.append((name, float(flops), float(nbytes), e0, e1, detail() if callable(detail) else detail))

    @staticmethod
    def _gemm_detail(d) -> dict:
        """profiling only: what a product launch was (shape, layout, operand formats, epilogue) -- read back from its descriptor"""
        lay = getattr(d, "_layout", None) or ("fwd" if d.a_kcontig and d.b_kcontig else ("dgrad" if d.a_kcontig else "wgrad"))
        fmt = ("pairs" if d.a_planes and d.b_planes and d.precision == _lib.GEMM_PRECISIONS["f32_f16x3"] else
               "wpairs" if d.b_planes and not d.a_planes and d.precision == _lib.GEMM_PRECISIONS["f32_f16x3"] else
               "planes" if d.a_planes or d.b_planes else "f32")
        epi = "".join(c for c, on in (("b", d.bias), ("e", d.act), ("x", d.aux or d.auxp), ("d", d.drop_p > 0), ("r", d.res or d.resp), ("2", d.C2 or d.C1p),
                                      ("p", d.pre), ("+", d.accumulate), ("m", d.out_amax or d.out_amax_parts), ("c", d.a_colsum)) if on)
        return {"M": d.M, "N": d.N, "K": d.K, "layout": lay, "fmt": fmt, "epi": epi or "-"}

    # ------------------------------------------------------------------ plumbing
    def _stream(self):
        # the raw handle of the current stream of the current device, straight from the C side: torch.cuda.current_stream() builds a
        # Stream object through three Python layers (3 - 4 us, twice per launch: 8 ms of host time per C2 train step, 1.2 ms per predict)
        return _raw_stream(torch._C._cuda_getDevice())

    def _workspace(self, nbytes: int, dev) -> torch.Tensor:
        key = (dev, self._stream())
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=dev)
            self._ws[key] = ws
        return ws

    # ------------------------------------------------------------------ dense
    def to_f32(self, t: torch.Tensor) -> torch.Tensor:
        """fp32 copy of a 2-d bf16 view (the few places where a bf16 tensor meets an fp32-only kernel)"""
        if t.dtype == torch.float32:
            return t
        out = torch.empty(t.shape, dtype=torch.float32, device=t.device)
        if t.numel():
            _chk(self.lib.grappa_convert_bf16_to_f32(self._stream(), t.shape[0], t.shape[1], t.data_ptr(), _f32_2d(t, "x", t.device, torch.bfloat16),
                                                     out.data_ptr(), out.shape[1]), "grappa_convert_bf16_to_f32")
        return out

    def to_bf16(self, t: torch.Tensor) -> torch.Tensor:
        if t.dtype == torch.bfloat16:
            return t
        out = torch.empty(t.shape, dtype=torch.bfloat16, device=t.device)
        if t.numel():
            _chk(self.lib.grappa_convert_f32_to_bf16(self._stream(), t.shape[0], t.shape[1], t.data_ptr(), _f32_2d(t, "x", t.device), out.data_ptr(),
                                                     out.shape[1]), "grappa_convert_f32_to_bf16")
        return out

    @staticmethod
    def _dma_ok(t: torch.Tensor, ld: int) -> bool:
        """a bf16 operand the LDS-DMA kernels can read: 16-byte aligned base, rows a multiple of 16 bytes apart"""
        return t.data_ptr() % 16 == 0 and ld % 8 == 0

    def gemm(self, a, b, out, *, M, N, K, a_kcontig=True, b_kcontig=True, bias=None, res=None, aux=None, pre=None, act=0,
             drop_p=0.0, drop_seed=0, accumulate=False, out2=None, a_colsum=None, precision=None, a_scales=None, b_scales=None, out_amax=False,
             res_ln=None, _defer=False):
        """C = epilogue(A B^T) (include/grappa_hip.h).  Operands and epilogue tensors may be float32 or -- the bf16 storage
        configuration -- bfloat16: a bf16 A (and, for the wgrad layout, B) is read by the LDS-DMA plane kernels as a one-plane
        operand when the shape allows it, otherwise converted to fp32 first; out / out2 / res / aux are written / read in their own
        element type by the shared epilogue.
        Precision "f32_f16x3": returns the `Amax` record of A (given as a_scales, completed as needed; in the forward layout with a
        completed as needed); otherwise None.  out_amax=True: the kernel also writes the row maxima of the final output and the
        return value becomes (record of A, record of the output) -- for an output that is the A operand of the next product."""
        dev = out.device
        d = _lib.GemmDesc()
        d.M, d.N, d.K = M, N, K
        d._layout = "fwd" if a_kcontig and b_kcontig else ("dgrad" if a_kcontig else "wgrad")      # (profiling only: pair operands turn every layout into k-contiguous rows)
        ar, ac = (M, K) if a_kcontig else (K, M)
        br, bc = (N, K) if b_kcontig else (K, N)
        # A in the pair format (its producer wrote it: a_scales.pairs): forward layout, default arithmetic; `a` itself may then be None
        a_pairs = getattr(a_scales, "pairs", None) if a_scales is not None else None
        if a_pairs is not None and not (a_kcontig and M > 32 and N > 32 and K % 32 == 0 and precision is None and self.gemm_precision_bwd is None
                                        and self.gemm_precision_name == "f32_f16x3" and b.dtype == torch.float32 and a_colsum is None
                                        and tuple(a_pairs.shape) == (M, 2 * K)):
            if a is None:
                raise ValueError("gemm: A was given in the pair format only, which this product cannot read")
            a_pairs = None
        if (a_pairs is None and tuple(a.shape) != (ar, ac)) or tuple(b.shape) != (br, bc) or tuple(out.shape) != (M, N):
            raise ValueError(f"gemm: shapes A{tuple(a.shape)} B{tuple(b.shape)} C{tuple(out.shape)} do not match M={M} N={N} K={K}")
        if M == 0 or N == 0:
            return (None, None) if out_amax else None
        if K == 0:
            raise ValueError("gemm: K == 0")
        if precision is not None:
            d.precision = _lib.GEMM_PRECISIONS[precision]
        elif a_kcontig and b_kcontig or self.gemm_precision_bwd is None:
            d.precision = self.gemm_precision
        else:
            d.precision = self.gemm_precision_bwd      # dgrad (B row-contiguous) and wgrad (both row-contiguous) products: backward pass only
        bf16 = torch.bfloat16
        big = M > 32 and N > 32
        # ---- operands
        planes_a = planes_b = None
        if a_pairs is not None:
            pass                                     # both operands are set below, from the pairs
        elif a_kcontig:
            # forward / dgrad: B is a weight matrix (fp32 parameter)
            if b.dtype != torch.float32:
                raise ValueError("gemm: the weight operand must be float32")
            if a.dtype == bf16:
                if big and K % 32 == 0 and self._dma_ok(a, a.stride(0)) and a_colsum is None:
                    planes_a = a                                                     # one plane = the bf16 tensor itself
                    planes_b = self._planes_of_weight(b, transposed=not b_kcontig)
                    d.precision = _lib.GEMM_PRECISIONS["bf16"]
                else:
                    a = self.to_f32(a)
            if planes_a is None and (self.weight_planes and b.requires_grad and big and K % 32 == 0 and a_colsum is None
                                     and d.precision not in (_lib.GEMM_PRECISIONS["f32"], _lib.GEMM_PRECISIONS["f32_f16x3"])
                                     and a.data_ptr() % 16 == 0 and a.stride(0) % 4 == 0):
                planes_b = self._planes_of_weight(b, transposed=not b_kcontig)      # fp32 activations x pre-split weight planes
        else:
            if b_kcontig:
                raise ValueError("gemm: layout a_kcontig=0, b_kcontig=1 is never needed by the path")
            if a.dtype == bf16 and b.dtype == bf16 and big and self._dma_ok(a, a.stride(0)) and self._dma_ok(b, b.stride(0)):
                planes_a, planes_b = a, b                                            # wgrad: both operands are bf16 activations
                d.precision = _lib.GEMM_PRECISIONS["bf16"]
            else:
                a, b = self.to_f32(a), self.to_f32(b)
        # fp32 A + the weight's pairs ("weight pairs": the library splits A's fragments in registers)
        w_only = (a_pairs is None and planes_a is None and planes_b is None and a_kcontig and big and M >= self.weight_pairs_min_rows
                  and d.precision == _lib.GEMM_PRECISIONS["f32_f16x3"] and precision is None and self.gemm_precision_bwd is None and K % 32 == 0
                  and a.dtype == torch.float32 and b.dtype == torch.float32 and b.requires_grad and a_colsum is None
                  and a.data_ptr() % 16 == 0 and a.stride(0) % 4 == 0 and a.stride(1) == 1
                  and M * a.stride(0) * 4 < 2 ** 32)      # (the LDS-DMA kernels address an operand through 32-bit offsets)
        if a_pairs is not None:
            # forward: B = the pairs of W (rows = out features); input gradient (b_kcontig False): B = the pairs of W^T (rows = in features)
            w_pairs = self._pairs_of_weight(b, transposed=not b_kcontig)
            wm = self._amax_of_weight(b)
            d.A, d.lda, d.a_planes = a_pairs.data_ptr(), a_pairs.stride(0), 1
            d.B, d.ldb, d.b_planes = w_pairs.data_ptr(), w_pairs.stride(0), 1
            d.a_kcontig, d.b_kcontig = 1, 1
            d.a_amax, d.b_amax = a_scales.row.data_ptr(), (wm.row if b_kcontig else wm.col).data_ptr()
        elif w_only:
            w_pairs = self._pairs_of_weight(b, transposed=not b_kcontig)
            wm = self._amax_of_weight(b)
            if a_scales is not None and a_scales.row is None and a_scales.parts is not None and a_scales.parts.numel() == a_scales.nseg * M:
                sa_w, am_w = a_scales, a_scales.parts          # the producer's per-segment partials: the kernel combines them (a_amax_nseg)
                d.a_amax_nseg = a_scales.nseg
            else:
                sa_w = self.amax(a, a_scales, rows=True)
                am_w = sa_w.row
                if am_w.numel() != M:
                    raise ValueError("gemm: operand maxima do not match the operands")
            d.A, d.lda = a.data_ptr(), _f32_2d(a, "A", dev)
            d.B, d.ldb, d.b_planes = w_pairs.data_ptr(), w_pairs.stride(0), 1
            d.a_kcontig, d.b_kcontig = 1, 1
            d.a_amax, d.b_amax = am_w.data_ptr(), (wm.row if b_kcontig else wm.col).data_ptr()
        elif planes_a is not None:
            d.A, d.lda, d.a_planes, d.a_plane_stride = planes_a.data_ptr(), planes_a.stride(0), 1, 0
            _f32_2d(planes_a, "A", dev, bf16)
        else:
            d.A, d.lda = a.data_ptr(), _f32_2d(a, "A", dev)
        if a_pairs is not None or w_only:
            pass
        elif planes_b is not None and planes_b.dim() == 3:                           # weight planes (3, rows_pad, cols_pad)
            d.B, d.ldb, d.b_planes, d.b_plane_stride = planes_b.data_ptr(), planes_b.stride(1), 1, planes_b.stride(0)
            d.a_kcontig, d.b_kcontig = 1, 1
        elif planes_b is not None:
            d.B, d.ldb, d.b_planes, d.b_plane_stride = planes_b.data_ptr(), planes_b.stride(0), 1, 0
            d.a_kcontig, d.b_kcontig = 0, 0
        else:
            d.B, d.ldb = b.data_ptr(), _f32_2d(b, "B", dev)
            d.a_kcontig, d.b_kcontig = int(a_kcontig), int(b_kcontig)
        # ---- the native fp32 kernel (precision "f32", or M / N <= 32) has no bf16 epilogue: run it on fp32 copies (tiny or non-default)
        native = planes_a is None and a_pairs is None and not w_only and (not big or d.precision == _lib.GEMM_PRECISIONS["f32"])
        if native and any(t is not None and t.dtype == bf16 for t in (out, out2, res, aux)):
            f = lambda t: None if t is None else (self.to_f32(t) if t.dtype == bf16 else t)      # noqa: E731
            o32 = torch.empty((M, N), dtype=torch.float32, device=dev) if out.dtype == bf16 else out
            o232 = None if out2 is None else (torch.empty((M, N), dtype=torch.float32, device=dev) if out2.dtype == bf16 else out2)
            self.gemm(a, b, o32, M=M, N=N, K=K, a_kcontig=a_kcontig, b_kcontig=b_kcontig, bias=bias, res=f(res), aux=f(aux), pre=pre, act=act,
                      drop_p=drop_p, drop_seed=drop_seed, accumulate=accumulate, out2=o232, a_colsum=a_colsum, precision=precision)
            for dst, src in ((out, o32), (out2, o232)):
                if dst is not None and dst is not src:
                    _chk(self.lib.grappa_convert_f32_to_bf16(self._stream(), M, N, src.data_ptr(), N, dst.data_ptr(), _f32_2d(dst, "out", dev, bf16)),
                         "grappa_convert_f32_to_bf16")
            return (None, None) if out_amax else None
        # ---- outputs and epilogue tensors, each in its own element type
        def plane_ok(t, name):
            ld = _f32_2d(t, name, dev, bf16)
            if t.data_ptr() % 8 or ld % 4:
                raise ValueError(f"gemm: bf16 {name} needs 8-byte aligned rows")
            return ld
        if out2 is not None and tuple(out2.shape) != (M, N):
            raise ValueError("gemm: out2 shape")
        final = out2 if out2 is not None else out
        if final.dtype == bf16:
            if accumulate:
                raise ValueError("gemm: accumulate needs a float32 output")
            d.Cp, d.ldcp, d.cp_nplanes = final.data_ptr(), plane_ok(final, "out"), 1
            if out2 is not None:                                                     # (out, out2) = (value before dropout / residual, final value)
                if out.dtype != bf16:
                    raise ValueError("gemm: out and out2 must share an element type")
                d.C1p, d.ldc1p = out.data_ptr(), plane_ok(out, "out (pre-dropout copy)")
        else:
            d.C, d.ldc = out.data_ptr(), _f32_2d(out, "C", dev)
            if out2 is not None:
                if out2.dtype != torch.float32:
                    raise ValueError("gemm: out and out2 must share an element type")
                d.C2, d.ldc2 = out2.data_ptr(), _f32_2d(out2, "C2", dev)
        if bias is not None:
            _flat(bias, "bias", dev)
            if bias.numel() != N:
                raise ValueError("gemm: bias length")
            d.bias = bias.data_ptr()
        if res is not None:
            if tuple(res.shape) != (M, N):
                raise ValueError("gemm: res shape")
            if res.dtype == bf16:
                d.resp, d.ldresp, d.resp_nplanes = res.data_ptr(), plane_ok(res, "res"), 1
            else:
                d.res, d.ldres = res.data_ptr(), _f32_2d(res, "res", dev)
        if res_ln is not None:
            # res holds the rows BEFORE a LayerNorm; the epilogue adds LayerNorm(res) = what the LayerNorm kernel would have written
            mean_, rstd_, gamma_, beta_ = res_ln
            if res is None or res.dtype != torch.float32 or not big or d.precision == _lib.GEMM_PRECISIONS["f32"]:
                raise ValueError("gemm: res_ln needs a float32 residual and a product of the split kernels (M, N > 32, not the native fp32 MFMA)")
            for t_, n_, k_ in ((mean_, "mean", M), (rstd_, "rstd", M), (gamma_, "gamma", N), (beta_, "beta", N)):
                _flat(t_, f"res_ln {n_}", dev)
                if t_.numel() != k_:
                    raise ValueError(f"gemm: res_ln {n_} length")
            d.res_ln_mean, d.res_ln_rstd, d.res_ln_gamma, d.res_ln_beta = mean_.data_ptr(), rstd_.data_ptr(), gamma_.data_ptr(), beta_.data_ptr()
        if aux is not None:
            if tuple(aux.shape) != (M, N):
                raise ValueError("gemm: aux shape")
            if aux.dtype == bf16:
                d.auxp, d.ldauxp, d.auxp_nplanes = aux.data_ptr(), plane_ok(aux, "aux"), 1
            else:
                d.aux, d.ldaux = aux.data_ptr(), _f32_2d(aux, "aux", dev)
        if pre is not None:
            if tuple(pre.shape) != (M, N):
                raise ValueError("gemm: pre shape")
            d.pre, d.ldpre = pre.data_ptr(), _f32_2d(pre, "pre", dev)
        if a_colsum is not None:
            _flat(a_colsum, "a_colsum", dev)
            if a_kcontig or a_colsum.numel() != M:
                raise ValueError("gemm: a_colsum needs the row-contiguous A layout and length M")
            d.a_colsum = a_colsum.data_ptr()
        d.act, d.drop_p, d.drop_seed, d.accumulate = int(act), float(drop_p), int(drop_seed) & (2 ** 64 - 1), int(accumulate)
        self._call_options(d)
        sa = so = None
        if a_pairs is not None:
            sa = a_scales
        elif w_only:
            sa = sa_w
        elif d.precision == _lib.GEMM_PRECISIONS["f32_f16x3"] and big and planes_a is None and planes_b is None:
            # power-of-two scales of both operands from their largest magnitudes along the reduced dimension
            if a_kcontig:
                wm = self._amax_of_weight(b)
                bm = wm.row if b_kcontig else wm.col
                if a_scales is not None and a_scales.row is None and a_scales.parts is not None and a_scales.parts.numel() == a_scales.nseg * M:
                    sa, am = a_scales, a_scales.parts              # the producer's partials: this product combines them (a_amax_nseg)
                    d.a_amax_nseg = a_scales.nseg
                else:
                    sa = self.amax(a, a_scales, rows=True)
                    am = sa.row
                    if am.numel() != M:
                        raise ValueError("gemm: operand maxima do not match the operands")
                if bm.numel() != N:
                    raise ValueError("gemm: operand maxima do not match the operands")
            elif self.wgrad_column_maxima:
                sa = self.amax(a, a_scales, cols=True)
                am, bm = sa.col, self.amax(b, b_scales, cols=True).col
            else:
                # weight gradient: the reduction runs over the rows (tokens) of both operands -> one scale per operand
                sa = self.amax(a, a_scales, tmax=True)
                am, bm = sa.tmax, self.amax(b, b_scales, tmax=True).tmax
                d.amax_bcast = 3
            d.a_amax, d.b_amax = am.data_ptr(), bm.data_ptr()
        if out_amax is True and final.dtype == torch.float32:          # (out_amax == "pair": the caller only wants the pair returned)
            if self.amax_parts and big and not native and planes_a is None and (planes_b is None or a_pairs is not None):
                nseg = (N + 31) // 32
                so = Amax(parts=torch.empty(nseg * M, dtype=torch.int32, device=dev), nseg=nseg)
                d.out_amax_parts = so.parts.data_ptr()
            else:
                so = Amax(row=torch.empty(M, dtype=torch.int32, device=dev))
                d.out_amax = so.row.data_ptr()
        need = self._ws_bytes(d, (M, N, K))
        ws = self._workspace(need, dev) if need else None
        el = lambda t: 0 if t is None else t.element_size()      # noqa: E731
        # algorithmic bytes of the fused call: both operands once, the result, and what the epilogue has to read / write beside it
        # (residual, saved activation for ELU', fp32 addend, the second output, an accumulated result's old value)
        epi = M * N * (el(res) + el(aux) + el(pre) + (el(out) if out2 is not None else 0) + (el(final) if accumulate else 0))
        nbytes = float(M * K * (4 if a_pairs is not None else el(a if planes_a is None else planes_a)) + N * K * (2 if planes_b is not None else 4)
                       + M * N * el(final) + epi)
        ret = (sa, so) if out_amax else sa
        if _defer:                               # gemm_group: the caller launches this product together with others
            return _PendingGemm(d, (M, N, K), 2.0 * M * N * K, nbytes, ret, dev, (a, b, out, out2, res, aux, pre, bias, a_pairs, sa, so, res_ln),
                                groupable=big and a_kcontig and a_colsum is None and planes_a is None and (planes_b is None or a_pairs is not None)
                                and not native and final.dtype == torch.float32)
        self._timed("gemm_f32", 2.0 * M * N * K, nbytes, lambda: self._launch_gemm(d, ws, dev, (M, N, K)), lambda: [self._gemm_detail(d)])
        return ret

    def _call_options(self, d) -> None:
        """the per-call options of a product (C ABI 10: they used to be process-wide setters of the library)"""
        d.drop_salt = self._salt_ptr
        d.plan_tail = 0 if self._tails is None else (1 if self._tails else 2)
        d.splitk_reduce = self.splitk_reduce
        if self.plan_override is not None:
            cfg, ns, tail = self.plan_override
            d.plan_cfg, d.plan_nsplit = (cfg + 1 if cfg >= 0 else 0), max(int(ns), 0)
            if tail == 0:
                d.plan_tail = 2
            elif tail == 1:
                d.plan_tail = 3

    def _ws_bytes(self, d, shape) -> int:
        """workspace of a product: per shape (enough for every tail / reduction setting), or asked per descriptor under a plan override"""
        if self.plan_override is not None:
            return self.lib.grappa_gemm_f32_workspace_bytes_desc(C.byref(d))
        need = self._ws_need.get(shape)
        if need is None:                         # (the query plans the product for every kernel family: 5 - 10 us, the same answer per shape)
            need = self._ws_need[shape] = self.lib.grappa_gemm_f32_workspace_bytes(*shape)
        return need

    def gemm_group(self, calls):
        """calls: [(args, kwargs)] of `gemm` -- independent products, e.g. the same product of the four writer heads -- launched as ONE grid
        where the library can (C ABI 8 grappa_gemm_f32_group: forward or input-gradient layout, one operand format), else one by one.
        -> the list of `gemm`'s return values.  At small batches one head's product leaves most of the chip idle for a whole tile time."""
        pend = [self.gemm(*a, **dict(k, _defer=True)) for a, k in calls]
        live = [p for p in pend if isinstance(p, _PendingGemm)]
        out = [p.ret if isinstance(p, _PendingGemm) else p for p in pend]
        i = 0
        while i < len(live):
            grp = [live[i]]
            if self.group_launches and live[i].groupable:
                for q in live[i + 1:i + _lib.GEMM_GROUP4_MAX]:
                    if q.groupable and q.key == live[i].key:
                        grp.append(q)
                    else:
                        break
            i += len(grp)
            if len(grp) >= 2 and self._launch_gemm_group(grp):
                continue
            for q in grp:
                need = self._ws_bytes(q.d, q.shape)
                ws = self._workspace(need, q.dev) if need else None
                self._timed("gemm_f32", q.flops, q.nbytes, lambda q=q, ws=ws: self._launch_gemm(q.d, ws, q.dev, q.shape), lambda q=q: [self._gemm_detail(q.d)])
        return out

    def _launch_gemm_group(self, grp) -> bool:
        n = len(grp)
        arr = (_lib.GemmDesc * n)()
        for dst, q in zip(arr, grp):
            C.memmove(C.byref(dst), C.byref(q.d), C.sizeof(_lib.GemmDesc))
        need = self.lib.grappa_gemm_f32_group_workspace_bytes(arr, n)
        ws = self._workspace(need, grp[0].dev) if need else None
        rc = [0]

        def launch():
            rc[0] = self.lib.grappa_gemm_f32_group(self._stream(), arr, n, _ptr(ws), ws.numel() if ws is not None else 0)
        self._timed("gemm_f32", sum(q.flops for q in grp), sum(q.nbytes for q in grp), launch, lambda: [self._gemm_detail(q.d) for q in grp])
        if rc[0] == -1:                          # GRAPPA_ERR_ARG: a combination the grouped entry does not take (nothing was launched)
            if self._prof:
                self._prof.pop()
            return False
        _chk(rc[0], "grappa_gemm_f32_group")
        return True

    def _launch_gemm(self, d, ws, dev, shape) -> None:
        rc = self.lib.grappa_gemm_f32(self._stream(), C.byref(d), _ptr(ws), ws.numel() if ws is not None else 0)
        if rc == -3:                             # GRAPPA_ERR_WORKSPACE: the cached size predates a plan setting (override, environment): ask again
            need = self.lib.grappa_gemm_f32_workspace_bytes_desc(C.byref(d))
            ws = self._workspace(need, dev) if need else None
            rc = self.lib.grappa_gemm_f32(self._stream(), C.byref(d), _ptr(ws), ws.numel() if ws is not None else 0)
        _chk(rc, "grappa_gemm_f32")

    # ------------------------------------------------------------------ weight gradients, grouped
    def gemm_wgrad(self, dz, x, dw, db=None, dz_scales=None, x_scales=None):
        """dW += dz^T x (dz (tokens, N'), x (tokens, K'), dW (N', K')), db += column sums of dz.  fp32 products with enough rows and
        columns are queued and launched as ONE grouped grid when 16 are waiting or the backward pass ends (autograd's end-of-pass
        callback; `flush_wgrads()` is also called by the gradient reducer and the optimiser): results are those of `gemm` up to the
        summation order of the K chunks.  Returns the `Amax` record of dz (precision "f32_f16x3") for the input-gradient product.
        An operand whose record carries `.pairs` (its producer wrote it in the pair format, every token row under its own scale) is read
        from there (C ABI 8) and its fp32 tensor may be None."""
        pz, px = getattr(dz_scales, "pairs", None), getattr(x_scales, "pairs", None)
        Np, Kp = dw.shape
        T = (dz if dz is not None else pz).shape[0]
        prec = self.gemm_precision if self.gemm_precision_bwd is None else self.gemm_precision_bwd
        f32ok = lambda t: t is None or t.dtype == torch.float32      # noqa: E731
        ok = (self.defer_wgrads and f32ok(dz) and f32ok(x) and Np > 32 and Kp > 32 and T > 0 and prec != _lib.GEMM_PRECISIONS["f32"])
        pairs_ok = ok and prec == _lib.GEMM_PRECISIONS["f32_f16x3"] and not self.wgrad_column_maxima
        if pz is not None and not (pairs_ok and Np % 32 == 0 and tuple(pz.shape) == (T, 2 * Np)):
            pz = None
        if px is not None and not (pairs_ok and Kp % 32 == 0 and tuple(px.shape) == (T, 2 * Kp)):
            px = None
        if (dz is None and pz is None) or (x is None and px is None):
            raise ValueError("gemm_wgrad: an operand was given in the pair format only, which this product cannot read")
        if not ok:
            return self.gemm(dz, x, dw, M=Np, N=Kp, K=T, a_kcontig=False, b_kcontig=False, accumulate=True, a_colsum=db,
                             a_scales=dz_scales, b_scales=x_scales)
        dev = dw.device
        if (dz is not None and tuple(dz.shape) != (T, Np)) or (x is not None and tuple(x.shape) != (T, Kp)) or (db is not None and db.numel() != Np):
            raise ValueError("gemm_wgrad: shapes")
        _f32_2d(dw, "dW", dev)
        if pz is None:
            _f32_2d(dz, "dz", dev)
        if px is None:
            _f32_2d(x, "x", dev)
        if db is not None:
            _flat(db, "db", dev)
        sdz = am = None
        if prec == _lib.GEMM_PRECISIONS["f32_f16x3"]:
            # row maxima now (they also serve the input-gradient product of dz); the whole-tensor maxima of the group in one launch at the flush
            if self.wgrad_column_maxima:
                sdz = self.amax(dz, dz_scales, rows=True, cols=True)
                am = (sdz, self.amax(x, x_scales, cols=True))
            else:
                has = lambda r: r is not None and (r.row is not None or r.parts is not None)      # noqa: E731
                sdz = dz_scales if (pz is not None or has(dz_scales)) else self.amax(dz, dz_scales, rows=True)
                am = (sdz, x_scales if (px is not None or has(x_scales)) else self.amax(x, x_scales, rows=True))
        task = self._queue_flush()                # the backward pass (autograd graph task) this product belongs to; -1 outside of one
        # one queue per backward pass and HIP stream (the writer heads run their backward passes on streams of their own): a full queue is
        # launched on the stream that filled it, what is left when the pass ends is launched together by flush_wgrads
        st = torch.cuda.current_stream()
        q = self._wq.setdefault((task, st.cuda_stream), (st, []))[1]
        q.append((dz if pz is None else None, x if px is None else None, dw, db, am, pz, px))
        if task < 0:                              # not inside a backward pass: nothing will call back
            self.flush_wgrads(-1)
            return sdz
        # a queue is launched when it is full -- or when the operands it keeps alive exceed the byte budget (at C3 / C4 sizes
        # sixteen (dz, x) pairs are tens of GB)
        if len(q) >= _lib.GEMM_GROUP_MAX or sum(self._item_bytes(it) for it in q) > self.wgrad_queue_bytes:
            del self._wq[(task, st.cuda_stream)]
            self._launch_wgrad_group(q)
        return sdz

    @staticmethod
    def _item_bytes(it) -> int:
        return sum(t.numel() * t.element_size() for t in (it[0], it[1], it[5], it[6]) if t is not None)

    @property
    def _wq_task(self):
        """the backward pass whose end-of-pass callback is pending (None: none) -- tests / tools"""
        return max(self._tasks) if self._tasks else None

    def _queue_flush(self) -> int:
        """-> autograd's graph task id of the running backward pass (-1 outside of one), after asking autograd to call flush_wgrads for
        that pass when it ends.  Every pass has queues of its own: a re-entrant pass inside a running one
        (torch.utils.checkpoint, torch.autograd.grad in a hook) launches its own products at its own end and leaves the outer pass's alone.
        A pass that DIED (an exception inside backward drops autograd's end-of-pass callbacks) leaves its queues behind; they are never
        launched -- its gradient buffer was abandoned with it -- and are discarded by `drop_deferred` (FlatParams.zero_grad) or by the
        first flush outside of any pass (FusedAdam.step, the gradient reducer)."""
        task = torch._C._current_graph_task_id()
        if task < 0:
            return -1
        if task not in self._tasks:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(lambda t=task: self.flush_wgrads(t))
            except RuntimeError:
                return -1
            # a TOP-LEVEL pass that starts while older ids are still pending: those passes are dead unless this one runs inside them;
            # a nested pass always starts from inside a running node of the outer one, i.e. with the engine's current task still alive --
            # which is not observable from here, so the older queues are only dropped where it is certain (see above)
            self._tasks.add(task)
        return task

    def drop_deferred(self) -> None:
        """forget queued weight-gradient products and LayerNorm reductions without launching them (leftovers of an aborted backward pass)"""
        self._join_aside()
        self._wq, self._lnq, self._tasks = {}, [], set()

    # ---- weight gradients beside the pass.  The GNN's backward pass is a chain of small products (8,233 atom rows at C2: 132 workgroups on
    # 256 CUs) that leaves half of the chip idle, and the weight gradients nobody waits for pile up behind it.  launch_wgrads_aside() takes
    # what is queued and launches it AT ONCE as grouped grids on a side stream, ordered behind the streams that produced the operands; the
    # operands stay referenced until flush_wgrads() has put the caller's stream behind the side stream again.
    def set_tail_launches(self, on: bool) -> None:
        """split-K tail launches of the products that follow (include/grappa_hip.h grappa_gemm_desc.plan_tail: sent with every product, C ABI 10).  The model turns
        them off while the writer heads keep several streams busy (a partial last round then runs beside another head's kernels: C2 step
        36.4 -> 36.0 ms) and on again on one stream (37.5 -> 37.4).  Ignored while pinned (pin_tail_launches; GRAPPA_PLAN_TAILS in the
        environment pins at start-up)."""
        if self._tails_pinned:
            return
        on = bool(on)
        self._tails = on

    def pin_tail_launches(self, on: Optional[bool]) -> None:
        """True / False: tail launches on / off whatever the model asks for (comparisons that need the same K cuts on one stream and on
        four); None: back to the model's choice"""
        self._tails_pinned = False
        if on is not None:
            self.set_tail_launches(on)
            self._tails_pinned = True

    def launch_wgrads_aside(self, all_streams: bool = False) -> None:
        if not self.wgrads_aside or not self._wq:
            return
        cur = torch.cuda.current_stream()
        task = torch._C._current_graph_task_id()
        keys = [k for k in self._wq if k[0] == task and (all_streams or k[1] == cur.cuda_stream)]      # this pass's queues only
        queues = [self._wq.pop(k) for k in keys]
        items = [it for _, q in queues for it in q]
        if not items:
            return
        key = (cur.device, cur.cuda_stream)
        side = self._side_streams.get(key)
        if side is None:
            side = self._side_streams[key] = torch.cuda.Stream(device=cur.device)
        for st, _ in queues:
            side.wait_stream(st)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for i in range(0, len(items), _lib.GEMM_GROUP_MAX):
                self._launch_wgrad_group(items[i:i + _lib.GEMM_GROUP_MAX])
        self._aside.append((side, items))

    def _join_aside(self) -> None:
        if self._aside:
            cur = torch.cuda.current_stream()
            for side, _ in self._aside:
                cur.wait_stream(side)
            self._aside = []

    def flush_wgrads(self, task: Optional[int] = None) -> None:
        """launch what a backward pass has queued: the grouped weight gradients and the LayerNorm parameter-gradient reductions.
        task: the pass (its end-of-pass callback passes its id); None: the running pass -- or, outside of any pass, nothing is left to
        launch (passes that ended launched their own) and whatever is still queued belongs to passes that died: discarded."""
        if task is None:
            task = torch._C._current_graph_task_id()
            if task < 0:
                if self._wq or self._lnq or self._tasks:
                    self._wq = {k: v for k, v in self._wq.items() if k[0] < 0}
                    self._lnq = [it for it in self._lnq if it[8] < 0]
                    self._tasks = set()
                task = -1
        self._tasks.discard(task)
        cur = torch.cuda.current_stream()
        keys = [k for k in self._wq if k[0] == task]
        if keys:
            items = []
            for k in keys:
                st, q = self._wq.pop(k)
                if st != cur:
                    cur.wait_stream(st)           # (operands queued on another stream: normally already ordered before this point, see ops.SplitHeadsFn)
                items += q
            for i in range(0, len(items), _lib.GEMM_GROUP_MAX):
                self._launch_wgrad_group(items[i:i + _lib.GEMM_GROUP_MAX])
        mine = [it for it in self._lnq if it[8] == task]
        if mine:
            self._lnq = [it for it in self._lnq if it[8] != task]
            for it in mine:
                if it[7] != cur:
                    cur.wait_stream(it[7])
            items = [it[:7] for it in mine]
            arr = (_lib.ColsumItem * len(items))()
            for d, (ws, nrows, W, pg, pb, _g, _b) in zip(arr, items):
                d.part, d.nrows, d.n, d.out, d.out2, d.n_first, d.accumulate = ws.data_ptr(), nrows, 2 * W, pg, pb, W, 1
            _chk(self.lib.grappa_colsum_partials_batched(self._stream(), arr, len(items)), "grappa_colsum_partials_batched")
        self._join_aside()                        # the gradients launched beside the pass are complete for whatever follows on this stream

    def _launch_wgrad_group(self, items) -> None:
        # one grid per load style and operand format: products whose operands allow 16-byte loads along their rows (aligned, leading
        # dimension % 4 == 0 and covering round_up(columns, 4)) run the faster kernel together; an odd one (513-wide tuple features)
        # would drag its whole group onto the dword-load kernel
        def vec(t):
            return t is None or (t.data_ptr() % 16 == 0 and t.stride(0) % 4 == 0 and (t.shape[1] + 3) // 4 * 4 <= t.stride(0))
        groups = {}
        for it in items:
            groups.setdefault(vec(it[0]) and vec(it[1]), []).append(it)        # (operand formats may mix inside a launch: C ABI 8)
        for part in groups.values():
            self._launch_wgrad_items(part)

    def _launch_wgrad_items(self, items) -> None:
        n = len(items)
        dev = items[0][2].device
        prec = self.gemm_precision if self.gemm_precision_bwd is None else self.gemm_precision_bwd
        arr = (_lib.GemmDesc * n)()
        flops = nbytes = 0.0
        if not self.wgrad_column_maxima:
            self._tmax_of([r for it in items if it[4] is not None for r in it[4]], dev)
        for d, (dz, x, dw, db, am, pz, px) in zip(arr, items):
            d.M, d.N = dw.shape
            d.K = (dz if dz is not None else pz).shape[0]
            d.a_kcontig, d.b_kcontig = 0, 0
            if pz is not None:
                d.A, d.lda, d.a_planes, d.a_rowmax = pz.data_ptr(), pz.stride(0), 1, am[0].row.data_ptr()
            else:
                d.A, d.lda = dz.data_ptr(), dz.stride(0)
            if px is not None:
                d.B, d.ldb, d.b_planes, d.b_rowmax = px.data_ptr(), px.stride(0), 1, am[1].row.data_ptr()
            else:
                d.B, d.ldb = x.data_ptr(), x.stride(0)
            d.C, d.ldc = dw.data_ptr(), dw.stride(0)
            d.a_colsum = None if db is None else db.data_ptr()
            d.accumulate, d.precision = 1, prec
            d.drop_salt, d.splitk_reduce = self._salt_ptr, self.splitk_reduce
            if am is not None:
                if self.wgrad_column_maxima:
                    d.a_amax, d.b_amax = am[0].col.data_ptr(), am[1].col.data_ptr()
                else:
                    d.a_amax, d.b_amax, d.amax_bcast = am[0].tmax.data_ptr(), am[1].tmax.data_ptr(), 3
            flops += 2.0 * d.M * d.N * d.K
            nbytes += 4.0 * (d.M * d.K + d.N * d.K + d.M * d.N)
        need = self.lib.grappa_gemm_f32_grouped_workspace_bytes(arr, n)
        ws = self._workspace(need, dev)
        self._timed("gemm_f32", flops, nbytes,
                    lambda: _chk(self.lib.grappa_gemm_f32_grouped(self._stream(), arr, n, ws.data_ptr(), ws.numel()), "grappa_gemm_f32_grouped"),
                    lambda: [self._gemm_detail(d) for d in arr])

    def colsum(self, x, out, accumulate=False) -> None:
        dev = out.device
        M, N = x.shape
        x = self.to_f32(x)
        ldx = _f32_2d(x, "x", dev)
        _flat(out, "out", dev)
        if out.numel() != N:
            raise ValueError("colsum: out length")
        ws = self._workspace(self.lib.grappa_colsum_workspace_bytes(M, N), dev)
        _chk(self.lib.grappa_colsum_f32(self._stream(), M, N, x.data_ptr(), ldx, out.data_ptr(), int(accumulate), ws.data_ptr(), ws.numel()),
             "grappa_colsum_f32")

    def act_dropout_bwd(self, dy, y, drop_p, drop_seed, dz, amax=None, pairs=False):
        """-> the `Amax` record (row maxima) of dz when the backward products run on fp16 pieces (amax=False: never), else None.
        pairs=True (fp32, N % 32 == 0): the rows are written in the pair format (record's .pairs); dz may then be None"""
        dev = dy.device
        M, N = dy.shape
        if pairs:
            if dy.dtype != torch.float32 or N % 32 or N > 2048 or (dz is not None and (dz.dtype != torch.float32 or tuple(dz.shape) != (M, N))) or \
                    (y is not None and (y.dtype != torch.float32 or tuple(y.shape) != (M, N))):
                raise ValueError("act_dropout_bwd: the pair format needs float32 rows of N % 32 == 0 (<= 2048) columns")
            row = torch.empty(M, dtype=torch.int32, device=dev)
            pr = torch.empty((M, 2 * N), dtype=torch.float16, device=dev)
            if M:
                _chk(self.lib.grappa_act_dropout_bwd_pairs_f32(self._stream(), M, N, dy.data_ptr(), _f32_2d(dy, "dy", dev), _ptr(y),
                                                               _f32_2d(y, "y", dev) if y is not None else 0, float(drop_p), int(drop_seed) & (2 ** 64 - 1),
                                                               _ptr(dz), _f32_2d(dz, "dz", dev) if dz is not None else 0, row.data_ptr(), pr.data_ptr(),
                                                               pr.stride(0), self._salt_ptr), "grappa_act_dropout_bwd_pairs_f32")
            return Amax(row=row, pairs=pr)
        if tuple(dz.shape) != (M, N) or (y is not None and tuple(y.shape) != (M, N)):
            raise ValueError("act_dropout_bwd: shapes")
        dt = _same_dtype(dy, y, dz)
        row = self._new_row_amax(dz, True, amax)
        args = (self._stream(), M, N, dy.data_ptr(), _f32_2d(dy, "dy", dev, dt), _ptr(y), _f32_2d(y, "y", dev, dt) if y is not None else 0,
                float(drop_p), int(drop_seed) & (2 ** 64 - 1), dz.data_ptr(), _f32_2d(dz, "dz", dev, dt))
        if row is not None:
            _chk(self.lib.grappa_act_dropout_bwd_amax_f32(*args, row.data_ptr(), self._salt_ptr), "grappa_act_dropout_bwd_amax_f32")
            return Amax(row=row)
        _chk(getattr(self.lib, f"grappa_act_dropout_bwd_{_sfx(dz)}")(*args, self._salt_ptr), "grappa_act_dropout_bwd")
        return None

    def add(self, x, z, y) -> None:
        dev = y.device
        for t, n in ((x, "x"), (z, "z"), (y, "y")):
            _flat(t, n, dev)
        if not (x.numel() == z.numel() == y.numel()):
            raise ValueError("add: sizes")
        _chk(self.lib.grappa_add_f32(self._stream(), x.numel(), x.data_ptr(), z.data_ptr(), y.data_ptr()), "grappa_add_f32")

    # ------------------------------------------------------------------ layer norm
    def layernorm_fwd(self, x, gamma, beta, y, mean, rstd, amax=None, pairs=False):
        """pairs=True (fp32, W % 32 == 0): the rows are ALSO written in the pair format (returned record's `.pairs`); y may then be None"""
        dev = x.device
        M, W = x.shape
        _flat(gamma, "gamma", dev), _flat(beta, "beta", dev)
        if gamma.numel() != W or beta.numel() != W or (y is not None and tuple(y.shape) != (M, W)):
            raise ValueError("layernorm: shapes")
        if pairs:
            if x.dtype != torch.float32 or W % 32 or (y is not None and y.dtype != torch.float32):
                raise ValueError("layernorm: the pair format needs float32 rows of W % 32 == 0 columns")
            row = torch.empty(M, dtype=torch.int32, device=dev)
            pr = torch.empty((M, 2 * W), dtype=torch.float16, device=dev)
            if M:
                _chk(self.lib.grappa_layernorm_fwd_pairs_f32(self._stream(), M, W, x.data_ptr(), _f32_2d(x, "x", dev), gamma.data_ptr(), beta.data_ptr(),
                                                             _ptr(y), _f32_2d(y, "y", dev) if y is not None else 0, _ptr(mean), _ptr(rstd),
                                                             row.data_ptr(), pr.data_ptr(), pr.stride(0)), "grappa_layernorm_fwd_pairs_f32")
            return Amax(row=row, pairs=pr)
        if y is None:
            raise ValueError("layernorm: y is required without pairs")
        if mean is not None:
            _flat(mean, "mean", dev), _flat(rstd, "rstd", dev)
            if mean.numel() != M or rstd.numel() != M:
                raise ValueError("layernorm: stats length")
        dt = _same_dtype(x, y)
        row = self._new_row_amax(y, False, amax)
        args = (self._stream(), M, W, x.data_ptr(), _f32_2d(x, "x", dev, dt), gamma.data_ptr(), beta.data_ptr(),
                y.data_ptr(), _f32_2d(y, "y", dev, dt), _ptr(mean), _ptr(rstd))
        if row is not None:
            _chk(self.lib.grappa_layernorm_fwd_amax_f32(*args, row.data_ptr()), "grappa_layernorm_fwd_amax_f32")
            return Amax(row=row)
        _chk(getattr(self.lib, f"grappa_layernorm_fwd_{_sfx(x)}")(*args), "grappa_layernorm_fwd")
        return None

    def layernorm_bwd(self, dy, x, mean, rstd, gamma, dx, dgamma, dbeta, accumulate=True, amax=None, drop=None):
        """drop=(p, seed): ALSO -> dz = the dropout backward (that mask, 1 / (1 - p)) of dx with its row maxima, written by the same
        launch (C ABI 9); the return value is then (record of dx, dz, record of dz).  `drop_fusable(x)` says when."""
        dev = dx.device
        M, W = x.shape
        for t, n, k in ((mean, "mean", M), (rstd, "rstd", M), (gamma, "gamma", W), (dgamma, "dgamma", W), (dbeta, "dbeta", W)):
            _flat(t, n, dev)
            if t.numel() != k:
                raise ValueError(f"layernorm_bwd: {n} length")
        if tuple(dy.shape) != (M, W) or tuple(dx.shape) != (M, W):
            raise ValueError("layernorm_bwd: shapes")
        dt = _same_dtype(dy, x, dx)
        # inside a backward pass the parameter gradients wait: the kernel leaves its per-block partial sums in a buffer of their own and
        # ONE launch reduces those of all LayerNorms when the pass ends (flush_wgrads) instead of two small launches per LayerNorm
        task = self._queue_flush() if (accumulate and self.defer_wgrads and self.defer_ln and M > 0) else -1
        defer = task >= 0 and all(q[3] != dgamma.data_ptr() for q in self._lnq if q[8] == task)      # (once per pass and parameter)
        need = self.lib.grappa_layernorm_bwd_workspace_bytes(M, W)
        ws = torch.empty(need, dtype=torch.uint8, device=dev) if defer else self._workspace(need, dev)
        row = self._new_row_amax(dx, True, amax)
        args = (self._stream(), M, W, dy.data_ptr(), _f32_2d(dy, "dy", dev, dt), x.data_ptr(), _f32_2d(x, "x", dev, dt),
                mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), dx.data_ptr(), _f32_2d(dx, "dx", dev, dt),
                dgamma.data_ptr(), dbeta.data_ptr(), 2 if defer else int(accumulate), ws.data_ptr(), ws.numel())
        if defer:
            self._lnq.append((ws, self.lib.grappa_layernorm_bwd_partial_rows(M), W, dgamma.data_ptr(), dbeta.data_ptr(), dgamma, dbeta,
                              torch.cuda.current_stream(), task))
        if drop is not None:
            p, seed = drop
            if not self.drop_fusable(x) or not (0.0 < p < 1.0):
                raise ValueError("layernorm_bwd: drop needs fp32 rows and the fp16-split arithmetic (drop_fusable)")
            dz = torch.empty_like(dx)
            zrow = torch.empty(M, dtype=torch.int32, device=dev)
            _chk(self.lib.grappa_layernorm_bwd_drop_f32(*args, _ptr(row), float(p), int(seed) & (2 ** 64 - 1), dz.data_ptr(), dz.stride(0), zrow.data_ptr(),
                                                        self._salt_ptr), "grappa_layernorm_bwd_drop_f32")
            return (Amax(row=row) if row is not None else None), dz, Amax(row=zrow)
        if row is not None:
            _chk(self.lib.grappa_layernorm_bwd_amax_f32(*args, row.data_ptr()), "grappa_layernorm_bwd_amax_f32")
            return Amax(row=row)
        _chk(getattr(self.lib, f"grappa_layernorm_bwd_{_sfx(x)}")(*args), "grappa_layernorm_bwd")
        return None

    def drop_fusable(self, x) -> bool:
        """may the LayerNorm backward over rows like x write the dropout backward of its result too (layernorm_bwd drop=)?"""
        return (self.fuse_ln_drop and not self.backward_pairs and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] > 0 and x.is_contiguous()
                and self.wants_amax(True))

    # ------------------------------------------------------------------ batched row-wise kernels (C ABI 8): the writer heads layer-locked
    @staticmethod
    def _rows_ok(*ts) -> bool:
        return all(t is None or (t.dtype == torch.float32 and t.dim() == 2 and t.is_contiguous() and t.data_ptr() % 16 == 0 and t.shape[1] % 4 == 0
                                 and t.shape[1] <= 2048) for t in ts)

    def layernorm_fwd_batched(self, xs, gammas, betas):
        """the LayerNorms of several tensors (<= 4) in ONE launch -> [(y, mean, rstd, record of y's row maxima or None)]; None if a tensor
        does not qualify (fp32, contiguous rows of W % 4 == 0 <= 2048 columns): the caller then goes one by one"""
        n = len(xs)
        if not (2 <= n <= _lib.ROW_BATCH_MAX) or not self._rows_ok(*xs) or any(x.shape[0] == 0 for x in xs):
            return None
        want = self.wants_amax(False)
        arr = (_lib.LnFwdItem * n)()
        out = []
        for it, x, g, b in zip(arr, xs, gammas, betas):
            M, W = x.shape
            dev = x.device
            _flat(g, "gamma", dev), _flat(b, "beta", dev)
            if g.numel() != W or b.numel() != W:
                raise ValueError("layernorm: shapes")
            y = torch.empty_like(x)
            mean, rstd = torch.empty(M, dtype=torch.float32, device=dev), torch.empty(M, dtype=torch.float32, device=dev)
            row = torch.empty(M, dtype=torch.int32, device=dev) if want else None
            it.M, it.W, it.x, it.ldx, it.gamma, it.beta, it.y, it.ldy = M, W, x.data_ptr(), x.stride(0), g.data_ptr(), b.data_ptr(), y.data_ptr(), y.stride(0)
            it.mean, it.rstd, it.y_amax = mean.data_ptr(), rstd.data_ptr(), _ptr(row)
            out.append((y, mean, rstd, Amax(row=row) if want else None))
        _chk(self.lib.grappa_layernorm_fwd_batched_f32(self._stream(), arr, n), "grappa_layernorm_fwd_batched_f32")
        return out

    def layernorm_bwd_batched(self, items):
        """items: [(dy, x, mean, rstd, gamma, dgamma, dbeta)] -> [(dx, record or None)] in ONE launch, the parameter gradients left as
        partials for the end-of-pass reduction (inside a backward pass only); None: go one by one"""
        n = len(items)
        if not (2 <= n <= _lib.ROW_BATCH_MAX) or not (self.defer_wgrads and self.defer_ln):
            return None
        if not self._rows_ok(*[t for it in items for t in (it[0], it[1])]) or any(it[1].shape[0] == 0 for it in items):
            return None
        task = self._queue_flush()
        if task < 0 or any(q[3] == it[5].data_ptr() for it in items for q in self._lnq if q[8] == task):
            return None
        if len({it[5].data_ptr() for it in items}) != n:
            return None
        want = self.wants_amax(True)
        arr = (_lib.LnBwdItem * n)()
        out = []
        cur = torch.cuda.current_stream()
        for a, (dy, x, mean, rstd, gamma, dgamma, dbeta) in zip(arr, items):
            M, W = x.shape
            dev = x.device
            for t, nme, k in ((mean, "mean", M), (rstd, "rstd", M), (gamma, "gamma", W), (dgamma, "dgamma", W), (dbeta, "dbeta", W)):
                _flat(t, nme, dev)
                if t.numel() != k:
                    raise ValueError(f"layernorm_bwd: {nme} length")
            if tuple(dy.shape) != (M, W):
                raise ValueError("layernorm_bwd: shapes")
            dx = torch.empty_like(x)
            ws = torch.empty(self.lib.grappa_layernorm_bwd_workspace_bytes(M, W), dtype=torch.uint8, device=dev)
            row = torch.empty(M, dtype=torch.int32, device=dev) if want else None
            a.M, a.W, a.dy, a.lddy, a.x, a.ldx = M, W, dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0)
            a.mean, a.rstd, a.gamma, a.dx, a.lddx, a.part, a.dx_amax = mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), dx.data_ptr(), dx.stride(0), ws.data_ptr(), _ptr(row)
            self._lnq.append((ws, self.lib.grappa_layernorm_bwd_partial_rows(M), W, dgamma.data_ptr(), dbeta.data_ptr(), dgamma, dbeta, cur, task))
            out.append((dx, Amax(row=row) if want else None))
        _chk(self.lib.grappa_layernorm_bwd_batched_f32(self._stream(), arr, n), "grappa_layernorm_bwd_batched_f32")
        return out

    def act_dropout_bwd_batched(self, items):
        """items: [(dy, y or None, drop_p, drop_seed)] -> [(dz, record or None)] in ONE launch; None: go one by one"""
        n = len(items)
        if not (2 <= n <= _lib.ROW_BATCH_MAX) or not self.wants_amax(True):
            return None
        if not self._rows_ok(*[t for it in items for t in (it[0], it[1])]) or any(it[0].shape[0] == 0 for it in items):
            return None
        arr = (_lib.ActDropoutItem * n)()
        out = []
        for a, (dy, y, p, seed) in zip(arr, items):
            M, N = dy.shape
            if y is not None and tuple(y.shape) != (M, N):
                raise ValueError("act_dropout_bwd: shapes")
            dz = torch.empty_like(dy)
            row = torch.empty(M, dtype=torch.int32, device=dy.device)
            a.M, a.N, a.dy, a.lddy, a.y, a.ldy = M, N, dy.data_ptr(), dy.stride(0), _ptr(y), y.stride(0) if y is not None else 0
            a.drop_p, a.drop_seed, a.dz, a.lddz, a.dz_amax = float(p), int(seed) & (2 ** 64 - 1), dz.data_ptr(), dz.stride(0), row.data_ptr()
            a.drop_salt = self._salt_ptr
            out.append((dz, Amax(row=row)))
        _chk(self.lib.grappa_act_dropout_bwd_batched_f32(self._stream(), arr, n), "grappa_act_dropout_bwd_batched_f32")
        return out

    def seqattn_batched(self, items, backward: bool):
        """forward: items [(qkv, s, T, nheads)] -> [(att, record)]; backward: [(qkv, dout, s, T, nheads)] -> [(dqkv, record)]; ONE launch;
        None: go one by one"""
        n = len(items)
        if not (2 <= n <= _lib.ROW_BATCH_MAX):
            return None
        arr = (_lib.SeqAttnItem * n)()
        out = []
        want = self.wants_amax(backward)
        for a, it in zip(arr, items):
            qkv = it[0]
            s, T, nheads = it[-3:]
            F = qkv.shape[1] // 3
            if qkv.dtype != torch.float32 or not qkv.is_contiguous() or qkv.shape != (s * T, 3 * F) or F % nheads or T == 0:
                return None
            dev = qkv.device
            if backward:
                dout = it[1]
                if dout.dtype != torch.float32 or not dout.is_contiguous() or tuple(dout.shape) != (s * T, F):
                    return None
                res = torch.empty_like(qkv)
                a.dout, a.dqkv = dout.data_ptr(), res.data_ptr()
            else:
                res = torch.empty((s * T, F), dtype=torch.float32, device=dev)
                a.out = res.data_ptr()
            row = torch.empty(s * T, dtype=torch.int32, device=dev) if want else None
            a.s, a.T, a.nheads, a.dh, a.qkv, a.amax = s, T, nheads, F // nheads, qkv.data_ptr(), _ptr(row)
            out.append((res, Amax(row=row) if want else None))
        fn = self.lib.grappa_seqattn_bwd_batched_f32 if backward else self.lib.grappa_seqattn_fwd_batched_f32
        _chk(fn(self._stream(), arr, n), "grappa_seqattn_batched_f32")
        return out

    # ------------------------------------------------------------------ graph
    def _csr_check(self, plan, N, dev):
        if plan.N != N or plan.indptr.device != dev or plan.indptr.dtype != torch.int32 or plan.indptr.numel() != N + 1:
            raise ValueError("graph plan does not match the feature tensor (atoms / device)")
        if plan.indices.numel() != plan.E or plan.rev.numel() != plan.E:
            raise ValueError("graph plan: edge arrays")

    def gat_fwd(self, plan, ft, H, D, out, alpha) -> None:
        dev = out.device
        N = ft.shape[0]
        self._csr_check(plan, N, dev)
        dt = _same_dtype(ft, out)
        _flat(ft, "ft", dev, dt), _flat(out, "out", dev, dt), _flat(alpha, "alpha", dev)
        if ft.shape[1] != H * D or out.shape != ft.shape or alpha.numel() != plan.E * H:
            raise ValueError("gat_fwd: shapes")
        # algorithmic bytes (element size b): one source row per edge + col index, one dst row read + one output row write + indptr per node
        eb = ft.element_size()
        nbytes = plan.E * (H * D * eb + 4) + N * (2 * H * D * eb + 4)
        fn = getattr(self.lib, f"grappa_gat_fwd_{_sfx(ft)}")
        self._timed("gat_fwd", 2.0 * plan.E * H * D * 2, nbytes,
                    lambda: _chk(fn(self._stream(), N, plan.E, H, D, plan.indptr.data_ptr(), plan.indices.data_ptr(),
                                    ft.data_ptr(), out.data_ptr(), alpha.data_ptr()), "grappa_gat_fwd"))

    def gat_bwd(self, plan, ft, out, alpha, dout, H, D, dft) -> None:
        dev = dft.device
        N = ft.shape[0]
        self._csr_check(plan, N, dev)
        dt = _same_dtype(ft, out, dout, dft)
        for t, n in ((ft, "ft"), (out, "out"), (dout, "dout"), (dft, "dft")):
            _flat(t, n, dev, dt)
        _flat(alpha, "alpha", dev)
        if ft.shape[1] != H * D or out.shape != ft.shape or dout.shape != ft.shape or dft.shape != ft.shape or alpha.numel() != plan.E * H:
            raise ValueError("gat_bwd: shapes")
        delta = torch.empty((N, H), dtype=torch.float32, device=dev)
        eb = ft.element_size()
        nbytes = plan.E * (2 * H * D * eb + 4) + N * (4 * H * D * eb + 4)
        fn = getattr(self.lib, f"grappa_gat_bwd_{_sfx(ft)}")
        self._timed("gat_bwd", 2.0 * plan.E * H * D * 5, nbytes,
                    lambda: _chk(fn(self._stream(), N, plan.E, H, D, plan.indptr.data_ptr(), plan.indices.data_ptr(),
                                    plan.rev.data_ptr(), ft.data_ptr(), out.data_ptr(), alpha.data_ptr(), dout.data_ptr(),
                                    dft.data_ptr(), delta.data_ptr()), "grappa_gat_bwd"))

    def neighbor_mean(self, plan, x, out, scale_by_neighbor: bool) -> None:
        dev = out.device
        N, F = x.shape
        self._csr_check(plan, N, dev)
        if out.shape != x.shape:
            raise ValueError("neighbor_mean: shapes")
        if x.dtype == torch.bfloat16:                  # bf16 storage configuration: out bf16 (a product's operand) or fp32 (a product's addend)
            _flat(x, "x", dev, torch.bfloat16), _flat(out, "out", dev, out.dtype)
            if out.dtype not in _ACT_DTYPES:
                raise ValueError("neighbor_mean: out must be bfloat16 or float32")
            _chk(self.lib.grappa_neighbor_mean_bf16(self._stream(), N, F, plan.indptr.data_ptr(), plan.indices.data_ptr(), x.data_ptr(), out.data_ptr(),
                                                    int(out.dtype == torch.float32), int(scale_by_neighbor)), "grappa_neighbor_mean_bf16")
            return
        _flat(x, "x", dev), _flat(out, "out", dev)
        _chk(self.lib.grappa_neighbor_mean_f32(self._stream(), N, F, plan.indptr.data_ptr(), plan.indices.data_ptr(), x.data_ptr(), out.data_ptr(),
                                               int(scale_by_neighbor)), "grappa_neighbor_mean_f32")

    def charge_encoding(self, q, dim, lo, hi, out, col0) -> None:
        dev = out.device
        _flat(q, "q", dev)
        N = q.numel()
        if out.shape[0] != N or col0 + dim > out.shape[1]:
            raise ValueError("charge_encoding: shapes")
        _chk(self.lib.grappa_charge_encoding_f32(self._stream(), N, q.data_ptr(), dim, float(lo), float(hi), out.data_ptr(), _f32_2d(out, "out", dev),
                                                 col0), "grappa_charge_encoding_f32")

    # ------------------------------------------------------------------ tuples
    def tuple_gather_fwd(self, a, idx, s, pe, x) -> None:
        dev = x.device
        T = idx.shape[0]
        W = x.shape[1]
        if idx.dtype != torch.int32 or idx.device != dev or not idx.is_contiguous() or (T and idx.shape[1] != s):
            raise ValueError("tuple_gather_fwd: idx must be contiguous int32 (T,s)")
        if x.shape[0] != s * T or a.shape[1] < W:
            raise ValueError("tuple_gather_fwd: shapes")
        if pe is not None:
            _flat(pe, "pe", dev)
            if pe.numel() != s:
                raise ValueError("tuple_gather_fwd: pe length")
        dt = _same_dtype(a, x)
        fn = getattr(self.lib, f"grappa_tuple_gather_fwd_{_sfx(x)}")
        _chk(fn(self._stream(), T, s, W, a.data_ptr(), _f32_2d(a, "a", dev, dt), idx.data_ptr(), _ptr(pe),
                x.data_ptr(), _f32_2d(x, "x", dev, dt)), "grappa_tuple_gather_fwd")

    def tuple_gather_bwd(self, inv_ptr, inv_rows, dx, da, has_pe: bool, accumulate=False) -> None:
        dev = da.device
        N, W = da.shape[0], dx.shape[1]
        if inv_ptr.dtype != torch.int32 or inv_ptr.numel() != N + 1 or inv_ptr.device != dev or inv_rows.dtype != torch.int32:
            raise ValueError("tuple_gather_bwd: inverse incidence")
        if da.shape[1] < W or inv_rows.numel() != dx.shape[0]:
            raise ValueError("tuple_gather_bwd: shapes")
        dt = _same_dtype(dx, da)
        fn = getattr(self.lib, f"grappa_tuple_gather_bwd_{_sfx(da)}")
        _chk(fn(self._stream(), N, W, inv_ptr.data_ptr(), inv_rows.data_ptr(), dx.data_ptr(),
                _f32_2d(dx, "dx", dev, dt), da.data_ptr(), _f32_2d(da, "da", dev, dt), int(has_pe), int(accumulate)),
             "grappa_tuple_gather_bwd")

    def seqattn_fwd(self, qkv, s, T, nheads, out, amax=None, pairs=False):
        if pairs:               # the output in the pair format ONLY (`out` is not written and may be None): inference
            dev = qkv.device
            F = qkv.shape[1] // 3
            if qkv.dtype != torch.float32 or qkv.shape != (s * T, 3 * F) or F % nheads or F % 32 or F > 512:
                raise ValueError("seqattn_fwd: the pair format needs float32 q, k, v of F % 32 == 0, F <= 512 columns each")
            _flat(qkv, "qkv", dev)
            row = torch.empty(s * T, dtype=torch.int32, device=dev)
            pr = torch.empty((s * T, 2 * F), dtype=torch.float16, device=dev)
            if T:
                _chk(self.lib.grappa_seqattn_fwd_pairs_f32(self._stream(), s, T, nheads, F // nheads, qkv.data_ptr(), pr.data_ptr(), pr.stride(0),
                                                           row.data_ptr()), "grappa_seqattn_fwd_pairs_f32")
            return Amax(row=row, pairs=pr)
        dev = out.device
        dt = _same_dtype(qkv, out)
        _flat(qkv, "qkv", dev, dt), _flat(out, "out", dev, dt)
        F = out.shape[1]
        if qkv.shape != (s * T, 3 * F) or out.shape[0] != s * T or F % nheads:
            raise ValueError("seqattn_fwd: shapes")
        row = self._new_row_amax(out, False, amax)
        args = (self._stream(), s, T, nheads, F // nheads, qkv.data_ptr(), out.data_ptr())
        if row is not None:
            _chk(self.lib.grappa_seqattn_fwd_amax_f32(*args, row.data_ptr()), "grappa_seqattn_fwd_amax_f32")
            return Amax(row=row)
        _chk(getattr(self.lib, f"grappa_seqattn_fwd_{_sfx(out)}")(*args), "grappa_seqattn_fwd")
        return None

    # ------------------------------------------------------------------ the fused writer-head layer (C ABI 11)
    def _packed_weight(self, w: torch.Tensor, transposed: bool = False) -> torch.Tensor:
        """W (out x in features) -- or W^T -- as bf16 in the MFMA fragment order of the fused writer layer (grappa_writer_pack_weight);
        cached per weight and refreshed when the weight changed, like _planes_of_weight"""
        R, Cc = w.shape
        key = (w.data_ptr(), R, Cc, "packT" if transposed else "pack")
        ver = (w._version, self._wepoch)
        hit = self._wplanes.get(key)
        if hit is not None and hit[2]() is not w:
            hit = None
        if hit is not None and hit[0] == ver:
            return hit[1]
        N, K = (Cc, R) if transposed else (R, Cc)
        if hit is not None:
            pk = hit[1]
        else:
            pk = torch.empty(N * K, dtype=torch.bfloat16, device=w.device)
            for k in [k for k, e in self._wplanes.items() if e[2]() is None]:
                del self._wplanes[k]
        _chk(self.lib.grappa_writer_pack_weight(self._stream(), N, K, w.data_ptr(), _f32_2d(w, "W", w.device), int(transposed), _lib.WRITER_BF16,
                                                pk.data_ptr()), "grappa_writer_pack_weight")
        self._wplanes[key] = (ver, pk, weakref.ref(w))
        return pk

    def writer_layer_ok(self, x: torch.Tensor, s: int, nheads: int, *params) -> bool:
        """can grappa_writer_head_fwd run this transformer layer?  (bf16 storage configuration, 512 features, 8 heads, tuples of 2 - 4 tokens,
        both LayerNorms present)"""
        if not self.fused_writer_layer or x.dtype != torch.bfloat16 or x.dim() != 2 or x.shape[1] != 512 or nheads != 8 or s not in (2, 3, 4):
            return False
        if x.shape[0] == 0:
            return False
        return all(p is not None and p.dtype == torch.float32 and p.is_contiguous() for p in params)

    def writer_layer_fwd(self, x, s, T, nheads, drop_p, seed1, seed2, n1_w, n1_b, w_in, b_in, w_o, b_o, nf_w, nf_b, w1, b1, w2, b2, out, save=None, gather=None):
        """out = the transformer layer of a writer head applied to the token table x (s*T, 512) in ONE launch (include/grappa_hip.h
        grappa_writer_head_fwd; reference models/network_utils.py:112-133, :44-54).  save: None (inference) or the tensors the unfused
        backward reads, written as by-products: dict(mean1, rstd1, x1, qkv, att, x2, meanf, rstdf, x3, u)."""
        dev = out.device
        M, Fd = out.shape
        d = _lib.WriterLayerDesc()
        d.s, d.T, d.F, d.nheads, d.dtype = s, T, Fd, nheads, _lib.WRITER_BF16
        _flat(out, "out", dev, torch.bfloat16)
        d.out = out.data_ptr()
        if gather is not None:
            # gather = (idx_tab (T, s) int32: table row of every token, x1_tab (rows, F): normalised table rows, qkv_tab (rows, 3F)): the first layer of a
            # head on (atom, position) rows -- x, the first LayerNorm and the q | k | v product are not part of the launch
            idx_tab, x1_tab, qkv_tab = gather
            if M != s * T or idx_tab.dtype != torch.int32 or tuple(idx_tab.shape) != (T, s) or not idx_tab.is_contiguous() or idx_tab.device != dev:
                raise ValueError("writer_layer_fwd: gather index")
            _flat(x1_tab, "x1_tab", dev, torch.bfloat16), _flat(qkv_tab, "qkv_tab", dev, torch.bfloat16)
            if x1_tab.shape[1] != Fd or tuple(qkv_tab.shape) != (x1_tab.shape[0], 3 * Fd):
                raise ValueError("writer_layer_fwd: gather tables")
            d.gather_idx, d.x1_tab, d.qkv_tab = idx_tab.data_ptr(), x1_tab.data_ptr(), qkv_tab.data_ptr()
        else:
            if tuple(x.shape) != (M, Fd) or M != s * T or out.dtype != x.dtype:
                raise ValueError("writer_layer_fwd: shapes")
            _flat(x, "x", dev, torch.bfloat16)
            d.x = x.data_ptr()
        for name, wt, shape in (("w_in_pk", w_in, (3 * Fd, Fd)), ("w_o_pk", w_o, (Fd, Fd)), ("w1_pk", w1, (Fd, Fd)), ("w2_pk", w2, (Fd, Fd))):
            if gather is not None and name == "w_in_pk":
                continue
            if tuple(wt.shape) != shape:
                raise ValueError(f"writer_layer_fwd: {name[:-3]} has shape {tuple(wt.shape)}, expected {shape}")
            setattr(d, name, self._packed_weight(wt).data_ptr())
        for name, v, n in (("b_in", b_in, 3 * Fd), ("b_o", b_o, Fd), ("b1", b1, Fd), ("b2", b2, Fd), ("n1_gamma", n1_w, Fd), ("n1_beta", n1_b, Fd),
                           ("nf_gamma", nf_w, Fd), ("nf_beta", nf_b, Fd)):
            if gather is not None and name.startswith("n1_"):
                continue
            _flat(v, name, dev)
            if v.numel() != n:
                raise ValueError(f"writer_layer_fwd: {name} length")
            setattr(d, name, v.data_ptr())
        d.drop_p, d.seed1, d.seed2 = float(drop_p), int(seed1) & (2 ** 64 - 1), int(seed2) & (2 ** 64 - 1)
        d.drop_salt = self._salt_ptr
        keep = None
        if save is not None:
            for name in ("meanf", "rstdf") if gather is not None else ("mean1", "rstd1", "meanf", "rstdf"):
                t = save[name]
                _flat(t, name, dev)
                if t.numel() != M:
                    raise ValueError(f"writer_layer_fwd: save[{name!r}] length")
                setattr(d, "save_" + name, t.data_ptr())
            tiles = self.lib.grappa_writer_head_tiles(s, T)
            d.x2_tiled = int(bool(save.get("x2_tiled", False)))
            for name in ("att", "x2", "x3", "u") if gather is not None else ("x1", "qkv", "att", "x2", "x3", "u"):
                t = save[name]
                _flat(t, name, dev, torch.bfloat16)
                want = (tiles * 64, Fd) if (name == "x2" and d.x2_tiled) else (M, 3 * Fd if name == "qkv" else Fd)
                if tuple(t.shape) != want:
                    raise ValueError(f"writer_layer_fwd: save[{name!r}] shape {tuple(t.shape)}, expected {want}")
                setattr(d, "save_" + name, t.data_ptr())
            keep = save
        nprod = 3 if gather is not None else 6
        flops = 2.0 * M * Fd * (nprod * Fd) + 4.0 * M * s * Fd
        nbytes = 2.0 * M * Fd * ((5 if gather is not None else 2) + ((5 if gather is not None else 9) if save is not None else 0)) + 2.0 * nprod * Fd * Fd
        self._timed("writer_layer", flops, nbytes, lambda: _chk(self.lib.grappa_writer_head_fwd(self._stream(), C.byref(d)), "grappa_writer_head_fwd"),
                    lambda: [{"M": M, "s": s, "save": keep is not None, "gather": gather is not None}])

    def writer_layer_bwd(self, dout, x, s, T, nheads, drop_p, seed1, seed2, saved, n1_w, n1_b, w_in, w_o, nf_w, nf_b, w1, w2, gather=None):
        """the input-gradient chain of the fused layer in ONE launch (grappa_writer_head_bwd) -> (dx, dz2, dz1, dzo, dqkv): dx = the gradient of
        the layer's input, the other four = the operands of the weight-gradient products (against saved u, x3, att, x1).  The LayerNorm
        parameter gradients (per-tile partials) are queued for the end-of-pass reduction like layernorm_bwd's, or reduced at once.
        saved: the dict `writer_layer_fwd` filled."""
        dev = dout.device
        M, Fd = dout.shape
        bf = torch.bfloat16
        if M != s * T:
            raise ValueError("writer_layer_bwd: shapes")
        d = _lib.WriterLayerBwdDesc()
        d.s, d.T, d.F, d.nheads, d.dtype = s, T, Fd, nheads, _lib.WRITER_BF16
        ntiles = self.lib.grappa_writer_head_tiles(s, T)
        d.x2_tiled = int(bool(saved.get("x2_tiled", False)))
        if gather is not None:
            # gather = (idx_tab (T, s) int32, qkv_tab (rows, 3F)): q | k | v from the table; returns dx2 (the skip branch's gradient) in place of dx
            idx_tab, qkv_tab = gather
            if idx_tab.dtype != torch.int32 or tuple(idx_tab.shape) != (T, s) or not idx_tab.is_contiguous() or qkv_tab.shape[1] != 3 * Fd:
                raise ValueError("writer_layer_bwd: gather arguments")
            d.gather_idx = idx_tab.data_ptr()
            tensors = (("dout", dout, (M, Fd)), ("qkv", qkv_tab, tuple(qkv_tab.shape)), ("x2", saved["x2"], (ntiles * 64, Fd) if d.x2_tiled else (M, Fd)),
                       ("u", saved["u"], (M, Fd)))
        else:
            tensors = (("dout", dout, (M, Fd)), ("x", x, (M, Fd)), ("qkv", saved["qkv"], (M, 3 * Fd)),
                       ("x2", saved["x2"], (ntiles * 64, Fd) if d.x2_tiled else (M, Fd)), ("u", saved["u"], (M, Fd)))
        for name, t, shape in tensors:
            _flat(t, name, dev, bf)
            if tuple(t.shape) != shape:
                raise ValueError(f"writer_layer_bwd: {name} shape")
            setattr(d, name, t.data_ptr())
        for name in ("meanf", "rstdf") if gather is not None else ("mean1", "rstd1", "meanf", "rstdf"):
            _flat(saved[name], name, dev)
            setattr(d, name, saved[name].data_ptr())
        _flat(nf_w, "nf_gamma", dev)
        d.nf_gamma = nf_w.data_ptr()
        if gather is None:
            _flat(n1_w, "n1_gamma", dev)
            d.n1_gamma = n1_w.data_ptr()
        for name, wt in (("w_in_tpk", w_in), ("w_o_tpk", w_o), ("w1_tpk", w1), ("w2_tpk", w2)):
            if gather is not None and name == "w_in_tpk":
                continue
            setattr(d, name, self._packed_weight(wt, transposed=True).data_ptr())
        d.drop_p, d.seed1, d.seed2, d.drop_salt = float(drop_p), int(seed1) & (2 ** 64 - 1), int(seed2) & (2 ** 64 - 1), self._salt_ptr
        new = lambda *sh: torch.empty(sh, dtype=bf, device=dev)      # noqa: E731
        dx, dz2, dz1, dzo, dqkv = new(M, Fd), new(M, Fd), new(M, Fd), new(M, Fd), new(M, 3 * Fd)
        parts = torch.empty((2, ntiles, 2, Fd), dtype=torch.float32, device=dev)
        d.dx, d.dz2, d.dz1, d.dzo, d.dqkv = dx.data_ptr(), dz2.data_ptr(), dz1.data_ptr(), dzo.data_ptr(), dqkv.data_ptr()
        d.ln1_part, d.lnf_part = parts[0].data_ptr(), parts[1].data_ptr()
        nprod = 3 if gather is not None else 6
        flops = 2.0 * M * Fd * (nprod * Fd) + 8.0 * M * s * Fd
        self._timed("writer_layer", flops, 2.0 * M * Fd * 15 + 2.0 * nprod * Fd * Fd,
                    lambda: _chk(self.lib.grappa_writer_head_bwd(self._stream(), C.byref(d)), "grappa_writer_head_bwd"),
                    lambda: [{"M": M, "s": s, "bwd": True, "gather": gather is not None}])
        for part, g, b in ((parts[1], nf_w, nf_b),) if gather is not None else ((parts[0], n1_w, n1_b), (parts[1], nf_w, nf_b)):
            self._reduce_ln_partials(part, ntiles, Fd, g, b)
        return dx, dz2, dz1, dzo, dqkv

    def _reduce_ln_partials(self, part, nrows, W, gamma, beta) -> None:
        """per-block partial sums [nrows][dgamma | dbeta][W] of a LayerNorm's parameter gradients -> accumulated into the parameters' gradient
        buffers: with the other LayerNorms' at the end of the backward pass where that queue is open (layernorm_bwd), else now"""
        from .ops import _pgrad
        dg = _pgrad(gamma) if gamma.requires_grad else torch.zeros_like(gamma)
        db = _pgrad(beta) if beta.requires_grad else torch.zeros_like(beta)
        task = self._queue_flush() if (self.defer_wgrads and self.defer_ln) else -1
        if task >= 0 and all(q[3] != dg.data_ptr() for q in self._lnq if q[8] == task):
            self._lnq.append((part, nrows, W, dg.data_ptr(), db.data_ptr(), dg, db, torch.cuda.current_stream(), task))
            return
        arr = (_lib.ColsumItem * 1)()
        a = arr[0]
        a.part, a.nrows, a.n, a.out, a.out2, a.n_first, a.accumulate = part.data_ptr(), nrows, 2 * W, dg.data_ptr(), db.data_ptr(), W, 1
        _chk(self.lib.grappa_colsum_partials_batched(self._stream(), arr, 1), "grappa_colsum_partials_batched")

    def seqattn_bwd(self, qkv, dout, s, T, nheads, dqkv, amax=None):
        dev = dqkv.device
        dt = _same_dtype(qkv, dout, dqkv)
        _flat(qkv, "qkv", dev, dt), _flat(dout, "dout", dev, dt), _flat(dqkv, "dqkv", dev, dt)
        F = dout.shape[1]
        if qkv.shape != (s * T, 3 * F) or dqkv.shape != qkv.shape or dout.shape[0] != s * T:
            raise ValueError("seqattn_bwd: shapes")
        row = self._new_row_amax(dqkv, True, amax)
        args = (self._stream(), s, T, nheads, F // nheads, qkv.data_ptr(), dout.data_ptr(), dqkv.data_ptr())
        if row is not None:
            _chk(self.lib.grappa_seqattn_bwd_amax_f32(*args, row.data_ptr()), "grappa_seqattn_bwd_amax_f32")
            return Amax(row=row)
        _chk(getattr(self.lib, f"grappa_seqattn_bwd_{_sfx(dqkv)}")(*args), "grappa_seqattn_bwd")
        return None

    def perm_concat_fwd(self, x, s, T, perms: Sequence[Sequence[int]], z) -> None:
        dev = z.device
        dt = _same_dtype(x, z)
        _flat(x, "x", dev, dt), _flat(z, "z", dev, dt)
        F, P = x.shape[1], len(perms)
        if x.shape[0] != s * T or z.shape != (P * T, s * F):
            raise ValueError("perm_concat_fwd: shapes")
        arr = (C.c_int * (P * s))(*[int(v) for p in perms for v in p])
        fn = getattr(self.lib, f"grappa_perm_concat_fwd_{_sfx(z)}")
        _chk(fn(self._stream(), s, T, F, P, arr, x.data_ptr(), z.data_ptr()), "grappa_perm_concat_fwd")

    def perm_concat_bwd(self, dz, s, T, perms, dx) -> None:
        dev = dx.device
        dt = _same_dtype(dz, dx)
        _flat(dz, "dz", dev, dt), _flat(dx, "dx", dev, dt)
        F, P = dx.shape[1], len(perms)
        if dx.shape[0] != s * T or dz.shape != (P * T, s * F):
            raise ValueError("perm_concat_bwd: shapes")
        arr = (C.c_int * (P * s))(*[int(v) for p in perms for v in p])
        fn = getattr(self.lib, f"grappa_perm_concat_bwd_{_sfx(dx)}")
        _chk(fn(self._stream(), s, T, F, P, arr, dz.data_ptr(), dx.data_ptr()), "grappa_perm_concat_bwd")

    def param_out_fwd(self, kind, o, T, P, n_per, gated, cutoff, consts, k, eq) -> None:
        dev = k.device
        _flat(o, "o", dev), _flat(consts, "consts", dev), _flat(k, "k", dev)
        if o.shape[0] != P * T:
            raise ValueError("param_out_fwd: shapes")
        if eq is not None:
            _flat(eq, "eq", dev)
        _chk(self.lib.grappa_param_out_fwd_f32(self._stream(), kind, T, P, n_per, int(gated), float(cutoff), o.data_ptr(), o.shape[1],
                                               consts.data_ptr(), k.data_ptr(), _ptr(eq)), "grappa_param_out_fwd_f32")

    def param_out_bwd(self, kind, o, T, P, n_per, gated, cutoff, consts, dk, deq, d_o) -> None:
        dev = d_o.device
        _flat(o, "o", dev), _flat(consts, "consts", dev), _flat(d_o, "d_o", dev)
        for t, n in ((dk, "dk"), (deq, "deq")):
            if t is not None:
                _flat(t, n, dev)
        if o.shape[0] != P * T or d_o.shape != o.shape:
            raise ValueError("param_out_bwd: shapes")
        _chk(self.lib.grappa_param_out_bwd_f32(self._stream(), kind, T, P, n_per, int(gated), float(cutoff), o.data_ptr(), o.shape[1],
                                               consts.data_ptr(), _ptr(dk), _ptr(deq), d_o.data_ptr()), "grappa_param_out_bwd_f32")

    def param_out_bwd_stats(self, kind, o, T, P, n_per, gated, cutoff, consts, dk, deq, d_consts) -> None:
        """dL/d(statistics of the output map) (learnable_statistics=True)"""
        dev = d_consts.device
        _flat(o, "o", dev), _flat(consts, "consts", dev), _flat(d_consts, "d_consts", dev)
        if o.shape[0] != P * T or d_consts.numel() != consts.numel():
            raise ValueError("param_out_bwd_stats: shapes")
        ws = self._workspace(self.lib.grappa_param_out_stats_workspace_bytes(T), dev)
        _chk(self.lib.grappa_param_out_bwd_stats_f32(self._stream(), kind, T, P, n_per, int(gated), float(cutoff), _ptr(o), o.shape[1] if o.dim() == 2 else 0,
                                                     consts.data_ptr(), _ptr(dk), _ptr(deq), d_consts.data_ptr(), ws.data_ptr(), ws.numel()),
             "grappa_param_out_bwd_stats_f32")

    # ------------------------------------------------------------------ MM energy
    def _mm_desc(self, plan, xyz, ks, eqs, n_per, offset_torsion):
        from .constants import TUPLE_LEVELS
        dev = xyz.device
        _flat(xyz, "xyz", dev)
        N, Cc = xyz.shape[0], xyz.shape[1]
        if N != plan.N or xyz.shape[2] != 3 or plan.indptr.device != dev:
            raise ValueError("mm: xyz does not match the batch plan")
        d = _lib.MMDesc()
        d.N, d.C, d.B = N, Cc, plan.B
        d.xyz = xyz.data_ptr()
        for l, lvl in enumerate(TUPLE_LEVELS):
            T = plan.T[lvl]
            d.T[l] = T
            d.idx[l] = plan.idx32[lvl].data_ptr()
            d.mol_ptr[l] = plan.mol_ptr[lvl].data_ptr()
            k = ks[l]
            _flat(k, f"k[{lvl}]", dev)
            if l < 2:
                if k.numel() != T:
                    raise ValueError(f"mm: k[{lvl}] length")
                _flat(eqs[l], f"eq[{lvl}]", dev)
                if eqs[l].numel() != T:
                    raise ValueError(f"mm: eq[{lvl}] length")
                d.eq[l] = eqs[l].data_ptr()
                d.n_per[l] = 0
            else:
                if k.numel() != T * n_per[l]:
                    raise ValueError(f"mm: k[{lvl}] must be (T,{n_per[l]})")
                d.n_per[l] = n_per[l]
            d.k[l] = k.data_ptr()
        d.offset_torsion = int(offset_torsion)
        d.inc_ptr, d.inc_code, d.atom_molptr = plan.inc_ptr.data_ptr(), plan.inc_code.data_ptr(), plan.atom_molptr.data_ptr()
        return d

    def mm_energy_fwd(self, plan, xyz, ks, eqs, n_per, offset_torsion, energy, term_energy, tuple_e=None, tuple_x=None) -> None:
        d = self._mm_desc(plan, xyz, ks, eqs, n_per, offset_torsion)
        te = _lib.VP4(*[_ptr(t) for t in (tuple_e or [None] * 4)])
        tx = _lib.VP4(*[_ptr(t) for t in (tuple_x or [None] * 4)])
        _chk(self.lib.grappa_mm_energy_fwd_f32(self._stream(), C.byref(d), energy.data_ptr(), _ptr(term_energy), C.byref(te), C.byref(tx)),
             "grappa_mm_energy_fwd_f32")

    def mm_gradient_fwd(self, plan, xyz, ks, eqs, n_per, grad) -> None:
        d = self._mm_desc(plan, xyz, ks, eqs, n_per, False)
        _flat(grad, "grad", xyz.device)
        if grad.shape != xyz.shape:
            raise ValueError("mm_gradient_fwd: grad shape")
        _chk(self.lib.grappa_mm_gradient_fwd_f32(self._stream(), C.byref(d), grad.data_ptr()), "grappa_mm_gradient_fwd_f32")

    def mm_bwd(self, plan, xyz, ks, eqs, n_per, offset_torsion, gE, gG, gks, geqs) -> None:
        d = self._mm_desc(plan, xyz, ks, eqs, n_per, offset_torsion)
        for t, n in ((gE, "gE"), (gG, "gG")):
            if t is not None:
                _flat(t, n, xyz.device)
        a = _lib.VP4(*[_ptr(t) for t in gks])
        b = _lib.VP4(*[_ptr(t) for t in geqs])
        _chk(self.lib.grappa_mm_bwd_f32(self._stream(), C.byref(d), _ptr(gE), _ptr(gG), C.byref(a), C.byref(b)), "grappa_mm_bwd_f32")

    # ------------------------------------------------------------------ loss
    def loss_ef(self, plan, energy, energy_ref, is_dummy, grad, grad_ref, wE, wG, inv_B, loss_mol, gE, gG) -> None:
        dev = loss_mol.device
        B = _loss_mols(plan)
        Cc = energy.shape[1] if energy is not None else grad.shape[1]
        for t, n in ((energy, "energy"), (energy_ref, "energy_ref"), (is_dummy, "is_dummy"), (grad, "grad"), (grad_ref, "grad_ref"),
                     (gE, "gE"), (gG, "gG"), (loss_mol, "loss_mol")):
            if t is not None:
                _flat(t, n, dev)
        _chk(self.lib.grappa_loss_ef_fwd_bwd_f32(self._stream(), B, Cc, plan.N, plan.atom_molptr.data_ptr(), _ptr(energy), _ptr(energy_ref),
                                                 _ptr(is_dummy), _ptr(grad), _ptr(grad_ref), float(wE), float(wG), float(inv_B),
                                                 loss_mol.data_ptr(), _ptr(gE), _ptr(gG)), "grappa_loss_ef_fwd_bwd_f32")

    def collate_gather(self, tables, B: int) -> None:
        """one launch for all tables of a batch (include/grappa_hip.h grappa_collate_batch).  tables: dicts with the tensors
        src, dst (4-byte element types), src_row (B,) int64, dst_row (B+1,) int64, optional p0 / p1 (int32), and width, mode, c0."""
        if not tables or B == 0:
            return
        dev = tables[0]["dst"].device
        arr = (_lib.CollateDesc * len(tables))()
        for d, t in zip(arr, tables):
            for k in ("src", "dst"):
                if t[k].element_size() not in (4, 8) or t[k].device != dev or not t[k].is_contiguous():
                    raise ValueError(f"collate table {k}: expected a contiguous 4- or 8-byte tensor on {dev}")
            for k, dt in (("src_row", torch.int64), ("dst_row", torch.int64), ("p0", torch.int32), ("p1", torch.int32)):
                v = t.get(k)
                if v is not None and (v.dtype != dt or v.device != dev or not v.is_contiguous()):
                    raise ValueError(f"collate table {k}: expected a contiguous {dt} tensor on {dev}")
            d.src, d.dst = t["src"].data_ptr(), t["dst"].data_ptr()
            d.src_row, d.dst_row = t["src_row"].data_ptr(), t["dst_row"].data_ptr()
            d.p0, d.p1 = _ptr(t.get("p0")), _ptr(t.get("p1"))
            d.c0, d.width, d.mode = int(t.get("c0", 0)), int(t["width"]), _lib.COLLATE_MODES[t["mode"]]
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        if dev.type == "cuda":
            host = host.pin_memory()       # (a pageable upload waits for everything queued on the stream before the host may go on)
        descs = host.to(dev, non_blocking=True)
        _chk(self.lib.grappa_collate_batch(self._stream(), descs.data_ptr(), arr, len(tables), int(B)), "grappa_collate_batch")
        self._keep = (descs, host)         # the copy and the kernel read them asynchronously: keep them alive until the next call

    def eval_se(self, plan, energy, energy_ref, is_dummy, grad, grad_ref, out) -> None:
        """out (B,4) = per molecule {se_E, n_E, se_G, n_G} (include/grappa_hip.h grappa_eval_se_f32)"""
        dev = out.device
        for t, n in ((energy, "energy"), (energy_ref, "energy_ref"), (is_dummy, "is_dummy"), (grad, "grad"), (grad_ref, "grad_ref"), (out, "out")):
            if t is not None:
                _flat(t, n, dev)
        # (a batch that ends in a padding molecule: the kernel is one workgroup per molecule -- run over the real ones, the caller zeroes `out`)
        _chk(self.lib.grappa_eval_se_f32(self._stream(), _loss_mols(plan), energy.shape[1], plan.N, plan.atom_molptr.data_ptr(), energy.data_ptr(),
                                         energy_ref.data_ptr(), _ptr(is_dummy), _ptr(grad), _ptr(grad_ref), out.data_ptr()), "grappa_eval_se_f32")

    def loss_param(self, plan, params, refs, fac, reg, pw, inv_B, loss_mol, gps) -> None:
        """params/refs/gps: lists of 6 tensors-or-None in the order n2_k, n2_eq, n3_k, n3_eq, n4_k, n4_improper_k."""
        dev = loss_mol.device
        lv = ["n2", "n2", "n3", "n3", "n4", "n4_improper"]
        d = _lib.PLossDesc()
        d.B = _loss_mols(plan)
        for l in range(6):
            p = params[l]
            d.mol_ptr[l] = plan.mol_ptr[lv[l]].data_ptr()
            if p is None:
                continue
            _flat(p, f"p[{l}]", dev)
            T = plan.T[lv[l]]
            w = p.numel() // T if T else (p.shape[1] if p.dim() == 2 else 1)
            d.p[l], d.width[l] = p.data_ptr(), max(w, 1)
            if refs[l] is not None:
                _flat(refs[l], f"ref[{l}]", dev)
                d.ref[l] = refs[l].data_ptr()
                d.ref_width[l] = max(refs[l].numel() // T if T else 1, 1)
            d.fac[l], d.reg[l] = float(fac[l]), float(reg[l])
        if pw is not None:
            _flat(pw, "pw", dev)
            d.pw = pw.data_ptr()
        d.inv_B = float(inv_B)
        g = _lib.VP6(*[_ptr(t) for t in gps])
        _chk(self.lib.grappa_loss_param_fwd_bwd_f32(self._stream(), C.byref(d), loss_mol.data_ptr(), C.byref(g)), "grappa_loss_param_fwd_bwd_f32")

    # ------------------------------------------------------------------ optimiser
    def sumsq(self, x, out, accumulate=False) -> None:
        dev = out.device
        _flat(x, "x", dev)
        ws = self._workspace(self.lib.grappa_sumsq_workspace_bytes(x.numel()), dev)
        _chk(self.lib.grappa_sumsq_f32(self._stream(), x.numel(), x.data_ptr(), out.data_ptr(), int(accumulate), ws.data_ptr(), ws.numel()),
             "grappa_sumsq_f32")

    def adam_step_dyn(self, p, g, m, v, lr_t, beta1, beta2, eps, weight_decay, step_t, grad_scale, sumsq, max_norm) -> None:
        """adam_step with the learning rate (float32 tensor of one element) and the step count (int32) read from device memory"""
        dev = p.device
        for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
            _flat(t, n, dev)
        _flat(lr_t, "lr", dev), _flat(step_t, "step", dev, torch.int32)
        _chk(self.lib.grappa_adam_step_dyn_f32(self._stream(), p.numel(), p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), lr_t.data_ptr(),
                                               float(beta1), float(beta2), float(eps), float(weight_decay), step_t.data_ptr(), float(grad_scale),
                                               _ptr(sumsq), float(max_norm)), "grappa_adam_step_dyn_f32")

    # ---- dropout salt (include/grappa_hip.h grappa_gemm_desc.drop_salt, C ABI 10): one 64-bit word of device memory mixed into every dropout
    # seed.  Nothing process-wide in the library: while enabled, this backend passes the word's address with every call that draws a mask;
    # a recorded graph keeps the address it was recorded with.
    def enable_dropout_salt(self, device=None) -> None:
        if getattr(self, "_salt", None) is None or (device is not None and self._salt.device != torch.device(device)):
            self._salt = torch.zeros(1, dtype=torch.int64, device=device if device is not None else "cuda")
        self._salt_ptr = self._salt.data_ptr()

    def disable_dropout_salt(self) -> None:
        """back to the seeds as given (calls made afterwards; recorded graphs keep reading the word they were recorded with)"""
        self._salt_ptr = None

    def bump_dropout_salt(self) -> None:
        """+1 on the device word (a kernel on the current stream: inside a capture it becomes a node of the graph)"""
        self._salt.add_(1)

    def adam_step(self, p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale, sumsq, max_norm) -> None:
        dev = p.device
        for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
            _flat(t, n, dev)
        _chk(self.lib.grappa_adam_step_f32(self._stream(), p.numel(), p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), float(lr),
                                           float(beta1), float(beta2), float(eps), float(weight_decay), int(step), float(grad_scale),
                                           _ptr(sumsq), float(max_norm)), "grappa_adam_step_f32")
