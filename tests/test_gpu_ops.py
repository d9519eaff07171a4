"""GPU (-m gpu): every entry point of the C ABI (through grappa_amd.backend.HipBackend, i.e. libgrappa_hip.so on
a real MI355X) against the oracle's restatement of the same op (oracle/ops_ref.py, CPU) on identical seeded inputs.
Tolerances are written next to each check; integer/index outputs must match exactly."""
import math
import os

import numpy as np
import pytest
import torch

import golden_utils as gu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from grappa_amd.backend import HipBackend
    return HipBackend()


@pytest.fixture(scope="module")
def ref():
    from oracle.ops_ref import RefBackend
    return RefBackend()


REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "op_errors.txt")


def _cmp(a_gpu, b_cpu, tol, what, floor_frac=1.0):
    """max|a-b| / max(|b|, floor_frac*max|b|) < tol.  Default floor_frac=1: error relative to the tensor's scale (the
    inputs are random normals, so individual elements are arbitrarily close to zero and a dot product of length K carries
    ~sqrt(K)*2^-24 of absolute rounding noise of that scale in ANY fp32 summation order)."""
    a = a_gpu.detach().cpu().numpy()
    b = b_cpu.detach().numpy()
    assert a.shape == b.shape, what
    assert np.isfinite(a).all(), f"{what}: non-finite values from the HIP kernel"
    err = gu.rel_err_scaled(a, b, floor_frac)
    err_el = gu.rel_err_scaled(a, b, 1e-2)
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        with open(REPORT, "a") as f:
            f.write(f"{what}: normwise {err:.3e} elementwise(floor 1% of max) {err_el:.3e} tol {tol}\n")
    except OSError:
        pass
    assert err < tol, f"{what}: rel err {err:.3e} >= {tol}"


def _graph(n_mols=6, n_confs=5, seed=0, start=300):
    """a batched graph from the committed molecule pool (CPU) and its CUDA copy"""
    from grappa_amd.datasets import build_batch_from_pool
    g = build_batch_from_pool(list(range(start, start + n_mols)), n_confs=n_confs, seed=seed)
    return g, g.to("cuda")


GEMM_CASES = [
    # M, N, K, a_kcontig, b_kcontig   (forward / dgrad / wgrad layouts, ragged edges, skinny N, split-K)
    (300, 512, 512, 1, 1), (257, 511, 256, 1, 1), (1000, 2048, 512, 1, 1), (130, 64, 85, 1, 1), (4096, 12, 256, 1, 1),
    (300, 512, 2048, 1, 0), (777, 256, 6, 1, 0), (129, 85, 64, 1, 0),
    (512, 512, 5000, 0, 0), (2048, 512, 3000, 0, 0), (12, 256, 4097, 0, 0), (64, 85, 333, 0, 0), (511, 256, 1500, 0, 0),
]


@pytest.mark.parametrize("M,N,K,ak,bk", GEMM_CASES)
def test_gemm_layouts(hip, ref, M, N, K, ak, bk):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn((M, K) if ak else (K, M), generator=g)
    B = torch.randn((N, K) if bk else (K, N), generator=g)
    out_r = torch.empty(M, N)
    ref.gemm(A, B, out_r, M=M, N=N, K=K, a_kcontig=bool(ak), b_kcontig=bool(bk))
    out_h = torch.empty(M, N, device="cuda")
    hip.gemm(A.cuda(), B.cuda(), out_h, M=M, N=N, K=K, a_kcontig=bool(ak), b_kcontig=bool(bk))
    torch.cuda.synchronize()
    # fp32 MFMA is an exact fmaf chain: error ~ 1e-7 * sqrt(K) relative to the row/column scale
    _cmp(out_h, out_r, 2e-5, f"gemm {M}x{N}x{K} ak={ak} bk={bk}")


@pytest.mark.parametrize("precision", ["f32", "f32_bf16x9", "f32_bf16x6"])
def test_gemm_epilogues(hip, ref, precision):
    g = torch.Generator().manual_seed(1)
    M, N, K = 333, 192, 160
    A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K)
    bias, res, aux, pre = torch.randn(N, generator=g), torch.randn(M, N, generator=g), torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
    cases = [dict(bias=bias, act=1), dict(bias=bias, res=res, drop_p=0.3, drop_seed=12345), dict(aux=aux),
             dict(bias=bias, act=1, drop_p=0.5, drop_seed=99, res=res, out2=True), dict(pre=pre, bias=bias, act=1, res=res, out2=True),
             dict(accumulate=True, bias=bias)]
    for kw in cases:
        kw = dict(kw)
        two = kw.pop("out2", False)
        o_r, o_h = torch.full((M, N), 0.5), torch.full((M, N), 0.5, device="cuda")
        o2_r, o2_h = (torch.zeros(M, N), torch.zeros(M, N, device="cuda")) if two else (None, None)
        ref.gemm(A, B, o_r, M=M, N=N, K=K, out2=o2_r, **kw)
        kw_h = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in kw.items()}
        hip.gemm(A.cuda(), B.cuda(), o_h, M=M, N=N, K=K, out2=o2_h, precision=precision, **kw_h)
        torch.cuda.synchronize()
        _cmp(o_h, o_r, 2e-5, f"gemm[{precision}] epilogue {sorted(kw)}")
        if two:
            _cmp(o2_h, o2_r, 2e-5, f"gemm epilogue out2 {sorted(kw)}")
            # the dropout mask itself must be the documented counter hash: identical zero pattern
            assert torch.equal((o2_h.cpu() - res) == 0, (o2_r - res) == 0) or kw.get("drop_p", 0) == 0


# tolerance of each arithmetic mode of the GEMM (include/grappa_hip.h GRAPPA_GEMM_*) against a float64 product, relative to max|C|
GEMM_MODE_TOL = {"f32": 1e-5, "f32_bf16x9": 1e-5, "f32_bf16x6": 1e-5, "f32_f16x3": 1e-5, "bf16x3": 1e-4, "bf16": 2e-2}


@pytest.mark.parametrize("precision", list(GEMM_MODE_TOL))
@pytest.mark.parametrize("M,N,K,ak,bk", GEMM_CASES + [(8300, 1536, 512, 1, 1), (8300, 512, 2048, 1, 0), (33, 33, 31, 1, 1), (640, 640, 64, 0, 0)])
def test_gemm_precision_modes(hip, M, N, K, ak, bk, precision):
    """bf16-split (x9 / x6) and fp16-split (f16x3) emulation of the fp32 product: they must be as close to the exact (float64) product
    as the native fp32 matrix instruction is; operands with a wide dynamic range (1e-3 .. 1e3 scales per row) exercise the low-order pieces."""
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + 1)
    A = torch.randn((M, K) if ak else (K, M), generator=g)
    B = torch.randn((N, K) if bk else (K, N), generator=g)
    A = A * torch.exp(torch.randn(A.shape, generator=g) * 2.0)        # log-normal magnitudes: exponents differ inside one dot product
    exact = ((A if ak else A.t()).double() @ (B if bk else B.t()).double().t())
    out_h = torch.empty(M, N, device="cuda")
    hip.gemm(A.cuda(), B.cuda(), out_h, M=M, N=N, K=K, a_kcontig=bool(ak), b_kcontig=bool(bk), precision=precision)
    torch.cuda.synchronize()
    # row-wise scale: every row of C is compared with that row's own magnitude
    err = ((out_h.cpu().double() - exact).abs().amax(dim=1) / exact.abs().amax(dim=1).clamp_min(1e-30)).max().item()
    try:
        with open(REPORT, "a") as f:
            f.write(f"gemm[{precision}] {M}x{N}x{K} ak={ak} bk={bk}: max row-relative error vs float64 {err:.3e} tol {GEMM_MODE_TOL[precision]}\n")
    except OSError:
        pass
    assert math.isfinite(err) and err < GEMM_MODE_TOL[precision], f"{precision} {M}x{N}x{K}: {err:.3e}"
    if precision in ("f32_bf16x9", "f32_bf16x6", "f32_f16x3"):
        # fp32-grade claim: no further from the exact product than the native fp32 matrix instruction on the same inputs
        # (both carry ~sqrt(K) * 2^-24 of accumulation-order noise; 2x + 5e-7 absorbs the difference in summation order)
        hip.gemm(A.cuda(), B.cuda(), out_h, M=M, N=N, K=K, a_kcontig=bool(ak), b_kcontig=bool(bk), precision="f32")
        torch.cuda.synchronize()
        err_native = ((out_h.cpu().double() - exact).abs().amax(dim=1) / exact.abs().amax(dim=1).clamp_min(1e-30)).max().item()
        assert err <= 2.0 * err_native + 5e-7, f"{precision} {M}x{N}x{K}: {err:.3e} vs native fp32 MFMA {err_native:.3e}"


@pytest.mark.parametrize("precision", ["f32", "f32_bf16x9", "f32_bf16x6"])
@pytest.mark.parametrize("M,N,K", [(512, 512, 5000), (1536, 512, 3001), (511, 256, 700), (12, 256, 4097), (2048, 512, 300)])
def test_gemm_wgrad_with_fused_bias_gradient(hip, ref, M, N, K, precision):
    """dW += dz^T x and db += colsum(dz) from one launch (with and without split-K)"""
    g = torch.Generator().manual_seed(M + N + K)
    ldz = (M + 3) // 4 * 4
    dz_full, x = torch.randn(K, ldz, generator=g), torch.randn(K, N, generator=g)
    dz = dz_full[:, :M]
    w_r, b_r = torch.ones(M, N), torch.ones(M)
    ref.gemm(dz, x, w_r, M=M, N=N, K=K, a_kcontig=False, b_kcontig=False, accumulate=True, a_colsum=b_r)
    w_h, b_h = torch.ones(M, N, device="cuda"), torch.ones(M, device="cuda")
    hip.gemm(dz_full.cuda()[:, :M], x.cuda(), w_h, M=M, N=N, K=K, a_kcontig=False, b_kcontig=False, accumulate=True, a_colsum=b_h,
             precision=precision)
    torch.cuda.synchronize()
    _cmp(w_h, w_r, 2e-5, f"wgrad[{precision}] {M}x{N}x{K}")
    _cmp(b_h, b_r, 2e-5, f"fused bias gradient[{precision}] {M}x{N}x{K}")


@pytest.mark.parametrize("cfg", [5, 6])
@pytest.mark.parametrize("M,N,K,ak,bk", [(1000, 512, 512, 1, 1), (257, 511, 256, 1, 1), (700, 384, 1536, 1, 0), (512, 512, 5000, 0, 0),
                                         (130, 64, 85, 1, 1), (1536, 512, 3001, 0, 0)])
def test_gemm_bf16x_tile_configurations(hip, ref, M, N, K, ak, bk, cfg):
    """both tile configurations of the bf16-split kernel (5: 128x128, 6: 256x128), forced through the plan override, with a fused
    epilogue and (wgrad layout) the fused bias gradient"""
    g = torch.Generator().manual_seed(M + 3 * N + K + cfg)
    A = torch.randn((M, K) if ak else (K, M), generator=g)
    B = torch.randn((N, K) if bk else (K, N), generator=g) / math.sqrt(K)
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    wgrad = not ak and not bk
    kw = dict(accumulate=True) if wgrad else dict(bias=bias, act=1, res=res)
    o_r, o_h = torch.full((M, N), 0.25), torch.full((M, N), 0.25, device="cuda")
    cs_r, cs_h = (torch.ones(M), torch.ones(M, device="cuda")) if wgrad else (None, None)
    ref.gemm(A, B, o_r, M=M, N=N, K=K, a_kcontig=bool(ak), b_kcontig=bool(bk), a_colsum=cs_r, **kw)
    kw_h = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in kw.items()}
    hip.plan_override = (cfg, 0, -1)
    try:
        hip.gemm(A.cuda(), B.cuda(), o_h, M=M, N=N, K=K, a_kcontig=bool(ak), b_kcontig=bool(bk), a_colsum=cs_h, precision="f32_bf16x6", **kw_h)
        torch.cuda.synchronize()
    finally:
        hip.plan_override = None
    _cmp(o_h, o_r, 2e-5, f"gemm[f32_bf16x6, cfg {cfg}] {M}x{N}x{K} ak={ak} bk={bk}")
    if wgrad:
        _cmp(cs_h, cs_r, 2e-5, f"fused bias gradient[cfg {cfg}] {M}x{N}x{K}")


def test_gemm_strided_views(hip, ref):
    g = torch.Generator().manual_seed(2)
    N_, R, Wp = 200, 256, 511
    h, w, b = torch.randn(N_, R, generator=g), torch.randn(Wp, R, generator=g) / 16, torch.randn(Wp, generator=g)
    a_r, a_h = torch.zeros(N_, 512), torch.zeros(N_, 512, device="cuda")
    ref.gemm(h, w, a_r[:, :Wp], M=N_, N=Wp, K=R, bias=b, act=1)
    hip.gemm(h.cuda(), w.cuda(), a_h[:, :Wp], M=N_, N=Wp, K=R, bias=b.cuda(), act=1)
    torch.cuda.synchronize()
    _cmp(a_h, a_r, 2e-5, "gemm into a strided view (ld 512, N 511)")


def test_colsum_actdrop_add(hip, ref):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(5000, 300, generator=g)
    o_r, o_h = torch.ones(300), torch.ones(300, device="cuda")
    ref.colsum(x, o_r, True), hip.colsum(x.cuda(), o_h, True)
    _cmp(o_h, o_r, 1e-5, "colsum")
    dy, y = torch.randn(700, 130, generator=g), torch.randn(700, 130, generator=g)
    for p, yy in ((0.0, y), (0.4, y), (0.4, None)):
        z_r, z_h = torch.empty(700, 130), torch.empty(700, 130, device="cuda")
        ref.act_dropout_bwd(dy, yy, p, 777, z_r)
        hip.act_dropout_bwd(dy.cuda(), None if yy is None else yy.cuda(), p, 777, z_h)
        _cmp(z_h, z_r, 1e-6, f"act_dropout_bwd p={p}")
    a, b = torch.randn(10001, generator=g), torch.randn(10001, generator=g)
    s_h = torch.empty(10001, device="cuda")
    hip.add(a.cuda(), b.cuda(), s_h)
    assert torch.equal(s_h.cpu(), a + b)


def test_dropout_hash_matches_oracle(hip):
    from oracle.ops_ref import dropout_keep
    idx = torch.arange(0, 5000, dtype=torch.int64)
    for seed in (0, 1, 2 ** 62 + 12345, 0x5DEECE66D):
        for p in (0.1, 0.5):
            want = dropout_keep(seed, idx, p)
            got = torch.tensor([hip.lib.grappa_dropout_keep(seed, int(i), p) for i in idx[:512]], dtype=torch.bool)
            assert torch.equal(got, want[:512])
            assert abs(float(want.float().mean()) - (1 - p)) < 0.03


@pytest.mark.parametrize("M,W", [(1000, 512), (37, 2048), (513, 64), (5, 1536), (2000, 256)])
def test_layernorm(hip, ref, M, W):
    g = torch.Generator().manual_seed(M + W)
    x = torch.randn(M, W, generator=g) * 2 + 0.5
    gamma, beta = 1 + 0.1 * torch.randn(W, generator=g), 0.1 * torch.randn(W, generator=g)
    dy = torch.randn(M, W, generator=g)
    y_r, m_r, r_r = torch.empty(M, W), torch.empty(M), torch.empty(M)
    ref.layernorm_fwd(x, gamma, beta, y_r, m_r, r_r)
    y_h, m_h, r_h = torch.empty(M, W, device="cuda"), torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    hip.layernorm_fwd(x.cuda(), gamma.cuda(), beta.cuda(), y_h, m_h, r_h)
    _cmp(y_h, y_r, 1e-5, "layernorm_fwd y"), _cmp(m_h, m_r, 1e-5, "mean"), _cmp(r_h, r_r, 1e-5, "rstd")
    dx_r, dg_r, db_r = torch.empty(M, W), torch.ones(W), torch.ones(W)
    ref.layernorm_bwd(dy, x, m_r, r_r, gamma, dx_r, dg_r, db_r, True)
    dx_h, dg_h, db_h = torch.empty(M, W, device="cuda"), torch.ones(W, device="cuda"), torch.ones(W, device="cuda")
    hip.layernorm_bwd(dy.cuda(), x.cuda(), m_h, r_h, gamma.cuda(), dx_h, dg_h, db_h, True)
    _cmp(dx_h, dx_r, 2e-5, "layernorm_bwd dx"), _cmp(dg_h, dg_r, 2e-5, "dgamma"), _cmp(db_h, db_r, 2e-5, "dbeta")


def test_layernorm_parameter_gradients_deferred_to_one_batched_reduction(hip, ref):
    """accumulate = 2: the kernel leaves its per-block partial sums in the caller's buffer; grappa_colsum_partials_batched reduces those
    of several LayerNorms in ONE launch (what the backend does with a backward pass's ~50 LayerNorms): the gradients match the
    oracle and the immediate reduction, and are the same bits run after run"""
    import ctypes
    from grappa_amd import _lib
    g = torch.Generator().manual_seed(77)
    cases = []
    for M, W in [(5000, 512), (83, 256), (20011, 2048), (3, 512), (9000, 1024)]:
        x = (torch.randn(M, W, generator=g) * 2 + 0.5).cuda()
        gamma, dy = (1 + 0.1 * torch.randn(W, generator=g)).cuda(), torch.randn(M, W, generator=g).cuda()
        y, mean, rstd = torch.empty(M, W, device="cuda"), torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
        hip.layernorm_fwd(x, gamma, torch.zeros(W, device="cuda"), y, mean, rstd)
        cases.append((M, W, x, gamma, dy, mean, rstd))
    stream = torch.cuda.current_stream().cuda_stream

    def deferred():
        outs, keep = [], []
        arr = (_lib.ColsumItem * len(cases))()
        for d, (M, W, x, gamma, dy, mean, rstd) in zip(arr, cases):
            dx, dg, db = torch.empty(M, W, device="cuda"), torch.ones(W, device="cuda"), torch.full((W,), 2.0, device="cuda")
            ws = torch.empty(hip.lib.grappa_layernorm_bwd_workspace_bytes(M, W), dtype=torch.uint8, device="cuda")
            rc = hip.lib.grappa_layernorm_bwd_f32(stream, M, W, dy.data_ptr(), W, x.data_ptr(), W, mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                                  dx.data_ptr(), W, None, None, 2, ws.data_ptr(), ws.numel())
            assert rc == 0
            d.part, d.nrows, d.n, d.out, d.out2, d.n_first, d.accumulate = ws.data_ptr(), hip.lib.grappa_layernorm_bwd_partial_rows(M), 2 * W, dg.data_ptr(), db.data_ptr(), W, 1
            outs.append((dx, dg, db))
            keep.append(ws)
        assert hip.lib.grappa_colsum_partials_batched(stream, arr, len(cases)) == 0
        torch.cuda.synchronize()
        return outs

    first, second = deferred(), deferred()
    for (M, W, x, gamma, dy, mean, rstd), (dx, dg, db), (dx2, dg2, db2) in zip(cases, first, second):
        assert torch.equal(dg, dg2) and torch.equal(db, db2) and torch.equal(dx, dx2)
        dx_i, dg_i, db_i = torch.empty(M, W, device="cuda"), torch.ones(W, device="cuda"), torch.full((W,), 2.0, device="cuda")
        hip.layernorm_bwd(dy, x, mean, rstd, gamma, dx_i, dg_i, db_i, True)           # outside a backward pass: reduced at once
        assert torch.equal(dx, dx_i)
        dx_r, dg_r, db_r = torch.empty(M, W), torch.ones(W), torch.full((W,), 2.0)
        ref.layernorm_bwd(dy.cpu(), x.cpu(), mean.cpu(), rstd.cpu(), gamma.cpu(), dx_r, dg_r, db_r, True)
        _cmp(dg, dg_r, 2e-5, f"deferred dgamma {M}x{W}"), _cmp(db, db_r, 2e-5, f"deferred dbeta {M}x{W}")
        _cmp(dg_i, dg_r, 2e-5, "dgamma"), _cmp(db_i, db_r, 2e-5, "dbeta")
    # argument checks: an item without an output, a second destination without a split column
    bad = (_lib.ColsumItem * 1)()
    bad[0].part, bad[0].nrows, bad[0].n, bad[0].out = first[0][1].data_ptr(), 1, 4, None
    assert hip.lib.grappa_colsum_partials_batched(stream, bad, 1) != 0
    bad[0].out, bad[0].out2, bad[0].n_first = first[0][1].data_ptr(), first[0][2].data_ptr(), 0
    assert hip.lib.grappa_colsum_partials_batched(stream, bad, 1) != 0
    assert hip.lib.grappa_colsum_partials_batched(stream, None, 0) == 0


@pytest.mark.parametrize("H,D", [(16, 32), (4, 16), (8, 64)])
def test_gat_fwd_bwd(hip, ref, H, D):
    g_cpu, g_gpu = _graph()
    pc, pg = g_cpu.plan(), g_gpu.plan()
    gen = torch.Generator().manual_seed(H * D)
    N, F = pc.N, H * D
    ft, dout = torch.randn(N, F, generator=gen), torch.randn(N, F, generator=gen)
    o_r, a_r, d_r = torch.empty(N, F), torch.empty(pc.E, H), torch.empty(N, F)
    ref.gat_fwd(pc, ft, H, D, o_r, a_r)
    ref.gat_bwd(pc, ft, o_r, a_r, dout, H, D, d_r)
    o_h, a_h, d_h = torch.empty(N, F, device="cuda"), torch.empty(pg.E, H, device="cuda"), torch.empty(N, F, device="cuda")
    hip.gat_fwd(pg, ft.cuda(), H, D, o_h, a_h)
    hip.gat_bwd(pg, ft.cuda(), o_h, a_h, dout.cuda(), H, D, d_h)
    _cmp(o_h, o_r, 1e-5, "gat out"), _cmp(a_h, a_r, 1e-5, "gat alpha"), _cmp(d_h, d_r, 5e-5, "gat dft")
    # softmax rows sum to one per destination and head
    deg = (pc.indptr[1:] - pc.indptr[:-1]).long()
    dst = torch.repeat_interleave(torch.arange(N), deg)
    sums = torch.zeros(N, H).index_add(0, dst, a_h.cpu())
    assert torch.allclose(sums, torch.ones(N, H), atol=1e-5)
    m_r, m_h = torch.empty(N, F), torch.empty(N, F, device="cuda")
    for flag in (False, True):
        ref.neighbor_mean(pc, ft, m_r, flag), hip.neighbor_mean(pg, ft.cuda(), m_h, flag)
        _cmp(m_h, m_r, 1e-5, f"neighbor_mean {flag}")


def test_charge_encoding(hip, ref):
    q = torch.linspace(-3, 3, 1001)
    o_r, o_h = torch.zeros(1001, 85), torch.zeros(1001, 85, device="cuda")
    ref.charge_encoding(q, 16, -2.0, 2.0, o_r, 69), hip.charge_encoding(q.cuda(), 16, -2.0, 2.0, o_h, 69)
    assert float((o_h.cpu() - o_r).abs().max()) < 2e-6


@pytest.mark.parametrize("lvl,s,W,pe", [("n2", 2, 512, False), ("n3", 3, 512, True), ("n4", 4, 64, True), ("n4_improper", 4, 512, True)])
def test_tuple_gather(hip, ref, lvl, s, W, pe):
    g_cpu, g_gpu = _graph()
    pc, pg = g_cpu.plan(), g_gpu.plan()
    gen = torch.Generator().manual_seed(s * W)
    N, T = pc.N, pc.T[lvl]
    a = torch.randn(N, W, generator=gen)
    pev = torch.tensor([0., 1., 1., 0.][:s]) if pe else None
    x_r, x_h = torch.empty(s * T, W), torch.empty(s * T, W, device="cuda")
    ref.tuple_gather_fwd(a, pc.idx32[lvl], s, pev, x_r)
    hip.tuple_gather_fwd(a.cuda(), pg.idx32[lvl], s, None if pev is None else pev.cuda(), x_h)
    assert torch.equal(x_h.cpu(), x_r)          # pure data movement: bit exact
    dx = torch.randn(s * T, W, generator=gen)
    d_r, d_h = torch.empty(N, W), torch.empty(N, W, device="cuda")
    ref.tuple_gather_bwd(pc.inv_ptr[lvl], pc.inv_rows[lvl], dx, d_r, pe, False)
    hip.tuple_gather_bwd(pg.inv_ptr[lvl], pg.inv_rows[lvl], dx.cuda(), d_h, pe, False)
    _cmp(d_h, d_r, 1e-5, "tuple_gather_bwd")


@pytest.mark.parametrize("s,nh,dh", [(2, 8, 64), (3, 8, 64), (4, 8, 64), (4, 4, 16), (3, 2, 128)])
def test_seqattn(hip, ref, s, nh, dh):
    T, F = 777, nh * dh
    gen = torch.Generator().manual_seed(s + nh + dh)
    qkv, dout = torch.randn(s * T, 3 * F, generator=gen), torch.randn(s * T, F, generator=gen)
    o_r, o_h = torch.empty(s * T, F), torch.empty(s * T, F, device="cuda")
    ref.seqattn_fwd(qkv, s, T, nh, o_r), hip.seqattn_fwd(qkv.cuda(), s, T, nh, o_h)
    _cmp(o_h, o_r, 1e-5, "seqattn_fwd")
    d_r, d_h = torch.empty(s * T, 3 * F), torch.empty(s * T, 3 * F, device="cuda")
    ref.seqattn_bwd(qkv, dout, s, T, nh, d_r), hip.seqattn_bwd(qkv.cuda(), dout.cuda(), s, T, nh, d_h)
    _cmp(d_h, d_r, 2e-5, "seqattn_bwd")


@pytest.mark.parametrize("s,perms", [(2, [[0, 1], [1, 0]]), (3, [[0, 1, 2], [2, 1, 0]]), (4, [[0, 1, 2, 3], [3, 1, 2, 0]]),
                                     (4, [[0, 1, 2, 3], [3, 1, 2, 0], [1, 3, 2, 0], [0, 3, 2, 1], [3, 0, 2, 1], [1, 0, 2, 3]])])
def test_perm_concat(hip, ref, s, perms):
    T, F, P = 501, 128, len(perms)
    gen = torch.Generator().manual_seed(s * P)
    x, dz = torch.randn(s * T, F, generator=gen), torch.randn(P * T, s * F, generator=gen)
    z_r, z_h = torch.empty(P * T, s * F), torch.empty(P * T, s * F, device="cuda")
    ref.perm_concat_fwd(x, s, T, perms, z_r), hip.perm_concat_fwd(x.cuda(), s, T, perms, z_h)
    assert torch.equal(z_h.cpu(), z_r)
    d_r, d_h = torch.empty(s * T, F), torch.empty(s * T, F, device="cuda")
    ref.perm_concat_bwd(dz, s, T, perms, d_r), hip.perm_concat_bwd(dz.cuda(), s, T, perms, d_h)
    _cmp(d_h, d_r, 1e-6, "perm_concat_bwd")


@pytest.mark.parametrize("kind,nout,n_per,gated", [(0, 2, 0, False), (0, 3, 0, False), (1, 2, 0, False), (2, 12, 6, True), (2, 3, 3, False)])
def test_param_out(hip, ref, kind, nout, n_per, gated):
    T, P = 1003, 2
    gen = torch.Generator().manual_seed(kind * 10 + nout)
    o = torch.randn(P * T, nout, generator=gen)
    if kind == 2:
        o[:50] *= 1e-5          # force the hard cutoff branch
        consts = torch.cat([0.1 + torch.rand(n_per, generator=gen), torch.randn(n_per, generator=gen)])
    elif kind == 0:
        consts = torch.tensor([6.3, 0.1953, 0.0, 4.7, 161.2, 0.0])
    else:
        consts = torch.tensor([0.0292, math.pi, 0.0, 3.97, 26.6, 0.0])
    shape_k = (T, n_per) if kind == 2 else (T,)
    k_r, k_h = torch.empty(shape_k), torch.empty(shape_k, device="cuda")
    eq_r, eq_h = (None, None) if kind == 2 else (torch.empty(T), torch.empty(T, device="cuda"))
    ref.param_out_fwd(kind, o, T, P, n_per, gated, 1e-4, consts, k_r, eq_r)
    hip.param_out_fwd(kind, o.cuda(), T, P, n_per, gated, 1e-4, consts.cuda(), k_h, eq_h)
    _cmp(k_h, k_r, 1e-5, "param_out k")
    if kind == 2:
        assert torch.equal(k_h.cpu() == 0, k_r == 0)
    else:
        _cmp(eq_h, eq_r, 1e-5, "param_out eq")
    dk, deq = torch.randn(shape_k, generator=gen), (None if kind == 2 else torch.randn(T, generator=gen))
    d_r, d_h = torch.empty(P * T, nout), torch.empty(P * T, nout, device="cuda")
    ref.param_out_bwd(kind, o, T, P, n_per, gated, 1e-4, consts, dk, deq, d_r)
    hip.param_out_bwd(kind, o.cuda(), T, P, n_per, gated, 1e-4, consts.cuda(), dk.cuda(), None if deq is None else deq.cuda(), d_h)
    _cmp(d_h, d_r, 2e-5, "param_out_bwd")
    # learnable_statistics=True: the gradient of the map's own constants (sums over all T tuples)
    c_r, c_h = torch.zeros_like(consts), torch.full_like(consts, 7.0).cuda()
    ref.param_out_bwd_stats(kind, o, T, P, n_per, gated, 1e-4, consts, dk, deq, c_r)
    hip.param_out_bwd_stats(kind, o.cuda(), T, P, n_per, gated, 1e-4, consts.cuda(), dk.cuda(), None if deq is None else deq.cuda(), c_h)
    _cmp(c_h, c_r, 2e-5, "param_out_bwd_stats", floor_frac=0.05)


@pytest.mark.parametrize("n_confs,offset", [(5, False), (32, False), (40, True), (1, False), (300, False)])
def test_mm_energy_gradient_backward(hip, ref, n_confs, offset):
    g_cpu, g_gpu = _graph(n_mols=5, n_confs=n_confs, seed=n_confs)
    pc, pg = g_cpu.plan(), g_gpu.plan()
    gen = torch.Generator().manual_seed(n_confs)
    xyz = g_cpu.nodes["n1"].data["xyz"].contiguous()
    n_per = [0, 0, 6, 3]
    ks = [700 + 100 * torch.rand(pc.T["n2"], generator=gen), 100 + 20 * torch.rand(pc.T["n3"], generator=gen),
          torch.randn(pc.T["n4"], 6, generator=gen), torch.randn(pc.T["n4_improper"], 3, generator=gen)]
    eqs = [1.2 + 0.1 * torch.randn(pc.T["n2"], generator=gen), 1.9 + 0.1 * torch.randn(pc.T["n3"], generator=gen), None, None]
    B, Cc, N = pc.B, n_confs, pc.N
    lv = ["n2", "n3", "n4", "n4_improper"]

    def run(be, plan, dev):
        c = lambda t: None if t is None else t.to(dev)
        e, terms, grad = torch.empty(B, Cc, device=dev), torch.empty(4, B, Cc, device=dev), torch.empty(N, Cc, 3, device=dev)
        te = [torch.empty(plan.T[l], Cc, device=dev) for l in lv]
        tx = [torch.empty(plan.T[l], Cc, device=dev) for l in lv]
        kk, ee = [c(k) for k in ks], [c(q) for q in eqs]
        be.mm_energy_fwd(plan, c(xyz), kk, ee, n_per, offset, e, terms, te, tx)
        be.mm_gradient_fwd(plan, c(xyz), kk, ee, n_per, grad)
        g2 = torch.Generator().manual_seed(5)
        gE, gG = torch.randn(B, Cc, generator=g2), torch.randn(N, Cc, 3, generator=g2)
        gks = [torch.zeros_like(k) for k in kk]
        geqs = [torch.zeros_like(ee[0]), torch.zeros_like(ee[1]), None, None]
        be.mm_bwd(plan, c(xyz), kk, ee, n_per, offset, c(gE), c(gG), gks, geqs)
        return e, terms, grad, te, tx, gks, geqs

    r = run(ref, pc, "cpu")
    h = run(hip, pg, "cuda")
    torch.cuda.synchronize()
    _cmp(h[0], r[0], 2e-5, "energy"), _cmp(h[1], r[1], 2e-5, "term energies"), _cmp(h[2], r[2], 5e-5, "gradient")
    for l in range(4):
        _cmp(h[3][l], r[3][l], 5e-5, f"tuple energy {lv[l]}"), _cmp(h[4][l], r[4][l], 2e-5, f"internal coordinate {lv[l]}")
        _cmp(h[5][l], r[5][l], 1e-4, f"gk {lv[l]}")
        if l < 2:
            _cmp(h[6][l], r[6][l], 1e-4, f"geq {lv[l]}")
    # translation invariance: zero net force per molecule and conformation
    ptr = pc.atom_molptr.long()
    seg = torch.repeat_interleave(torch.arange(B), ptr[1:] - ptr[:-1])
    net = torch.zeros(B, Cc, 3).index_add(0, seg, h[2].cpu())
    assert float(net.abs().max()) < 1e-2 * float(h[2].abs().max())


def test_loss_kernels(hip, ref):
    g_cpu, g_gpu = _graph(n_mols=7, n_confs=6, seed=9)
    pc, pg = g_cpu.plan(), g_gpu.plan()
    gen = torch.Generator().manual_seed(11)
    B, Cc, N = pc.B, 6, pc.N
    e, er = torch.randn(B, Cc, generator=gen) * 5, torch.randn(B, Cc, generator=gen) * 5
    gr, grr = torch.randn(N, Cc, 3, generator=gen) * 10, torch.randn(N, Cc, 3, generator=gen) * 10
    dummy = torch.zeros(B, Cc)
    dummy[1, 4:] = 1
    dummy[3, 2:] = 1
    for dm in (None, dummy):
        lm_r, gE_r, gG_r = torch.zeros(B), torch.zeros(B, Cc), torch.zeros(N, Cc, 3)
        ref.loss_ef(pc, e, er, dm, gr, grr, 1.0, 0.8, 1.0 / B, lm_r, gE_r, gG_r)
        lm_h, gE_h, gG_h = torch.zeros(B, device="cuda"), torch.zeros(B, Cc, device="cuda"), torch.zeros(N, Cc, 3, device="cuda")
        hip.loss_ef(pg, e.cuda(), er.cuda(), None if dm is None else dm.cuda(), gr.cuda(), grr.cuda(), 1.0, 0.8, 1.0 / B, lm_h, gE_h, gG_h)
        _cmp(lm_h, lm_r, 1e-5, "loss_mol"), _cmp(gE_h, gE_r, 2e-5, "gE"), _cmp(gG_h, gG_r, 1e-5, "gG")
    lv = ["n2", "n2", "n3", "n3", "n4", "n4_improper"]
    width = [1, 1, 1, 1, 6, 3]
    params = [torch.randn(pc.T[l], w, generator=gen).squeeze(-1) if w == 1 else torch.randn(pc.T[l], w, generator=gen) for l, w in zip(lv, width)]
    refs = [torch.randn_like(p) for p in params]
    refs[4] = torch.randn(pc.T["n4"], 4, generator=gen)        # narrower reference: zero padded
    refs[2][:3] = float("nan")
    refs[5] = None
    fac, reg = [1e-3, 1.0, 1e-2, 1.0, 1e-4, 1.0], [0, 0, 0, 0, 1e-3, 2e-3]
    pw = torch.full((B,), 1e-3)
    pw[2] = 0.0
    lm_r, lm_h = torch.zeros(B), torch.zeros(B, device="cuda")
    gp_r = [torch.zeros_like(p) for p in params]
    gp_h = [torch.zeros_like(p).cuda() for p in params]
    ref.loss_param(pc, params, refs, fac, reg, pw, 1.0 / B, lm_r, gp_r)
    hip.loss_param(pg, [p.cuda() for p in params], [None if r_ is None else r_.cuda() for r_ in refs], fac, reg, pw.cuda(), 1.0 / B, lm_h, gp_h)
    _cmp(lm_h, lm_r, 1e-5, "param loss")
    for l in range(6):
        _cmp(gp_h[l], gp_r[l], 2e-5, f"param grad {l}")


def test_adam_and_sumsq(hip, ref):
    gen = torch.Generator().manual_seed(4)
    n = 1_000_003
    p, g = torch.randn(n, generator=gen), torch.randn(n, generator=gen) * 3
    m, v = torch.zeros(n), torch.zeros(n)
    ph, gh, mh, vh = p.cuda(), g.cuda(), m.cuda(), v.cuda()
    ss_r, ss_h = torch.zeros(1), torch.zeros(1, device="cuda")
    for step in (1, 2, 3):
        ref.sumsq(g, ss_r, False), hip.sumsq(gh, ss_h, False)
        _cmp(ss_h, ss_r, 1e-5, "sumsq")
        ref.adam_step(p, g, m, v, 1e-3, 0.9, 0.999, 1e-8, 0.0, step, 1.0, ss_r, 10.0)
        hip.adam_step(ph, gh, mh, vh, 1e-3, 0.9, 0.999, 1e-8, 0.0, step, 1.0, ss_h, 10.0)
    _cmp(ph, p, 2e-5, "adam p"), _cmp(mh, m, 2e-5, "adam m"), _cmp(vh, v, 2e-5, "adam v")
    # against torch.optim.Adam + clip_grad_norm_ (what the reference's trainer does)
    q = torch.nn.Parameter(torch.randn(1000, generator=gen))
    q0 = q.detach().clone()
    opt = torch.optim.Adam([q], lr=1e-3)
    m2, v2, qh = torch.zeros(1000, device="cuda"), torch.zeros(1000, device="cuda"), q0.cuda()
    for step in (1, 2):
        gq = torch.randn(1000, generator=gen) * 5
        q.grad = gq.clone()
        torch.nn.utils.clip_grad_norm_([q], 10.0)
        opt.step()
        hip.sumsq(gq.cuda(), ss_h, False)
        hip.adam_step(qh, gq.cuda(), m2, v2, 1e-3, 0.9, 0.999, 1e-8, 0.0, step, 1.0, ss_h, 10.0)
    _cmp(qh, q.detach(), 1e-5, "adam vs torch.optim.Adam + clip_grad_norm_")


def test_unsupported_shapes_fail_loudly(hip):
    from grappa_amd.backend import GrappaHipError
    x = torch.randn(8, 30, device="cuda")           # W % 4 != 0
    with pytest.raises((GrappaHipError, ValueError)):
        hip.layernorm_fwd(x, torch.ones(30, device="cuda"), torch.zeros(30, device="cuda"), torch.empty_like(x), None, None)
    with pytest.raises((GrappaHipError, ValueError)):
        hip.gemm(torch.randn(4, 4), torch.randn(4, 4, device="cuda"), torch.empty(4, 4, device="cuda"), M=4, N=4, K=4)   # CPU tensor
