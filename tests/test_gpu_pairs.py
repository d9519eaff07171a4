"""GPU (-m gpu): the pair-format product (csrc/gemm_pairs.hip; include/grappa_hip.h, ABI 5) through the C ABI: operands split once by
grappa_split_pairs_f32 (rows of A, rows of W or -- for the input-gradient layout -- rows of W^T) give the SAME BITS as the fp32-operand
fp16-split product of the same K split, and float64-grade errors on ragged shapes, scaled rows and the fused epilogues.  Where K % 16 == 0
`check` also runs the "weight pairs" form (fp32 A as it is + the pairs of W) and asserts the all-pairs product's bits."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("M,N,K,dgrad,scaled,epi", [
    (256, 128, 32, False, False, {}), (300, 200, 64, False, False, {}), (257, 129, 96, False, False, {}), (1000, 512, 512, False, False, {}),
    (4096, 1536, 512, False, False, {}), (777, 256, 1536, False, False, {}), (1000, 512, 512, True, False, {}),
    (1000, 512, 512, False, True, {}), (1000, 512, 512, False, False, dict(bias=1, act=1)), (1000, 512, 512, False, False, dict(bias=1, res=1)),
    (33, 40, 48, False, False, {}), (20000, 64, 512, False, False, {}), (3000, 512, 1536, True, False, {})])
def test_pair_format_product_equals_the_fp32_operand_product_bit_for_bit(M, N, K, dgrad, scaled, epi):
    import gemm_pairs_check as gp
    gen = torch.Generator(device="cuda")
    gen.manual_seed(M * 7 + N)
    same = gp.check(M, N, K, gen, dgrad=dgrad, scale_rows=scaled, **epi)          # (asserts the float64 error inside; returns bit equality)
    # the two kernels plan their K cuts independently (the pair kernel has one tile shape): where the cuts agree the bits do
    assert same or (M, N, K) in ((20000, 64, 512), (3000, 512, 1536), (33, 40, 48))


def test_pair_layout_and_rejections():
    """element (r, k): HI at r * ld + 32 * (k // 16) + k % 16, LO 16 further; (HI + LO) * 2^-s reproduces the value to 2^-22 of the row maximum"""
    import ctypes as C
    import gemm_pairs_check as gp
    from grappa_amd import _lib
    gen = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn((37, 50), generator=gen, device="cuda") * torch.exp2(torch.randint(-20, 20, (37, 1), generator=gen, device="cuda").float())
    am = gp.amax(x)
    p = gp.split_pairs(x, am)
    assert p.shape == (37, 2 * 64) and p.dtype == torch.float16
    k = torch.arange(50, device="cuda")
    col = 32 * (k // 16) + k % 16
    shift = 141 - ((am >> 23) & 0xff)
    back = (p[:, col].double() + p[:, col + 16].double()) * torch.exp2(-shift.double())[:, None]
    rowmax = x.abs().amax(dim=1, keepdim=True).double()
    assert float(((back - x.double()).abs() / rowmax).max()) < 2.0 ** -21
    assert float(p[:, col].abs().max()) < 2.0 ** 15 and float(p[:, col].abs().amax(dim=1).min()) >= 2.0 ** 14       # every row's largest element in [2^14, 2^15)
    pad = torch.ones(128, dtype=torch.bool, device="cuda")
    pad[col] = False
    pad[col + 16] = False
    assert float(p[:, pad].abs().max()) == 0.0                       # the padding beyond C stays zero
    # one operand in pairs, the other not: refused
    d = _lib.GemmDesc()
    out = torch.empty((37, 37), device="cuda")
    d.M, d.N, d.K, d.a_kcontig, d.b_kcontig = 37, 37, 50, 1, 1
    d.A, d.lda, d.a_planes = p.data_ptr(), p.stride(0), 1
    d.B, d.ldb = x.data_ptr(), x.stride(0)
    d.C, d.ldc = out.data_ptr(), out.stride(0)
    d.a_amax, d.b_amax = am.data_ptr(), am.data_ptr()
    d.precision = _lib.GEMM_PRECISIONS["f32_f16x3"]
    ws = gp.ws_for(37, 37, 50)
    assert gp.lib.grappa_gemm_f32(gp.stream(), C.byref(d), ws.data_ptr(), ws.numel()) == -1
