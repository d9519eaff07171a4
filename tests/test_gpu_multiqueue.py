"""GPU (-m gpu): the shipped kernels under queue sharing.  Four HIP streams each run GEMM -> LayerNorm chains of this library at the
writers' token counts; every LayerNorm output must equal the result of the same chain run alone, bit for bit.  This is the op-level
reproducer of the multi-queue deviation (tools/stream_order_probe.py): a test-only LayerNorm kernel with packed
fp32 instructions fails it in 100+ of 160 cases; the shipped library (compiled without them, tests/test_capi_symbols.py) must not."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", ["f32_f16x3", "bf16x3"])
def test_layernorm_beside_gemms_on_four_streams_equals_its_solo_result(precision):
    from grappa_amd.backend import get_backend
    be = get_backend()
    old = be.gemm_precision_name
    be.set_gemm_precision(precision)
    try:
        Ms, W = [83328, 44325, 28248, 17158], 512
        Wm = torch.randn(512, 512, generator=torch.Generator().manual_seed(1)).cuda() / 22.6
        data = []
        for M in Ms:
            x = torch.randn(M, W, device="cuda", generator=torch.Generator(device="cuda").manual_seed(M)) * 2 + 0.3
            g, b = torch.randn(W, device="cuda"), torch.randn(W, device="cuda")
            ref, mean, rstd, t0 = torch.empty_like(x), torch.empty(M, device="cuda"), torch.empty(M, device="cuda"), torch.empty_like(x)
            be.gemm(x, Wm, t0, M=M, N=512, K=512, res=x)
            be.layernorm_fwd(t0, g, b, ref, mean, rstd, amax=False)
            data.append((x, g, b, ref, t0))
        torch.cuda.synchronize()
        streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in Ms[1:]]
        bad = total = 0
        for trial in range(5):
            outs = []
            for st in streams[1:]:
                st.wait_stream(streams[0])
            for (x, g, b, ref, t0), st in zip(data, streams):
                with torch.cuda.stream(st):
                    for rep in range(4):
                        t = torch.empty_like(x)
                        be.gemm(x, Wm, t, M=x.shape[0], N=512, K=512, res=x)
                        y, m2, r2 = torch.empty_like(x), torch.empty(x.shape[0], device="cuda"), torch.empty(x.shape[0], device="cuda")
                        be.layernorm_fwd(t if rep % 2 else t0, g, b, y, m2, r2, amax=False)      # on the concurrent product's output / on a constant input
                        outs.append((y, ref, t, t0))
            torch.cuda.synchronize()
            for y, ref, t, t0 in outs:
                total += 1
                bad += int(not torch.equal(y, ref)) + int(not torch.equal(t, t0))
        assert total == 80 and bad == 0, f"{bad} of {total} chains differ from their solo result under four queues"
    finally:
        be.set_gemm_precision(old)
