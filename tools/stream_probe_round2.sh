# the round-1 multi-queue deviation, looked for again with this round's GEMM kernels: tools/stream_order_probe.py under the test-only
# library whose LayerNorm forward takes two rows per trip (tools/ln_two_rows_variant.py), for the arithmetics the probe named
L=build/variants/libgrappa_hip_tworow.so
for p in bf16x3 f32_bf16x6 f32_f16x3 f32; do
  echo "== two-row LayerNorm, $p"
  GRAPPA_HIP_LIB=$L GRAPPA_GEMM_PRECISION=$p timeout -k 10 120 python tools/stream_order_probe.py 2>&1 | grep -v amdgpu.ids | tail -4
  echo "== two-row LayerNorm on a constant input, $p"
  PROBE_CONST_INPUT=1 GRAPPA_HIP_LIB=$L GRAPPA_GEMM_PRECISION=$p timeout -k 10 120 python tools/stream_order_probe.py 2>&1 | grep -v amdgpu.ids | tail -4
done
echo "== shipped library (one row per trip), bf16x3"
GRAPPA_GEMM_PRECISION=bf16x3 timeout -k 10 120 python tools/stream_order_probe.py 2>&1 | grep -v amdgpu.ids | tail -2
