# the round-1 multi-queue deviation against this round's kernels: tools/stream_order_probe.py under test-only libraries whose LayerNorm
# forward takes two rows per trip WITHOUT the row-maxima code (tools/ln_two_rows_variant.py --no-maxima: round 1's form of the kernel),
# for every arithmetic, on a constant input, and beside GEMM kernels with one part knocked out (GB_KNOCK)
V=build/variants
run() { GRAPPA_HIP_LIB=$1 GRAPPA_GEMM_PRECISION=$2 timeout -k 10 120 python tools/stream_order_probe.py 2>&1 | tail -1; }
for p in bf16x3 f32_bf16x6 f32_f16x3 f32; do echo "== two-row LayerNorm, $p"; run $V/libgrappa_hip_tworow.so $p; done
echo "== two-row LayerNorm on a constant input, bf16x3"; PROBE_CONST_INPUT=1 run $V/libgrappa_hip_tworow.so bf16x3
for k in nomfma nolds noepi noglobal nosplit; do
  [ -f $V/libgrappa_hip_tworow_$k.so ] && { echo "== two-row LayerNorm (constant input) beside GEMMs with $k, bf16x3"; PROBE_CONST_INPUT=1 run $V/libgrappa_hip_tworow_$k.so bf16x3; }
done
echo "== shipped library (one row per trip), bf16x3"; GRAPPA_GEMM_PRECISION=bf16x3 timeout -k 10 120 python tools/stream_order_probe.py 2>&1 | tail -1
# the same two-row kernel with its wave sums on the DPP path + v_readlane instead of ds_bpermute (the LDS crossbar)
[ -f $V/libgrappa_hip_tworow_dpp.so ] && for p in bf16x3 f32_f16x3; do echo "== two-row LayerNorm, DPP sums (no ds_bpermute), $p"; PROBE_CONST_INPUT=1 run $V/libgrappa_hip_tworow_dpp.so $p; done
