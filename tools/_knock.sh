for v in base nostore nodescale; do
  echo "== $v"
  if [ $v = base ]; then L=; else L=build/variants/libgrappa_hip_bx_$v.so; fi
  GRAPPA_HIP_LIB=$L timeout -k 10 100 python tools/gemm_f16x3_check.py --bench-only 2>&1 | grep -E "^  M|sum" | sed 's/f32 [0-9.]* ms *[0-9.]* TF  f32_bf16x6/x6/' | cut -c1-150
done
