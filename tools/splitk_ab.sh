# C2 step with split-K summed by a reduction launch / inside the product's launch, and the two timing knock-outs of the latter
# (build/variants/libgrappa_hip_sk{1,2}.so = gemm_bf16x_h3.hip compiled with -DSK_EXP=1: no fences, 2: fences + ticket, no sum)
set -e
B="python bench.py --no-cpu-baseline --no-extras --alt-precision= --steps 20 --warmup 5"
show() { python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(b['ms_per_step'],2), 'ms/step, products', round(b['roofline']['kernel_ms_per_step'],2))"; }
GRAPPA_SPLITK_IN_KERNEL=0 $B 2>/dev/null | show "reduction launch           "
GRAPPA_SPLITK_IN_KERNEL=1 $B 2>/dev/null | show "in-kernel (fences + sum)   "
for k in 1 2; do
  if [ -f build/variants/libgrappa_hip_sk$k.so ]; then
    GRAPPA_SPLITK_IN_KERNEL=1 GRAPPA_HIP_LIB=build/variants/libgrappa_hip_sk$k.so $B 2>/dev/null | show "in-kernel, SK_EXP=$k         "
  fi
done
