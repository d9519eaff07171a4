#!/bin/bash
# GPU box: the C2 step with the products' results stored nontemporally (build/variants/libgrappa_hip_nt1.so: GB_NT_STORE=1; nt2: + nontemporal
# loads of the residual / saved activation) against the shipped library.  Build the variants first (or:
#   hipcc <csrc/Makefile flags> -DGB_NT_STORE=1 -c csrc/gemm_bf16x_h3.hip -o build/variants/h3_nt1.o; hipcc -shared ... the other objects)
cd "$(dirname "$0")/.."
for v in ${VARIANTS:-"" nt1 nt2 "" nt1 nt2}; do
  [ "$v" = shipped ] && v=""
  echo "== variant ${v:-shipped}"
  lib=""; [ -n "$v" ] && lib="$PWD/build/variants/libgrappa_hip_$v.so"
  GRAPPA_HIP_LIB=$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --alt-precision '' 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line)
        print('ms/step', round(r['ms_per_step'], 2), 'loss', r['final_loss'], 'gemm ms (one queue)', round(r['roofline']['kernel_ms_per_step'], 2))
"
done
