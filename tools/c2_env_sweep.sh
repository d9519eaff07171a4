#!/bin/bash
# tools/c2_env_sweep.sh OUT "ENV1" "ENV2" ... -- on the GPU box: the C2 train step (10 steps after 3 warm-up) under each environment setting, one box, in order
OUT=$1; shift
mkdir -p gpurun_out
for V in "$@"; do
  env $V python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --alt-precision "" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"$V\", round(r[\"ms_per_step\"],2), round(r[\"roofline\"][\"frac\"],4), round(r[\"roofline\"][\"kernel_ms_per_step\"],2))" >> $OUT
done
cat $OUT
