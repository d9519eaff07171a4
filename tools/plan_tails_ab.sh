#!/bin/bash
# GPU box: the C2 step with and without the products' split-K tail launches (GRAPPA_PLAN_TAILS), writer heads on four streams and on one
cd "$(dirname "$0")/.."
for hs in 4 1; do
for tails in 1 0 1 0; do
  echo "== GRAPPA_HEAD_STREAMS=$hs GRAPPA_PLAN_TAILS=$tails"
  GRAPPA_HEAD_STREAMS=$hs GRAPPA_PLAN_TAILS=$tails python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --alt-precision '' 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line)
        print('ms/step', round(r['ms_per_step'], 2), 'loss', r['final_loss'], 'gemm ms (one queue)', round(r['roofline']['kernel_ms_per_step'], 2), 'launches', r['roofline']['launches_per_step'])
"
done
done
