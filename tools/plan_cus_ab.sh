#!/bin/bash
# GPU box: the C2 step with the products planning for fewer CUs (the writer heads run on four streams: each product fills a share of the chip)
cd "$(dirname "$0")/.."
for cus in 256 128 96 64 256 128; do
  echo "== GRAPPA_PLAN_CUS=$cus"
  GRAPPA_PLAN_CUS=$cus python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --alt-precision '' 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line)
        print('ms/step', round(r['ms_per_step'], 2), 'loss', r['final_loss'], 'gemm ms', round(r['roofline']['kernel_ms_per_step'], 2), 'launches', r['roofline']['launches_per_step'])
"
done
