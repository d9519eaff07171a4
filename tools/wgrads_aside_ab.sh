#!/bin/bash
# GPU box: the C2 step with the GNN's (and the heads' leftover) weight gradients launched on a side stream beside the GNN's backward pass
cd "$(dirname "$0")/.."
for v in 0 1 0 1; do
  echo "== GRAPPA_WGRADS_ASIDE=$v"
  GRAPPA_WGRADS_ASIDE=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --alt-precision '' 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line)
        print('ms/step', round(r['ms_per_step'], 2), 'final_loss', r['final_loss'])
"
done
