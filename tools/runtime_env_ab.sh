#!/bin/bash
# GPU box: the C2 step under HIP runtime settings that touch launch latency and queue mapping (the engine uses 4 head streams + side streams)
cd "$(dirname "$0")/.."
run() {
  echo "== $*"
  env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --alt-precision '' 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line)
        print('ms/step', round(r['ms_per_step'], 2), 'loss', r['final_loss'])
"
}
for rep in 1 2; do
run X=1
run GPU_MAX_HW_QUEUES=8
run GPU_MAX_HW_QUEUES=2
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run GPU_MAX_HW_QUEUES=8 HIP_FORCE_DEV_KERNARG=1
done
