#!/bin/bash
# GPU box: recorded epochs at batch 32 (tools/b32_epochs.py) under a list of environments, one process each
cd "$(dirname "$0")/.."
out=gpurun_out/b32_ab_${1:-run}.txt
: > $out
while read -r line; do
  [ -z "$line" ] && continue
  res=$(env $line python tools/b32_epochs.py 4 4 2>/dev/null | grep "recorded epoch [23]" | awk '{print $7}' | tr '\n' ' ')
  echo "$line : ms/step epochs 2,3 = $res" | tee -a $out
done <<< "$CONFIGS"
