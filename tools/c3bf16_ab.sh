#!/bin/bash
# GPU box: the C3 train step in the bf16 storage configuration under a list of environments, one process each, alternating twice
cd "$(dirname "$0")/.."
out=gpurun_out/c3bf16_ab_${1:-run}.txt
: > $out
run() { res=$(env "$@" python bench.py --no-extras --no-cpu-baseline --alt-precision '' --workload C3-espaloma-b1024 --act-dtype bf16 --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print(round(j['ms_per_step'],2), round(r['frac'],4), round(r['kernel_ms_per_step'],2))"); echo "$* : $res" | tee -a $out; }
for rep in 1 2; do
  while read -r line; do [ -n "$line" ] && run $line; done <<< "$CONFIGS"
done
