#!/bin/bash
# A/B on the GPU box: the GNN's 8,233-row products (C2) through the weight-pairs / pair kernels instead of the fp32-operand kernel
set -e
mkdir -p gpurun_out
O=gpurun_out/r6_ab_gnn_pairs.txt
: > $O
for cfg in "" "GRAPPA_WPAIRS_MIN_ROWS=8000" "GRAPPA_WPAIRS_MIN_ROWS=8000 GRAPPA_PAIRS_MIN_ROWS=8000" "" "GRAPPA_WPAIRS_MIN_ROWS=8000" "GRAPPA_WPAIRS_MIN_ROWS=8000 GRAPPA_PAIRS_MIN_ROWS=8000"; do
    echo "[$cfg]" >> $O
    env $cfg python bench.py --no-cpu-baseline --no-extras --alt-precision "" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('  ms_per_step', round(d['ms_per_step'],3), 'final_loss', d.get('final_loss'), 'products ms', round(d['roofline']['kernel_ms_per_step'],2))" >> $O
done
env GRAPPA_WPAIRS_MIN_ROWS=8000 GRAPPA_PAIRS_MIN_ROWS=8000 python bench.py --no-cpu-baseline --no-extras --alt-precision "" --shape-table gpurun_out/r6_gnn_pairs_shapes.txt > /dev/null 2>&1
cat $O
grep " 8233 " gpurun_out/r6_gnn_pairs_shapes.txt | head -24 | cut -c1-130
