#!/bin/bash
# A/B on the GPU box: split-K tail launches for the GNN's products (the GNN runs alone on the chip) while the heads run without
set -e
mkdir -p gpurun_out
O=gpurun_out/r6_ab_gnn_tails.txt
: > $O
for v in 1 0 1 0; do
    echo -n "GRAPPA_GNN_TAILS=$v  " >> $O
    GRAPPA_GNN_TAILS=$v python bench.py --no-cpu-baseline --no-extras --alt-precision "" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ms_per_step', round(d['ms_per_step'],3), 'final_loss', d.get('final_loss'), 'products one queue', round(d['roofline']['kernel_ms_per_step'],2))" >> $O
done
GRAPPA_GNN_TAILS=1 python bench.py --no-cpu-baseline --no-extras --alt-precision "" --shape-table gpurun_out/r6_gnn_tails_shapes.txt > /dev/null 2>&1
cat $O
grep " 8233 " gpurun_out/r6_gnn_tails_shapes.txt | head -10 | cut -c1-130
