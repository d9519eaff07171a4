# A/B on one box: first transformer layer of the writers on (atom, position) rows (default) vs on tokens (GRAPPA_FIRST_LAYER_ROWS=0)
set -e
B="python bench.py --no-cpu-baseline --no-extras --alt-precision= --steps 20 --warmup 5"
show() { python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(b['ms_per_step'],2), 'ms/step; products', round(b['roofline']['kernel_ms_per_step'],2), 'ms,', round(b['roofline']['achieved'],1), 'TFLOP/s')"; }
for i in 1 2 3; do
  GRAPPA_FIRST_LAYER_ROWS=0 $B 2>/dev/null | show "tokens          "
  GRAPPA_FIRST_LAYER_ROWS=1 $B 2>/dev/null | show "(atom, pos) rows"
done
