# A/B on one box: LayerNorm parameter-gradient reductions deferred to one batched launch per backward pass (default) vs two small
# launches per LayerNorm (GRAPPA_DEFER_LN_REDUCTIONS=0)
set -e
B="python bench.py --no-cpu-baseline --no-extras --alt-precision= --steps 20 --warmup 5"
show() { python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(b['ms_per_step'],2), 'ms/step')"; }
for i in 1 2 3; do
  GRAPPA_DEFER_LN_REDUCTIONS=0 $B 2>/dev/null | show "per LayerNorm "
  GRAPPA_DEFER_LN_REDUCTIONS=1 $B 2>/dev/null | show "one batched   "
done
