# A/B on one box: the four writer heads on one stream (default) vs on four streams (GRAPPA_HEAD_STREAMS=4, opt-in: DESIGN.md section 6
# "Multi-queue deviation")
set -e
B="python bench.py --no-cpu-baseline --no-extras --alt-precision= --steps 20 --warmup 5"
show() { python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(b['ms_per_step'],2), 'ms/step')"; }
for i in 1 2; do
  GRAPPA_HEAD_STREAMS=1 $B 2>/dev/null | show "one stream  "
  GRAPPA_HEAD_STREAMS=4 $B 2>/dev/null | show "four streams"
done
