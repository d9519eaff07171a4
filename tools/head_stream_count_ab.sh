cd "$(dirname "$0")/.."
for hs in 4 2 3 4 2 3; do
  echo "== GRAPPA_HEAD_STREAMS=$hs"
  GRAPPA_HEAD_STREAMS=$hs python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --alt-precision '' 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line)
        print('ms/step', round(r['ms_per_step'], 2), 'loss', r['final_loss'])
"
done
