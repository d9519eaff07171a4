for m in 3 1 0; do
  echo "== GRAPPA_EPI_FAST=$m"
  GRAPPA_EPI_FAST=$m timeout -k 10 200 python bench.py --workload C3-espaloma-b1024 --act-dtype bf16 --no-extras --no-cpu-baseline --alt-precision "" --steps 3 --warmup 1 2>&1 | grep -E "timed region" 
done
