# the library compiled without packed fp32 instructions (build/variants/libgrappa_hip_nopk.so: every csrc/*.hip with
# -Xclang -target-feature -Xclang -packed-fp32-ops) beside the shipped one: C2 step on one stream and on four
set -e
B="python bench.py --no-cpu-baseline --no-extras --alt-precision= --steps 20 --warmup 5"
show() { python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(b['ms_per_step'],2), 'ms/step')"; }
L=build/variants/libgrappa_hip_nopk.so
for i in 1 2; do
  $B 2>/dev/null | show "shipped library, one stream        "
  GRAPPA_HIP_LIB=$L $B 2>/dev/null | show "no packed fp32, one stream         "
  GRAPPA_HIP_LIB=$L GRAPPA_HEAD_STREAMS=4 $B 2>/dev/null | show "no packed fp32, four streams       "
done
