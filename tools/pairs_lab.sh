#!/bin/bash
# GPU box: A/B of the pair-format GEMM variants, one process per environment (the switches are read once per process)
cd "$(dirname "$0")/.."
out=gpurun_out/pairs_lab_${1:-run}.txt
: > $out
run() { echo "--- $*" >> $out; env "$@" python tools/pairs_lab.py --tag "$*" $LAB_ARGS >> $out 2>&1 || echo "FAILED rc=$?" >> $out; }
while read -r line; do [ -n "$line" ] && run $line; done <<< "$CONFIGS"
tail -3 $out
