#!/bin/bash
# GPU box: A/B of the pair-format GEMM variants, one process per environment (the switches are read once per process)
cd "$(dirname "$0")/.."
out=gpurun_out/pairs_lab_${1:-run}.txt
: > $out
run() { echo "--- $*" >> $out; env "$@" python tools/pairs_lab.py --tag "$*" $LAB_ARGS >> $out 2>&1 || echo "FAILED rc=$?" >> $out; }
run GRAPPA_PAIRS_PERSIST=0
run GRAPPA_PAIRS_PERSIST=1 GRAPPA_PAIRS_STAGGER=0
run GRAPPA_PAIRS_PERSIST=1
run GRAPPA_PAIRS_PERSIST=1 GRAPPA_PAIRS_STAGGER=3
run GRAPPA_PAIRS_PERSIST=0 GRAPPA_PAIRS_TILE=256
run GRAPPA_PAIRS_PERSIST=1 GRAPPA_PAIRS_TILE=256
run GRAPPA_PAIRS_PERSIST=0
tail -5 $out
