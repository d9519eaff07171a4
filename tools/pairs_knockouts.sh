#!/bin/bash
# GPU box: times the pair-format GEMM with parts of the kernel removed (GQ_KNOCK: 1 no LDS-DMA after the prologue, 2 no MFMAs, 3 no epilogue)
set -e
cd "$(dirname "$0")/../grappa_amd/csrc"
FLAGS="-O3 -fPIC -std=c++17 --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops -Wno-unused-function"
OTHERS=$(ls *.o | grep -v gemm_pairs.o)
for k in ${KNOCKS:-0 1 2 3}; do
  /opt/rocm/bin/hipcc $FLAGS -DGQ_KNOCK=$k -c gemm_pairs.hip -o /tmp/gemm_pairs_k$k.o 2> >(grep -v "is not a recognized feature" >&2)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/libgrappa_k$k.so $OTHERS /tmp/gemm_pairs_k$k.o
  echo "== GQ_KNOCK=$k"
  GRAPPA_HIP_LIB=/tmp/libgrappa_k$k.so python ../../tools/gemm_pairs_check.py --timing-only ${PAIRS_ARGS}
done
