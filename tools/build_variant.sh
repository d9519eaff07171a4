#!/bin/bash
# tools/build_variant.sh N: libgrappa_hip_expN.so = the library with gemm_bf16x.hip compiled under -DGRAPPA_EXP=N (kernel A/B runs;
# select it with GRAPPA_HIP_LIB=grappa_amd/libgrappa_hip_expN.so)
set -e
cd "$(dirname "$0")/../grappa_amd/csrc"
make -j8 >/dev/null
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wall -Wno-unused-function -DGRAPPA_EXP=$1 -c gemm_bf16x.hip -o /tmp/gemm_bf16x_exp$1.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libgrappa_hip_exp$1.so gemm_f32.o /tmp/gemm_bf16x_exp$1.o rowwise.o graph.o tuples.o mm_energy.o loss.o
echo built libgrappa_hip_exp$1.so
