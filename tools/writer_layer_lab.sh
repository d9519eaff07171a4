#!/bin/bash
# build lab variants of the library that differ in csrc/writer_layer.hip only (-DWL_LAB_...): tools/lab/libgrappa_hip_<name>.so
# usage: tools/writer_layer_lab.sh name "-DWL_LAB_NO_WLOAD" [name2 "flags2" ...]
set -e
cd "$(dirname "$0")/../grappa_amd/csrc"
mkdir -p ../../tools/lab
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops -Wall -Wno-unused-function $flags -c writer_layer.hip -o /tmp/wl_$name.o 2> >(grep -v "is not a recognized feature" >&2)
  objs=$(ls *.o | grep -v writer_layer.o)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/lab/libgrappa_hip_$name.so $objs /tmp/wl_$name.o
  echo built tools/lab/libgrappa_hip_$name.so
done
