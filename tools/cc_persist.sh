#!/bin/bash
# compile one csrc unit with the resource-usage remarks and keep its ISA in /tmp (development aid)
cd /root/repo/grappa_amd/csrc
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops -Wno-unused-function -Rpass-analysis=kernel-resource-usage --save-temps=obj $EXTRA -c ${1:-gemm_pairs_persist}.hip -o /tmp/${1:-gemm_pairs_persist}.o 2>&1 | grep -E "warning|error|Function Name|VGPRs:|ScratchSize|VGPRs Spill|SGPRs Spill|Occupancy" | grep -v "splitk_reduce" 
