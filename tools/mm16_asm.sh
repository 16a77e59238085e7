#!/bin/bash
# compile matmul16.hip to /tmp/probe/mm16.s and print VGPR / spill counts of the N>128 kernels
cd /root/repo/graph_neural_net_amd/csrc || exit 1
mkdir -p /tmp/probe
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I../../include -I. -S --cuda-device-only matmul16.hip -o /tmp/probe/mm16.s ${EXTRA} 2>&1 | grep -v hip-link | head -20
grep "vgpr_count\|vgpr_spill\|\.name:.*matmul" /tmp/probe/mm16.s | paste - - - | awk '{print $2, $4, $6}' | grep "Li8"
awk '/^_ZN12_GLOBAL__N_124chan_matmul_fwd16_kernelILi8ELi7/,/s_endpgm/' /tmp/probe/mm16.s > /tmp/probe/f87.s
