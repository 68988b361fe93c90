#!/bin/bash
# experiment builds of csrc/wino4t.hip (-DT_EXP=bits: 1 no input transform, 2 no products, 4 no epilogue, 8 no LDS-DMA) linked into
# copies of libdenet_hip.so under tools/exp/_w4t/ (git-ignored, travels with gpurun); W4T_LIB=<path> python tools/exp/w4t_check.py
cd "$(dirname "$0")/../.."
C=denet_amd/csrc
mkdir -p tools/exp/_w4t
OBJS=$(ls $C/*.o | grep -v wino4t.o)
for bits in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DT_EXP=$bits -c $C/wino4t.hip -o tools/exp/_w4t/wino4t_$bits.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/_w4t/libdenet_hip_texp$bits.so $OBJS tools/exp/_w4t/wino4t_$bits.o || exit 1
done
ls -la tools/exp/_w4t/*.so
