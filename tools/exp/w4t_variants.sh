#!/bin/bash
# experiment builds of csrc/wino4t.hip linked into copies of libdenet_hip.so under tools/exp/_w4t/ (git-ignored, travels with
# gpurun): every argument is name:flags, e.g. texp5:-DT_EXP=5 (T_EXP bits: 1 no input transform, 2 no products, 4 no epilogue,
# 8 no LDS-DMA, 16 stores into a 1 MB window) or d5:-DT_D_=5; W4T_LIB=tools/exp/_w4t/libdenet_hip_<name>.so python tools/exp/w4t_check.py
cd "$(dirname "$0")/../.."
C=denet_amd/csrc
mkdir -p tools/exp/_w4t
OBJS=$(ls $C/*.o | grep -v wino4t.o)
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function ${flags//,/ } -c $C/wino4t.hip -o tools/exp/_w4t/wino4t_$name.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/_w4t/libdenet_hip_$name.so $OBJS tools/exp/_w4t/wino4t_$name.o || exit 1
done
ls tools/exp/_w4t/*.so
