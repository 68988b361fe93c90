#!/bin/bash
# debug build of csrc/wino4f.hip with the in-kernel s_memtime stamps (-DW4_TRACE) linked into libdenet_hip.so;
# `rm denet_amd/csrc/wino4f.o.sha256; python denet_amd/build.py` restores the product build
cd "$(dirname "$0")/../.."
C=denet_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DW4_TRACE $1 -c $C/wino4f.hip -o $C/wino4f.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libdenet_hip.so $C/runtime.o $C/igemm.o $C/bn.o $C/pool.o $C/elementwise.o $C/dss.o $C/samples.o $C/detect.o $C/winograd.o $C/wino2f.o $C/wino4f.o $C/wino4g.o $C/stem.o $C/gemm3b.o $C/augment.o $C/image.o
