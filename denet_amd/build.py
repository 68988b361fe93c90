"""Builds libdenet_hip.so (the C-ABI of include/denet_hip.h) for gfx950 with hipcc.

The library is built IN-TREE (denet_amd/csrc/libdenet_hip.so) so that it travels with the repository
snapshot to the GPU box; hipcc cross-compiles gfx950 code objects without a GPU present.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libdenet_hip.so")

SOURCES = ["runtime.hip", "igemm.hip", "bgemm.hip", "bn.hip", "pool.hip", "elementwise.hip", "dss.hip", "samples.hip", "detect.hip", "winograd.hip", "wino2f.hip", "gemm3b.hip",
           "augment.hip", "image.hip"]
# files whose integer results must not depend on FMA contraction
NO_CONTRACT = {"dss.hip", "samples.hip", "detect.hip", "augment.hip", "image.hip"}
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    objs = []
    headers = [os.path.join(CSRC, "common.h")]
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + COMMON + (["-ffp-contract=off"] if src in NO_CONTRACT else []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
