"""Builds libdenet_hip.so (the C-ABI of include/denet_hip.h) for gfx950 with hipcc.

The library is built IN-TREE (denet_amd/csrc/libdenet_hip.so) so that it travels with the repository
snapshot to the GPU box; hipcc cross-compiles gfx950 code objects without a GPU present.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libdenet_hip.so")

SOURCES = ["runtime.hip", "igemm.hip", "bn.hip", "pool.hip", "elementwise.hip", "dss.hip", "samples.hip", "detect.hip", "winograd.hip", "wino2f.hip", "wino4f.hip", "wino4t.hip", "dgrad_s2.hip", "wino4g.hip", "stem.hip", "gemm3b.hip",
           "augment.hip", "image.hip"]
# files whose integer results must not depend on FMA contraction
NO_CONTRACT = {"dss.hip", "samples.hip", "detect.hip", "augment.hip", "image.hip"}
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc_version(hipcc):
    try:
        return subprocess.run([hipcc, "--version"], check=True, capture_output=True, text=True).stdout
    except (OSError, subprocess.CalledProcessError):
        return "unknown"


def _digest(parts, files):
    """content hash of (command line, compiler version, every input file): mtimes say nothing after an rsync / checkout"""
    h = hashlib.sha256()
    for p in parts:
        h.update(p.encode())
        h.update(b"\0")
    for f in files:
        with open(f, "rb") as fh:
            h.update(hashlib.sha256(fh.read()).digest())
    return h.hexdigest()


def _stale(target, digest):
    """a target is current iff it exists and the digest recorded next to it (<target>.sha256) is the one of its inputs"""
    stamp = target + ".sha256"
    if not (os.path.exists(target) and os.path.exists(stamp)):
        return True
    with open(stamp) as f:
        return f.read().strip() != digest


def _stamp(target, digest):
    with open(target + ".sha256", "w") as f:
        f.write(digest + "\n")


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    version = _hipcc_version(hipcc)
    objs, todo = [], []
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "bn_final.h"), os.path.join(HERE, "..", "include", "denet_hip.h")]
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        flags = COMMON + (["-ffp-contract=off"] if src in NO_CONTRACT else [])
        digest = _digest([version] + flags, [s] + headers)
        if force or _stale(o, digest):
            todo.append(([hipcc] + flags + ["-c", s, "-o", o], o, digest))
    # the stale translation units side by side (one hipcc process each; the largest files first)
    todo.sort(key=lambda t: -os.path.getsize(t[0][-3]))
    running = []
    jobs = max(1, min(len(todo), int(os.environ.get("DENET_BUILD_JOBS", os.cpu_count() or 1))))

    def reap(block):
        for ent in list(running):
            proc, cmd, o, digest = ent
            rc = proc.wait() if block else proc.poll()
            if rc is None:
                continue
            running.remove(ent)
            if rc != 0:
                for other in running:
                    other[0].kill()
                raise subprocess.CalledProcessError(rc, cmd)
            _stamp(o, digest)
            if block:
                return

    for cmd, o, digest in todo:
        while len(running) >= jobs:
            reap(block=True)
        if verbose:
            print(" ".join(cmd), flush=True)
        running.append((subprocess.Popen(cmd), cmd, o, digest))
    while running:
        reap(block=True)
    link = ["--offload-arch=gfx950", "-shared", "-fPIC"]
    digest = _digest([version] + link, objs)
    if force or _stale(LIB, digest):
        cmd = [hipcc] + link + ["-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        _stamp(LIB, digest)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
