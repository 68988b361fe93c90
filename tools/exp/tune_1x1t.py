"""times the four launch configurations of the forward kernel on the geometries of the head layers' data gradient
(ops.conv_dgrad over the transposed filter): records for denet_amd/tuned/gfx950.json"""
import ctypes
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from denet_amd import ops
from denet_amd.lib import load

L = load()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


ops._load_tuned_once()
for (N, C, K) in ((32, 4736, 1536), (32, 1536, 1024), (16, 4736, 1536), (16, 1536, 1024), (64, 4736, 1536), (64, 1536, 1024)):
    dy = torch.randn(N, 24, 24, K, device="cuda")
    w = torch.randn(K, 1, 1, C, device="cuda") * 0.02
    dx = torch.empty(N, 24, 24, C, device="cuda")
    key = [0, N, 24, 24, K, C, 1, 1, 1, 1, 0]
    ops._TUNED.add((0, ops.conv_geom((N, 24, 24, K), (C, 1, 1, K), 1, 0, None)))
    res = []
    for tile, nbuf in ((0, 1), (0, 2), (1, 1), (1, 2)):
        rec = (ctypes.c_int * 14)(*(key + [tile, nbuf, 0]))
        assert L.denet_tune_import(rec, 1) == 0
        res.append((round(timed(lambda: ops.conv_dgrad(dy, w, (N, 24, 24, C), stride=1, pad=0, cache={}, out=dx))), tile, nbuf))
    old = ops.DGRAD_1X1T_GFLOP
    ops.DGRAD_1X1T_GFLOP = 0.0
    plain = round(timed(lambda: ops.conv_dgrad(dy, w, (N, 24, 24, C), stride=1, pad=0, cache={}, out=dx)))
    ops.DGRAD_1X1T_GFLOP = old
    print(key, "us, tile, nbuf:", sorted(res), "data-gradient kernel as tuned:", plain)
