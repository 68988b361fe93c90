"""the data gradient of the 1x1 head convolutions as the data-gradient mode runs it (the filter read k-major) and as a FORWARD
pass over the transposed filter (ops.conv_dgrad, DGRAD_1X1T_GFLOP): interleaved, so that both see the same clocks"""
import ctypes
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from denet_amd import ops
from denet_amd.lib import load

L = load()


def timed(fn, n=10):
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
torch.manual_seed(0)
for (C, K) in ((4736, 1536), (1536, 1024), (1024, 512)):
    dy = torch.randn(32, 24, 24, K, device="cuda")
    w = torch.randn(K, 1, 1, C, device="cuda") * 0.02
    dx = torch.empty(32, 24, 24, C, device="cuda")
    key = [0, 32, 24, 24, K, C, 1, 1, 1, 1, 0]
    ops._TUNED.add((0, ops.conv_geom((32, 24, 24, K), (C, 1, 1, K), 1, 0, None)))
    rec = (ctypes.c_int * 14)(*(key + [0, 2, 0]))
    assert L.denet_tune_import(rec, 1) == 0
    ta, tb = [], []
    for rep in range(6):
        ops.DGRAD_1X1T_GFLOP = 0.0
        ta.append(timed(lambda: ops.conv_dgrad(dy, w, (32, 24, 24, C), stride=1, pad=0, cache={}, out=dx)))
        ops.DGRAD_1X1T_GFLOP = 1e-9
        tb.append(timed(lambda: ops.conv_dgrad(dy, w, (32, 24, 24, C), stride=1, pad=0, cache={}, out=dx)))
    print("C %5d K %5d: data-gradient mode %s us, forward over w^T (incl. the transposition) %s us"
          % (C, K, [round(t) for t in ta], [round(t) for t in tb]))
