"""first-layer kernels (csrc/stem.hip) against the generic implicit-GEMM kernels at the benchmark size: us per launch"""
import ctypes
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from denet_amd import ops
from denet_amd.lib import load, ptr, check, stream_ptr

L = load()
N, H, W, K = 32, 512, 512, 64
OH, OW = H // 2, W // 2
x = torch.rand(N, H, W, 4, device="cuda")
w = torch.randn(K, 7, 8, 4, device="cuda") * 0.1
w[:, :, 7] = 0
w[..., 3] = 0
bias = torch.randn(K, device="cuda")
y = torch.empty(N, OH, OW, K, device="cuda")
dy = torch.randn(N, OH, OW, K, device="cuda")
dw = torch.empty(K, 7, 8, 4, device="cuda")
st = torch.zeros(1 << 22, dtype=torch.float64, device="cuda")
rows = ctypes.c_int(0)
ws = torch.empty(64 << 20, device="cuda")


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


def fwd():
    check(L.denet_conv_fwd_stats(ptr(x), ptr(w), ptr(bias), None, ptr(y), ptr(st), st.numel() * 8, ctypes.byref(rows), N, H, W, 4, K, 7, 8, 7, 2, 3,
                                 OH, OW, stream_ptr()))


def wgrad():
    check(L.denet_conv_wgrad(ptr(x), ptr(dy), ptr(dw), ptr(ws), ws.numel() * 4, N, H, W, 4, K, 7, 8, 7, 2, 3, OH, OW, stream_ptr()))


print("DENET_STEM=%s  fwd+stats %.1f us (rows %d)  wgrad %.1f us" % (os.environ.get("DENET_STEM", "3"), timed(fwd), rows.value, timed(wgrad)))
