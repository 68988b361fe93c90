"""fused 64-channel F(2x2) kernel (csrc/wino2f.hip) at the benchmark size: us per launch of the forward pass"""
import ctypes
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from denet_amd import ops
from denet_amd.lib import load, ptr, check, stream_ptr

L = load()
N, H, W, C = 32, 128, 128, 64
x = torch.randn(N, H, W, C, device="cuda")
w = torch.randn(C, 3, 3, C, device="cuda") * 0.05
u = ops.conv_wino_filter(w, 2, dgrad=False)
y = torch.empty(N, H, W, C, device="cuda")
add = torch.randn(N, H, W, C, device="cuda")
st = torch.zeros(1 << 20, dtype=torch.float64, device="cuda")
rows = ctypes.c_int(0)


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


def plain():
    check(L.denet_conv_wino2f(ptr(x), ptr(u), None, None, ptr(y), 0, None, 0, ctypes.byref(rows), N, H, W, C, C, stream_ptr()))


def full():
    check(L.denet_conv_wino2f(ptr(x), ptr(u), None, ptr(add), ptr(y), 0, ptr(st), st.numel() * 8, ctypes.byref(rows), N, H, W, C, C, stream_ptr()))


print("forward: plain %.1f us  add + statistics %.1f us" % (timed(plain), timed(full)))


# the data-gradient form: accumulated add + the backward sums of a batch norm (input bx, output by, ReLU)
from denet_amd.ops import BnSums
bx, by_ = torch.randn(N, H, W, C, device="cuda"), torch.randn(N, H, W, C, device="cuda")
g1, b1, m1, i1 = (torch.rand(C, device="cuda") + 0.5), torch.randn(C, device="cuda"), torch.randn(C, device="cuda"), torch.rand(C, device="cuda") + 0.5
for yv in (None, by_):
    sums = BnSums(bx, yv, g1, b1, m1, i1, True)
    cs = sums.c_struct()

    def dg():
        check(L.denet_conv_wino2f_sums(ptr(x), ptr(u), None, ptr(add), ptr(y), 0, ptr(st), st.numel() * 8, ctypes.byref(rows), ctypes.byref(cs), N, H, W, C,
                                       C, stream_ptr()))
    print("  data gradient + add + backward sums (%s): %.1f us" % ("mask from y" if yv is not None else "mask recomputed", timed(dg)))
