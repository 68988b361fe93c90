"""csrc/dgrad_s2.hip (3x3 stride-2 data gradient, four parity classes per workgroup) against the implicit-GEMM kernel and fp64:
difference and time per launch at the three geometries of DeNet-34 (B = 32). usage: python tools/exp/s2_check.py [B]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as Fn
from denet_amd import ops
from denet_amd.lib import load, ptr, stream_ptr, check

L = load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
GEOMS = [("l2", 128, 128, 64, 128), ("l3", 64, 64, 128, 256), ("l4", 32, 32, 256, 512)]


def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def s2(dy, pk, add, shape, sums=None, st=None):
    N, H, W, C = shape
    K = dy.shape[3]
    dx = torch.empty(N, H, W, C, device="cuda")
    rows = ctypes.c_int(0)
    so = sums.c_struct() if sums is not None else None
    check(L.denet_conv_dgrad_s2(ptr(dy), ptr(pk), ptr(add), ptr(dx), ctypes.byref(so) if so is not None else None, ptr(st),
                                st.numel() * 8 if st is not None else 0, ctypes.byref(rows), N, H, W, C, K, stream_ptr()), "s2")
    return dx, rows.value


for name, H, W, C, K in GEOMS:
    torch.manual_seed(1)
    w = torch.randn(K, 3, 3, C, device="cuda") * 0.05
    dy = torch.randn(B, H // 2, W // 2, K, device="cuda")
    add = torch.randn(B, H, W, C, device="cuda")
    pk = torch.empty(9 * K * C, device="cuda")
    check(L.denet_conv_dgrad_s2_pack(ptr(w), ptr(pk), C, K, stream_ptr()), "pack")
    # fp64 on two images
    xd = torch.zeros(2, C, H, W, dtype=torch.float64, requires_grad=True)
    y = Fn.conv2d(xd, w.double().cpu().permute(0, 3, 1, 2), None, stride=2, padding=1)
    ref = torch.autograd.grad(y, xd, dy[:2].double().cpu().permute(0, 3, 1, 2))[0].permute(0, 2, 3, 1)
    dx2, _ = s2(dy[:2].contiguous(), pk, None, (2, H, W, C))
    e64 = float((dx2.double().cpu() - ref).abs().max() / ref.abs().max())
    # the implicit-GEMM kernel, full batch, with add
    g = ops.conv_geom((B, H, W, C), w.shape, 2, 1)
    dxi = torch.empty(B, H, W, C, device="cuda")
    check(L.denet_conv_dgrad(ptr(dy), ptr(w), ptr(add), ptr(dxi), *g, stream_ptr()), "igemm")
    dxs, _ = s2(dy, pk, add, (B, H, W, C))
    ei = float((dxs - dxi).abs().max() / dxi.abs().max())
    # backward sums
    xb = torch.randn(B, H, W, C, device="cuda"); gam = torch.rand(C, device="cuda") + 0.5; bet = torch.randn(C, device="cuda")
    mu = xb.reshape(-1, C).mean(0); isd = 1.0 / xb.reshape(-1, C).std(0); yb = torch.relu((xb - mu) * isd * gam + bet)
    bsum = ops.BnSums(xb, yb, gam, bet, mu, isd, True)
    st = torch.zeros(L.denet_conv_dgrad_s2_stats_rows(B, H, W) * 2 * C, dtype=torch.float64, device="cuda")
    dxs2, rows = s2(dy, pk, add, (B, H, W, C), sums=bsum, st=st)
    gq = torch.where(yb > 0, dxs2, torch.zeros_like(dxs2)).double().reshape(-1, C)
    xh = ((xb - mu) * isd).double().reshape(-1, C)
    s_ = st.view(rows, 2, C).sum(0)
    es = (float((s_[0] - gq.sum(0)).abs().max() / gq.abs().sum(0).max()), float((s_[1] - (gq * xh).sum(0)).abs().max() / (gq * xh).abs().sum(0).max()))
    t_i = timeit(lambda: check(L.denet_conv_dgrad(ptr(dy), ptr(w), ptr(add), ptr(dxi), *g, stream_ptr()), "igemm"))
    t_s = timeit(lambda: s2(dy, pk, add, (B, H, W, C)))
    t_s2 = timeit(lambda: s2(dy, pk, add, (B, H, W, C), sums=bsum, st=st))
    flops = 2.0 * 9 * B * (H // 2) * (W // 2) * C * K
    print("%-3s vs fp64 %.2e  vs igemm %.2e  equal with sums %s  sums %.1e %.1e | igemm %.1f us (%.0f TF/s)  dgrad_s2 %.1f us (%.0f TF/s)  + sums %.1f us" % (
        name, e64, ei, bool(torch.equal(dxs, dxs2)), es[0], es[1], t_i, flops / t_i / 1e6, t_s, flops / t_s / 1e6, t_s2), flush=True)
