"""Tile-parallel fused F(4x4) kernel (csrc/wino4t.hip) against the fused F(2x2) kernel (wino2f.hip) and the direct kernel on the
64-channel stage's geometry: difference to an fp64 reference on a small batch, difference between the implementations at the
full batch, statistics / backward-sums epilogues, time per launch (each kernel alone on the chip).
usage: python tools/exp/w4t_check.py [B]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import denet_amd.lib as _lib
if os.environ.get("W4T_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["W4T_LIB"])          # an experiment build (tools/exp/w4t_variants.sh)
from denet_amd import ops
from denet_amd.lib import load, ptr, stream_ptr, check

L = load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
GEOMS = [("s1_128", 128, 128, 64, 64), ("s1_64", 64, 64, 64, 64), ("c128_64", 64, 64, 128, 128), ("c256_32", 32, 32, 256, 256)]
if os.environ.get("W4T_GEOMS"):
    GEOMS = [g for g in GEOMS if g[0] in os.environ["W4T_GEOMS"].split(",")]


def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def packed_u(w, dgrad):
    K, _, _, C = w.shape
    u = ops.conv_wino_filter(w, 4, dgrad=dgrad)          # [36][out][red]
    out_c, red_c = (C, K) if dgrad else (K, C)
    pk = torch.empty_like(u)
    check(L.denet_conv_wino4t_pack(ptr(u), ptr(pk), red_c, out_c, stream_ptr()), "pack")
    return pk


def w4t(x, pk, bias, add, relu, K, stats=None, sums=None):
    N, H, W, C = x.shape
    y = torch.empty(N, H, W, K, device="cuda")
    rows = ctypes.c_int(0)
    so = sums.c_struct() if sums is not None else None
    check(L.denet_conv_wino4t_sums(ptr(x), ptr(pk), ptr(bias), ptr(add), ptr(y), int(relu), ptr(stats), stats.numel() * 8 if stats is not None else 0,
                                   ctypes.byref(rows), ctypes.byref(so) if so is not None else None, N, H, W, C, K, stream_ptr()), "w4t")
    return y, rows.value


def ref64(x, w, bias, add, relu):
    xx = x.double().cpu().permute(0, 3, 1, 2)
    ww = w.double().cpu().permute(0, 3, 1, 2)
    y = torch.nn.functional.conv2d(xx, ww, None, 1, 1).permute(0, 2, 3, 1)
    if bias is not None:
        y = y + bias.double().cpu()
    if add is not None:
        y = y + add.double().cpu()
    return torch.relu(y) if relu else y


def relerr(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max())


for name, H, W, C, K in GEOMS:
    torch.manual_seed(1)
    x = torch.randn(B, H, W, C, device="cuda")
    w = torch.randn(K, 3, 3, C, device="cuda") * 0.05
    bias = torch.randn(K, device="cuda")
    add = torch.randn(B, H, W, K, device="cuda")
    dy = torch.randn(B, H, W, K, device="cuda")
    pk = packed_u(w, False)
    pkd = packed_u(w, True)
    # ---- fp64 reference on two images (forward with bias / add / relu; plain)
    r = ref64(x[:2], w, bias, add[:2], True)
    y, _ = w4t(x[:2].contiguous(), pk, bias, add[:2].contiguous(), True, K)
    e1 = relerr(y, r)
    r = ref64(x[:2], w, None, None, False)
    y, _ = w4t(x[:2].contiguous(), pk, None, None, False, K)
    e2 = relerr(y, r)
    # data gradient: dx = conv(dy, rot180(w) with channels swapped)
    wd = w.flip(1, 2).permute(3, 1, 2, 0).contiguous()
    r = ref64(dy[:2], wd, None, None, False)
    dx, _ = w4t(dy[:2].contiguous(), pkd, None, None, False, C)
    e3 = relerr(dx, r)
    print("%-8s fp64 (2 images): fwd+bias+add+relu %.2e  fwd %.2e  dgrad %.2e" % (name, e1, e2, e3), flush=True)
    # ---- full batch against the direct kernel; statistics epilogue
    yd = ops.empty(B, H, W, K)
    g = ops.conv_geom(x.shape, w.shape, 1, 1)
    check(L.denet_conv_fwd_act(ptr(x), ptr(w), ptr(bias), ptr(add), ptr(yd), 0, *g, stream_ptr()), "direct")
    st = torch.zeros(L.denet_conv_wino4t_stats_rows(B, H, W) * 2 * K, dtype=torch.float64, device="cuda")
    y, rows = w4t(x, pk, bias, add, False, K, stats=st)
    sums = st.view(rows, 2, K).sum(0)
    yy = y.double().reshape(-1, K)
    es = float((sums[0] - yy.sum(0)).abs().max() / yy.sum(0).abs().max())
    eq = float((sums[1] - (yy * yy).sum(0)).abs().max() / (yy * yy).sum(0).abs().max())
    print("         full batch vs direct %.2e   stats rows %d: sum %.1e  sum sq %.1e" % (relerr(y, yd), rows, es, eq), flush=True)
    # backward sums epilogue (EP 2): dx = gradient of the output of a BN+ReLU layer with input xb
    xb = torch.randn(B, H, W, C, device="cuda"); gam = torch.rand(C, device="cuda") + 0.5; bet = torch.randn(C, device="cuda")
    mu = xb.reshape(-1, C).mean(0); isd = 1.0 / xb.reshape(-1, C).std(0); yb = torch.relu((xb - mu) * isd * gam + bet)
    accd = torch.randn(B, H, W, C, device="cuda")
    bsum = ops.BnSums(xb, yb, gam, bet, mu, isd, True)
    st2 = torch.zeros(L.denet_conv_wino4t_stats_rows(B, H, W) * 2 * C, dtype=torch.float64, device="cuda")
    dx, rows = w4t(dy, pkd, None, accd, False, C, stats=st2, sums=bsum)
    gq = torch.where(yb > 0, dx, torch.zeros_like(dx)).double().reshape(-1, C)
    xh = ((xb - mu) * isd).double().reshape(-1, C)
    s2 = st2.view(rows, 2, C).sum(0)
    print("         backward sums: sum(g) %.1e  sum(g xhat) %.1e" % (float((s2[0] - gq.sum(0)).abs().max() / gq.sum(0).abs().max()),
                                                                     float((s2[1] - (gq * xh).sum(0)).abs().max() / (gq * xh).sum(0).abs().max())), flush=True)
    # ---- time
    t_f = timeit(lambda: w4t(x, pk, None, None, False, K, stats=st))
    t_d = timeit(lambda: w4t(dy, pkd, None, accd, False, C, stats=st2, sums=bsum))
    t_p = timeit(lambda: w4t(x, pk, None, None, False, K))
    flops = 2.0 * 36 * (B * H * W / 16) * C * K
    line = "         wino4t: fwd+stats %.1f us (%.1f TF/s)  dgrad+add+sums %.1f us  plain %.1f us" % (t_f, flops / t_f / 1e6, t_d, t_p)
    if C == 64:
        u2 = ops.conv_wino_filter(w, 2, dgrad=False)
        u2d = ops.conv_wino_filter(w, 2, dgrad=True)
        stw = torch.zeros(1 << 22, dtype=torch.float64, device="cuda")
        cache = {}
        t2f = timeit(lambda: ops.conv_wino_fwd(x, w, None, None, tile=ops.FUSED2, u=u2, stats=(stw, cache)))
        t2d = timeit(lambda: ops.conv_wino_dgrad(dy, w, add=accd, tile=ops.FUSED2, u=u2d, sums=bsum, cache={}))
        line += "   | wino2f: fwd+stats %.1f us  dgrad+add+sums %.1f us" % (t2f, t2d)
    print(line, flush=True)
