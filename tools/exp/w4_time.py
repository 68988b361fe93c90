"""Fused F(4x4) product + output-transform kernel (csrc/wino4f.hip) against the un-fused passes: same inputs, difference and
time per pass (every kernel alone on the chip). usage: python tools/exp/w4_time.py [B]"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from denet_amd import ops
from denet_amd.lib import load

L = load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
GEOMS = [("l2_3x3", 64, 64, 128, 128), ("up2_3x3", 64, 64, 256, 128), ("l3_3x3", 32, 32, 256, 256), ("up1_3x3", 32, 32, 512, 256),
         ("l4_3x3", 16, 16, 512, 512)]
if os.environ.get("W4_GEOMS"):
    GEOMS = [g for g in GEOMS if g[0] in os.environ["W4_GEOMS"].split(",")]


def timeit(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for name, H, W, C, K in GEOMS:
    torch.manual_seed(1)
    x = torch.randn(B, H, W, C, device="cuda")
    w = torch.randn(K, 3, 3, C, device="cuda") * 0.05
    bias = torch.randn(K, device="cuda")
    add = torch.randn(B, H, W, K, device="cuda")
    dy = torch.randn(B, H, W, K, device="cuda")
    u = ops.conv_wino_filter(w, 4, dgrad=False)
    xb = torch.randn(B, H, W, C, device="cuda"); gam = torch.rand(C, device="cuda") + 0.5; bet = torch.randn(C, device="cuda")
    mu = xb.reshape(-1, C).mean(0); isd = 1.0 / xb.reshape(-1, C).std(0); yb = torch.relu((xb - mu) * isd * gam + bet)
    accd = torch.randn(B, H, W, C, device="cuda")
    ud = ops.conv_wino_filter(w, 4, dgrad=True)
    res = {}
    for mode in (0, 64, 32, 33, 34):
        L.denet_conv_wino4f_mode(mode)
        cache = {}
        st = torch.zeros(1 << 22, dtype=torch.float64, device="cuda")
        y = ops.conv_wino_fwd(x, w, bias, add, tile=4, u=u, stats=(st, cache))
        stt, rows = cache["bn_stats"]
        sums = stt[: rows * 2 * K].view(rows, 2, K).sum(0).clone()
        dx = ops.conv_wino_dgrad(dy, w, tile=4, u=ud)
        tf = timeit(lambda: ops.conv_wino_fwd(x, w, bias, add, tile=4, u=u, stats=(st, cache)))
        td = timeit(lambda: ops.conv_wino_dgrad(dy, w, tile=4, u=ud))
        # the fused kernel alone (event pair around its launch)
        kf = kd = 0.0
        if mode:
            ms, v = ctypes.c_float(), [ctypes.c_int() for _ in range(4)]
            for which in (0, 1):
                L.denet_conv_profile(1)
                for _ in range(5):
                    if which == 0:
                        if os.environ.get("W4_STEP_LIKE"):      # what a training step runs: statistics, no bias / add
                            ops.conv_wino_fwd(x, w, None, None, tile=4, u=u, stats=(st, cache))
                        else:
                            ops.conv_wino_fwd(x, w, bias, add, tile=4, u=u, stats=(st, cache))
                    elif os.environ.get("W4_STEP_LIKE"):        # ... the backward sums of the batch norm in front + the accumulated add
                        bsum = ops.BnSums(xb, yb, gam, bet, mu, isd, True)
                        ops.conv_wino_dgrad(dy, w, add=accd, tile=4, u=ud, sums=bsum, cache={})
                    else:
                        ops.conv_wino_dgrad(dy, w, tile=4, u=ud)
                torch.cuda.synchronize()
                tot = 0.0
                for i in range(L.denet_conv_profile_count()):
                    L.denet_conv_profile_read(i, ctypes.byref(ms), *[ctypes.byref(a) for a in v])
                    if v[0].value == 14:
                        tot += ms.value
                L.denet_conv_profile(0)
                if which == 0:
                    kf = tot / 5 * 1e3
                else:
                    kd = tot / 5 * 1e3
        res[mode] = (y.clone(), dx.clone(), sums, tf, td, rows, kf, kd)
    L.denet_conv_wino4f_mode(-1)
    y0, dx0, s0, tf0, td0, r0 = res[0][:6]
    flop = 2.0 * B * H * W * C * K * 9 / 4
    line = "%-8s B=%d  unfused fwd %6.1f us dgrad %6.1f us |" % (name, B, tf0, td0)
    for mode in (64, 32, 33, 34):
        y1, dx1, s1, tf1, td1, r1, kf, kd = res[mode]
        ey = float((y1 - y0).abs().max() / y0.abs().max())
        ed = float((dx1 - dx0).abs().max() / dx0.abs().max())
        es = float(((s1 - s0).abs() / (s0.abs() + 1e-3 * s0.abs().max())).max())
        line += " TB%d fwd %6.1f (kernel %5.1f = %3.0f TF) dgrad %6.1f (kernel %5.1f = %3.0f TF) err y %.1e dx %.1e sums %.1e |" % (
            mode, tf1, kf, flop / max(kf, 1e-3) / 1e6, td1, kd, flop / max(kd, 1e-3) / 1e6, ey, ed, es)
    print(line, flush=True)
