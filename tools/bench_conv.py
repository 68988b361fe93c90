"""Micro-benchmark of the implicit-GEMM conv kernels on the DeNet-34 skip layer shapes (B=32, 512x512)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from denet_amd import ops

B = int(os.environ.get("B", 32))
# (name, H, W, C, K, R, stride, pad, count)
SHAPES = [
    ("stem7x7", 512, 512, 4, 64, 7, 2, 3, 1),
    ("l1_3x3", 128, 128, 64, 64, 3, 1, 1, 6),
    ("l2_3x3s2", 128, 128, 64, 128, 3, 2, 1, 1),
    ("l2_3x3", 64, 64, 128, 128, 3, 1, 1, 7),
    ("l3_3x3s2", 64, 64, 128, 256, 3, 2, 1, 1),
    ("l3_3x3", 32, 32, 256, 256, 3, 1, 1, 11),
    ("l4_3x3s2", 32, 32, 256, 512, 3, 2, 1, 1),
    ("l4_3x3", 16, 16, 512, 512, 3, 1, 1, 5),
    ("up1_3x3", 32, 32, 512, 256, 3, 1, 1, 1),
    ("up2_3x3", 64, 64, 256, 128, 3, 1, 1, 1),
    ("dnc_1x1", 64, 64, 128, 128, 1, 1, 0, 1),
    ("head1", 24, 24, 4736, 1536, 1, 1, 0, 1),
    ("head2", 24, 24, 1536, 1024, 1, 1, 0, 1),
    ("head3", 24, 24, 1024, 768, 1, 1, 0, 1),
    ("head4", 24, 24, 768, 512, 1, 1, 0, 1),
]

def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

tot = {"fwd": 0, "dgrad": 0, "wgrad": 0}
totflop = 0
for name, H, W, C, K, R, st, pad, cnt in SHAPES:
    S = 8 if C == 4 else R
    x = torch.randn(B, H, W, C, device="cuda")
    w = torch.randn(K, R, S, C, device="cuda") * 0.05
    OH = (H + 2 * pad - R) // st + 1
    dy = torch.randn(B, OH, OH, K, device="cuda")
    flop = 2.0 * B * OH * OH * K * R * R * (3 if C == 4 else C)
    t_f = timeit(lambda: ops.conv_fwd(x, w, stride=st, pad=pad, s_real=R))
    t_w = timeit(lambda: ops.conv_wgrad(x, dy, tuple(w.shape), stride=st, pad=pad, s_real=R))
    if C >= 32:
        t_d = timeit(lambda: ops.conv_dgrad(dy, w, tuple(x.shape), stride=st, pad=pad, s_real=R))
    else:
        t_d = 0.0
    print("%-10s fwd %7.3f ms %6.1f TF | dgrad %7.3f ms %6.1f TF | wgrad %7.3f ms %6.1f TF  (x%d)" % (
        name, t_f, flop / t_f / 1e9, t_d, (flop / t_d / 1e9 if t_d else 0), t_w, flop / t_w / 1e9, cnt), flush=True)
    g = ops.conv_geom(x.shape, w.shape, st, pad, R)
    for tile in (2, 4):
        if ops.conv_wino_ok(g, tile):
            tw_f = timeit(lambda: ops.conv_wino_fwd(x, w, tile=tile))
            tw_d = timeit(lambda: ops.conv_wino_dgrad(dy, w, tile=tile))
            tw_w = timeit(lambda: ops.conv_wino_wgrad(x, dy, tile=tile))
            print("           winograd F%d: fwd %7.3f ms (x%.2f) | dgrad %7.3f ms (x%.2f) | wgrad %7.3f ms (x%.2f)" % (tile, tw_f, t_f / tw_f, tw_d, t_d / tw_d, tw_w, t_w / tw_w), flush=True)
    if ops.AUTOTUNE:
        import ctypes
        cfg = []
        for mode in (0, 1, 2):
            v = [ctypes.c_int(-1) for _ in range(3)]
            ops._L().denet_conv_tuned(mode, B, H, W, C, K, R, S, R, st, pad, *[ctypes.byref(a) for a in v])
            cfg.append("%s/n%d/r%d" % ({0: "128x128", 1: "128x64", -1: "-"}[v[0].value], v[1].value, v[2].value))
        print("           tuned: fwd %s | dgrad %s | wgrad %s" % tuple(cfg), flush=True)
    tot["fwd"] += t_f * cnt; tot["dgrad"] += t_d * cnt; tot["wgrad"] += t_w * cnt
    totflop += flop * cnt
print("total ms:", tot, "sum %.1f ms" % sum(tot.values()), " fwd GFLOP %.1f" % (totflop / 1e9))
print("avg TF fwd %.1f dgrad %.1f wgrad %.1f" % (totflop / tot["fwd"] / 1e9, totflop / tot["dgrad"] / 1e9, totflop / tot["wgrad"] / 1e9))
print("img/s bound from conv only: %.1f" % (B / (sum(tot.values()) / 1e3)))
