"""F(4x4) filter-gradient kernel (csrc/wino4g.hip) against the generic batched split-K path: same inputs, difference against fp64
and time (whole pass incl. both transforms; products alone from the event pairs around their launches).
usage: python tools/exp/w4g_time.py [B]"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as Fn
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


def products_us(fn, want_mode):
    ms, v = ctypes.c_float(), [ctypes.c_int() for _ in range(4)]
    L.denet_conv_profile(1)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    tot = 0.0
    for i in range(L.denet_conv_profile_count()):
        L.denet_conv_profile_read(i, ctypes.byref(ms), *[ctypes.byref(a) for a in v])
        if v[0].value == want_mode:
            tot += ms.value
    L.denet_conv_profile(0)
    return tot / 5 * 1e3


for name, H, W, C, K in GEOMS:
    torch.manual_seed(1)
    x = torch.randn(B, H, W, C, device="cuda")
    dy = torch.randn(B, H, W, K, device="cuda")
    # fp64 reference of the filter gradient (a few images at a time)
    ref = torch.zeros(K, C, 3, 3, dtype=torch.float64, device="cuda")
    for i in range(0, B, 4):
        xd = x[i:i + 4].double().permute(0, 3, 1, 2).contiguous()
        wd = torch.zeros(K, C, 3, 3, dtype=torch.float64, device="cuda", requires_grad=True)
        y = Fn.conv2d(xd, wd, None, padding=1)
        ref += torch.autograd.grad(y, wd, dy[i:i + 4].double().permute(0, 3, 1, 2))[0]
    ref = ref.permute(0, 2, 3, 1)
    flop = 2.0 * B * H * W * C * K * 9 / 4
    line = "%-8s B=%d |" % (name, B)
    for mode in (0, 1):
        L.denet_conv_wino4g_mode(mode)
        fn = lambda: ops.conv_wino_wgrad(x, dy, tile=4)
        dw = fn().clone()
        err = float((dw.double() - ref).abs().max() / ref.abs().max())
        t = timeit(fn)
        k = products_us(fn, 15 if mode else 2)
        line += " %s pass %6.1f us, products %6.1f us (%3.0f TF), err %.1e |" % ("wino4g " if mode else "generic", t, k, flop / max(k, 1e-3) / 1e6, err)
    L.denet_conv_wino4g_mode(-1)
    print(line, flush=True)
