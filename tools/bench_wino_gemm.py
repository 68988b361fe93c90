"""Times the Winograd F(4x4,3x3) forward / data-gradient passes of the backbone stages (batched component GEMMs dominate)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from denet_amd import ops
def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
B = 32
torch.manual_seed(0)
ONLY = os.environ.get("SHAPE")
for name, HW, C, K in [("l1", 128, 64, 64), ("l2", 64, 128, 128), ("l3", 32, 256, 256), ("l4", 16, 512, 512), ("up1", 32, 512, 256), ("up2", 64, 256, 128)]:
    if ONLY and name != ONLY: continue
    x = torch.randn(B, HW, HW, C, device="cuda"); w = torch.randn(K, 3, 3, C, device="cuda") * 0.05
    dy = torch.randn(B, HW, HW, K, device="cuda")
    u = ops.conv_wino_filter(w, 4, dgrad=False); ud = ops.conv_wino_filter(w, 4, dgrad=True)
    y = ops.conv_wino_fwd(x, w, tile=4, u=u)
    t = timeit(lambda: ops.conv_wino_fwd(x, w, tile=4, u=u))
    td = timeit(lambda: ops.conv_wino_dgrad(dy, w, tile=4, u=ud))
    print("%-4s fwd %.3f ms dgrad %.3f ms  checksum %.6e" % (name, t, td, float(y.double().sum())), flush=True)
