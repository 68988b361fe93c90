import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from denet_amd import ops
os.environ.setdefault("X", "1")
def timeit(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for name, B, HW, C, K in [("l1", 32, 128, 64, 64), ("r34_l1", 64, 56, 64, 64)]:
    x = torch.randn(B, HW, HW, C, device="cuda"); dy = torch.randn(B, HW, HW, K, device="cuda")
    ref = None
    for tile in (4, 2):
        if HW % tile: continue
        dw = ops.conv_wino_wgrad(x, dy, tile=tile)
        t = timeit(lambda: ops.conv_wino_wgrad(x, dy, tile=tile))
        d = ops.conv_wgrad(x, dy, (K, 3, 3, C), stride=1, pad=1)
        err = float((dw - d).abs().max() / d.abs().max())
        print("%-7s F%d wgrad %.3f ms  (%s)  err vs direct %.1e" % (name, tile, t, ops._last_igemm_name(), err), flush=True)
