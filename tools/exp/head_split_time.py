"""the first head GEMM (1x1 convolution 4736 -> 1536 on 32 x 24 x 24 RoIs) as one launch and as two over a batch split: the one
launch is 1728 tiles on 512 workgroup slots = 3.4 rounds (the last one a third full)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from denet_amd import ops

C, K = 4736, 1536
torch.manual_seed(0)
x = torch.randn(32, 24, 24, C, device="cuda")
w = torch.randn(K, 1, 1, C, device="cuda") * 0.02
y = torch.empty(32, 24, 24, K, device="cuda")


def timed(fn, n=10):
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


print("one launch: %.0f us" % timed(lambda: ops.conv_fwd(x, w, out=y, cache={"train": True}, bn_stats=True)))
for n1 in (28, 30, 24, 16):
    def split():
        ops.conv_fwd(x[:n1], w, out=y[:n1], cache={"train": True}, bn_stats=True)
        ops.conv_fwd(x[n1:], w, out=y[n1:], cache={"train": True}, bn_stats=True)
    print("split %d + %d: %.0f us" % (n1, 32 - n1, timed(split)))

