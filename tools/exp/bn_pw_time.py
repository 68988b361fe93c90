"""the pointwise batch-norm passes (forward apply, backward apply) on the step's tensor sizes: us and TB/s"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from denet_amd import ops


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


torch.manual_seed(0)
for shape in ((32, 24, 24, 1536), (32, 24, 24, 1024), (32, 64, 64, 128), (32, 32, 32, 256), (32, 128, 128, 64), (32, 16, 16, 512)):
    C = shape[-1]
    x = torch.randn(shape, device="cuda")
    dy = torch.randn(shape, device="cuda")
    gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
    rm, rs = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    y, sm, si = ops.bn_fwd_train(x, gamma, beta, rm, rs, relu=True)
    mb = x.numel() * 4 / 1e6
    tf = timed(lambda: ops.bn_fwd_train(x, gamma, beta, rm, rs, relu=True))
    tb = timed(lambda: ops.bn_bwd(x, None, dy, gamma, sm, si, relu=True, beta=beta))
    print("%s (%.0f MB): forward statistics + apply %.0f us, backward sums + apply %.0f us" % (shape, mb, tf, tb))
