import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from denet_amd import ops
from wino2f_test import timeit
N, H, W, C = 32, 256, 256, 64
x = torch.randn(N, H, W, C, device="cuda")
gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.3
rm, rs = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
yp, arg, sm, si = ops.bn_relu_pool_fwd_train(x, gamma, beta, rm, rs, 3, 2, 1)
dyp = torch.randn_like(yp)
print("fwd fused  %.1f us" % timeit(lambda: ops.bn_relu_pool_fwd_train(x, gamma, beta, rm, rs, 3, 2, 1)))
print("bwd fused  %.1f us" % timeit(lambda: ops.bn_relu_pool_bwd(x, dyp, arg, gamma, beta, sm, si, 3, 2, 1)))
def sep_f():
    y, a, b = ops.bn_fwd_train(x, gamma, beta, rm, rs, relu=True)
    return ops.maxpool_fwd(y, 3, 2, 1)
def sep_b():
    dy = ops.maxpool_bwd(dyp, arg, tuple(x.shape), 3, 2, 1)
    return ops.bn_bwd(x, None, dy, gamma, sm, si, relu=True, beta=beta)
print("fwd separate %.1f us" % timeit(sep_f))
print("bwd separate %.1f us" % timeit(sep_b))
