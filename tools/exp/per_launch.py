"""every matrix-kernel launch of one DeNet-34 skip training step (B=32, 512x512), alone on one stream: symbol, us, GFLOP executed,
TFLOP/s - in launch order (forward, then the backward sweep). usage: python tools/exp/per_launch.py [min_us]"""
import ctypes
import random
import sys

import torch

sys.path.insert(0, ".")
from denet_amd import ops
from denet_amd.model import zoo

min_us = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
m = zoo.denet34(32, "skip", 512, class_num=80, seed=1)
m.build_train_func("nesterov")
x, metas = zoo.synthetic_batch(32, 512, 80, seed=1)
xd = torch.from_numpy(x).cuda()
random.seed(1)
for it in range(4):
    m.train_step(xd, metas, 0, it, 0.1, [0.9], 1e-4)
torch.cuda.synchronize()
acc = {}
N = 3
for rep in range(N):
    prof = ops.KernelProfile()
    ops.PROFILE = prof
    m.train_step(xd, metas, 0, 4 + rep, 0.1, [0.9], 1e-4)
    ops.PROFILE = None
    torch.cuda.synchronize()
    L = ops._L()
    n = L.denet_conv_profile_count()
    ms, v = ctypes.c_float(), [ctypes.c_int() for _ in range(4)]
    for i in range(n):
        L.denet_conv_profile_read(i, ctypes.byref(ms), *[ctypes.byref(q) for q in v])
        name = ops.kernel_symbol(*[q.value for q in v])
        e = acc.setdefault(i, [name, 0.0, prof.flops[i]])
        e[1] += ms.value / N
    L.denet_conv_profile(0)
tot = 0.0
for i in sorted(acc):
    name, msv, fl = acc[i]
    tot += msv
    if msv * 1e3 >= min_us:
        print("%3d %-40s %8.1f us %8.2f GFLOP %7.1f TF/s" % (i, name, msv * 1e3, fl / 1e9, fl / (msv * 1e-3) / 1e12))
print("total %.3f ms over %d launches" % (tot, len(acc)))
