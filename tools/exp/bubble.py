"""is the device idle behind the stem convolution at the start of a training step? Events: A behind the first layer's kernels, B in front of
the second layer's first launch; elapsed(A, B) = time the compute stream had nothing queued (0 when the host is ahead).
Also the host time of the step-start work (filter prefetch launches, begin_step host work)."""
import random
import sys
import time

import torch

sys.path.insert(0, ".")
from denet_amd import ops
from denet_amd.model import zoo

m = zoo.denet34(32, "skip", 512, class_num=80, seed=1)
m.build_train_func("nesterov")
x, metas = zoo.synthetic_batch(32, 512, 80, seed=1)
xd = torch.from_numpy(x).cuda()
random.seed(1)
for it in range(4):
    m.train_step(xd, metas, 0, it, 0.1, [0.9], 1e-4)
torch.cuda.synchronize()
first, second = m.layers[1], m.layers[2]
print("layers:", first.type_name, second.type_name)
pairs, host = [], []
f1, f2 = first.forward, second.forward
state = {}


def fwd1(ctx, *a, **k):
    r = f1(ctx, *a, **k)
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    state["a"] = e
    state["t"] = time.perf_counter()
    return r


def fwd2(ctx, *a, **k):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    host.append(time.perf_counter() - state["t"])
    pairs.append((state["a"], e))
    return f2(ctx, *a, **k)


first.forward, second.forward = fwd1, fwd2
t0 = time.perf_counter()
N = 20
for it in range(4, 4 + N):
    m.train_step(xd, metas, 0, it, 0.1, [0.9], 1e-4)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("ms/step %.3f" % (1e3 * dt / N))
b = [a.elapsed_time(e) for a, e in pairs]
print("bubble ms per step: mean %.3f  min %.3f  max %.3f" % (sum(b) / len(b), min(b), max(b)))
print("host ms between the two launches: mean %.3f" % (1e3 * sum(host) / len(host)))
