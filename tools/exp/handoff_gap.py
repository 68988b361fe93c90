"""device idle time in the RoI hand-off of a training step: event A behind the proposal's device-to-host copy (recorded right before
the host starts waiting for it), event B in front of the sparse gather's launch; elapsed(A, B) = the compute stream had nothing to run"""
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
dns = [l for l in m.layers if l.type_name == "denet-sparse"][0]
pairs = []
state = {}
w0, s0 = ops.wait_stream, ops.sparse_fwd


def wait_stream(*a, **k):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    state["a"] = e
    t = time.perf_counter()
    r = w0(*a, **k)
    state["t_wait"] = time.perf_counter() - t
    state["t_ret"] = time.perf_counter()
    return r


def sparse_fwd(*a, **k):
    if "a" in state:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        pairs.append((state.pop("a"), e, dict(dns.handoff_modes), state["t_wait"], time.perf_counter() - state["t_ret"]))
    return s0(*a, **k)


ops.wait_stream, ops.sparse_fwd = wait_stream, sparse_fwd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
t0 = time.perf_counter()
for it in range(N):
    m.train_step(xd, metas, 0, it, 0.1, [0.9], 1e-4)
torch.cuda.synchronize()
print("ms/step %.3f" % (1e3 * (time.perf_counter() - t0) / N))
prev = None
for i, (a, b, modes, tw, th) in enumerate(pairs):
    mode = "?" if prev is None else [k for k in modes if modes[k] != prev[k]]
    prev = modes
    if i < 30 or i % 10 == 0:
        print("step %3d  gap %.3f ms  %s   host: waited %.3f ms for the copy, then %.3f ms until the gather's launch" % (i, a.elapsed_time(b), mode, 1e3 * tw, 1e3 * th))
