"""where the host's time between the proposal's arrival and the gather's launch goes (fast hand-off): checkpoints around the calls"""
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
T = {}
rows = []
pc = time.perf_counter


def wrap(obj, name, key):
    f = getattr(obj, name)

    def g(*a, **k):
        t = pc()
        r = f(*a, **k)
        T[key] = T.get(key, 0.0) + pc() - t
        return r
    setattr(obj, name, g)


w0, s0 = ops.wait_stream, ops.sparse_fwd


def wait_stream(*a, **k):
    r = w0(*a, **k)
    T.clear()
    T["t0"] = pc()
    return r


def sparse_fwd(*a, **k):
    if "t0" in T:
        t0 = T.pop("t0")
        rows.append((pc() - t0, dict(T), dict(dns.handoff_modes)))
        T.clear()
    return s0(*a, **k)


ops.wait_stream, ops.sparse_fwd = wait_stream, sparse_fwd
_sh = dns._short_handoff


def _sh_t0(*a, **k):
    if "t0" not in T:
        T.clear()
        T["t0"] = pc()
    return _sh(*a, **k)


dns._short_handoff = _sh_t0
wrap(dns, "_short_handoff", "short_handoff")
wrap(dns, "_device_edit", "device_edit")
wrap(dns, "_fast_handoff", "fast_handoff")
wrap(dns, "_log_get_samples", "log")
for it in range(5):
    m.train_step(xd, metas, 0, it, 0.1, [0.9], 1e-4)
bufs = dns.__dict__.get("_ho_bufs")
if bufs is not None:
    fn = bufs["fn"]

    def timed_fn(*a):
        t = pc()
        r = fn(*a)
        T["native"] = pc() - t
        return r
    bufs["fn"] = timed_fn
for it in range(5, 30):
    m.train_step(xd, metas, 0, it, 0.1, [0.9], 1e-4)
torch.cuda.synchronize()
for i, (tot, parts, modes) in enumerate(rows):
    print("%2d total %.0f us  " % (i, 1e6 * tot) + "  ".join("%s %.0f" % (k, 1e6 * v) for k, v in parts.items()))
