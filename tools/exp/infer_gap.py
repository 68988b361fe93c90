"""device idle time in the RoI hand-off of INFERENCE (get_detections): events either side, per batch size"""
import sys
import time

import numpy
import torch

sys.path.insert(0, ".")
from denet_amd import ops
from denet_amd.model import zoo

for B in (1, 32):
    model = zoo.denet34(B, "skip", 512, 80)
    rng = numpy.random.RandomState(3)
    dnc = [l for l in model.layers if l.type_name == "denet-corner"][0].layers[-1]
    w = dnc.omega.get_value().copy(); w[:4] = rng.normal(0, 0.3, w[:4].shape); dnc.omega.set_value(w)
    b = dnc.beta.get_value().copy(); b[:4] = 4.0; dnc.beta.set_value(b)
    dnd = [l for l in model.layers if l.type_name == "denet-detect"][0]
    hw = dnd.layers[0].omega.get_value().copy(); hw[:] = rng.normal(0, 0.02, hw.shape); dnd.layers[0].omega.set_value(hw)
    x, metas = zoo.synthetic_batch(B, 512, 80, seed=1)
    xd = torch.from_numpy(x).cuda()
    params = {"prThreshold": 0.05, "nmsThreshold": 0.5, "useSoftNMS": 0}
    pairs, state = [], {}
    w0, s0 = ops.wait_stream, ops.sparse_fwd

    def wait_stream(*a, **k):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        state["a"] = e
        r = w0(*a, **k)
        state["t"] = time.perf_counter()
        return r

    def sparse_fwd(*a, **k):
        if "a" in state:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            pairs.append((state.pop("a"), e, time.perf_counter() - state["t"]))
        return s0(*a, **k)

    ops.wait_stream, ops.sparse_fwd = wait_stream, sparse_fwd
    for _ in range(4):
        dnd.get_detections(model, xd, metas, params)
    torch.cuda.synchronize()
    pairs.clear()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        r = dnd.get_detections(model, xd, metas, params)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    gaps = [a.elapsed_time(e) for a, e, _ in pairs]
    host = [h for _, _, h in pairs]
    print("B=%d: %.2f ms per batch (%.0f img/s); hand-off gap mean %.3f ms (host part %.3f ms), RoIs in the last batch %d" % (
        B, 1e3 * dt, B / dt, sum(gaps) / len(gaps), 1e3 * sum(host) / len(host), sum(len(bx) for bx in dnd.sparse_layer.sample_boxes)))
    ops.wait_stream, ops.sparse_fwd = w0, s0
