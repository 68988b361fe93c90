"""batch-1 inference (DeNet-34 skip 512x512, get_detections): wall per image, host time of the forward launches, kernels per image
(under rocprofv3: python tools/rocpd_stats.py on the trace gives the device-busy share)"""
import os
import sys
import time

import numpy
import torch

sys.path.insert(0, ".")
from denet_amd.model import zoo

B = 1
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
for _ in range(5):
    r = dnd.get_detections(model, xd, metas, params)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
t0 = time.perf_counter()
for _ in range(n):
    r = dnd.get_detections(model, xd, metas, params)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("B=1: %.3f ms per image, %.1f Hz" % (dt * 1e3, 1 / dt))
# the forward alone: host time to queue it (no sync inside), then device time
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
hs, ds = [], []
for _ in range(20):
    torch.cuda.synchronize()
    e0.record()
    t0 = time.perf_counter()
    model.forward(xd, None, train=False)
    hs.append(time.perf_counter() - t0)
    e1.record()
    torch.cuda.synchronize()
    ds.append(e0.elapsed_time(e1))
print("forward only: host %.3f ms to queue, device span %.3f ms" % (1e3 * sum(hs) / len(hs), sum(ds) / len(ds)))
