"""host phases of the RoI hand-off (DeNetSparseLayer.phase_ms) and the device-side gap between the proposal kernels and the
gather, cold corner detector (the headline regime), averaged over 20 steps"""
import os
import sys
import random
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from denet_amd.model import zoo

model = zoo.denet34(32, "skip", 512, class_num=80, seed=1)
model.build_train_func("nesterov")
x, metas = zoo.synthetic_batch(32, 512, 80, seed=1)
xd = torch.from_numpy(x).cuda()
random.seed(1)
dns = [l for l in model.layers if l.type_name == "denet-sparse"][0]
for it in range(4):
    model.train_step(xd, metas, 0, it, 0.1, [0.9], 1e-4)
torch.cuda.synchronize()
acc = {}
import time
t0 = time.perf_counter()
for it in range(4, 24):
    model.train_step(xd, metas, 0, it, 0.1, [0.9], 1e-4)
    for k, v in dns.phase_ms.items():
        acc[k] = acc.get(k, 0.0) + v
torch.cuda.synchronize()
print("ms/step %.3f" % ((time.perf_counter() - t0) / 20 * 1e3), {k: round(v / 20, 3) for k, v in sorted(acc.items())}, "cold hits", getattr(dns, "cold_hits", None))

print("device-side edits", getattr(dns, "device_edits", 0))

import denet_amd.layer.denet_sparse as ds
import math
orig = ds.DeNetSparseLayer._device_edit


def probe(self, hcount):
    de, pf, prep = self.__dict__.get("_dev_edit"), self.__dict__.get("_prefetch"), self.__dict__.get("_prep")
    hc = hcount.numpy()
    why = ("no upload" if de is None else "no prefetch" if pf is None else "no prep" if prep is None else
           "sum 0" if int(hc.sum()) == 0 else "max %d" % int(hc.max()) if int(hc.max()) > 519 else "moved" if not pf["mirror"].fresh() else "ok")
    probe.log.append((why, int(hc.sum())))
    return orig(self, hcount)


probe.log = []
ds.DeNetSparseLayer._device_edit = probe
for it in range(24, 34):
    model.train_step(xd, metas, 0, it, 0.1, [0.9], 1e-4)
print(probe.log)
