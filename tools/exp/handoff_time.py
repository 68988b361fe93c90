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
print("ms/step %.3f" % ((time.perf_counter() - t0) / 20 * 1e3), {k: round(v / 20, 3) for k, v in sorted(acc.items())}, "hand-off modes", dns.handoff_modes)

print("device-side edits", getattr(dns, "device_edits", 0))
