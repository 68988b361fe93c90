import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy, torch
from denet_amd.model import zoo
from denet_amd import ops
B = 32
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
for _ in range(3): dnd.get_detections(model, xd, metas, params)
torch.cuda.synchronize()
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10): dnd.get_detections(model, xd, metas, params)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
t0 = time.perf_counter()
for _ in range(10):
    model.forward(xd, None, train=False); torch.cuda.synchronize()
print("forward only (incl. RoI hand-off): %.2f ms" % ((time.perf_counter() - t0) * 100))
