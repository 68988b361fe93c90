"""where a B = 1 get_detections call with hard NMS spends its time (bench.py inference leg: 6.5 ms against 2.6 ms with soft-NMS)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy, torch
from denet_amd import ops
from denet_amd.model import zoo

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
model = zoo.denet34(B, "skip", 512, class_num=80, seed=1)
zoo.warm_corner_head(model, 4.0, 0.3)
rng = numpy.random.RandomState(3)
dnd = [l for l in model.layers if l.type_name == "denet-detect"][0]
dnd.layers[0].omega.set_value(rng.normal(0, 0.02, dnd.layers[0].omega.value.shape))
x, metas = zoo.synthetic_batch(B, 512, 80, seed=1)
xd = torch.from_numpy(x).cuda()
for soft in (0, 1, 0):
    params = {"prThreshold": 0.05, "nmsThreshold": 0.5, "useSoftNMS": soft}
    for _ in range(3):
        r = dnd.get_detections(model, xd, metas, params)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        r = dnd.get_detections(model, xd, metas, params)
    torch.cuda.synchronize()
    print("soft", soft, "ms per call %.3f" % ((time.perf_counter() - t0) / 20 * 1e3), "detections", sum(len(i["detections"]) for i in r))
import cProfile, pstats
params = {"prThreshold": 0.05, "nmsThreshold": 0.5, "useSoftNMS": 0}
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    dnd.get_detections(model, xd, metas, params)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
