"""Inference throughput of DeNetDetectLayer.get_detections (README.md:118-128 quotes Hz for this path)."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch
from denet_amd.model import zoo
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
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
    for soft in (0, 1):
        params = {"prThreshold": 0.05, "nmsThreshold": 0.5, "useSoftNMS": soft}
        for _ in range(3): r = dnd.get_detections(model, xd, metas, params)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 10
        for _ in range(n): r = dnd.get_detections(model, xd, metas, params)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        print("B=%d softNMS=%d: %.2f ms/batch, %.1f img/s (Hz), %d detections in the last batch" % (B, soft, dt * 1e3, B / dt, sum(len(i["detections"]) for i in r)), flush=True)
