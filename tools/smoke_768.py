"""DeNet-34 skip re-targeted to 768x768 / 1296 RoIs per image (README.md:145 of the reference): a few training steps and a
detection pass, to show nothing in the path is tied to the 512 / 576 geometry."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from denet_amd.model import zoo
B, IMG = 8, 768
desc = zoo.DENET34_SKIP_DESC.replace("DNS[7,24,0.01,0.1]", "DNS[7,36,0.01,0.1]")
model = zoo.denet34(B, "skip", IMG, class_num=80, seed=1, head_desc=desc)
x, metas = zoo.synthetic_batch(B, IMG, seed=2)
model.build_train_func("nesterov")
random.seed(1)
for it in range(6):
    if it == 3:
        torch.cuda.synchronize(); t0 = time.time()
    cost, costs = model.train_step(x, metas, 0, it, 0.01, [0.9], 1e-4)
    assert np.isfinite(cost)
torch.cuda.synchronize()
dt = (time.time() - t0) / 3
dns = [l for l in model.layers if l.type_name == "denet-sparse"][0]
print("768x768, %d RoIs/image: %.1f ms/step = %.1f img/s, cost %.3f, output %s" % (dns.sample_count, 1e3 * dt, B / dt, cost, dns.output_shape))
dets = model.layers[-1].get_detections(model, x, metas, {"prThreshold": 0.5, "nmsThreshold": 0.5})
print("detection pass ok:", [len(d["detections"]) for d in dets])
