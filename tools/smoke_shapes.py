"""Odd batch sizes and image sizes through a full training step + detection pass (nothing in the path may assume the
512 / 32 geometry): Winograd eligibility falls back to the direct kernels where a feature map is not a tile multiple."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from denet_amd.model import zoo
for B, IMG in ((3, 160), (5, 224), (1, 96), (7, 352)):
    model = zoo.denet34(B, "skip", IMG, class_num=80, seed=1)
    x, metas = zoo.synthetic_batch(B, IMG, seed=2)
    model.build_train_func("nesterov")
    random.seed(1)
    costs = [model.train_step(x, metas, 0, it, 0.01, [0.9], 1e-4)[0] for it in range(3)]
    assert np.isfinite(costs).all(), costs
    dets = model.layers[-1].get_detections(model, x, metas, {"prThreshold": 0.5, "nmsThreshold": 0.5})
    print("B=%d %dx%d: costs %s, detection pass ok (%d images)" % (B, IMG, IMG, ["%.3f" % c for c in costs], len(dets)), flush=True)
