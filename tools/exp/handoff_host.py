"""where the host time of the RoI hand-off goes (trimming regime: the first steps of the synthetic run): per call, us"""
import os
import sys
import random
import time
import collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from denet_amd.model import zoo
from denet_amd import ops
import denet_amd.layer.denet_sparse as ds

acc = collections.defaultdict(float)


def wrap(obj, name, label=None):
    f = getattr(obj, name)

    def g(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        acc[label or name] += time.perf_counter() - t
        return r
    setattr(obj, name, g)


model = zoo.denet34(32, "skip", 512, class_num=80, seed=1)
model.build_train_func("nesterov")
x, metas = zoo.synthetic_batch(32, 512, 80, seed=1)
xd = torch.from_numpy(x).cuda()
random.seed(1)
dns = [l for l in model.layers if l.type_name == "denet-sparse"][0]
for it in range(3):
    model.train_step(xd, metas, 0, it, 0.1, [0.9], 1e-4)
wrap(ops, "samples_finish_host")
wrap(ops, "wait_stream")
wrap(ds.DeNetSparseLayer, "_finish_samples")
wrap(ds.DeNetSparseLayer, "_edit_and_upload_native")
wrap(ds.DeNetSparseLayer, "_native_edit_stream")
wrap(ds.DeNetSparseLayer, "edit_samples_native")
wrap(ds.DeNetSparseLayer, "get_target")
wrap(ds.DeNetSparseLayer, "_device_samples")
n = 15
for it in range(3, 3 + n):
    model.train_step(xd, metas, 0, it, 0.1, [0.9], 1e-4)
torch.cuda.synchronize()
print({k: round(v / n * 1e6) for k, v in acc.items()})
