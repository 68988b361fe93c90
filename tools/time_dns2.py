import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch
from denet_amd.model import zoo
from denet_amd import ops
B = 32
model = zoo.denet34(B, "skip", 512)
random.seed(1)
x, metas = zoo.synthetic_batch(B, 512)
model.build_train_func("nesterov")
xd = torch.from_numpy(x).cuda()
dns = model.layers[31]
T = {}
def wrap(obj, name, key):
    fn = getattr(obj, name)
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); T.setdefault(key, []).append((time.perf_counter() - t) * 1e6); return r
    setattr(obj, name, w)
wrap(ops, "build_samples", "launch build_samples")
wrap(ops, "wait_stream", "wait_stream (GPU drain + D2H)")
wrap(ops, "samples_finish_host", "samples_finish_host")
wrap(dns, "_device_samples", "_device_samples total")
wrap(dns, "edit_samples_native", "edit native")
wrap(dns, "_edit_and_upload_native", "edit+upload total")
wrap(dns, "get_target", "DNS get_target total")
for it in range(20):
    model.train_step(xd, metas, 0, it, 0.1, [0.9], 1e-4)
for k, v in T.items():
    v = v[5:]
    print("%-34s mean %8.1f us  min %8.1f" % (k, sum(v) / len(v), min(v)))
