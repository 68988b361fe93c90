"""config 2 (ResNet-34 224x224, B=64): N training steps for a rocprofv3 kernel trace (steps end with solver_kernel)
usage: rocprofv3 --kernel-trace -d out -o kt -- python tools/exp/cfg2_prof.py [steps]"""
import random
import sys
import time

import torch

sys.path.insert(0, ".")
from denet_amd import ops
from denet_amd.model import zoo

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
alone = len(sys.argv) > 2 and sys.argv[2] == "alone"
m = zoo.resnet34(64, 224, 1000)
x, metas = zoo.synthetic_batch(64, 224, 1000, seed=1, image_class=True)
m.build_train_func("nesterov")
xd = torch.from_numpy(x).cuda()
random.seed(1)
for it in range(3):
    m.train_step(xd, metas, 0, it, 0.05, [0.9], 1e-4)
torch.cuda.synchronize()
if alone:
    ops.WGRAD_STREAM = False          # every kernel alone on one stream
t0 = time.perf_counter()
for it in range(3, 3 + steps):
    m.train_step(xd, metas, 0, it, 0.05, [0.9], 1e-4)
torch.cuda.synchronize()
print("ms/step %.3f  img/s %.1f" % (1e3 * (time.perf_counter() - t0) / steps, 64 * steps / (time.perf_counter() - t0)))
from denet_amd.model import audit
with audit.KernelAudit(m) as ka:
    m.train_step(xd, metas, 0, 99, 0.05, [0.9], 1e-4)
for g, e in ka.summary().items():
    print(e["layers"], g, e["fwd"], e["bwd"])
