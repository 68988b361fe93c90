"""config 5 (DeNet-101 wide 512x512 B=16, DND.JB): steps for a kernel trace + the per-layer kernel audit
usage: python tools/exp/cfg5_prof.py [steps] [alone]"""
import random
import sys
import time

import torch

sys.path.insert(0, ".")
import os
from denet_amd import lib as dlib
if os.environ.get("OLD_LIB"):
    dlib.LIB_PATH = os.environ["OLD_LIB"]
from denet_amd import ops
from denet_amd.model import zoo, audit

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
alone = len(sys.argv) > 2 and sys.argv[2] == "alone"
m = zoo.denet101(16, "wide", 512, 80, head_desc=zoo.DENET101_WIDE_DESC.replace("DND[0.5,1,1]", "DND.JB[0.5,1,1]"))
print("undecided passes:", len(audit.decisions_cover(m)))
x, metas = zoo.synthetic_batch(16, 512, 80, seed=1)
m.build_train_func("nesterov")
xd = torch.from_numpy(x).cuda()
random.seed(1)
for it in range(3):
    m.train_step(xd, metas, 0, it, 0.05, [0.9], 1e-4)
torch.cuda.synchronize()
if alone:
    ops.WGRAD_STREAM = False
t0 = time.perf_counter()
for it in range(3, 3 + steps):
    m.train_step(xd, metas, 0, it, 0.05, [0.9], 1e-4)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("ms/step %.3f  img/s %.1f" % (1e3 * dt / steps, 16 * steps / dt))
if os.environ.get("NO_AUDIT"):
    sys.exit(0)
with audit.KernelAudit(m) as ka:
    m.train_step(xd, metas, 0, 99, 0.05, [0.9], 1e-4)
for g, e in ka.summary().items():
    print(e["layers"], g, e["fwd"], e["bwd"])
