"""Runs a few DeNet-34 skip training steps on synthetic data and prints timings (development aid)."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch
from denet_amd.model import zoo

B = int(os.environ.get("B", 32))
IMG = int(os.environ.get("IMG", 512))
steps = int(os.environ.get("STEPS", 5))
t = time.time()
model = zoo.denet34(B, "skip", IMG)
print("build %.1fs" % (time.time() - t), flush=True)
random.seed(1)
x, metas = zoo.synthetic_batch(B, IMG)
model.build_train_func("nesterov")
xd = torch.from_numpy(x).cuda()
torch.cuda.synchronize()
for it in range(steps):
    torch.cuda.synchronize(); t0 = time.time()
    cost, costs = model.train_step(xd, metas, 0, it, 0.1, [0.9], 1e-4)
    torch.cuda.synchronize(); dt = time.time() - t0
    print("it %d cost %.5f %s  %.1f ms  %.1f img/s" % (it, cost, ["%.5f" % c for c in costs], dt * 1e3, B / dt), flush=True)
print("max mem GB", torch.cuda.max_memory_allocated() / 2**30)
