"""sustained run: N training steps of the headline configuration; images/sec per block of 50 steps and peak device memory
(a leak or a slowly growing workspace would show here, not in the 20-step bench)"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from denet_amd.model import zoo
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
model = zoo.denet34(32, "skip", 512, class_num=80, seed=1)
model.build_train_func("nesterov")
x, metas = zoo.synthetic_batch(32, 512, 80, seed=1)
xd = torch.from_numpy(x).cuda()
random.seed(1)
for it in range(3):
    model.train_step(xd, metas, 0, it, 0.1, [0.9], 1e-4)
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(3, 3 + n):
    c, _ = model.train_step(xd, metas, 0, it, 0.1, [0.9], 1e-4)
    if (it - 2) % 50 == 0:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        print("steps %4d-%4d: %.1f img/s, cost %.4f, allocated %.2f GB, reserved %.2f GB, peak %.2f GB" % (
            it - 49, it, 50 * 32 / (t1 - t0), c, torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30,
            torch.cuda.max_memory_allocated() / 2**30), flush=True)
        t0 = t1
