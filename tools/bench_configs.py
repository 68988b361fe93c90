"""Secondary configurations of BASELINE.json / SURVEY §8(d) (not bench.py lines: parity-test cases whose throughput is
recorded for DESIGN.md): cfg 1 three-layer CIFAR CNN, cfg 2 ResNet-34 224x224, cfg 5 DeNet-101 wide (B=16, 2304 RoIs).
Algorithmic FLOPs per step = 2 x MACs of every convolution launch (fwd + dgrad + wgrad), summed live."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch
from denet_amd import ops
from denet_amd.model import zoo

def run(name, model, x, metas, solver, steps=8, warm=3):
    model.build_train_func(solver)
    xd = torch.from_numpy(x).cuda()
    it = 0
    for _ in range(warm):
        model.train_step(xd, metas, 0, it, 0.05, [0.9], 1e-4); it += 1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        cost, _ = model.train_step(xd, metas, 0, it, 0.05, [0.9], 1e-4); it += 1
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    prof = ops.KernelProfile(); ops.PROFILE = prof
    model.train_step(xd, metas, 0, it, 0.05, [0.9], 1e-4)
    ops.PROFILE = None
    agg = prof.summary()
    flops = sum(a["flops"] for a in agg.values()); kms = sum(a["ms"] for a in agg.values())
    B = x.shape[0]
    print("%-22s B=%-3d %8.2f ms/step %9.1f img/s | conv %.1f GFLOP/img/step, whole step %.1f TFLOP/s, conv kernels %.2f ms = %.1f TFLOP/s | cost %.4f"
          % (name, B, dt * 1e3, B / dt, flops / B / 1e9, flops / dt / 1e12, kms, flops / kms / 1e9, cost), flush=True)

which = sys.argv[1:] or ["cifar3", "resnet34", "denet101"]
random.seed(1)
if "cifar3" in which:
    m = zoo.cifar3(32); x, metas = zoo.synthetic_batch(32, 32, 10, seed=1, image_class=True)
    run("cfg1 cifar3 32x32", m, x, metas, "sgd", steps=50)
if "resnet34" in which:
    m = zoo.resnet34(64, 224, 1000); x, metas = zoo.synthetic_batch(64, 224, 1000, seed=1, image_class=True)
    run("cfg2 resnet34 224", m, x, metas, "nesterov")
if "denet101" in which:
    m = zoo.denet101(16, "wide", 512, 80, head_desc=zoo.DENET101_WIDE_DESC.replace("DND[0.5,1,1]", "DND.JB[0.5,1,1]"))
    x, metas = zoo.synthetic_batch(16, 512, 80, seed=1)
    run("cfg5 denet101-wide 512", m, x, metas, "nesterov", steps=5, warm=2)
if "denet34" in which:
    m = zoo.denet34(32, "skip", 512, 80); x, metas = zoo.synthetic_batch(32, 512, 80, seed=1)
    run("cfg3 denet34-skip 512", m, x, metas, "nesterov")
