"""two independent runs of N training steps of the headline configuration (B=32, 512x512) from the same seed: parameters,
momentum and running statistics must agree bit for bit (no float atomics, fixed launch configurations) - with PRESS=1 a side
stream keeps the memory system saturated during the second run (the condition under which the store hazard of
csrc/wino4f.hip showed, EXPERIMENTS.md). usage: python tools/soak_determinism.py [steps]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from denet_amd.model import zoo
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
press = os.environ.get("PRESS", "1") == "1"
x, metas = zoo.synthetic_batch(32, 512, 80, seed=1)
xd = torch.from_numpy(x).cuda()
side = torch.cuda.Stream()
big_a = torch.empty(1 << 27, device="cuda")
big_b = torch.empty(1 << 27, device="cuda")
states = []
for run in range(2):
    model = zoo.warm_corner_head(zoo.denet34(32, "skip", 512, class_num=80, seed=1))
    model.build_train_func("nesterov")
    random.seed(1)
    t0 = time.perf_counter()
    for it in range(n):
        if run == 1 and press:
            with torch.cuda.stream(side):
                for _ in range(12):
                    big_b.copy_(big_a, non_blocking=True)
        c, _ = model.train_step(xd, metas, 0, it, 0.02, [0.9], 1e-4)
    torch.cuda.synchronize()
    print("run %d: %d steps, %.1f img/s, final cost %.6f" % (run, n, n * 32 / (time.perf_counter() - t0), c), flush=True)
    states.append((model.P.clone(), model.M.clone(), model.S.clone(), c))
    del model
same = all(torch.equal(a, b) for a, b in zip(states[0][:3], states[1][:3]))
print("bit-identical state after %d steps (second run %s): %s" % (n, "under memory pressure" if press else "alone", same))
sys.exit(0 if same else 1)
