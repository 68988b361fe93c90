"""Host-side timeline of steady-state training steps: when (relative to the step's start on the host) the stem is queued, the
filter prefetch and the begin_step host work end, the RoI hand-off wait starts / ends, backward is queued and train_step returns."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from denet_amd import ops
from denet_amd.model import zoo, model_cnn
B = 32
model = zoo.denet34(B, "skip", 512, class_num=80, seed=1)
x, metas = zoo.synthetic_batch(B, 512, 80, seed=1)
xd = torch.from_numpy(x).cuda()
model.build_train_func("nesterov")
marks = []
T = time.perf_counter
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        marks.append((label + ">", T()))
        r = f(*a, **k)
        marks.append((label + "<", T()))
        return r
    setattr(obj, name, g)
wrap(ops, "wino_prefetch_filters", "prefetch")
wrap(ops, "wait_stream", "handoff_wait")
wrap(model, "backward", "backward")
wrap(model.layers[1], "forward", "stem_conv")
wrap(model.layers[2], "forward", "stem_bn")
for l in model.layers[2:]:
    if hasattr(l, "begin_step"):
        pass
random.seed(1)
for it in range(6):
    model.train_step(xd, metas, 0, it, 0.1, [0.9], 1e-4)
for it in range(6, 10):
    marks.clear()
    t0 = T()
    model.train_step(xd, metas, 0, it, 0.1, [0.9], 1e-4)
    t1 = T()
    print("step %d: %.2f ms on the host: " % (it, (t1 - t0) * 1e3) + "  ".join("%s %.2f" % (n, (t - t0) * 1e3) for n, t in marks))
torch.cuda.synchronize()
