"""which batch-norm layers of a training step still run their own backward reduction pass (no sums from the data-gradient pass)"""
import os
import sys
import random
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from denet_amd.model import zoo
from denet_amd import ops
import denet_amd.layer.batch_norm as bnm

model = zoo.denet34(32, "skip", 512, class_num=80, seed=1)
model.build_train_func("nesterov")
x, metas = zoo.synthetic_batch(32, 512, 80, seed=1)
xd = torch.from_numpy(x).cuda()
random.seed(1)
for it in range(3):
    model.train_step(xd, metas, 0, it, 0.1, [0.9], 1e-4)
log = []
orig = bnm.BatchNormLayer.backward


def probe(self, ctx, want_dres=False):
    if self.enabled and not getattr(self, "_pooled", False):
        out_act = self._save[3]
        sums = out_act.grad_sums
        if sums is None or sums.partial is None:
            log.append((tuple(self.input.data.shape), "relu" if self._save[2] else "-", "res" if self._save[4] else "-",
                        "no request" if sums is None else "request not served"))
    return orig(self, ctx, want_dres)


bnm.BatchNormLayer.backward = probe
model.train_step(xd, metas, 0, 3, 0.1, [0.9], 1e-4)
torch.cuda.synchronize()
for l in log:
    print(l)
print(len(log), "fallbacks")
