import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from denet_amd import ops
from denet_amd.model import zoo
from oracle import model as OM

B, IMG = 2, 128
model = zoo.denet34(B, "skip", IMG, class_num=80, seed=1)
rng = np.random.RandomState(5)
dconv = model.layers[40].layers[0]
dconv.omega.set_value(rng.normal(0, 0.05, dconv.omega.value.shape))
x, metas = zoo.synthetic_batch(B, IMG, seed=2)
om = OM.OracleModel(model.export_json(), B)
model.build_train_func("nesterov")
random.seed(100)
# product: forward + backward only (no solver) to compare raw gradients
ctx = model.forward(x, metas, True)
model.backward(ctx)
random.seed(100)
for p in om.params(): p.g = None
om.forward(x, metas, True)
OM.backprop(om.cost_roots)

def walk_p(layers):
    out = []
    for l in layers:
        if l.type_name == "conv":
            out.append(("conv.w", l.omega))
            if l.use_bias: out.append(("conv.b", l.beta))
        elif l.type_name in ("batchnorm", "batchnorm-relu"):
            out.append(("bn.g", l.omega)); out.append(("bn.b", l.beta))
        sub = [s for s in l.layers if s.type_name != "initial"]
        out += walk_p(sub)
    return out
pp = walk_p(model.layers[1:])
op = om.params()
print(len(pp), len(op))
for i, ((name, p), o) in enumerate(zip(pp, op)):
    g = p.get_grad(); og = o.g
    scale = np.abs(og).max() + 1e-30
    print("%3d %-8s %-18s rel err %.2e  scale %.2e" % (i, name, str(g.shape), np.abs(g - og).max() / scale, scale))

print("---- mask mismatches (product y>0 vs oracle y>0) per top-level layer")
for i, layer in enumerate(model.layers[1:], 1):
    a = layer.output
    if a.data is None or a.data.dim() != 4: continue
    pa = ops.nhwc_to_nchw(a.data, a.shape[1]).cpu().numpy()
    oa = om.acts[i]
    mm = int(((pa > 0) != (oa > 0)).sum())
    print(i, layer.type_name, "mask mismatches", mm, "of", pa.size, "max abs diff %.2e" % np.abs(pa - oa).max(), "scale %.2e" % np.abs(oa).max())
