"""Phase timing of one train step (host vs device) — development aid."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch
from denet_amd.model import zoo, model_cnn
from denet_amd import layer as layer_mod

B = int(os.environ.get("B", 32))
model = zoo.denet34(B, "skip", 512)
random.seed(1)
x, metas = zoo.synthetic_batch(B, 512)
model.build_train_func("nesterov")
xd = torch.from_numpy(x).cuda()
for it in range(2):
    model.train_step(xd, metas, 0, it, 0.1, [0.9], 1e-4)
torch.cuda.synchronize()

def sync_time(fn):
    torch.cuda.synchronize(); t = time.time(); r = fn(); torch.cuda.synchronize(); return (time.time() - t) * 1e3, r

# per-layer forward / backward with syncs
layer_mod.set_train(True)
for a in model.acts: a.grad = None
model._upload_input(xd)
ctx = model_cnn.StepContext(model)
tf = {}
for i, layer in enumerate(model.layers[1:], 1):
    t0 = time.time()
    target = layer.get_target(model, xd, metas)
    if target is not None:
        layer.set_target(ctx, target[0], target[1])
    th = (time.time() - t0) * 1e3
    td, _ = sync_time(lambda: layer.forward(ctx))
    tf[i] = (th, td)
tl = []
for i, layer in enumerate(model.cost_layers):
    td, _ = sync_time(lambda: layer.loss_backward(ctx, model.cost_buf[2 * i:2 * i + 2]))
    tl.append(td)
tb = {}
for i in range(len(model.layers) - 1, 0, -1):
    td, _ = sync_time(lambda: model.layers[i].backward(ctx))
    tb[i] = td
ts, _ = sync_time(lambda: model._device_step.__self__ and None)
from denet_amd import ops
ts, _ = sync_time(lambda: ops.solver_step(model.P[:model.n_trainable], model.M[:model.n_trainable], model.G[:model.n_trainable], model.n_weights, 0.1, 0.9, 1, 1e-4, 1, 1.0))
print("%3s %-16s %8s %8s %8s" % ("i", "layer", "host_ms", "fwd_ms", "bwd_ms"))
sh = sf = sb = 0
for i, layer in enumerate(model.layers[1:], 1):
    print("%3d %-16s %8.2f %8.2f %8.2f" % (i, layer.type_name, tf[i][0], tf[i][1], tb[i]))
    sh += tf[i][0]; sf += tf[i][1]; sb += tb[i]
print("sum host %.1f fwd %.1f bwd %.1f loss %.2f solver %.2f" % (sh, sf, sb, sum(tl), ts))
