import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from denet_amd import ops
from denet_amd.model import zoo

def run(mode, img=128, tile=4, wgrad_stream=True):
    ops.WGRAD_STREAM = wgrad_stream
    ops.LINK_BN = False
    random.seed(7)
    model = zoo.warm_corner_head(zoo.denet34(2, "skip", img, class_num=80, seed=1), 4.0, 0.3)
    model.build_train_func("nesterov")
    x, metas = zoo.synthetic_batch(2, img, seed=11)
    model.train_step(x, metas, 0, 0, 0.02, [0.9], 1e-4)
    for (m, g) in list(ops._WINO):
        if ops.conv_wino_ok(g, tile):
            ops._WINO[(m, g)] = tile
    ops.LINK_BN = mode != "off"
    orig_b, orig_f = ops.conv_backward_linked, ops.bn_fwd_train_link
    if mode == "fwd":
        ops.conv_backward_linked = lambda *a, **k: None
    if mode == "bwd":
        import denet_amd.layer.batch_norm as BN
        # forward: never lazy
        ops_fwd = ops.bn_fwd_train
        def no_link(x, gamma, beta, rm, rs, pre, momentum=0.9, eps=1e-5, relu=False, res=None):
            y, sm, si = ops_fwd(x, gamma, beta, rm, rs, momentum, eps, relu=relu, res=res, pre=pre)
            l = ops.BnLink(False, x, res, None, gamma, beta, sm, si, None, relu)
            l.result = y
            return l, sm, si
        ops.bn_fwd_train_link = no_link
    ops.LINK_COUNT[:] = [0, 0]
    for it in (1, 2):
        model.train_step(x, metas, 0, it, 0.02, [0.9], 1e-4)
    torch.cuda.synchronize()
    ops.conv_backward_linked, ops.bn_fwd_train_link = orig_b, orig_f
    return model.P.clone(), list(ops.LINK_COUNT)

ref, _ = run("off")
for mode in ("fwd", "bwd", "both"):
    for ws in (True, False):
        p, cnt = run(mode, wgrad_stream=ws)
        print(mode, "wgrad stream", ws, "equal", torch.equal(p, ref), "max diff %.3g" % float((p - ref).abs().max()), cnt)
