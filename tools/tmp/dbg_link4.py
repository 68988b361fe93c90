import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from denet_amd import ops
from denet_amd.model import zoo
import denet_amd.layer.convolution as CV

def run(mode, img=128, tile=4):
    ops.WGRAD_STREAM = False
    ops.LINK_BN = False
    random.seed(7)
    model = zoo.warm_corner_head(zoo.denet34(2, "skip", img, class_num=80, seed=1), 4.0, 0.3)
    model.build_train_func("nesterov")
    x, metas = zoo.synthetic_batch(2, img, seed=11)
    model.train_step(x, metas, 0, 0, 0.0, [0.9], 0.0)
    for (m, g) in list(ops._WINO):
        if ops.conv_wino_ok(g, tile):
            ops._WINO[(m, g)] = tile
    ops.LINK_BN = mode != "off"
    ob, of = ops.conv_backward_linked, ops._conv_wino_fwd_linked
    if mode in ("lazy", "fwd"):
        ops.conv_backward_linked = lambda *a, **k: None
    if mode in ("lazy", "bwd"):
        def nolink(link, g, tile, w, bias, add, out, cache, bn_stats):
            xx = link.materialise()
            return ops.conv_fwd(xx, w, bias=bias, add=add, stride=1, pad=1, s_real=3, out=out, cache=cache, bn_stats=bn_stats)
        ops._conv_wino_fwd_linked = nolink
    model.train_step(x, metas, 0, 1, 0.0, [0.9], 0.0)     # lr 0: parameters stay, gradients are what we compare
    torch.cuda.synchronize()
    ops.conv_backward_linked, ops._conv_wino_fwd_linked = ob, of
    return model, model.G.clone(), model.S.clone()

mref, gref, sref = run("off")
for mode in ("lazy", "fwd", "bwd", "both"):
    m, g, s = run(mode)
    print(mode, "G equal", torch.equal(g, gref), "S equal", torch.equal(s, sref), "max dG %.3g" % float((g - gref).abs().max()))
    if not torch.equal(g, gref):
        bad = []
        for layer, lo, hi in m.layer_weight_range:
            if not torch.equal(g[lo:hi], gref[lo:hi]):
                bad.append((getattr(layer, "layer_index", "?"), type(layer).__name__, lo, float((g[lo:hi] - gref[lo:hi]).abs().max())))
        print("   weight grads differing in", len(bad), "of", len(m.layer_weight_range), "ranges; last ones (backward order first):", bad[-4:])
m, g, s = run("lazy")
d = (s != sref).nonzero().flatten()
print("S size", s.numel(), "differing", d.numel(), "first idx", d[:5].tolist(), "vals", s[d[:5]].tolist(), sref[d[:5]].tolist())
from denet_amd.model.model_cnn import walk_layers
for l in walk_layers(m.layers):
    if l.type_name in ("batchnorm", "batchnorm-relu") and getattr(l, "enabled", True):
        a = l.mean.dev; 
        off = a.data_ptr() - m.S.data_ptr()
        n = a.numel()
        i0 = off // 4
        if not torch.equal(s[i0:i0 + n], sref[i0:i0 + n]):
            print("mean differs in BN layer", l.layer_index, l.type_name, n, "input shape", l.input_shape); break
m2, g2, s2 = run("off")
print("off vs off: S equal", torch.equal(s2, sref), "G equal", torch.equal(g2, gref))
