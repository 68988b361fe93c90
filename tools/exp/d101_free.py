"""DeNet-101 wide 512x512 B=1..2 free-running forward vs the oracle, per layer: element-wise p99.99 / max-norm, with the Winograd
passes (static policy) and with the direct kernels only. usage: python tools/exp/d101_free.py [B]"""
import random
import sys

import numpy as np

sys.path.insert(0, ".")
from denet_amd import ops
from denet_amd.model import zoo
from oracle import model as OM
from tests.test_parity_gpu import _warm_corner_head, _product_acts

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
desc = zoo.DENET101_WIDE_DESC.replace("DND[0.5,1,1]", "DND.JB[0.5,1,1]")
x, metas = zoo.synthetic_batch(B, 512, seed=3)
ref = None
for wino in (4, 0):
    ops.WINOGRAD = wino
    ops.POLICY = ops.static_policy
    ops._WINO.clear()
    ops.BN_POOL_FUSE = False
    model = zoo.denet101(B, "wide", 512, class_num=80, seed=1, head_desc=desc)
    rng = np.random.RandomState(5)
    dnd = [l for l in model.layers if l.type_name == "denet-detect"][0]
    dnd.layers[0].omega.set_value(rng.normal(0, 0.05, dnd.layers[0].omega.value.shape))
    _warm_corner_head(model, 4.0, 0.3)
    if ref is None:
        om = OM.OracleModel(model.export_json(), B)
    model.build_train_func("nesterov")
    random.seed(9)
    cost, costs = model.train_step(x, metas, 0, 0, 0.05, [0.9], 1e-4)
    dns = [l for l in model.layers if l.type_name == "denet-sparse"][0]
    if ref is None:
        random.seed(9)
        om.train_step(x, metas, 0, 0.05, 0.9, 1e-4, "nesterov", sample_override=dns.sample_bbox_list)
        ref = om.acts
    print("WINOGRAD =", wino, "cost", cost)
    for i, a in sorted(_product_acts(model).items()):
        b = np.asarray(ref[i], np.float64)
        d = np.abs(np.asarray(a, np.float64) - b)
        rms = float(np.sqrt(np.mean(b * b)))
        stat = (d / (np.abs(b) + rms + 1e-30)).reshape(-1)
        q = float(np.quantile(stat, 0.9999)) if stat.size >= 10000 else float(stat.max())
        print("  L%-3d %-14s %-22s p99.99 %.2e  max-norm %.2e" % (i, model.layers[i].type_name, a.shape, q, d.max() / (np.abs(b).max() + 1e-12)))
