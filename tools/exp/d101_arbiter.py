"""DeNet-101 wide 512x512 B=1 free-running forward: the product (fp32 HIP kernels) and the fp32 oracle, each against the SAME
restatement evaluated in float64 (oracle/model.py: float64_arbiter), per layer: element-wise p99.99 / max-norm.
usage: python tools/exp/d101_arbiter.py [B]"""
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


def stats(a, b):
    b = np.asarray(b, np.float64)
    d = np.abs(np.asarray(a, np.float64) - b)
    rms = float(np.sqrt(np.mean(b * b)))
    stat = (d / (np.abs(b) + rms + 1e-30)).reshape(-1)
    q = float(np.quantile(stat, 0.9999)) if stat.size >= 10000 else float(stat.max())
    return q, float(d.max() / (np.abs(b).max() + 1e-12))


model = zoo.denet101(B, "wide", 512, class_num=80, seed=1, head_desc=desc)
rng = np.random.RandomState(5)
dnd = [l for l in model.layers if l.type_name == "denet-detect"][0]
dnd.layers[0].omega.set_value(rng.normal(0, 0.05, dnd.layers[0].omega.value.shape))
_warm_corner_head(model, 4.0, 0.3)
j = model.export_json()
om = OM.OracleModel(j, B)
model.build_train_func("nesterov")
random.seed(9)
cost, costs = model.train_step(x, metas, 0, 0, 0.05, [0.9], 1e-4)
dns = [l for l in model.layers if l.type_name == "denet-sparse"][0]
dnc = [l for l in model.layers if l.type_name == "denet-corner"][0]
random.seed(9)
c32 = om.forward_costs(x, metas, sample_override=dns.sample_bbox_list)
with OM.float64_arbiter():
    om64 = OM.OracleModel(j, B)
    random.seed(9)
    c64 = om64.forward_costs(x, metas, sample_override=dns.sample_bbox_list)
print("costs: product", cost, costs, "oracle fp32", c32, "fp64", c64)
print("taps equal", np.array_equal(om.taps[0], om64.taps[0]) and np.array_equal(om.taps[1], om64.taps[1]))
acts = _product_acts(model)
for i in sorted(acts):
    p = stats(acts[i], om64.acts[i])
    o = stats(om.acts[i], om64.acts[i])
    po = stats(acts[i], om.acts[i])
    print("  L%-3d %-14s %-20s product-vs-fp64 p99.99 %.2e max %.2e | oracle32-vs-fp64 %.2e %.2e | product-vs-oracle32 %.2e %.2e" % (
        i, model.layers[i].type_name, acts[i].shape, p[0], p[1], o[0], o[1], po[0], po[1]))
p = stats(dnc.corner_pr.cpu().numpy(), om64.corner_pr)
o = stats(om.corner_pr, om64.corner_pr)
print("  corner_pr product-vs-fp64 %.2e %.2e | oracle32-vs-fp64 %.2e %.2e" % (p[0], p[1], o[0], o[1]))
