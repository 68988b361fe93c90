"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/layers.py for the rules and the parity-pin statement).

A CPU interpreter for models in the reference's own serialisation (the `.mdl.gz` JSON layer list,
denet/model/model_cnn.py:159-203 and the per-layer export_json methods): it rebuilds the graph the reference
would build from those dictionaries and runs one training step of it with the numpy restatements of
oracle/layers.py — forward, the host-side target builders, the costs, reverse-mode gradients and the solver.

  conv            denet/layer/convolution.py:55-89         batchnorm(-relu) denet/layer/batch_norm.py:50-79,
  activation      denet/layer/activation.py:31-34                           batch_norm_relu.py:34-54
  pool / pool-inv denet/layer/pool.py:28-40, pool_inv.py:21-26
  resnet          denet/layer/resnet.py:52-113             skip-src / skip  denet/layer/skip.py:19-24, 78-86
  denet-corner    denet/layer/denet_corner.py:39-58,81-134 denet-sparse     denet/layer/denet_sparse.py:60-96,164-206
  denet-detect    denet/layer/denet_detect.py:60-107,147-313   regression   denet/layer/regression.py:53-98
  train step      denet/model/model_cnn.py:407-445, solver :282-331
"""
import copy
import ctypes
import math
import os
import random

import numpy as np

from . import layers as L

F32 = np.float32
_HERE = os.path.dirname(os.path.abspath(__file__))


class float64_arbiter:
    """context manager: the SAME restatement evaluated in float64 (every `F32` of oracle/layers.py and of this file becomes
    numpy.float64; index work - RoI taps, targets - keeps its float32 / integer arithmetic). Not the reference's arithmetic: an
    ARBITER for whole-network comparisons in which two float32 evaluations (the product's and this oracle's) drift apart with
    depth - each side is then measured against the value both approximate (tests/test_parity_gpu.py, DeNet-101 wide). Build the
    OracleModel INSIDE the context (parameters are stored in the active precision)."""

    def __enter__(self):
        global F32
        self.saved = (F32, L.F32)
        F32 = L.F32 = np.float64
        return self

    def __exit__(self, *exc):
        global F32
        F32, L.F32 = self.saved
        return False


# ---------------------------------------------------------------------------------------------------------
# minimal reverse-mode tape
# ---------------------------------------------------------------------------------------------------------
class T:
    def __init__(self, v, parents=(), bw=None):
        self.v = v
        self.g = None
        self.parents = parents
        self.bw = bw

    def add_grad(self, g):
        self.g = g if self.g is None else (self.g + g)


class P:
    """parameter in the reference layout"""

    def __init__(self, v, is_weight):
        self.v = np.array(v, dtype=F32)
        self.g = None
        self.m = np.zeros_like(self.v)
        self.is_weight = is_weight


def backprop(outputs):
    order, seen = [], set()

    def visit(n):
        if id(n) in seen:
            return
        seen.add(id(n))
        for p in n.parents:
            visit(p)
        order.append(n)

    for o in outputs:
        visit(o)
    for n in reversed(order):
        if n.g is not None and n.bw is not None:
            n.bw(n.g)


def pad_of(border, k):
    if border == "valid":
        return 0
    if border == "full":
        return k - 1
    if border == "half":
        return k // 2
    if border == "same":
        return (k - 1) // 2
    if isinstance(border, (list, tuple)):
        return int(border[0])
    return int(border)


# ---------------------------------------------------------------------------------------------------------
# ops
# ---------------------------------------------------------------------------------------------------------
def op_conv(x, w, b, stride, pad, need_dx=True):
    y = L.conv2d(x.v, w.v, None if b is None else b.v, stride, pad)

    def bw(g):
        dx, dw, db = L.conv2d_grad(x.v, w.v, g, stride, pad, need_dx)
        w.g = dw if w.g is None else w.g + dw
        if b is not None:
            b.g = db.astype(F32) if b.g is None else b.g + db
        if need_dx:
            x.add_grad(dx)

    return T(y, (x,), bw)


def op_bn(x, node, train, relu):
    gamma, beta = node["gamma"], node["beta"]
    if train:
        y, mean, invstd = L.bn_train(x.v, gamma.v, beta.v, node["eps"])
        node["new_stats"] = L.bn_running_update(node["mean"], node["stdinv"], mean, invstd, node["momentum"])
    else:
        y = L.bn_test(x.v, gamma.v, beta.v, node["mean"], node["stdinv"], node["eps"])
        mean = invstd = None
    if relu:
        y = L.relu(y)
    t = T(y, (x,), None)

    def bw(g):
        if relu:
            g = L.relu_grad(t.v, g)     # mask from the (possibly teacher-forced) output
        dx, dg, db = L.bn_grad(x.v, g, gamma.v, mean, invstd)
        gamma.g, beta.g = dg, db
        x.add_grad(dx)

    t.bw = bw
    return t


def op_relu(x):
    t = T(L.relu(x.v), (x,), None)
    t.bw = lambda g: x.add_grad(L.relu_grad(t.v, g))
    return t


def op_add(a, b):
    def bw(g):
        a.add_grad(g)
        b.add_grad(g)

    return T((a.v + b.v).astype(F32), (a, b), bw)


def op_pool(x, mode, k, s, p):
    if mode == "max":
        y, arg = L.pool_max(x.v, k, s, p)
        return T(y, (x,), lambda g: x.add_grad(L.pool_max_grad(g, arg, x.v.shape, k, s, p)))
    y = L.pool_avg(x.v, k, s, p)
    return T(y, (x,), lambda g: x.add_grad(L.pool_avg_grad(g, x.v.shape, k, s, p)))


def op_deconv(x, w, b, stride, pad, need_dx=True):
    y = L.deconv2d(x.v, w.v, None if b is None else b.v, stride, pad)

    def bw(g):
        dx, dw, db = L.deconv2d_grad(x.v, w.v, g, stride, pad, need_dx)
        w.g = dw if w.g is None else w.g + dw
        if b is not None:
            b.g = db.astype(F32) if b.g is None else b.g + db
        if need_dx:
            x.add_grad(dx)

    return T(y, (x,), bw)


def op_border(x, b):
    return T(L.border(x.v, b), (x,), lambda g: x.add_grad(L.border_grad(g, b)))


def op_dropout(x, rate, seed):
    m = L.dropout_mask(x.v.shape, rate, seed)
    return T((x.v * m).astype(F32), (x,), lambda g: x.add_grad((g * m).astype(F32)))


def op_crop_mirror(x, crop, geom):
    return T(L.crop_mirror(x.v, crop, geom), (x,), lambda g: x.add_grad(L.crop_mirror_grad(g, x.v.shape, geom)))


def op_concat(a, b):
    ca = a.v.shape[1]

    def bw(g):
        a.add_grad(np.ascontiguousarray(g[:, :ca]))
        b.add_grad(np.ascontiguousarray(g[:, ca:]))

    return T(np.concatenate([a.v, b.v], axis=1), (a, b), bw)


def op_pool_inv(x, size):
    return T(L.pool_inv(x.v, size), (x,), lambda g: x.add_grad(L.pool_inv_grad(g, size)))


# ---------------------------------------------------------------------------------------------------------
# model
# ---------------------------------------------------------------------------------------------------------
def _bn_node(j):
    return {"gamma": P(j["gamma"], False), "beta": P(j["bias"], False), "mean": np.array(j["mean"], dtype=F32),
            "stdinv": np.array(j["std"], dtype=F32), "momentum": j["momentum"], "eps": j["eps"],
            "enabled": j.get("enabled", True)}


def _conv_node(j):
    return {"w": P(j["weight"], True), "b": P(j["bias"], False) if j["useBias"] else None,
            "stride": int(j["stride"][0]), "pad": pad_of(j["border"], int(j["shape"][2])),
            "enabled": j.get("enabled", True)}


class OracleModel:
    def __init__(self, json_obj, batch_size, tap_rule=0, rng_seed=0):
        self.batch_size = batch_size
        self.class_num = json_obj["classNum"]
        self.tap_rule = tap_rule
        self.rng_seed = rng_seed   # base key of the `D` / `CM` counter generator (ModelCNN.rng_seed)
        self.iteration = 0
        self.nodes = [self._build(j) for j in json_obj["layers"]]
        self.acts = {}
        self.sample_bbox_list = None
        self.force = None          # teacher forcing: list of arrays / None in op order (see _forced)
        self.force_err = []

    def _forced(self, t, name):
        """Teacher forcing for per-op parity: replace the value of op output `t` by the product's tensor for the
        same op, recording the relative error of what the oracle computed from the (forced) inputs. With every op
        output forced, each op sees exactly the product's inputs and the ReLU masks of the backward pass are the
        product's, so forward AND backward are compared op by op instead of through 40 layers of rounding."""
        if self.force is None:
            return t
        v = self.force[self._fpos]
        self._fpos += 1
        if v is not None:
            v = np.asarray(v, dtype=F32)
            assert v.shape == t.v.shape, (name, v.shape, t.v.shape)
            scale = float(np.abs(v).max()) + 1e-30
            self.force_err.append((name, float(np.abs(t.v - v).max()) / scale))
            t.v = v
        return t

    def _build(self, j):
        t = j["type"]
        n = {"type": t}
        if t == "conv":
            n.update(_conv_node(j))
        elif t in ("batchnorm", "batchnorm-relu"):
            n.update(_bn_node(j))
        elif t == "activation":
            n["activation"] = j["activation"]
        elif t == "pool":
            n.update(mode=j["mode"], k=int(j["size"][0]), s=int(j["stride"][0]), p=int(j["pad"][0]))
        elif t == "pool-inv":
            n["size"] = tuple(j["size"])
        elif t == "resnet":
            n.update(version=j["version"], bottleneck=j["bottleneck"], activation=j["activation"])
            n["layers"] = [self._build(s) for s in j["layers"] if s["type"] not in ("initial", "identity")]
        elif t == "deconv":
            n.update(w=P(j["weight"], True), b=P(j["bias"], False) if j["useBias"] else None,
                     stride=int(j["stride"][0]), pad=int(j["shape"][2]) // 2)
        elif t == "border":
            n["border"] = tuple(int(v) for v in j["border"])
        elif t == "dropout":
            n["rate"] = float(j["dropoutRate"])
        elif t == "crop-mirror":
            n.update(crop=tuple(j["crop"]), mirror=float(j["mirror"]), flip=float(j["flip"]))
        elif t in ("skip-src", "skip"):
            n["index"] = j["index"]
            n["mode"] = j.get("combineMode", "proj-add")
            n["layers"] = [self._build(s) for s in j.get("layers", []) if s["type"] != "initial"]
        elif t == "denet-corner":
            n.update(sample_feat=j["sampleFeat"], cost_factor=j["costFactor"], use_center=j["useCenter"])
            n["conv"] = self._build([s for s in j["layers"] if s["type"] == "conv"][0])
        elif t == "denet-sparse":
            n.update(gs=j["gridSize"], sn=j["sampleNum"], sample_gt=j["sampleGT"], local_max=j["localMax"],
                     thr=j["cornerThreshold"], random_sample=j["randomSample"])
        elif t == "denet-detect":
            ot = j["overlapThreshold"]
            ot = (float(ot[0]), float(ot[1])) if isinstance(ot, (list, tuple)) else (float(ot), float(ot))
            n.update(cost_factor=j["costFactor"], bbox_factor=j["bboxFactor"], class_num=j["classNum"], thresholds=ot,
                     jointfit=j["useJointFitness"], bounded=j["useBoundedIoU"], indfit_factor=j.get("fitnessFactor", 0.0))
            n["conv"] = self._build([s for s in j["layers"] if s["type"] == "conv"][0])
        elif t in ("regression", "split", "identity"):
            pass
        else:
            raise NotImplementedError("oracle: layer type " + t)
        return n

    # ---- parameter enumeration -------------------------------------------------------------------------
    def params(self):
        out = []

        def rec(n):
            if n["type"] == "deconv" or (n["type"] == "conv" and n["enabled"]):
                out.append(n["w"])
                if n["b"] is not None:
                    out.append(n["b"])
            elif n["type"] in ("batchnorm", "batchnorm-relu") and n["enabled"]:
                out.extend([n["gamma"], n["beta"]])
            for s in n.get("layers", []):
                rec(s)
            if "conv" in n:
                rec(n["conv"])

        for n in self.nodes:
            rec(n)
        return out

    def bn_nodes(self):
        out = []

        def rec(n):
            if n["type"] in ("batchnorm", "batchnorm-relu") and n["enabled"]:
                out.append(n)
            for s in n.get("layers", []):
                rec(s)

        for n in self.nodes:
            rec(n)
        return out

    # ---- forward ---------------------------------------------------------------------------------------
    def _resnet(self, n, x, train):
        pre = "pre-activation" in n["version"]
        fused = ("bnrelu" in n["version"]) and n["activation"] == "relu"
        nb = 1 if fused else 2
        n_main = (nb if pre else 0) + 1 + nb + 1 + ((nb + 1) if n["bottleneck"] > 0 else 0) + (0 if pre else 1)
        main, sc = n["layers"][:n_main], n["layers"][n_main:]
        h = x
        first_bn_out = None
        for i, s in enumerate(main):
            h = self._apply(s, h, train)
            if pre and i == 0:
                first_bn_out = h
        if sc:
            r = first_bn_out if pre else x
            for s in sc:
                r = self._apply(s, r, train)
        else:
            r = x
        y = op_add(r, h)
        if not pre and n["activation"] != "none":
            y = op_relu(y)
        return self._forced(y, "resnet-out")

    def _apply(self, n, x, train, li=0):
        t = n["type"]
        if t == "deconv":
            return self._forced(op_deconv(x, n["w"], n["b"], n["stride"], n["pad"], need_dx=x is not self.x_in), t)
        if t == "border":
            return self._forced(op_border(x, n["border"]), t)
        if t == "dropout":
            if not train:
                return x
            return self._forced(op_dropout(x, n["rate"], L.layer_seed(self.rng_seed, li, self.iteration)), t)
        if t == "crop-mirror":
            N, _, H, W = x.v.shape
            geom = L.crop_mirror_geom(N, H, W, n["crop"][0], n["crop"][1], n["mirror"], n["flip"], train,
                                      L.layer_seed(self.rng_seed, li, self.iteration))
            return self._forced(op_crop_mirror(x, n["crop"], geom), t)
        if t == "conv":
            return self._forced(op_conv(x, n["w"], n["b"], n["stride"], n["pad"], need_dx=x is not self.x_in), t)
        if t == "batchnorm":
            return self._forced(op_bn(x, n, train, False), t) if n["enabled"] else x
        if t == "batchnorm-relu":
            return self._forced(op_bn(x, n, train, True), t)
        if t == "activation":
            return x if n["activation"] == "none" else self._forced(op_relu(x), t)
        if t == "pool":
            return self._forced(op_pool(x, "max" if n["mode"] == "max" else "avg", n["k"], n["s"], n["p"]), t)
        if t == "pool-inv":
            return self._forced(op_pool_inv(x, n["size"]), t)
        if t == "resnet":
            return self._resnet(n, x, train)
        raise NotImplementedError(t)

    def forward(self, x_nchw, metas=None, train=True, sample_override=None):
        """runs all layers; returns list of cost terms [(name, value)]; gradients are seeded on the tape"""
        self.x_in = T(np.asarray(x_nchw, dtype=F32))
        self._fpos = 0
        self.force_err = []
        h = self.x_in
        self.acts = {}
        self.costs = []
        self.cost_roots = []
        taps_src = {}
        corner = None
        B = self.batch_size
        for li, n in enumerate(self.nodes, 1):
            t = n["type"]
            if t in ("split", "identity"):
                pass
            elif t == "skip-src":
                taps_src[n["index"]] = h
            elif t == "skip":
                tap = taps_src[n["index"]]
                if n["mode"] == "concat":
                    h = self._forced(op_concat(h, tap), "skip")
                    self.acts[li] = h.v
                    continue
                if n["layers"]:
                    tap = self._apply(n["layers"][0], tap, train)
                h = self._forced(op_add(h, tap), "skip")
            elif t == "denet-corner":
                cn = 5 if n["use_center"] else 4
                conv_out = self._apply(n["conv"], h, train)
                pr = L.corner_pr(conv_out.v[:, :cn])
                corner = {"node": n, "conv_out": conv_out, "pr": pr, "cn": cn}
                self.corner_pr = pr
                if train and metas is not None:
                    target = L.corner_target(metas, pr.shape)
                    self.corner_target = target
                    cost, g = L.corner_cost(target, pr, n["cost_factor"])
                    self.costs.append(("denet-corner", cost))
                    full = np.zeros_like(conv_out.v)
                    full[:, :cn] = g
                    conv_out.add_grad(full)
                    self.cost_roots.append(conv_out)
            elif t == "denet-sparse":
                sn, gs = n["sn"], n["gs"]
                if sample_override is not None:
                    lists = [list(s) for s in sample_override]
                else:
                    lists = oracle_build_samples(corner["pr"], n["thr"], sn, 1024, n["local_max"])
                    if train:
                        lists = L.edit_samples(lists, metas, sn * sn, n["random_sample"], n["sample_gt"])
                self.sample_bbox_list = lists
                bbox = L.bbox_array(lists, B, sn)
                self.sample_bbox = bbox
                cn = corner["cn"]
                conv_out = corner["conv_out"]
                fmap = conv_out.v[:, cn:]
                out, taps = L.sparse_sample(fmap, bbox, gs, self.tap_rule)
                self.taps = taps

                def bw(g, conv_out=conv_out, taps=taps, fshape=fmap.shape, gs=gs, cn=cn):
                    full = np.zeros_like(conv_out.v)
                    full[:, cn:] = L.sparse_sample_grad(g, taps, fshape, gs)
                    conv_out.add_grad(full)

                h = self._forced(T(out, (conv_out,), bw), "sparse")
            elif t == "denet-detect":
                out = self._apply(n["conv"], h, train)
                self.detect_out = out
                if train and metas is not None:
                    sn = out.v.shape[2]
                    fit = 5 if n["jointfit"] else 6
                    s0 = (n["class_num"] * fit + 1) if n["jointfit"] else n["class_num"] + 1
                    use_reg = n["bbox_factor"] > 0.0
                    use_indfit = n["indfit_factor"] > 0.0
                    tg = L.detect_target(metas, self.sample_bbox_list, B, sn, n["class_num"], n["thresholds"], use_reg,
                                         n["jointfit"], use_indfit)
                    det_t, valid, reg_t = tg[:3]
                    self.detect_target = (det_t, valid, reg_t)
                    dc, bc, g = L.detect_cost(out.v, det_t, valid, reg_t, self.sample_bbox, s0, n["cost_factor"],
                                              n["bbox_factor"], n["bounded"])
                    if use_indfit:
                        fc, fg = L.indfit_cost(out.v, tg[3], s0 + (4 if use_reg else 0), n["indfit_factor"])
                        bc += fc
                        g = g + fg
                    self.costs.append(("denet-detect", dc + bc))
                    self.detect_cost_terms = (dc, bc)
                    out.add_grad(g)
                    self.cost_roots.append(out)
            elif t == "regression":
                if train and metas is not None:
                    classes = np.array([m["image_class"] for m in metas])
                    cost, g = L.regression_cost(h.v, classes)
                    self.costs.append(("regression", cost))
                    h.add_grad(g)
                    self.cost_roots.append(h)
            else:
                h = self._apply(n, h, train, li)
            self.acts[li] = h.v
        self.out = h
        return self.costs

    def forward_costs(self, x_nchw, metas, sample_override=None):
        """the forward half of train_step alone (training mode: batch statistics, targets, cost terms) -> (total, [terms]); no
        backward sweep, no update - what a free-running comparison of activations / corner map / costs needs (a third of the time)"""
        costs = self.forward(x_nchw, metas, True, sample_override)
        return sum(c for _, c in costs), [c for _, c in costs]

    def train_step(self, x_nchw, metas, iteration, lr, momentum, decay, solver="nesterov", sample_override=None):
        """denet/model/model_cnn.py:407-445 + the solver updates of :282-331"""
        for p in self.params():
            p.g = None
        self.iteration = iteration
        costs = self.forward(x_nchw, metas, True, sample_override)
        backprop(self.cost_roots)
        for p in self.params():
            g = p.g if p.g is not None else np.zeros_like(p.v)
            if solver == "adam":
                if getattr(p, "v2", None) is None:
                    p.v2 = np.zeros_like(p.v)
                p.v, p.m, p.v2 = L.adam_update(p.v, p.m, p.v2, g.astype(F32), lr, momentum, iteration, decay, p.is_weight)
            else:
                p.v, p.m = L.solver_update(p.v, p.m, g.astype(F32), lr, momentum, iteration, decay, p.is_weight, solver)
        for n in self.bn_nodes():
            if "new_stats" in n:
                n["mean"], n["stdinv"] = n.pop("new_stats")
        total = sum(c for _, c in costs)
        return total, [c for _, c in costs]


# ---------------------------------------------------------------------------------------------------------
# binding of the C++ restatement (oracle/build_samples.cc)
# ---------------------------------------------------------------------------------------------------------
_lib = None


def oracle_lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle.so is missing: run `make -C oracle` (or __graft_entry__.build())")
        _lib = ctypes.CDLL(path)
    return _lib


def oracle_build_samples_raw(corner_pr, corner_threshold, sample_num, max_corners=1024, local_max=0, cluster_threshold=1.0):
    pr = np.ascontiguousarray(corner_pr, dtype=F32)
    B, _, Cn, H, W = pr.shape
    S = sample_num * sample_num
    out = np.zeros((B, S, 5), F32)
    box = np.zeros((B, S, 4), np.int32)
    absd = np.zeros((B, S), F32)
    cnt = np.zeros(B, np.int32)
    f = oracle_lib().oracle_build_samples
    f.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                           ctypes.c_float] + [ctypes.c_void_p] * 4
    f.restype = ctypes.c_int
    rc = f(pr.ctypes.data, B, Cn, H, W, float(corner_threshold), sample_num, max_corners, local_max, float(cluster_threshold),
           out.ctypes.data, box.ctypes.data, absd.ctypes.data, cnt.ctypes.data)
    if rc != 0:
        raise RuntimeError("oracle_build_samples failed")
    return out, box, absd, cnt


def oracle_build_samples(corner_pr, corner_threshold, sample_num, max_corners=1024, local_max=0, cluster_threshold=1.0):
    """list[B] of list[(pr, (x0,y0,x1,y1))] like c_code.build_samples (denet/layer/denet_sparse.cc:587-592)"""
    out, _, _, cnt = oracle_build_samples_raw(corner_pr, corner_threshold, sample_num, max_corners, local_max,
                                              cluster_threshold)
    res = []
    for b in range(out.shape[0]):
        rows = out[b, :cnt[b]].tolist()
        res.append([(r[0], (r[1], r[2], r[3], r[4])) for r in rows])
    return res


def oracle_cluster_ranked(ranked, cluster_threshold, output_num):
    """apply_cluster + final ranking (denet_sparse.cc:165-242, 543-545) on an already ranked candidate list [n, 5] (pr, x0, y0,
    x1, y1) -> clustered list [m, 5]; see oracle/build_samples.cc"""
    r = np.ascontiguousarray(ranked, dtype=F32)
    out = np.zeros((max(output_num, 1), 5), F32)
    cnt = ctypes.c_int(0)
    f = oracle_lib().oracle_cluster_ranked
    f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    f.restype = ctypes.c_int
    if f(r.ctypes.data, int(r.shape[0]), float(cluster_threshold), int(output_num), out.ctypes.data, ctypes.addressof(cnt)) != 0:
        raise RuntimeError("oracle_cluster_ranked failed")
    return out[:cnt.value]
