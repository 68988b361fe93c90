"""Generates tests/golden/layer_method_fixtures.json by EXECUTING METHODS OF THE REFERENCE'S LAYER CLASSES in the build container:

    python tests/golden/make_layer_method_fixtures.py          # needs /root/reference; never runs on the GPU box

The reference's layer modules cannot be imported (each one imports Theano at module level, and Theano is not installed).
Several of their methods, however, are plain host code - numpy / math / random / `denet.common` only. This script parses the
module with `ast`, takes the FunctionDef node of such a method, compiles THAT NODE (no source text is read into a string,
stored or written anywhere) and calls the resulting function with a plain attribute holder as `self`. What it records is
data: the inputs of a scenario and what the reference's own code returned for them.

  executed                                   reference lines               what it pins
  DeNetCornerLayer.get_target                denet_corner.py:81-123        corner-target rasteriser: round-half-even cell, x1 - 1,
                                                                           max(x0, .), on-screen tests, centre point, 1 - p plane,
                                                                           normaliser W * H * corner_num
  DeNetSparseLayer.get_target                denet_sparse.py:164-207       training-time RoI list editing: keep count
                                                                           n = S - floor(random_sample * S), random.sample trim, the 4
                                                                           uniform draws per random box and their order, ground
                                                                           truth written from the tail backwards, stdlib random
                                                                           stream position afterwards
  DeNetDetectLayer.get_target                denet_detect.py:147-236       RoI -> class / fitness-bin / box-regression targets (index ->
   (its LOOP; see below)                                                   (i, j) mapping, joint and independent fitness bins, argmax
                                                                           object for the regression, normalisation, packing order)

  not executable this way, and why
  DeNetDetectLayer.get_target's overlap matrix   theano_util.get_overlap_iou (theano_util.py:38-59) evaluates a COMPILED Theano
                                             graph. The method is run with that one call answered by the numpy fp32 evaluation
                                             of the graph's expression (min / max / subtract / multiply / divide in the order of
                                             theano_util.py:43-50) - third-party arithmetic restated, so this fixture pins the loop,
                                             not the IoU rounding.
  DeNetSparseLayer.get_samples / get_bbox_array / set_samples, DeNetDetectLayer.get_detections
                                             call `c_code.*` (the C++ module Theano's cmodule builds from denet_sparse.cc /
                                             denet_detect.cc, which needs Theano's headers) and compiled Theano functions.
  RegressionLayer.get_target                 reads `theano.config.floatX` (regression.py:94).
  every cost() / get_errors()                symbolic Theano graphs.

Only data is written: no reference source text."""
import ast
import copy
import json
import math
import os
import random
import sys
import types

import numpy

REF = "/root/reference"
sys.path.insert(0, REF)
import denet.common as common  # noqa: E402
import denet.common.logging as ref_logging  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ref_logging.init()
ref_logging.setLevel("ERROR")


def reference_method(rel_path, class_name, method_name, namespace):
    """the function object of `class_name.method_name` of the reference file, compiled from its AST node alone"""
    path = os.path.join(REF, rel_path)
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == class_name:
            for item in node.body:
                if isinstance(item, ast.FunctionDef) and item.name == method_name:
                    mod = ast.Module(body=[item], type_ignores=[])
                    ns = dict(namespace)
                    exec(compile(mod, path, "exec"), ns)
                    return ns[method_name], (item.lineno, item.end_lineno)
    raise KeyError("%s.%s not found in %s" % (class_name, method_name, rel_path))


def f32_overlap_iou(obj_bboxs, sample_bboxs):
    """theano_util.py:38-59 with the compiled graph's expression evaluated by numpy in float32 (see the module docstring)"""
    if len(obj_bboxs) == 0 or len(sample_bboxs) == 0:
        return None
    x = numpy.array(obj_bboxs, dtype=numpy.float32)
    y = numpy.array(sample_bboxs, dtype=numpy.float32)
    x_area = (x[:, 2] - x[:, 0]) * (x[:, 3] - x[:, 1])
    y_area = (y[:, 2] - y[:, 0]) * (y[:, 3] - y[:, 1])
    dx = numpy.maximum(numpy.minimum(x[:, None, 2], y[None, :, 2]) - numpy.maximum(x[:, None, 0], y[None, :, 0]), numpy.float32(0))
    dy = numpy.maximum(numpy.minimum(x[:, None, 3], y[None, :, 3]) - numpy.maximum(x[:, None, 1], y[None, :, 1]), numpy.float32(0))
    inter = dx * dy
    union = x_area[:, None] + y_area[None, :] - inter
    return inter / union


def rand_boxes(rng, n, lo=-0.15, hi=1.15):
    a = numpy.sort(rng.uniform(lo, hi, (n, 2, 2)), axis=1)          # [n, (min, max), (x, y)]
    return [(float(b[0, 0]), float(b[0, 1]), float(b[1, 0]), float(b[1, 1])) for b in a]


# ------------------------------------------------------------------------------------------------------------------------
def corner_scenarios():
    fn, lines = reference_method("denet/layer/denet_corner.py", "DeNetCornerLayer", "get_target", {"numpy": numpy})
    rng = numpy.random.RandomState(81123)
    out = []
    for (B, H, W, center) in [(2, 64, 64, False), (3, 16, 16, True), (2, 5, 7, False), (2, 9, 4, True), (1, 32, 32, False)]:
        cn = 5 if center else 4
        metas = []
        for b in range(B):
            boxes = rand_boxes(rng, int(rng.randint(0, 7)))
            # cell edges and half-cell positions: Python's round() is half-to-even, x1 - 1 can fall below x0, 1.0 maps to W - 1
            k = rng.randint(0, W + 1, 4)
            boxes.append((k[0] / W, k[1] / H if k[1] <= H else 1.0, min(1.0, (k[0] + k[2] % 3) / W), min(1.0, (k[1] % H + 1) / H)))
            boxes.append(((k[0] + 0.5) / W, (k[1] % H + 0.5) / H, (k[2] + 0.5) / W, (k[3] % H + 1.5) / H))
            boxes.append((0.0, 0.0, 1.0, 1.0))
            boxes.append((0.25, 0.25, 0.25, 0.25))                # zero area
            boxes.append((-0.3, 0.1, 0.2, 1.4))                   # partly off screen
            if b == 0:
                boxes = boxes[:3]
            metas.append({"bbox": [tuple(float(v) for v in bx) for bx in boxes]})
        if B > 1:
            metas[-1] = {"bbox": []}                              # an image without objects
        holder = types.SimpleNamespace(corner_shape=(B, 2, cn, H, W), width=W, height=H, use_center=center, corner_num=cn,
                                       dropout=0.0)
        yt_index, yt_value = fn(holder, None, None, copy.deepcopy(metas))
        t = yt_value.reshape(B, 2, cn, H, W)
        pos = numpy.argwhere(t[:, 1] > 0)
        vals = sorted(set(float(v) for v in numpy.unique(t)))
        assert yt_index.size == 0 and yt_index.dtype == numpy.int64 and yt_value.dtype == numpy.float32
        # plane 0 is the complement of plane 1 everywhere (checked here, so the fixture only stores the positive cells)
        assert numpy.array_equal(t[:, 0] > 0, ~(t[:, 1] > 0))
        out.append({"B": B, "H": H, "W": W, "use_center": center, "metas": metas, "positive_cells": pos.tolist(),
                    "distinct_values": vals, "sum": float(yt_value.astype(numpy.float64).sum())})
    return {"reference": "denet/layer/denet_corner.py:%d-%d DeNetCornerLayer.get_target" % lines, "cases": out}


# ------------------------------------------------------------------------------------------------------------------------
def sparse_scenarios():
    ns = {"numpy": numpy, "math": math, "random": random, "common": common, "logging": ref_logging}
    fn, lines = reference_method("denet/layer/denet_sparse.py", "DeNetSparseLayer", "get_target", ns)
    rng = numpy.random.RandomState(164207)
    out = []
    for case, (sn, random_sample, sample_gt, B) in enumerate([(4, 0.0, True, 3), (4, 0.25, True, 4), (6, 0.1, True, 3), (5, 0.5, False, 2),
                                                             (24, 0.1, True, 2), (3, 1.0, True, 2)]):
        S = sn * sn
        n_keep = S - math.floor(random_sample * S)
        counts = [0, S, max(0, n_keep - 2), n_keep, min(S, n_keep + 1), S // 2][:B] if case % 2 == 0 else \
                 [S, 1, n_keep + 1 if n_keep < S else S, 0][:B]
        samples, metas = [], []
        for b in range(B):
            n = counts[b % len(counts)]
            boxes = rand_boxes(rng, n, 0.0, 1.0)
            prs = numpy.sort(rng.uniform(0.0, 0.5, n))[::-1]
            samples.append([(float(numpy.float32(p)), bx) for p, bx in zip(prs, boxes)])
            gt = rand_boxes(rng, int(rng.randint(0, 5)), 0.0, 1.0)
            if b == 1 and n > 0:
                gt.append(boxes[0])                       # a ground-truth box the detector found exactly
            metas.append({"bbox": gt})
        seed = 1000 + case
        captured = {}
        holder = types.SimpleNamespace(sample_count=S, random_sample=random_sample, sample_gt=sample_gt,
                                       get_samples=lambda data_x, train=False, _s=samples: copy.deepcopy(_s),
                                       set_samples=lambda lists, _c=captured: _c.setdefault("lists", copy.deepcopy(lists)))
        random.seed(seed)
        ret = fn(holder, None, None, copy.deepcopy(metas))
        after = [random.random() for _ in range(3)]
        lists = captured["lists"]
        assert all(len(l) == S for l in lists)
        out.append({"sample_num": sn, "random_sample": random_sample, "sample_gt": sample_gt, "seed": seed,
                    "samples": [[[p, list(bx)] for p, bx in l] for l in samples], "metas": metas,
                    "returned_is_none": ret is None,
                    "edited": [[[float(p), [float(v) for v in bx]] for p, bx in l] for l in lists],
                    "random_after": after})
    return {"reference": "denet/layer/denet_sparse.py:%d-%d DeNetSparseLayer.get_target" % lines, "cases": out}


# ------------------------------------------------------------------------------------------------------------------------
def detect_scenarios():
    ns = {"numpy": numpy, "math": math, "random": random, "common": common, "logging": ref_logging,
          "theano_util": types.SimpleNamespace(get_overlap_iou=f32_overlap_iou)}
    fn, lines = reference_method("denet/layer/denet_detect.py", "DeNetDetectLayer", "get_target", ns)
    rng = numpy.random.RandomState(147236)
    out = []
    variants = [  # class_num, sample_num, thresholds, jointfit, bbox_reg, indfit
        (5, 4, (0.5, 0.5), False, True, False),
        (80, 6, (0.5, 0.3), False, True, False),
        (3, 5, (0.5, 0.5), True, True, False),
        (4, 4, (0.4, 0.6), False, False, True),
        (6, 3, (0.5, 0.5), True, False, False),
    ]
    for case, (ncls, sn, thr, jointfit, bbox_reg, indfit) in enumerate(variants):
        fnum = 5 if jointfit else 6            # denet_detect.py:59-66
        B, S = 3, sn * sn
        metas, lists = [], []
        for b in range(B):
            n_gt = [3, 0, 5][b]
            gt = rand_boxes(rng, n_gt, 0.0, 1.0)
            cls = [int(c) for c in rng.randint(0, ncls, n_gt)]
            n = [S, S, S - 2][b] if case != 1 else [S, 0, S][b]
            boxes = rand_boxes(rng, n, 0.0, 1.0)
            # RoIs near the ground truth at graded IoU (all fitness bins), one exact copy, one with IoU exactly at a threshold
            for i, g in enumerate(gt):
                if 3 * i + 2 < n:
                    for j, shrink in enumerate((0.02, 0.12, 0.3)):
                        w, h = g[2] - g[0], g[3] - g[1]
                        boxes[3 * i + j] = (g[0] + shrink * w, g[1], g[2], g[3] - shrink * h * 0.5)
            if n_gt and n:
                boxes[-1] = gt[0]
            metas.append({"bbox": gt, "class": cls})
            lists.append([(0.0, tuple(float(v) for v in bx)) for bx in boxes])
        null_class = ncls * fnum if jointfit else ncls
        det_shape = (B, null_class + 1, sn, sn)
        holder = types.SimpleNamespace(det_shape=det_shape, null_class=null_class, use_bbox_reg=bbox_reg, use_indfit=indfit,
                                       use_jointfit=jointfit, batch_size=B, sample_num=sn, fitness_num=fnum,
                                       indfit_shape=(B, fnum, sn, sn), overlap_threshold=thr, class_num=ncls,
                                       sparse_layer=types.SimpleNamespace(sample_bbox_list=lists, sample_num=sn))
        yt_index, yt_value = fn(holder, None, None, copy.deepcopy(metas))
        assert yt_index.size == 0 and yt_value.dtype == numpy.float32
        out.append({"class_num": ncls, "sample_num": sn, "overlap_threshold": list(thr), "use_jointfit": jointfit,
                    "use_bbox_reg": bbox_reg, "use_indfit": indfit, "fitness_num": fnum, "metas": metas,
                    "sample_bbox_list": [[[p, list(bx)] for p, bx in l] for l in lists],
                    "yt_value_f32_hex": yt_value.astype("<f4").tobytes().hex(), "yt_len": int(yt_value.size)})
    return {"reference": "denet/layer/denet_detect.py:%d-%d DeNetDetectLayer.get_target (loop executed; overlap matrix = numpy fp32 "
                         "evaluation of the graph of denet/common/theano_util.py:38-59)" % lines, "cases": out}


if __name__ == "__main__":
    fix = {"corner_target": corner_scenarios(), "sparse_get_target": sparse_scenarios(), "detect_target": detect_scenarios()}
    path = os.path.join(HERE, "layer_method_fixtures.json")
    with open(path, "w") as f:
        json.dump(fix, f)
    print("wrote", path, os.path.getsize(path), "bytes;", {k: len(v["cases"]) for k, v in fix.items()})
