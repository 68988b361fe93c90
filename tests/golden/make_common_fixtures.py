"""Generates tests/golden/common_fixtures.json and tests/golden/ref_codec.mdl.gz by IMPORTING the reference's
`denet.common` package (the only part of the reference that imports without Theano) in the build container:

    python tests/golden/make_common_fixtures.py          # needs /root/reference; never runs on the GPU box

The fixtures hold inputs and the reference's outputs for the host helpers that sit on the hot path's boundary:
  convert_num      denet/common/__init__.py:142-149   (layer-description parameters, model_cnn.py:130)
  get_params_dict  denet/common/__init__.py:200-208   (predict parameters)
  ndarray_unpack   denet/common/__init__.py:125-133   (flat target vectors, model_cnn.py:560, denet_detect.py:247)
  overlap / overlap_rel / overlap_iou   :91-110       (RoI coverage, denet_sparse.py:176)
  numpy_to_json / json_to_gz            denet/common/json_util.py:8-37  (the .mdl.gz parameter codec)
  multi.shared.ModelUpdate.set_mean_*   denet/multi/shared.py:105-119   (the parameter averaging of model-train-multi,
                                        train_multi.py:131-137; the build's DataParallel.average_state must give the same mean)
Only data is written: no reference source text."""
import json
import os
import sys

import numpy

sys.path.insert(0, "/root/reference")
import denet.common as C  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
rng = numpy.random.RandomState(20260928)

fix = {}
strings = ["64", "7", "-3", "0.01", "1e-3", "5.", "abc", "1.0", "0x10", " 12 ", "3,4", "", "True", "nan", "inf", "-0.5e2"]
fix["convert_num"] = [{"in": s, "out": C.convert_num(s), "type": type(C.convert_num(s)).__name__} for s in strings]
for e in fix["convert_num"]:
    if isinstance(e["out"], float) and (e["out"] != e["out"] or e["out"] in (float("inf"), float("-inf"))):
        e["out"] = repr(e["out"])
pstr = ["prThreshold=0.05,nmsThreshold=0.3,useSoftNMS=1", "scale,cornerMax=512", "a=b,c", "x=1e-2"]
fix["get_params_dict"] = [{"in": s, "out": C.get_params_dict(s)} for s in pstr]

v = rng.randn(2 * 3 * 4 + 5 + 6).astype(numpy.float32)
shapes = [(2, 3, 4), (5,), (2, 3)]
out = C.ndarray_unpack(v, shapes)
fix["ndarray_unpack"] = {"v": v.tolist(), "shapes": [list(s) for s in shapes], "out": [o.tolist() for o in out]}

pairs = []
for _ in range(64):
    a = numpy.sort(rng.uniform(-0.2, 1.2, (2, 2)), axis=0).T.reshape(-1)  # x0,y0,x1,y1 (may leave the unit square)
    b = numpy.sort(rng.uniform(0, 1, (2, 2)), axis=0).T.reshape(-1)
    a, b = tuple(float(x) for x in a), tuple(float(x) for x in b)
    pairs.append({"a": a, "b": b, "overlap": C.overlap(a, b), "iou": C.overlap_iou(a, b), "rel": C.overlap_rel(a, b),
                  "iou_unit": C.overlap_iou(a), "rel_unit": C.overlap_rel(a)})
pairs.append({"a": (0.1, 0.1, 0.1, 0.5), "b": (0.0, 0.0, 1.0, 1.0), "overlap": C.overlap((0.1, 0.1, 0.1, 0.5), (0, 0, 1, 1)),
              "iou": C.overlap_iou((0.1, 0.1, 0.1, 0.5), (0, 0, 1, 1)), "rel": C.overlap_rel((0.1, 0.1, 0.1, 0.5), (0, 0, 1, 1)),
              "iou_unit": C.overlap_iou((0.1, 0.1, 0.1, 0.5)), "rel_unit": C.overlap_rel((0.1, 0.1, 0.1, 0.5))})
fix["overlap"] = pairs

# parameter codec: arrays encoded by the reference, with their plain values beside them
arrays = {
    "f32_4d": rng.randn(3, 2, 3, 3).astype(numpy.float32),
    "f64_1d": rng.randn(7),
    "i64_2d": rng.randint(-5, 5, (2, 5)),
    "f32_0d": numpy.array(1.5, dtype=numpy.float32),
    "f32_empty": numpy.zeros((0, 4), dtype=numpy.float32),
    "f32_fortran": numpy.asfortranarray(rng.randn(3, 4).astype(numpy.float32)),
}
fix["codec"] = {k: {"encoded": C.numpy_to_json(a), "dtype": str(a.dtype), "shape": list(a.shape),
                    "values": a.astype(numpy.float64).reshape(-1).tolist()} for k, a in arrays.items()}

# a model file as the reference writes it (json_to_gz): one conv layer dictionary with the keys of
# ConvLayer.export_json (convolution.py:126-136) and one BN dictionary (batch_norm.py:109-121)
w = rng.randn(32, 3, 3, 3).astype(numpy.float32)
b = rng.randn(32).astype(numpy.float32)
model = {"layers": [
    {"type": "conv", "shape": [32, 3, 3, 3], "stride": [1, 1], "border": "half", "enabled": True, "useBias": True, "bias": b, "weight": w},
    {"type": "batchnorm", "enabled": True, "momentum": 0.9, "eps": 1e-5, "mean": rng.randn(32).astype(numpy.float32),
     "std": rng.rand(32).astype(numpy.float32) + 0.5, "gamma": rng.randn(32).astype(numpy.float32), "bias": rng.randn(32).astype(numpy.float32)},
]}
C.json_to_gz(os.path.join(HERE, "ref_codec.mdl.gz"), model)
fix["mdl"] = {"weight": w.reshape(-1).tolist(), "bias": b.tolist(), "mean": model["layers"][1]["mean"].tolist(),
              "std": model["layers"][1]["std"].tolist()}

# DatasetAbstract.export / shuffle (denet/dataset/__init__.py:196-201, 349-366): array-type samples, last batch padded
# with random.randint draws; the ids of the exported samples and the generator state afterwards are the contract
import random  # noqa: E402
import denet.dataset as D  # noqa: E402
ds = D.DatasetAbstract()
samples = [rng.rand(3, 4, 4).astype(numpy.float32) for _ in range(7)]
ds.data = [("s%d" % i, samples[i], {"id": i, "image_class": i % 3}) for i in range(7)]
random.seed(42)
ds.shuffle()
x, metas, n = ds.export(4)
fix["dataset"] = {"samples": [a.reshape(-1).tolist() for a in samples], "seed": 42, "batch": 4, "size": n,
                  "ids": [m["id"] for m in metas], "x_shape": list(x.shape), "x_sum": float(x.astype(numpy.float64).sum()),
                  "next_random": random.random()}

# the reference's multi-GPU exchange: every worker's update targets are summed into a zeroed ModelUpdate in worker order and
# multiplied by 1/N (shared.py:105-119: fill 0, add_array per worker, mul_value(1.0 / N))
import tempfile  # noqa: E402
import denet.multi.shared as SH  # noqa: E402
dims = [(3, 5), (17,), (2, 2, 3)]
with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as tf:
    json.dump({"input": [2, 3, 8, 8], "output": [2, 10], "dims": [{"shape": list(d)} for d in dims]}, tf)
fix["multi_mean"] = []
for workers in (2, 3):
    ups = []
    for wk in range(workers):
        u = SH.ModelUpdate(tf.name)
        for a, d in zip(u.updates, dims):
            a.get_array()[...] = (rng.randn(*d) * 10 ** rng.uniform(-3, 3)).astype(numpy.float32)
        ups.append(u)
    mean = SH.ModelUpdate(tf.name)
    mean.set_mean_init()
    for u in ups:
        mean.set_mean_update(u)
    mean.set_mean_finish()
    fix["multi_mean"].append({"workers": [[a.get_array().reshape(-1).tolist() for a in u.updates] for u in ups],
                              "mean": [a.get_array().reshape(-1).tolist() for a in mean.updates]})
os.unlink(tf.name)

with open(os.path.join(HERE, "common_fixtures.json"), "w") as f:
    json.dump(fix, f)
print("written", os.path.getsize(os.path.join(HERE, "common_fixtures.json")), "bytes")
