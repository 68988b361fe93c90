"""CPU tests of the host logic (no GPU compute): model-desc grammar, graph assembly, JSON surface, targets,
RoI editing, the C-ABI export table and the N>1 data-parallel path over gloo."""
import ctypes
import json
import os
import random
import re
import subprocess
import sys

import numpy as np
import pytest

from denet_amd import lib as dlib
from denet_amd.common import json_util
from denet_amd.model import model_cnn, modify, zoo
from oracle import layers as OL

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def denet_small():
    return zoo.denet34(2, "skip", 128, class_num=80, seed=1)


def test_denet34_skip_shape_table():
    """parse_desc + the model-modify sequence of papers/dss/denet34.sh reproduce the SURVEY §8 table"""
    m = zoo.denet34(32, "skip", 512)
    tab = json.load(open(os.path.join(GOLDEN, "denet34_skip_shapes.json")))
    assert len(m.layers) == len(tab["layers"])
    for (idx, tname, shape), layer in zip(tab["layers"], m.layers):
        assert layer.type_name == tname, (idx, layer.type_name, tname)
        assert tuple(layer.output_shape[1:]) == tuple(shape), (idx, layer.output_shape, shape)
    n = sum(p.value.size for l in m.layers for p in l.weights() + l.biases())
    assert abs(n / 1e6 - tab["trainable_parameters_millions"]) < 0.01
    # weights() then biases(): BN gamma/beta are biases (no decay), conv omega is a weight
    rsn = m.layers[4]
    assert len(rsn.weights()) == 2 and len(rsn.biases()) == 4 and len(rsn.updates()) == 4
    dnc = m.layers[30]
    conv = dnc.layers[-1]
    assert np.all(conv.omega.value[:4] == 0) and np.all(conv.beta.value[:4] == 5.0) and conv.pad == 0
    dnd = m.layers[40]
    assert np.all(dnd.layers[0].omega.value == 0) and dnd.layers[0].filter_shape[0] == 85


def test_desc_grammar_and_errors():
    m = model_cnn.ModelCNN()
    m.batch_size = 4
    m.class_num = 10
    m.build(zoo.CIFAR3_DESC, (3, 32, 32), "relu", "half", ["he-backward"])
    names = [l.type_name for l in m.layers]
    assert names == ["initial", "conv", "batchnorm", "activation", "pool", "conv", "batchnorm", "activation", "pool",
                     "conv", "batchnorm", "activation", "pool", "conv", "regression"]
    assert m.layers[-1].output_shape == (4, 10)
    assert m.layers[12].mode == "average_inc_pad" and m.layers[12].output_shape == (4, 512, 1, 1)
    with pytest.raises(Exception):
        m.build_layer("ZZ[1]", m.layers, "relu", "half", "he-backward")
    with pytest.raises(Exception):   # bare P has no size in this version of the reference (pool.py:51)
        m.build_layer("P.A", list(m.layers[:4]), "relu", "half", "he-backward")
    # C.X[filters,kh,kw,sh,sw], C.B = bias ON (code wins over README)
    ls = list(m.layers[:1])
    m.build_layer("C.BX[16,3,3,2,2]", ls, "relu", "half", "he-backward")
    assert ls[-1].use_bias and ls[-1].stride == (2, 2) and ls[-1].output_shape == (4, 16, 16, 16)


def test_json_roundtrip_and_layout(denet_small, tmp_path):
    m = denet_small
    j = m.export_json()
    assert j["version"] == 3 and j["layers"][0]["type"] == "conv"
    assert set(["shape", "stride", "border", "enabled", "useBias", "bias", "weight"]) <= set(j["layers"][0].keys())
    assert set(["momentum", "eps", "mean", "std", "gamma", "bias"]) <= set(j["layers"][1].keys())
    f = str(tmp_path / "m.mdl.gz")
    model_cnn.save_to_file(m, f)
    m2 = model_cnn.load_from_file(f, 2)
    for a, b in zip(m.layers, m2.layers):
        assert a.type_name == b.type_name and a.output_shape == b.output_shape
        for pa, pb in zip(a.params(), b.params()):
            np.testing.assert_array_equal(pa.value, pb.value)
    # device layout: flipped KRSC with channel padding
    conv = m.layers[1]
    d = conv.omega.to_dev_layout()
    assert d.shape == (64, 7, 8, 4)
    w = conv.omega.value
    assert d[5, 1, 2, 1] == w[5, 1, 5, 4] and np.all(d[:, :, 7, :] == 0) and np.all(d[:, :, :, 3] == 0)
    np.testing.assert_array_equal(conv.omega.from_dev_layout(d), w)


def test_modify_operations():
    np.random.seed(3)
    m = model_cnn.ModelCNN()
    m.batch_size = 2
    m.class_num = 10
    m.build("C.B[32,3] BN A nRSN.O[2,32,3] P.A[8] R", (3, 8, 8), "relu", "half", ["he-backward"])
    m2 = modify.convert_bn_relu(m)
    assert [l.type_name for l in m2.layers][:4] == ["initial", "conv", "batchnorm-relu", "resnet"]
    assert "bnrelu" in m2.layers[3].version and m2.layers[3].layers[2].type_name == "batchnorm-relu"
    np.testing.assert_array_equal(m.layers[1].omega.value, m2.layers[1].omega.value)
    m3 = modify.layer_remove(m2, 3)
    assert len(m3.layers) == len(m2.layers) - 3
    m4 = modify.layer_insert(m3, ["3:SKIPSRC.X[0]"])
    assert m4.layers[3].type_name == "skip-src" and m4.layers[3].has_split
    m5 = modify.layer_append(m4, "C[32,3] SKIP[0] BNA")
    assert [l.type_name for l in m5.layers][-3:] == ["conv", "skip", "batchnorm-relu"]


def test_random_mirror_matches_stdlib():
    from denet_amd.layer.denet_sparse import py_random_doubles
    random.seed(11)
    a = [random.random() for _ in range(257)]
    nxt = random.uniform(1, 2)
    random.seed(11)
    b = py_random_doubles(257)
    assert np.array_equal(np.array(a), b) and random.uniform(1, 2) == nxt


def test_roi_editing_matches_reference_loop(denet_small):
    dns = denet_small.layers[31]
    _, metas = zoo.synthetic_batch(2, 128, seed=4)
    rng = np.random.RandomState(3)
    prs, boxes, lists = [], [], []
    for b, k in enumerate([40, 576]):
        bx = np.sort(rng.uniform(0, 1, (k, 4)).astype(np.float32).astype(np.float64), axis=1)
        pr = rng.uniform(0, 0.5, k).astype(np.float32).astype(np.float64)
        prs.append(pr)
        boxes.append(bx)
        lists.append([(float(p), tuple(v)) for p, v in zip(pr.tolist(), bx.tolist())])
    random.seed(7)
    p2, b2 = dns.edit_samples(prs, boxes, metas)
    after = random.random()
    random.seed(7)
    ref = OL.edit_samples(lists, metas, 576, 0.1, True)
    assert random.random() == after
    for b in range(2):
        assert np.array_equal(np.array([s[1] for s in ref[b]]), b2[b])
        assert np.array_equal(np.array([s[0] for s in ref[b]]), p2[b])
    # bbox array == build_bbox_array
    dns.sample_pr, dns.sample_boxes = p2, b2
    np.testing.assert_array_equal(dns._bbox_array(b2), OL.bbox_array(ref, 2, 24))


@pytest.mark.parametrize("random_sample", [0.1, 0.0, 0.5])
def test_native_roi_editing_matches_reference_loop(denet_small, random_sample):
    """denet_host_edit_samples (one native call per batch) == the reference's Python loop on stdlib random"""
    dns = denet_small.layers[31]
    old = dns.random_sample
    dns.random_sample = random_sample
    try:
        _, metas = zoo.synthetic_batch(2, 128, seed=4)
        rng = np.random.RandomState(5)
        S = 576
        det = np.zeros((2, S, 5), np.float32)
        cnt = np.array([S, 31], np.int32)
        lists = []
        for b in range(2):
            k = cnt[b]
            det[b, :k, 1:] = np.sort(rng.uniform(0, 1, (k, 4)), axis=1)
            det[b, :k, 0] = rng.uniform(0, 0.5, k)
            lists.append([(float(r[0]), tuple(float(v) for v in r[1:])) for r in det[b, :k]])
        assert dns._native_edit_ok(metas)
        f32 = np.empty((2 * S, 4), np.float32)
        random.seed(9)
        pr, bx = dns.edit_samples_native(det, cnt, metas, f32)
        after = random.random()
        random.seed(9)
        ref = OL.edit_samples(lists, metas, S, random_sample, True)
        assert random.random() == after, "generator stream diverged"
        for b in range(2):
            assert np.array_equal(np.array([s[1] for s in ref[b]]), bx[b])
            assert np.array_equal(np.array([s[0] for s in ref[b]]), pr[b])
        np.testing.assert_array_equal(f32.reshape(2, 24, 24, 4), OL.bbox_array(ref, 2, 24))
    finally:
        dns.random_sample = old


@pytest.mark.parametrize("jointfit", [False, True, "indfit"])
def test_detect_and_corner_targets_match_reference_loops(jointfit):
    indfit = jointfit == "indfit"
    jointfit = jointfit is True
    head = zoo.DENET34_SKIP_DESC.replace("DND[0.5,1,1]", "DND.J[0.5,1,1]") if jointfit else None
    if indfit:
        head = zoo.DENET34_SKIP_DESC.replace("DND[0.5,1,1]", "DND[0.5,1,1,0.5]")
    m = zoo.denet34(2, "skip", 128, head_desc=head)
    dns, dnd, dnc = m.layers[31], m.layers[40], m.layers[30]
    _, metas = zoo.synthetic_batch(2, 128, seed=5)
    random.seed(2)
    prs = [np.zeros(0), np.zeros(0)]
    boxes = [np.zeros((0, 4)), np.zeros((0, 4))]
    dns.sample_pr, dns.sample_boxes = dns.edit_samples(prs, boxes, metas)
    # jittered copies of the ground truth: many IoU > t0 matches, multi-label cells, all fitness bins
    rng = np.random.RandomState(8)
    for b in range(2):
        gt = np.asarray(metas[b]["bbox"], dtype=np.float64)
        for k in range(200):
            g = gt[k % len(gt)]
            dns.sample_boxes[b][k] = np.clip(g + rng.normal(0, 0.04 * (1 + k % 5), 4), 0, 1)
    n_det, n_valid, n_reg = dnd.build_targets_numpy(metas)
    n_det, n_valid, n_reg = n_det.copy(), n_valid.copy(), n_reg.copy()
    n_fit = dnd._fit_target.copy() if indfit else None
    d_det, d_valid, d_reg = dnd.build_targets(metas)                       # native host path
    assert np.array_equal(n_det, d_det) and np.array_equal(n_valid, d_valid) and np.array_equal(n_reg, d_reg)
    if indfit:
        assert dnd.s2 == 6 and np.array_equal(n_fit, dnd._fit_target) and (n_fit[:, 1:] > 0).sum() > 100
    assert (d_det[:, :-1] > 0).sum() > 100
    idx, val = dnd.get_target(m, None, metas)
    tg = OL.detect_target(metas, dns.sample_bbox_list, 2, 24, 80, (0.5, 0.5), True, jointfit, indfit)
    det_t, valid, reg_t = tg[:3]
    ref = np.concatenate([det_t.flatten(), valid.flatten(), reg_t.flatten()] + ([tg[3].flatten()] if indfit else []))
    assert idx.size == 0 and val.dtype == np.float32
    np.testing.assert_array_equal(val, ref)
    assert (det_t.sum(axis=1) > 0).all() and valid.sum() > 0
    _, cval = dnc.get_target(m, None, metas)
    np.testing.assert_array_equal(cval.reshape(dnc.corner_shape), OL.corner_target(metas, dnc.corner_shape))


def test_host_helpers_match_imported_reference():
    """fixtures written by tests/golden/make_common_fixtures.py, which IMPORTS the reference's denet.common in the
    build container (the one reference package that loads without Theano)"""
    import json
    from denet_amd import common as C
    from denet_amd.common import json_util as J
    fix = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "common_fixtures.json")))
    for e in fix["convert_num"]:
        got = C.convert_num(e["in"])
        assert type(got).__name__ == e["type"], e
        assert (repr(got) == e["out"]) if isinstance(e["out"], str) and e["type"] == "float" else (got == e["out"]), e
    for e in fix["get_params_dict"]:
        assert C.get_params_dict(e["in"]) == e["out"], e
    u = fix["ndarray_unpack"]
    got = C.ndarray_unpack(np.array(u["v"], np.float32), [tuple(s) for s in u["shapes"]])
    for g, r in zip(got, u["out"]):
        assert np.array_equal(g, np.array(r, np.float32))
    for e in fix["overlap"]:
        a, b = tuple(e["a"]), tuple(e["b"])
        assert C.overlap(a, b) == e["overlap"] and C.overlap_iou(a, b) == e["iou"] and C.overlap_rel(a, b) == e["rel"]
        assert C.overlap_iou(a) == e["iou_unit"] and C.overlap_rel(a) == e["rel_unit"]
    for name, e in fix["codec"].items():
        arr = J.numpy_from_json(e["encoded"])
        assert str(arr.dtype) == e["dtype"] and list(arr.shape) == e["shape"], name
        assert np.array_equal(arr.astype(np.float64).reshape(-1), np.array(e["values"])), name
        back = J.numpy_from_json(json.loads(json.dumps(J.numpy_to_json(arr))))
        assert back.dtype == arr.dtype and np.array_equal(back, arr), name
    mdl = J.json_from_gz(os.path.join(os.path.dirname(__file__), "golden", "ref_codec.mdl.gz"))
    conv, bn = mdl["layers"]
    assert conv["weight"].dtype == np.float32 and conv["weight"].shape == (32, 3, 3, 3)
    assert np.array_equal(conv["weight"].reshape(-1), np.array(fix["mdl"]["weight"], np.float32))
    assert np.array_equal(conv["bias"], np.array(fix["mdl"]["bias"], np.float32))
    assert np.array_equal(bn["std"], np.array(fix["mdl"]["std"], np.float32))
    # dataset surface of train_epoch: same shuffle permutation, same padding draws, same generator state afterwards
    from denet_amd.model.train import ArrayDataset
    d = fix["dataset"]
    arrs = np.array(d["samples"], np.float32).reshape(-1, 3, 4, 4)
    ds = ArrayDataset(arrs, [{"id": i, "image_class": i % 3} for i in range(len(arrs))], 3)
    random.seed(d["seed"])
    ds.shuffle()
    x, metas, n = ds.export(d["batch"])
    assert n == d["size"] and list(x.shape) == d["x_shape"] and [m["id"] for m in metas] == d["ids"]
    assert abs(float(x.astype(np.float64).sum()) - d["x_sum"]) < 1e-9 and random.random() == d["next_random"]
    for k, m in enumerate(metas):
        assert np.array_equal(x[k], arrs[m["id"]])
    # the reference's layer dictionaries load into the product's layers (same JSON keys)
    from denet_amd.model.model_cnn import ModelCNN
    m = ModelCNN()
    m.batch_size, m.class_num = 1, 2
    m.build("C.B[32,3] BN", (3, 8, 8), "relu", "half", ["he-backward"])
    m.layers[1].import_json(conv)
    m.layers[2].import_json(bn)
    assert np.array_equal(m.layers[1].omega.get_value(), conv["weight"])
    assert np.array_equal(m.layers[2].stdinv.get_value(), bn["std"])      # "std" holds the running INVERSE std


def test_cabi_exports_every_declared_symbol():
    """the shared library loads and exports exactly what include/denet_hip.h declares (no compute calls)"""
    hdr = open(os.path.join(ROOT, "include", "denet_hip.h")).read()
    declared = set(re.findall(r"\b(denet_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(dlib.SIGNATURES.keys()), declared ^ set(dlib.SIGNATURES.keys())
    lib = dlib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.denet_abi_version() == 1
    # argument validation happens before any device work: callable on a CPU-only box
    assert lib.denet_conv_fwd(None, None, None, None, None, 1, 8, 8, 32, 33, 3, 3, 3, 1, 1, 8, 8, None) == -1000
    assert b"multiple of 32" in lib.denet_last_error()


def test_committed_launch_configurations_are_consistent():
    """denet_amd/tuned/gfx950.json (tools/tune.py): every record has the 11-field key + 3 values denet_tune_import expects, and a
    direct / Winograd / fused decision only names an algorithm whose kernel covers that geometry (an ineligible entry would make
    the pass fail at launch on every process that loads the file)"""
    import json
    path = os.path.join(ROOT, "denet_amd", "tuned", "gfx950.json")
    d = json.load(open(path))
    assert d["kernels"] and all(len(r) == 14 and all(isinstance(v, int) for v in r) for r in d["kernels"])
    assert len({tuple(r[:11]) for r in d["kernels"]}) == len(d["kernels"])
    seen = set()
    n_fused = 0
    for mode, g, tile in d["winograd"]:
        assert mode in (0, 1, 2, 3) and len(g) == 12 and tile in (0, 2, 4, 22, 44)       # mode 3: the inference forward pass
        assert (mode, tuple(g)) not in seen
        seen.add((mode, tuple(g)))
        N, H, W, C, K, R, S, s_real, stride, pad, OH, OW = g
        if tile:
            assert (R, S, s_real, stride, pad) == (3, 3, 3, 1, 1) and (OH, OW) == (H, W)
        if tile in (2, 4):                              # ops.conv_wino_ok: a map that is no multiple of the tile runs on ceil(H / tile) tiles
            assert H >= tile and W >= tile and C % 32 == 0 and K % 32 == 0
            from denet_amd import ops
            assert ops.conv_wino_ok(tuple(g), tile)
        if tile == 22:                                  # csrc/wino2f.hip: denet_conv_wino2f_ok / _wgrad_ok
            n_fused += 1
            assert mode != 3
            ci, co = (C, K) if mode == 0 else (K, C)
            assert H % 2 == 0 and W % 2 == 0 and ci == 64 and co % 64 == 0
            if mode == 2:
                assert C == 64 and K == 64
        if tile == 44:                                  # csrc/wino4t.hip: denet_conv_wino4t_ok (forward / data gradient only)
            n_fused += 1
            red, out = (K, C) if mode == 1 else (C, K)
            assert mode in (0, 1, 3) and H % 4 == 0 and W % 4 == 0 and red % 16 == 0 and out % 64 == 0
    assert n_fused >= 3                                 # the 64-channel stage of the benchmark configuration, all three passes


def test_product_fails_loudly_without_library(monkeypatch):
    monkeypatch.setattr(dlib, "_lib", None)
    monkeypatch.setattr(dlib, "LIB_PATH", "/nonexistent/libdenet_hip.so")
    with pytest.raises(dlib.DenetHipError):
        dlib.load()


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "denet_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cc")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("oracle's", ""), os.path.join(dirpath, f)


DP_WORKER = r'''
import os, sys, torch
sys.path.insert(0, %r)
from denet_amd.multi import DataParallel

class FakeModel: pass

rank = int(os.environ["RANK"])
dp = DataParallel(backend="gloo", bucket_bytes=4 * 100, scale_fn=lambda t, s: t.mul_(s))
m = FakeModel()
sizes = [64, 192, 128, 64, 320]
layers = [object() for _ in sizes]
off, rng = 0, []
for l, n in zip(layers, sizes):
    rng.append((l, off, off + n)); off += n
m.layer_weight_range = rng
m.n_weights, m.n_trainable = off, off + 64
torch.manual_seed(rank)
m.G = torch.randn(m.n_trainable)
m.S = torch.full((32,), float(rank))
m.P = torch.full((8,), float(rank)); m.M = m.P.clone()
g_local = m.G.clone()
dp.broadcast_state(m)
assert float(m.P[0]) == 0.0 and float(m.S[0]) == 0.0
m.S = torch.full((32,), float(rank))      # per-rank BN running statistics after a local step
dp.begin_step(m)
buckets = dp._buckets
assert buckets[0][1] == m.n_weights and buckets[-1][0] == 0
assert sum(hi - lo for lo, hi, _ in buckets) == m.n_weights
for l in reversed(layers):
    dp.layer_done(m, l)
dp.finish_step(m)
# one collective per bucket: the bias gradients and the BN statistics ride in the last one, nothing is issued in finish_step
assert dp.collectives_per_step() == len(buckets) and len(buckets) >= 3, (dp.collectives_per_step(), len(buckets))
assert dp.bytes_per_step() == 4 * (m.n_trainable + 32)
others = []
for r in range(dp.world_size):
    torch.manual_seed(r); others.append(torch.randn(m.n_trainable))
assert torch.allclose(m.G, sum(others), atol=1e-6)
assert torch.allclose(m.S, torch.full((32,), 0.5))
assert dp.max_over_ranks(float(rank)) == 1.0
# --batch-size-factor > 1: local steps, then parameters / momentum / BN statistics averaged (train_multi.py:96-145)
m.P = torch.full((8,), 1.0 + rank); m.M = torch.full((8,), 10.0 * rank); m.S = torch.full((32,), 3.0 - rank)
dp.average_state(m)
assert torch.allclose(m.P, torch.full((8,), 1.5)) and torch.allclose(m.M, torch.full((8,), 5.0))
assert torch.allclose(m.S, torch.full((32,), 2.5))
# the same averaging against the REFERENCE's own (denet.multi.shared.ModelUpdate.set_mean_*, fixture generated by importing
# it: tests/golden/make_common_fixtures.py): two workers, bit-identical mean
import json
fx = json.load(open(os.path.join(%r, "tests", "golden", "common_fixtures.json")))["multi_mean"][0]
flat = lambda parts: torch.tensor([v for p in parts for v in p], dtype=torch.float32)
m.P = flat(fx["workers"][rank]); m.M = m.P.clone(); m.S = m.P.clone()
dp.average_state(m)
assert torch.equal(m.P, flat(fx["mean"])), (m.P - flat(fx["mean"])).abs().max()
print("rank", rank, "ok")
'''


def test_data_parallel_gloo_world2(tmp_path):
    script = tmp_path / "dp_worker.py"
    script.write_text(DP_WORKER % (ROOT, ROOT))
    port = 29500 + (os.getpid() % 1000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=180)
        assert p.returncode == 0, out.decode()


def test_denet101_wide_builds_with_roi_clustering():
    """the v2 models the reference advertises ("corner clustering", README.md:132; 48 x 48 RoIs, papers/dss/denet101.sh:19,
    README.md:145) are DNS with nmsThreshold < 1 at sample_num 48: the layer asks the device proposal for the 10 * 48^2 best
    candidates (denet_sparse.cc:171-175) - more than one LDS sort holds, which used to raise"""
    desc = zoo.DENET101_WIDE_DESC
    assert "DNS[7,48,0.01,0.1]" in desc
    m = zoo.denet101(1, "wide", 128, class_num=80, seed=1, head_desc=desc.replace("DNS[7,48,0.01,0.1]", "DNS[7,48,0.01,0.1,0,0.7]"))
    dns = [l for l in m.layers if l.type_name == "denet-sparse"][0]
    assert dns.cluster and dns.sample_count == 2304 and dns.proposal_count == 23040
    assert dns.export_json()["nmsThreshold"] == 0.7
    L = dlib.load()
    assert L.denet_build_samples_workspace_bytes(1, 4, 32, 32, 1024, 23040) > L.denet_build_samples_workspace_bytes(1, 4, 32, 32, 1024, 7936)


def _distinct_corner_map(seed, B, H, W, per_type, Cn=4):
    """corner maps whose candidate scores are all distinct (no saturated log-probabilities): `per_type` firing cells
    per corner type with P in (0.02, 0.6), background P in (1e-4, 2e-4)"""
    rng = np.random.RandomState(seed)
    P = rng.uniform(1e-4, 2e-4, (B, Cn, H, W))
    for b in range(B):
        for c in range(Cn):
            idx = rng.choice(H * W, per_type, replace=False)
            P[b, c].reshape(-1)[idx] = rng.uniform(0.02, 0.6, per_type)
    return np.ascontiguousarray(np.stack([np.log(1 - P), np.log(P)], axis=1).astype(np.float32))


@pytest.mark.parametrize("thr", [0.3, 0.5, 0.7])
def test_native_roi_clustering_vs_oracle(thr):
    """apply_cluster (denet_sparse.cc:165-242): the native host routine of the product on the oracle's ranked candidate
    list against the oracle's own clustered proposal (sn = 6: 36 outputs from the 360 best of ~3000 candidates)"""
    import ctypes
    from oracle import model as OM
    L = dlib.load()
    sn, S = 6, 36
    pr = _distinct_corner_map(3, 3, 32, 32, 70)
    # the ranked candidate list = the unclustered proposal with a large sample_num (19^2 = 361 >= 10 * 36)
    full, _, _, cnt = OM.oracle_build_samples_raw(pr, 0.01, 19, 1024, 0, 1.0)
    ref, _, _, rcnt = OM.oracle_build_samples_raw(pr, 0.01, sn, 1024, 0, thr)
    plain, _, _, _ = OM.oracle_build_samples_raw(pr, 0.01, sn, 1024, 0, 1.0)
    changed = 0
    for b in range(pr.shape[0]):
        assert cnt[b] == 361
        cand = np.ascontiguousarray(full[b, :10 * S])
        assert len(np.unique(cand[:, 0])) == len(cand), "the test map must not produce score ties"
        out = np.zeros((S, 5), np.float32)
        n_out = ctypes.c_int(0)
        rc = L.denet_host_cluster_samples(cand.ctypes.data_as(ctypes.c_void_p), 10 * S, thr, S,
                                          out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n_out))
        assert rc == 0 and n_out.value == rcnt[b]
        assert np.array_equal(out[:n_out.value], ref[b, :rcnt[b]]), b
        changed += int(not np.array_equal(ref[b, :rcnt[b]], plain[b, :rcnt[b]]))
    assert changed > 0, "clustering never changed the selection: the test exercises nothing"
    assert L.denet_host_cluster_samples(cand.ctypes.data_as(ctypes.c_void_p), 10, thr, S, out.ctypes.data_as(ctypes.c_void_p),
                                        ctypes.byref(n_out)) == -1000


def test_model_modify_cli_replays_the_denet_recipe(tmp_path):
    """the two model-modify commands of papers/dss/denet34.sh:87-88 (skip variant) through the CLI, on a small
    ResNet-style classifier: same layer list, shapes and weights as the programmatic surgery"""
    np.random.seed(4)
    m = model_cnn.ModelCNN()
    m.batch_size = 2
    m.class_num = 10
    m.build("C.B[32,7,2] BN A P[3,2,1] nRSN.O[2,32,3] nRSN.O[2,64,3,2] nRSN.O[2,128,3,2] P.A[2] R.TB", (3, 64, 64), "relu",
            "half", ["he-backward"])
    src, mid, dst = str(tmp_path / "cls.mdl.gz"), str(tmp_path / "skipsrc.mdl.gz"), str(tmp_path / "initial.mdl.gz")
    model_cnn.save_to_file(m, src)
    head = "PI[2] C[64,3] SKIP[1] BNA PI[2] C[32,3] SKIP[0] BNA DNC[16,100] DNS[3,4,0.01,0.1] C.B[64,1] BNA DND[0.5,1,1]"
    assert modify.main(["--input", src, "--output", mid, "--modify-bn", "1", "0.9", "1e-5", "--convert-bn-relu",
                        "--use-cudnn-pool", "--class-num", "20", "--image-size", "128", "128", "--layer-remove", "3",
                        "--layer-insert", "6:SKIPSRC.X[0]", "7:SKIPSRC.X[1]"]) == 0
    assert modify.main(["--input", mid, "--output", dst, "--layer-append"] + head.split()) == 0
    got = model_cnn.load_from_file(dst, 2)
    ref = modify.modify_bn(m, 1, 0.9, 1e-5)
    ref = modify.convert_bn_relu(ref)
    ref = modify.layer_remove(ref, 3)
    ref = modify.set_class_num(ref, 20)
    ref = modify.set_image_size(ref, 128, 128)
    ref = modify.layer_insert(ref, ["6:SKIPSRC.X[0]", "7:SKIPSRC.X[1]"])
    np.random.seed(23455)          # the CLI seeds numpy with --seed before it builds the new layers
    random.seed(23455)
    ref = modify.layer_append(ref, head)
    assert [l.type_name for l in got.layers] == [l.type_name for l in ref.layers]
    assert [l.output_shape for l in got.layers] == [l.output_shape for l in ref.layers]
    assert got.class_num == 20 and tuple(got.data_shape) == (3, 128, 128)
    assert got.layers[-1].type_name == "denet-detect" and got.layers[6].type_name == "skip-src" and got.layers[6].has_split
    for a, b in zip(got.layers[:6], ref.layers[:6]):
        for pa, pb in zip(a.params(), b.params()):
            np.testing.assert_array_equal(pa.value, pb.value)
    # --merge and --modify-layer
    out2 = str(tmp_path / "merged.mdl.gz")
    assert modify.main(["--input", dst, "--output", out2, "--merge", "--modify-layer", "denet-sparse", "corner_threshold=0.05"]) == 0
    mm = model_cnn.load_from_file(out2, 2)
    assert not any(getattr(l, "has_split", False) for l in mm.layers if l.type_name == "skip-src")
    assert [l for l in mm.layers if l.type_name == "denet-sparse"][0].corner_threshold == 0.05


def test_train_multi_restart_lookup(tmp_path):
    """--restart: newest checkpoint of the run and the epoch to continue with (load_restart_args, train_multi.py:242-268)"""
    from denet_amd.model.train_multi import find_restart
    prefix = str(tmp_path / "run")
    with pytest.raises(Exception):
        find_restart(prefix)
    for name in ("_epoch000_final", "_epoch001_subset003", "_epoch001_final", "_epoch002_subset001"):
        open(prefix + name + ".mdl.gz", "w").close()
    assert find_restart(prefix) == (prefix + "_epoch002_subset001.mdl.gz", 2, 2)
    os.remove(prefix + "_epoch002_subset001.mdl.gz")
    assert find_restart(prefix) == (prefix + "_epoch001_subset003.mdl.gz", 1, 4)      # sorted(): "subset" > "final", as in the reference
    os.remove(prefix + "_epoch001_subset003.mdl.gz")
    assert find_restart(prefix) == (prefix + "_epoch001_final.mdl.gz", 2, 0)


def test_train_multi_sharding():
    """model-train-multi: a global iteration is world x F batches of consecutive samples; worker r takes the F consecutive
    batches r*F .. r*F+F-1 (train_multi.py:113-119); the last partial iteration is padded with random.randint draws like
    export(batch_size = world x B x F) (train_multi.py:53-55, dataset/__init__.py:349-354), identically on every rank"""
    import random
    from denet_amd.model.train_multi import _Shard

    class Data:
        subset_size, subset_total_size = 23, 41
        images = list(range(41))
    world, B = 3, 4
    shards = [_Shard(Data, r, world, B) for r in range(world)]
    for subset, (lo, hi) in enumerate([(0, 23), (23, 41)]):
        parts = []
        for s in shards:
            random.seed(5)
            parts.append(s.images_of_subset(subset))
        n = hi - lo
        n_glob = -(-n // (world * B))
        assert all(len(p) == n_glob * B for p in parts)
        # the reference's export: first n samples in order, then randint pads
        random.seed(5)
        full = list(range(lo, hi)) + [lo + random.randint(0, n - 1) for _ in range(n_glob * world * B - n)]
        for k in range(n_glob):
            for r in range(world):
                assert parts[r][k * B:(k + 1) * B] == full[(k * world + r) * B:(k * world + r + 1) * B]
    # --batch-size-factor 2: an iteration = world x F batches, worker r takes batches r*F, r*F+1 of it
    world, B, F = 2, 3, 2
    Data.subset_size, Data.subset_total_size, Data.images = 40, 40, list(range(40))
    parts = []
    for r in range(world):
        random.seed(9)
        parts.append(_Shard(Data, r, world, B, F).images_of_subset(0))
    assert parts[0][:6] == [0, 1, 2, 3, 4, 5] and parts[1][:6] == [6, 7, 8, 9, 10, 11]
    assert parts[0][6:12] == [12, 13, 14, 15, 16, 17] and parts[1][6:12] == [18, 19, 20, 21, 22, 23]
    assert len(parts[0]) == len(parts[1]) == -(-40 // (world * F * B)) * F * B
    random.seed(9)
    pads = [random.randint(0, 39) for _ in range(48 - 40)]
    assert (parts[0] + parts[1])[:0] == [] and sorted(parts[0][:18] + parts[1][:18]) == list(range(36))
    assert parts[0][18:] == [36, 37, 38, 39] + pads[:2] and parts[1][18:] == pads[2:]


# ---- methods of the reference's layer classes, executed in the build container (tests/golden/make_layer_method_fixtures.py) ----
def _layer_method_fixtures():
    with open(os.path.join(ROOT, "tests", "golden", "layer_method_fixtures.json")) as f:
        return json.load(f)


def test_corner_target_matches_executed_reference_method():
    """DeNetCornerLayer.get_target of the build against the reference's own method (denet_corner.py:81-123), which the fixture
    script compiled from its AST node and ran: positive cells, the two distinct target values, and the oracle's restatement"""
    fix = _layer_method_fixtures()["corner_target"]
    m = zoo.denet34(1, "skip", 128)
    dnc = [l for l in m.layers if l.type_name == "denet-corner"][0]
    assert len(fix["cases"]) >= 5
    for c in fix["cases"]:
        B, H, W, cn = c["B"], c["H"], c["W"], 5 if c["use_center"] else 4
        dnc.corner_shape, dnc.width, dnc.height, dnc.use_center, dnc.corner_num = (B, 2, cn, H, W), W, H, c["use_center"], cn
        for got in (dnc.get_target(m, None, c["metas"])[1].reshape(B, 2, cn, H, W), OL.corner_target(c["metas"], (B, 2, cn, H, W))):
            assert got.dtype == np.float32
            assert np.argwhere(got[:, 1] > 0).tolist() == c["positive_cells"]
            assert np.array_equal(got[:, 0] > 0, ~(got[:, 1] > 0))
            assert sorted(set(float(v) for v in np.unique(got))) == c["distinct_values"]
            assert float(got.astype(np.float64).sum()) == c["sum"]


def _fixture_model(sn, random_sample, sample_gt, B, class_num=80, dnd="DND[0.5,1,1]"):
    head = zoo.DENET34_SKIP_DESC.replace("DNS[7,24,0.01,0.1]", "DNS%s[7,%d,0.01,%r]" % ("" if sample_gt else ".G", sn, random_sample))
    return zoo.denet34(B, "skip", 128, class_num=class_num, head_desc=head.replace("DND[0.5,1,1]", dnd))


def test_roi_editing_matches_executed_reference_method():
    """the training-time RoI list editing (random.sample trim, random boxes, ground truth from the tail) of the build - the
    Python path, the native host call and the oracle - against DeNetSparseLayer.get_target of the reference itself
    (denet_sparse.py:164-207, executed by the fixture script): edited lists value for value, and the stdlib generator must
    stand at the same position afterwards"""
    fix = _layer_method_fixtures()["sparse_get_target"]
    assert len(fix["cases"]) >= 6
    for c in fix["cases"]:
        sn, B = c["sample_num"], len(c["metas"])
        S = sn * sn
        m = _fixture_model(sn, c["random_sample"], c["sample_gt"], B)
        dns = [l for l in m.layers if l.type_name == "denet-sparse"][0]
        assert dns.sample_count == S and dns.sample_gt == c["sample_gt"] and dns.random_sample == c["random_sample"]
        want = [[(p, tuple(bx)) for p, bx in l] for l in c["edited"]]
        prs = [np.array([s[0] for s in l], dtype=np.float64) for l in c["samples"]]
        boxes = [np.array([s[1] for s in l], dtype=np.float64).reshape(-1, 4) for l in c["samples"]]
        # Python path
        random.seed(c["seed"])
        out_pr, out_bx = dns.edit_samples(prs, boxes, c["metas"])
        assert [random.random() for _ in range(3)] == c["random_after"]
        got = [[(float(p), tuple(b)) for p, b in zip(pr.tolist(), bx.tolist())] for pr, bx in zip(out_pr, out_bx)]
        assert got == want
        # native host call (denet_host_edit_samples): the detector rows are float32 there, like the reference's C++ hand-over
        det = np.zeros((B, S, 5), dtype=np.float32)
        cnt = np.zeros(B, dtype=np.int32)
        for b, l in enumerate(c["samples"]):
            cnt[b] = len(l)
            for i, (p, bx) in enumerate(l):
                det[b, i, 0], det[b, i, 1:] = p, bx
        if dns._native_edit_ok(c["metas"]):
            random.seed(c["seed"])
            f32 = np.zeros((B * S, 4), dtype=np.float32)
            n_pr, n_bx = dns.edit_samples_native(det, cnt, c["metas"], f32)
            assert [random.random() for _ in range(3)] == c["random_after"]
            for b in range(B):
                w_pr = np.array([s[0] for s in want[b]])
                w_bx = np.array([s[1] for s in want[b]])
                from_det = w_pr != 1.0                    # detector / random rows passed through float32; ground truth is exact
                assert np.array_equal(n_pr[b], w_pr)
                assert np.array_equal(n_bx[b][~from_det], w_bx[~from_det])
                assert np.array_equal(n_bx[b][from_det].astype(np.float32), w_bx[from_det].astype(np.float32))
            np.testing.assert_array_equal(f32.reshape(B, S, 4), np.array([[s[1] for s in l] for l in want], dtype=np.float32))
        # the oracle's restatement
        random.seed(c["seed"])
        ol = OL.edit_samples([[(p, tuple(bx)) for p, bx in l] for l in c["samples"]], c["metas"], S, c["random_sample"], c["sample_gt"])
        assert [random.random() for _ in range(3)] == c["random_after"]
        assert [[(float(p), tuple(float(v) for v in bx)) for p, bx in l] for l in ol] == want


def test_prefetched_generator_outputs_give_the_same_editing():
    """DeNetSparseLayer._prefetch_random: the editing that walks through generator outputs drawn ahead of the hand-off
    (denet_host_mt_prefetch / denet_host_edit_samples_stream) against the editing on the live generator: same lists, same
    upload array and the stdlib generator in the identical state afterwards (trim / no trim / empty / full lists, several
    steps in a row so that refills of the 624-word state fall at every kind of position)"""
    from denet_amd.layer import roi_handoff as DS
    saved = DS.PREFETCH_RANDOM
    try:
        for rs in (0.1, 0.5):
            m = _fixture_model(24, rs, True, 4)
            dns = [l for l in m.layers if l.type_name == "denet-sparse"][0]
            _, metas = zoo.synthetic_batch(4, 128, 80, seed=1)
            B, S = 4, 576
            det = np.random.RandomState(3).rand(B, S, 5).astype(np.float32)
            for counts in ([0, 0, 0, 0], [576, 10, 519, 520], [576, 576, 576, 576], [3, 0, 519, 1]):
                cnt = np.array(counts, dtype=np.int32)
                res = []
                for pre in (True, False):
                    DS.PREFETCH_RANDOM = pre
                    random.seed(5)
                    random.random()
                    for _ in range(3):
                        dns.begin_step(metas)
                        assert (dns._prefetch is not None) == pre
                        f32 = np.zeros((B * S, 4), dtype=np.float32)
                        pr, bx = dns.edit_samples_native(det, cnt, metas, f32)
                    res.append((pr.copy(), bx.copy(), f32.copy(), random.getstate()))
                a, b = res
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), (rs, counts)
                assert a[3] == b[3], "generator state differs"
    finally:
        DS.PREFETCH_RANDOM = saved


def test_uniform_table_of_a_prefetched_stretch_is_random_random():
    """denet_host_mt_uniforms: entry p of the table is the double `random.random()` returns when the stdlib generator stands at output
    p of the prefetched stretch (checked against the stdlib itself at even AND odd positions: getrandbits(32) moves it by one), and the
    hand-off that reads the table (denet_host_handoff_boxes_stream_u) writes the same bytes and stops at the same cursor as the one that
    forms the doubles in place - trimmed / short / empty lists, and a stretch that runs dry inside the random boxes"""
    import ctypes
    L = dlib.load()
    random.seed(11)
    random.random()
    st = random.getstate()
    key = np.array(st[1][:-1], dtype=np.uint32)
    pos = np.array([st[1][-1]], dtype=np.int32)
    n = 5000
    out, snaps, first, ns = np.empty(n, np.uint32), np.empty((n // 624 + 3, 624), np.uint32), np.empty(n // 624 + 3, np.int64), ctypes.c_int(0)
    assert L.denet_host_mt_prefetch(key.ctypes.data, pos.ctypes.data, n, out.ctypes.data, snaps.ctypes.data, first.ctypes.data,
                                    snaps.shape[0], ctypes.byref(ns)) == 0
    uni = np.empty(n, np.float64)
    assert L.denet_host_mt_uniforms(out.ctypes.data, n, uni.ctypes.data) == 0
    p = 0
    for i in range(700):
        assert random.random() == uni[p], (i, p)
        p += 2
        if i % 7 == 3:                       # one output: the next double starts at an odd position
            assert random.getrandbits(32) == int(out[p])
            p += 1
    B, S, H, W = 5, 96, 64, 48
    n_keep = S - S // 4
    rng = np.random.RandomState(2)
    x0, y0 = rng.randint(0, W - 1, (B, S)), rng.randint(0, H - 1, (B, S))
    box = np.ascontiguousarray(np.stack([x0, y0, np.minimum(W - 1, x0 + rng.randint(0, 9, (B, S))),
                                         np.minimum(H - 1, y0 + rng.randint(0, 9, (B, S)))], -1).astype(np.int32))
    gt, off = rng.rand(B * 2, 4), (np.arange(B + 1) * 2).astype(np.int32)
    ws = np.empty(2 * S, np.int32)
    for counts, n_stream in (([96, 80, 0, 73, 96], n), ([96, 96, 96, 96, 96], n), ([0, 0, 0, 0, 0], n), ([10, 96, 3, 0, 50], 900)):
        cnt = np.array(counts, np.int32)
        res = []
        for table in (None, uni):
            f32 = np.full((B, S, 4), -1.0, np.float32)
            cur, dry = ctypes.c_long(3), ctypes.c_int(0)
            assert L.denet_host_handoff_boxes_stream_u(out.ctypes.data, n_stream, ctypes.byref(cur), ctypes.byref(dry), box.ctypes.data,
                                                       cnt.ctypes.data, H, W, B, S, n_keep, gt.ctypes.data, off.ctypes.data, 1, ws.ctypes.data,
                                                       f32.ctypes.data, table.ctypes.data if table is not None else None) == 0
            res.append((f32.tobytes(), cur.value, dry.value))
        assert res[0] == res[1], counts
        assert res[0][2] == (1 if n_stream == 900 else 0)
    # the kept boxes are cell coordinates turned into fractions of the map: (x0 / W, y0 / H, (x1 + 1) / W, (y1 + 1) / H) as float32 of
    # the double quotient, up to the largest map the tables cover (256 cells: coordinate 256 itself is a legal x0 / y0), and a
    # coordinate beyond the map is an argument error, not a read past the tables
    Hb = Wb = 256
    bx = np.array([[[0, 0, 0, 0], [256, 256, 255, 255], [17, 200, 255, 201], [3, 4, 5, 6]]], np.int32)
    cnt1, f1 = np.array([4], np.int32), np.zeros((1, 4, 4), np.float32)
    cur, dry = ctypes.c_long(0), ctypes.c_int(0)
    ws1 = np.empty(8, np.int32)
    args = lambda b: (out.ctypes.data, n, ctypes.byref(cur), ctypes.byref(dry), b.ctypes.data, cnt1.ctypes.data, Hb, Wb, 1, 4, 4, None,
                      np.zeros(2, np.int32).ctypes.data, 0, ws1.ctypes.data, f1.ctypes.data, uni.ctypes.data)
    assert L.denet_host_handoff_boxes_stream_u(*args(bx)) == 0
    want = np.stack([bx[0, :, 0] / Wb, bx[0, :, 1] / Hb, (bx[0, :, 2] + 1) / Wb, (bx[0, :, 3] + 1) / Hb], -1).astype(np.float32)
    assert np.array_equal(f1[0], want)
    bad = bx.copy()
    bad[0, 2, 2] = 257
    assert L.denet_host_handoff_boxes_stream_u(*args(bad)) == -1000
    bad[0, 2, 2] = -1
    assert L.denet_host_handoff_boxes_stream_u(*args(bad)) == -1000


def test_detect_targets_match_executed_reference_method():
    """RoI -> class / fitness / box-regression targets: DeNetDetectLayer.get_target of the build (native host path and numpy
    path) and the oracle against the reference's own method loop (denet_detect.py:147-236, executed by the fixture script with
    the IoU matrix of theano_util.py:38-59 evaluated in numpy float32), byte for byte"""
    fix = _layer_method_fixtures()["detect_target"]
    assert len(fix["cases"]) >= 5
    for c in fix["cases"]:
        sn, ncls, B = c["sample_num"], c["class_num"], len(c["metas"])
        dnd_desc = "DND%s[0.5,1,%d%s]" % (".J" if c["use_jointfit"] else "", 1 if c["use_bbox_reg"] else 0, ",0.5" if c["use_indfit"] else "")
        m = _fixture_model(sn, 0.1, True, B, class_num=ncls, dnd=dnd_desc)
        dns = [l for l in m.layers if l.type_name == "denet-sparse"][0]
        dnd = [l for l in m.layers if l.type_name == "denet-detect"][0]
        dnd.overlap_threshold = tuple(c["overlap_threshold"])
        assert (dnd.use_jointfit, dnd.use_bbox_reg, dnd.use_indfit) == (c["use_jointfit"], c["use_bbox_reg"], c["use_indfit"])
        lists = [[(p, tuple(bx)) for p, bx in l] for l in c["sample_bbox_list"]]
        dns.sample_pr, dns.sample_boxes = dns._from_lists(lists)
        want = np.frombuffer(bytes.fromhex(c["yt_value_f32_hex"]), dtype="<f4")
        assert want.size == c["yt_len"]
        idx, val = dnd.get_target(m, None, c["metas"])
        assert idx.size == 0 and val.dtype == np.float32
        np.testing.assert_array_equal(val, want)
        tg = OL.detect_target(c["metas"], lists, B, sn, ncls, tuple(c["overlap_threshold"]), c["use_bbox_reg"], c["use_jointfit"],
                              c["use_indfit"])
        parts = [tg[0].flatten()] + ([tg[1].flatten(), tg[2].flatten()] if c["use_bbox_reg"] else []) + \
                ([tg[3].flatten()] if c["use_indfit"] else [])
        np.testing.assert_array_equal(np.concatenate(parts), want)


def test_desc_grammar_matches_the_executed_reference_parser():
    """The `TYPE.TAGS[args]` operator surface against the reference itself: tests/golden/make_desc_fixtures.py executed the reference's
    `ModelCNN.build_layer` (model_cnn.py:122-146) and the `parse_desc` of its 19 registry classes (layer_types.py:17-25) with the
    layer constructors recorded; the same tokens go through the build's parser with ITS constructors recorded the same way. Per
    token: the same class(es), in the same number, with the same constructor arguments (defaults, tag semantics, argument
    types), the same registry order, and an exception for the tokens the reference rejects"""
    import inspect
    import types
    from denet_amd import layer as layer_pkg
    from denet_amd.layer.layer_types import layer_types
    with open(os.path.join(GOLDEN, "desc_fixtures.json")) as f:
        fix = json.load(f)
    assert [c.__name__ for c in layer_types] == fix["registry_order"]
    stub_shape = tuple(fix["stub_shape"])
    log = []
    classes = {c.__name__: c for c in layer_types}
    modules = [m for name, m in sys.modules.items() if name.startswith("denet_amd.layer") and m is not None]
    patched = []

    def recorder(cls):
        names = list(inspect.signature(cls.__init__).parameters)[1:]

        def construct(*args, **kwargs):
            named = dict(zip(names, args))
            named.update(kwargs)
            named.pop("layers", None)
            log.append({"class": cls.__name__, "args": {k: (list(v) if isinstance(v, tuple) else v) for k, v in named.items()}})
            return types.SimpleNamespace(output_shape=stub_shape, type_name=cls.__name__)
        return construct
    try:
        for m in modules:
            for name, cls in classes.items():
                if m.__dict__.get(name) is cls:
                    patched.append((m, name, cls))
                    setattr(m, name, recorder(cls))
        model = model_cnn.ModelCNN()
        model.class_num = 80
        assert len(fix["cases"]) >= 50
        for c in fix["cases"]:
            layers = [types.SimpleNamespace(output_shape=stub_shape, type_name="initial")]
            del log[:]
            if "error" in c:
                with pytest.raises(Exception):
                    model.build_layer(c["token"], layers, "relu", "half", "he-backward")
                continue
            model.build_layer(c["token"], layers, "relu", "half", "he-backward")
            assert len(layers) - 1 == c["layers_appended"], c["token"]
            assert [x["class"] for x in log] == [x["class"] for x in c["calls"]], c["token"]
            for got, want in zip(log, c["calls"]):
                for k, v in want["args"].items():
                    assert k in got["args"], (c["token"], "constructor argument the reference passes is unknown here", k)
                    assert got["args"][k] == v and type(got["args"][k]) is type(v), (c["token"], k, got["args"][k], v)
                assert set(got["args"]) == set(want["args"]), (c["token"], got["args"], want["args"])
    finally:
        for m, name, cls in patched:
            setattr(m, name, cls)


def test_every_environment_switch_is_registered():
    """denet_amd/switches.py lists every DENET_* environment switch the sources read (Python: os.environ.get, C++: getenv), with its
    default - the one place that says what the product default is; bench.py reports the switches a run was taken with against it"""
    import re
    from denet_amd import switches
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    found = set()
    for base in ("denet_amd", "tests", "bench.py", "__graft_entry__.py"):
        path = os.path.join(root, base)
        files = [path] if os.path.isfile(path) else [os.path.join(d, f) for d, _, fs in os.walk(path) for f in fs
                                                      if f.endswith((".py", ".hip", ".h")) and "native" not in d]
        for f in files:
            with open(f, errors="ignore") as fh:
                found |= set(re.findall(r'(?:getenv\(|environ\.get\(|environ\[)"(DENET_[A-Z0-9_]+)"', fh.read()))
    missing = sorted(found - set(switches.SWITCHES))
    assert not missing, "switches read by the sources but not listed in denet_amd/switches.py: %s" % missing
    stale = sorted(k for k in switches.SWITCHES if k not in found)
    assert not stale, "listed switches nothing reads any more: %s" % stale
    assert switches.changes_kernels({"DENET_WINO4F": "0"}) and not switches.changes_kernels({"DENET_BUILD_JOBS": "2", "DENET_FORCE_DP": "1"})


def test_tune_merge_replaces_only_the_remeasured_keys():
    """tools/tune.py --only X --merge-into FILE: denet34-skip and cifar3 both run batch 32 - re-measuring one must
    not delete the other's records; exactly the re-measured geometry keys are replaced"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("tune_tool", os.path.join(ROOT, "tools", "tune.py"))
    saved = {k: os.environ.get(k) for k in ("DENET_TUNE_CACHE", "DENET_TUNE")}
    try:
        tune = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(tune)
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    key = lambda mode, n, h, c: [mode, n, h, h, c, c, 3, 3, 3, 1, 1]
    geom = lambda n, h, c: [n, h, h, c, c, 3, 3, 3, 1, 1, h, h]
    old_k = [key(0, 32, 64, 128) + [1, 1, 1],        # denet34-skip, batch 32
             key(2, 32, 64, 128) + [2, 2, 2],
             key(0, 32, 16, 256) + [3, 3, 3],        # cifar3, batch 32 as well
             key(0, 64, 56, 64) + [4, 4, 4],         # resnet34
             key(5, 32, 64, 128) + [9, 9, 9]]        # a batched-product record (mode > 2)
    old_w = [[0, geom(32, 64, 128), 4], [0, geom(32, 16, 256), 2], [1, geom(64, 56, 64), 22]]
    rec = [key(0, 32, 16, 256) + [7, 7, 7], key(5, 32, 64, 128) + [8, 8, 8], key(6, 32, 16, 256) + [5, 5, 5]]
    wino = {(0, tuple(geom(32, 16, 256))): 4}
    kept, add, wkept, wadd = tune.merge_records(old_k, old_w, rec, wino)
    merged = sorted(kept + add)
    assert key(0, 32, 64, 128) + [1, 1, 1] in merged and key(2, 32, 64, 128) + [2, 2, 2] in merged      # the other batch-32 config stays
    assert key(0, 32, 16, 256) + [7, 7, 7] in merged and key(0, 32, 16, 256) + [3, 3, 3] not in merged  # the re-measured key replaced
    assert key(0, 64, 56, 64) + [4, 4, 4] in merged
    assert key(5, 32, 64, 128) + [9, 9, 9] in merged and key(5, 32, 64, 128) + [8, 8, 8] not in merged  # product records: only added
    assert key(6, 32, 16, 256) + [5, 5, 5] in merged
    assert len(merged) == 6
    w = sorted(wkept + wadd)
    assert [0, geom(32, 64, 128), 4] in w and [1, geom(64, 56, 64), 22] in w and [0, geom(32, 16, 256), 4] in w and len(w) == 3
