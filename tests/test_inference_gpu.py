"""Inference tail: class/fitness/box decode + per-class NMS of the detection head against the oracle
(oracle/layers.py:detect_outputs restates denet_detect.py:76-100,330-349; oracle/build_samples.cc restates
build_detections_nms of denet_detect.cc:99-173). Membership, class, order and boxes are exact on the same decoded
arrays; the decode itself (exp/log) is floating point: 1e-5 relative."""
import ctypes

import numpy as np
import pytest
import torch

from denet_amd import ops
from denet_amd.model import zoo
from oracle import layers as OL
from oracle import model as OM

pytestmark = pytest.mark.gpu


def _oracle_nms(det_pr, fitness, bbox, counts, B, sn, C1, pr_thr, nms_thr, soft):
    """arrays in the product's [B*S, C1] / [B*S, 4] layout -> list[B] of (pr, cls, box) rows from the oracle"""
    S = sn * sn
    det = np.ascontiguousarray(det_pr.reshape(B, sn, sn, C1).transpose(0, 3, 1, 2), dtype=np.float32)
    fit = np.ascontiguousarray(fitness.reshape(B, sn, sn, C1).transpose(0, 3, 1, 2), dtype=np.float32)
    bx = np.ascontiguousarray(bbox.reshape(B, sn, sn, 4), dtype=np.float32)
    num = np.ascontiguousarray(counts, dtype=np.int32)
    max_out = S * (C1 - 1)
    out = np.zeros((B, max_out, 6), np.float32)
    cnt = np.zeros(B, np.int32)
    f = OM.oracle_lib().oracle_build_detections_nms
    f.argtypes = [ctypes.c_float, ctypes.c_float, ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_void_p] * 2
    f(pr_thr, nms_thr, int(soft), det.ctypes.data, fit.ctypes.data, bx.ctypes.data, num.ctypes.data, B, C1, sn, max_out,
      out.ctypes.data, cnt.ctypes.data)
    return [out[b, :cnt[b]] for b in range(B)]


def _clustered_boxes(rng, B, S):
    centres = rng.rand(B, 6, 2)
    pick = rng.randint(0, 6, (B, S))
    c = np.take_along_axis(centres, pick[..., None].repeat(2, -1), 1) + rng.normal(0, 0.02, (B, S, 2))
    wh = 0.2 + 0.05 * rng.rand(B, S, 2)
    return np.concatenate([c - wh / 2, c + wh / 2], -1).astype(np.float32)


def _product_lists(keep, fit_h, box_h, B, S):
    res = []
    for b in range(B):
        cls_idx, roi_idx = np.nonzero(keep[b])
        rows = b * S + roi_idx
        res.append(np.concatenate([np.exp(fit_h[rows, cls_idx])[:, None], cls_idx[:, None].astype(np.float32),
                                   box_h[rows]], 1).astype(np.float32).reshape(-1, 6))
    return res


@pytest.mark.parametrize("jointfit", [0, 1])
@pytest.mark.parametrize("sn", [4, 24])
def test_detect_decode_and_nms_vs_oracle(hip, jointfit, sn):
    rng = np.random.RandomState(10 + sn + jointfit)
    B, C, S = 3, 7, sn * sn
    s0 = C * 5 + 1 if jointfit else C + 1
    CP = (s0 + 4 + 31) // 32 * 32
    t0 = 0.5
    logits = np.zeros((B * S, CP), np.float32)
    logits[:, :s0] = rng.normal(0, 2.5, (B * S, s0))
    logits[:, s0:s0 + 4] = rng.normal(0, 0.2, (B * S, 4))
    roi = _clustered_boxes(rng, B, S).reshape(B * S, 4)
    counts = np.array([S, S // 3, 0], np.int32)
    det_pr, fitness, bbox = ops.detect_decode(torch.from_numpy(logits).cuda(), torch.from_numpy(roi).cuda(), C, jointfit, 4, t0)
    det_h, fit_h, box_h = det_pr.cpu().numpy(), fitness.cpu().numpy(), bbox.cpu().numpy()
    # decode against the numpy restatement (NCHW like the reference)
    out_nchw = logits[:, :s0 + 4].reshape(B, sn, sn, s0 + 4).transpose(0, 3, 1, 2)
    o_det, o_fit, o_box = OL.detect_outputs(out_nchw, roi.reshape(B, sn, sn, 4), C, bool(jointfit), t0)
    np.testing.assert_allclose(det_h.reshape(B, sn, sn, C + 1).transpose(0, 3, 1, 2), o_det, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(fit_h.reshape(B, sn, sn, C + 1).transpose(0, 3, 1, 2), o_fit, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(box_h.reshape(B, sn, sn, 4), o_box, rtol=1e-5, atol=1e-6)
    # NMS: exact against the oracle on the SAME decoded arrays
    for pr_thr, nms_thr in ((0.1, 0.5), (0.3, 0.3), (0.1, 1.0), (0.2, 0.0)):
        keep = ops.detect_nms(det_pr, fitness, bbox, torch.from_numpy(counts).cuda(), B, S, C, pr_thr, nms_thr).cpu().numpy()
        got = _product_lists(keep, fit_h, box_h, B, S)
        ref = _oracle_nms(det_h, fit_h, box_h, counts, B, sn, C + 1, pr_thr, nms_thr, False)
        for b in range(B):
            assert got[b].shape == ref[b].shape, (b, pr_thr, nms_thr, got[b].shape, ref[b].shape)
            assert np.array_equal(got[b][:, 1:], ref[b][:, 1:])
            np.testing.assert_allclose(got[b][:, 0], ref[b][:, 0], rtol=2e-6)
        if 0 < nms_thr < 1:
            assert 0 < sum(len(g) for g in got) < int((det_h[:, :C] >= np.log(np.float32(pr_thr))).sum()), "NMS did nothing"
    assert len(got[2]) == 0


@pytest.mark.parametrize("nreg", [4, 0])
def test_detect_decode_independent_fitness_vs_oracle(hip, nreg):
    """independent fitness head at inference (denet_detect.py:396-401): fitness += log E[val]"""
    rng = np.random.RandomState(21 + nreg)
    B, C, sn = 2, 7, 6
    S, s0, nfit, t0 = sn * sn, C + 1, 6, 0.5
    CP = 32
    logits = np.zeros((B * S, CP), np.float32)
    logits[:, :s0] = rng.normal(0, 2.0, (B * S, s0))
    logits[:, s0:s0 + nreg] = rng.normal(0, 0.2, (B * S, nreg))
    logits[:, s0 + nreg:s0 + nreg + nfit] = rng.normal(0, 1.5, (B * S, nfit))
    roi = _clustered_boxes(rng, B, S).reshape(B * S, 4)
    det_pr, fitness, bbox = ops.detect_decode(torch.from_numpy(logits).cuda(), torch.from_numpy(roi).cuda(), C, 0, nreg, t0,
                                              nfit=nfit)
    out_nchw = logits[:, :s0 + nreg + nfit].reshape(B, sn, sn, -1).transpose(0, 3, 1, 2)
    o_det, o_fit, o_box = OL.detect_outputs(out_nchw, roi.reshape(B, sn, sn, 4), C, False, t0, use_bbox_reg=nreg == 4,
                                            nfit=nfit)
    np.testing.assert_allclose(det_pr.cpu().numpy().reshape(B, sn, sn, C + 1).transpose(0, 3, 1, 2), o_det, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(fitness.cpu().numpy().reshape(B, sn, sn, C + 1).transpose(0, 3, 1, 2), o_fit, rtol=1e-5, atol=4e-6)
    assert float(np.abs(o_fit - o_det).max()) > 0.1      # the fitness term is really there


def test_soft_nms_host_vs_oracle(hip):
    rng = np.random.RandomState(3)
    B, sn, C = 1, 12, 1
    S = sn * sn
    box = _clustered_boxes(rng, B, S).reshape(S, 4)
    fit = np.log(rng.uniform(0.02, 1.0, (S, 2))).astype(np.float32)
    counts = np.array([S - 5], np.int32)
    for nms_thr in (0.3, 0.6):
        ref = _oracle_nms(fit, fit, box, counts, B, sn, C + 1, 0.05, nms_thr, True)[0]
        cand = np.arange(counts[0])[fit[:counts[0], 0] >= np.log(np.float32(0.05))]
        order, score = ops.soft_nms_host(fit[cand, 0], box[cand], nms_thr)
        assert len(order) == len(ref) and 0 < len(ref) < len(cand)
        assert np.array_equal(box[cand[order]], ref[:, 2:])
        np.testing.assert_allclose(np.exp(score), ref[:, 0], rtol=2e-6)


@pytest.mark.parametrize("B,sn,C,nms_thr", [(3, 12, 5, 0.4), (2, 24, 80, 0.5), (1, 48, 3, 0.6), (2, 12, 4, 1.5), (2, 12, 4, 0.3)])
def test_soft_nms_device_vs_host_and_oracle(hip, B, sn, C, nms_thr):
    """denet_soft_nms_batch (one wave per (class, image)) against the host call and the C++ oracle: the same detections in the
    same order with the same scores, bit for bit - incl. exact score ties (duplicated candidates), images without candidates,
    a threshold outside (0, 1) (no suppression: list order) and 48x48 RoIs (36 candidates per lane)"""
    rng = np.random.RandomState(11 + sn + C)
    S = sn * sn
    box = _clustered_boxes(rng, B, S).reshape(B * S, 4)
    fit = np.log(rng.uniform(0.005, 1.0, (B * S, C + 1))).astype(np.float32)
    det = (fit + rng.normal(0, 0.3, fit.shape)).astype(np.float32)
    # exact ties: copies of a candidate's score (and of a whole candidate) later in the list
    fit[7::13, 0] = fit[5, 0]
    box[9] = box[5]
    fit[9, :] = fit[5, :]
    det[9, :] = det[5, :]
    counts = np.array([S - 3 * b for b in range(B)], np.int32)
    if B > 1:
        det[S:S + counts[1], C - 1] = -50.0          # an (image, class) pair without candidates
    if nms_thr == 0.3:
        counts[0] = 0                                  # an image without RoIs
    pr_thr = 0.05
    t = lambda a: torch.from_numpy(a).cuda()
    sc_d, cls_d, row_d, per_d = ops.soft_nms_batch(t(det), t(fit), t(box), t(counts), B, S, C, pr_thr, nms_thr)
    sc_h, cls_h, row_h, per_h = ops.soft_nms_batch_host(det, fit, box, counts, B, S, C, pr_thr, nms_thr)
    assert np.array_equal(per_d, per_h) and per_h.sum() > 0
    assert np.array_equal(cls_d, cls_h) and np.array_equal(row_d, row_h)
    assert np.array_equal(sc_d.view(np.uint32), sc_h.view(np.uint32))
    ref = _oracle_nms(det, fit, box, counts, B, sn, C + 1, pr_thr, nms_thr, True)
    lo = 0
    for b in range(B):
        hi = lo + int(per_d[b])
        assert hi - lo == len(ref[b])
        assert np.array_equal(box[row_d[lo:hi]], ref[b][:, 2:]) and np.array_equal(cls_d[lo:hi], ref[b][:, 1].astype(np.int32))
        np.testing.assert_allclose(np.exp(sc_d[lo:hi]), ref[b][:, 0], rtol=2e-6)
        lo = hi


@pytest.mark.parametrize("soft", [0, 1, 2])
def test_denet34_get_detections_vs_oracle(hip, soft, monkeypatch):
    """whole inference path on DeNet-34 skip (128x128): test-mode forward -> corner detector RoIs -> head -> NMS
    (soft = 1: Gaussian soft-NMS as a batch of 2 runs it, on host copies; 2: forced onto the device kernel)"""
    if soft == 2:
        monkeypatch.setenv("DENET_SOFT_NMS_HOST", "0")
        soft = 1
    from tests.test_parity_gpu import _warm_corner_head, rel_close
    B, IMG = 2, 128
    model = zoo.denet34(B, "skip", IMG, class_num=20, seed=1)
    rng = np.random.RandomState(5)
    dnd = model.layers[40]
    dconv = dnd.layers[0]
    dconv.omega.set_value(rng.normal(0, 0.3, dconv.omega.value.shape))
    _warm_corner_head(model, 4.0, 0.3)
    x, metas = zoo.synthetic_batch(B, IMG, seed=2)
    params = {"prThreshold": 0.08, "nmsThreshold": 0.5, "cornerThreshold": 0.02, "useSoftNMS": soft}
    # test-mode BN on untrained running statistics blows the activations up: rescale the head filters so that the
    # class logits are O(1) and the box regression outputs O(0.1), like a trained head
    dnd.get_detections(model, x, metas, params)
    raw = dnd.conv.output.data.float().cpu().numpy().reshape(-1, dnd.conv.kp)
    s0 = dnd.s0
    w = dconv.omega.get_value().copy()
    w[:s0] *= 2.0 / raw[:, :s0].std()
    w[s0:s0 + 4] *= 0.2 / raw[:, s0:s0 + 4].std()
    dconv.omega.set_value(w)
    _, total = check_detections_vs_oracle(model, x, metas, params)
    assert total > 0, "no detections: the test exercises nothing"


def check_detections_vs_oracle(model, x, metas, params, om=None):
    """one batch through the product's get_detections against the oracle, stage by stage (denet_detect.py:316-424):
    RoI proposal exact on the product's corner map (tie groups as sets), test-mode forward of oracle/model.py on the same RoIs
    (corner map 1e-3, decoded class / box arrays 1e-3), threshold + NMS exact on the product's decoded arrays. Returns
    (results, number of detections). `om`: an OracleModel of the same weights (built from model.export_json() when None)."""
    from tests.test_parity_gpu import rel_close
    B = model.batch_size
    by_type = lambda t: [l for l in model.layers if l.type_name == t][0]
    dnd, dns, cl = by_type("denet-detect"), by_type("denet-sparse"), by_type("denet-corner")
    soft = int(params.get("useSoftNMS", 0))
    results = dnd.get_detections(model, x, metas, params)
    # RoI proposal: exact on the product's corner map
    lists = OM.oracle_build_samples(cl.corner_pr.cpu().numpy(), params["cornerThreshold"], dns.sample_num, 1024, 0)
    got_lists = dns.sample_bbox_list
    for g, r in zip(got_lists, lists):        # order inside a group of exactly equal scores is unspecified (std::partial_sort)
        assert [p for p, _ in g] == [p for p, _ in r]
        i = 0
        while i < len(r):
            j = i
            while j + 1 < len(r) and r[j + 1][0] == r[i][0]:
                j += 1
            if not (j + 1 == len(r) and len(r) == dns.sample_count):   # a tie group cut by the top-K boundary
                assert sorted(bx for _, bx in g[i:j + 1]) == sorted(bx for _, bx in r[i:j + 1])
            i = j + 1
    lists = got_lists
    # head: oracle forward in test mode on the same RoIs
    if om is None:
        om = OM.OracleModel(model.export_json(), B)
    om.forward(x, None, train=False, sample_override=lists)
    rel_close(cl.corner_pr.cpu().numpy(), om.corner_pr, 1e-3, "corner_pr (test mode)")
    det_pr, fitness, bbox, counts = dnd.last_outputs
    assert counts.tolist() == [len(l) for l in lists]
    sn = dns.sample_num
    t0 = dnd._thresholds()[0]
    C = dnd.class_num
    o_det, o_fit, o_box = OL.detect_outputs(om.detect_out.v, om.sample_bbox, C, bool(dnd.use_jointfit), t0)
    C1 = C + 1
    valid = (np.arange(sn * sn)[None] < counts[:, None]).reshape(B, 1, sn, sn)
    d = det_pr.cpu().numpy().reshape(B, sn, sn, C1).transpose(0, 3, 1, 2)
    np.testing.assert_allclose(np.where(valid, d, 0), np.where(valid, o_det, 0), rtol=1e-3, atol=1e-3)
    bx = bbox.cpu().numpy().reshape(B, sn, sn, 4)
    v4 = valid.reshape(B, sn, sn, 1)
    np.testing.assert_allclose(np.where(v4, bx, 0), np.where(v4, o_box, 0), rtol=1e-3, atol=1e-4)
    # threshold + NMS: exact on the product's decoded arrays
    ref = _oracle_nms(det_pr.cpu().numpy(), fitness.cpu().numpy(), bbox.cpu().numpy(), counts, B, sn, C1,
                      params["prThreshold"], params["nmsThreshold"], soft)
    total = 0
    for b in range(B):
        dets = results[b]["detections"]
        assert results[b]["meta"] is metas[b]
        assert len(dets) == len(ref[b])
        total += len(dets)
        for (pr, cls, box), r in zip(dets, ref[b]):
            assert cls == int(r[1]) and np.array_equal(np.array(box, np.float32), r[2:])
            assert abs(pr - r[0]) <= 2e-6 * r[0]
    return results, total


def test_inference_bn_folding_matches_unfolded(hip):
    """inference folds every batch norm that sits directly behind a convolution into that convolution's filters
    (denet_bn_fold) and moves ReLU / residual add into its epilogue (denet_conv_fwd_act): same activations as the
    layer-by-layer test-mode pass to rounding, for top-level pairs and inside the residual blocks"""
    from denet_amd import ops
    from tests.test_parity_gpu import _warm_corner_head
    B, IMG = 2, 128
    model = zoo.denet34(B, "skip", IMG, class_num=20, seed=1)
    rng = np.random.RandomState(5)
    _warm_corner_head(model, 4.0, 0.3)
    # non-trivial running statistics and affine parameters so that the fold has something to get wrong
    for l in model_cnn_walk(model):
        if l.type_name in ("batchnorm", "batchnorm-relu"):
            C = l.mean.value.shape[0]
            l.mean.set_value(rng.normal(0, 0.2, C).astype(np.float32))
            l.stdinv.set_value(rng.uniform(0.7, 1.4, C).astype(np.float32))
            l.omega.set_value(rng.uniform(0.5, 1.5, C).astype(np.float32))
            l.beta.set_value(rng.normal(0, 0.2, C).astype(np.float32))
    x, _ = zoo.synthetic_batch(B, IMG, seed=2)
    outs = []
    saved = ops.INFER_FOLD
    try:
        for fold in (False, True):
            ops.INFER_FOLD = fold
            for l in model_cnn_walk(model):
                l.__dict__.pop("_plan", None)
            model.forward(x, None, train=False)
            dnc = [l for l in model.layers if l.type_name == "denet-corner"][0]
            res = [l for l in model.layers if l.type_name == "resnet"]
            outs.append([ops.nhwc_to_nchw(a.output.data, a.output_shape[1]).cpu().numpy() for a in (res[0], res[7], res[-1])] +
                        [dnc.corner_pr.cpu().numpy(), ops.nhwc_to_nchw(model.layers[2].output.data, 64).cpu().numpy()])
    finally:
        ops.INFER_FOLD = saved
    for a, b in zip(*outs):
        scale = float(np.abs(a).max())
        assert float(np.abs(a - b).max()) <= 2e-5 * scale + 1e-6, (float(np.abs(a - b).max()), scale)
    assert float(np.abs(outs[0][0]).max()) > 0


def test_inference_forward_at_the_batch_of_the_tuned_file(hip):
    """B = 32, 512x512 - the geometry at which the committed tuned file decides the algorithms, including its MODE 3 entries (the
    inference forward pass alone: the tile-parallel fused F(4x4) kernel, csrc/wino4t.hip, on the 128 / 256-channel layers, where
    training keeps the component-walk kernel) and algorithm 44 on the 64-channel stage. The test-mode forward with those decisions
    against the same forward with the tile-parallel kernel switched off (DENET_WINO4T = 0: fused F(2x2) / component-walk F(4x4)
    kernels): corner map and class logits agree to 2e-4 of their scale, the per-layer audit shows the kernels really differ."""
    from denet_amd import ops
    from denet_amd.model import audit
    from tests.test_parity_gpu import _warm_corner_head
    B = 32
    model = zoo.denet34(B, "skip", 512, class_num=80, seed=1)
    _warm_corner_head(model, 4.0, 0.3)
    rng = np.random.RandomState(3)
    dnd = [l for l in model.layers if l.type_name == "denet-detect"][0]
    dnc = [l for l in model.layers if l.type_name == "denet-corner"][0]
    dnd.layers[0].omega.set_value(rng.normal(0, 0.02, dnd.layers[0].omega.value.shape))
    x, metas = zoo.synthetic_batch(B, 512, 80, seed=1)
    xd = torch.from_numpy(x).cuda()
    ops._load_tuned_once()
    saved = (ops.WINO4T, dict(ops._WINO))
    outs, kernels = [], []
    try:
        for on in (True, False):
            if not on:
                ops.WINO4T = 0
                for k in [k for k, v in ops._WINO.items() if v == ops.FUSED4]:
                    del ops._WINO[k]                       # (what load_tuned does with entries the switches exclude: decided afresh)
            for l in model_cnn_walk(model):
                l.__dict__.pop("_plan", None)
            with audit.KernelAudit(model) as ka:
                model.forward(xd, None, train=False)
                torch.cuda.synchronize()
            kernels.append({g: tuple(e["fwd"]) for g, e in ka.summary().items()})
            dns = [l for l in model.layers if l.type_name == "denet-sparse"][0]
            outs.append((dnc.corner_pr.cpu().numpy().copy(), dnc.conv.output.data.float().cpu().numpy().copy(),
                         dnd.conv.output.data.float().cpu().numpy().copy(), [list(l) for l in dns.sample_bbox_list]))
    finally:
        ops.WINO4T = saved[0]
        ops._WINO.clear()
        ops._WINO.update(saved[1])
    used = [g for g, k in kernels[0].items() if any(n.startswith("wino4t_kernel") for n in k)]
    assert len(used) >= 5, kernels[0]                      # the 64-channel stage + the four mode-3 geometries
    assert not any(n.startswith("wino4t_kernel") for k in kernels[1].values() for n in k)
    # corner map and the corner layer's convolution output (corner logits + the sampling features): everything the backbone feeds
    # the detector with; the head's logits row by row only where both runs proposed the same RoI lists (the proposal ranks by
    # scores that differ in the last bits between the two kernel sets)
    for a, b in zip(outs[0][:2], outs[1][:2]):
        scale = float(np.abs(b).max())
        assert float(np.abs(a - b).max()) <= 2e-4 * scale, (float(np.abs(a - b).max()), scale)
    if outs[0][3] == outs[1][3]:
        a, b = outs[0][2], outs[1][2]
        assert float(np.abs(a - b).max()) <= 2e-4 * float(np.abs(b).max())


def model_cnn_walk(model):
    from denet_amd.model.model_cnn import walk_layers
    return walk_layers(model.layers)
