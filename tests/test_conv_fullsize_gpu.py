"""Numeric parity of the convolution passes AT THE GEOMETRIES THE BENCHMARK TIMES (DeNet-34 skip, B=32, 512x512; and
ResNet-34 224x224, B=64): for every distinct geometry of tests/golden/denet34_skip_shapes.json the pass that the
autotuner picks (direct tile / loop structure, Winograd F(2x2) / F(4x4), split-K rounds) AND the heuristic direct kernel
are compared with an fp64 reference of the same sums, max-norm and element-wise.

Reference op: denet/layer/convolution.py:80-83 (forward) and tensor.grad of it, model_cnn.py:318 (data / filter gradient);
north_star budget: 1e-3 relative for fp32 activations. Measured (MI355X, round 2): direct kernels <= 4.1e-6 max-norm
(up1 forward, 4608-term sums), F(2x2) <= 8e-7 (fused kernel of the 64-channel layers: 4e-7), F(4x4) <= 2.1e-5 max-norm / 2.5e-5 p99.9 (l4 / up1); asserted with a margin:
1e-5 / 2e-5 / 8e-5. The per-layer numbers of a run are written to gpurun_out/conv_fullsize_parity.json.

Metrics (both against the fp64 result `r`, s = max|r|):
  max-norm   max|got - r| / s
  p99.9      99.9th percentile of |got - r| / (|r| + rms(r))     (element-wise; the rms floor keeps the sums that cancel to
             ~0 - a filter gradient is a sum of 5e5 signed terms - from dominating the statistic)
"""
import json
import os

import pytest
import torch
import torch.nn.functional as Fn

pytestmark = pytest.mark.gpu

# name, B, H, W, C(phys), K(phys), R, S(phys), S_real, stride, pad, logical C
GEOMS = [
    ("stem7x7", 32, 512, 512, 4, 64, 7, 8, 7, 2, 3),
    ("l1_3x3", 32, 128, 128, 64, 64, 3, 3, 3, 1, 1),
    ("l2_3x3s2", 32, 128, 128, 64, 128, 3, 3, 3, 2, 1),
    ("l2_1x1s2", 32, 128, 128, 64, 128, 1, 1, 1, 2, 0),
    ("l2_3x3", 32, 64, 64, 128, 128, 3, 3, 3, 1, 1),
    ("l3_3x3s2", 32, 64, 64, 128, 256, 3, 3, 3, 2, 1),
    ("l3_1x1s2", 32, 64, 64, 128, 256, 1, 1, 1, 2, 0),
    ("l3_3x3", 32, 32, 32, 256, 256, 3, 3, 3, 1, 1),
    ("l4_3x3s2", 32, 32, 32, 256, 512, 3, 3, 3, 2, 1),
    ("l4_1x1s2", 32, 32, 32, 256, 512, 1, 1, 1, 2, 0),
    ("l4_3x3", 32, 16, 16, 512, 512, 3, 3, 3, 1, 1),
    ("up1_3x3", 32, 32, 32, 512, 256, 3, 3, 3, 1, 1),
    ("up2_3x3", 32, 64, 64, 256, 128, 3, 3, 3, 1, 1),
    ("dnc_1x1", 32, 64, 64, 128, 128, 1, 1, 1, 1, 0),
    ("head1", 32, 24, 24, 4736, 1536, 1, 1, 1, 1, 0),
    ("head2", 32, 24, 24, 1536, 1024, 1, 1, 1, 1, 0),
    ("head3", 32, 24, 24, 1024, 768, 1, 1, 1, 1, 0),
    ("head4", 32, 24, 24, 768, 512, 1, 1, 1, 1, 0),
    ("dnd_1x1", 32, 24, 24, 512, 96, 1, 1, 1, 1, 0),
    # config 2: ResNet-34 at 224x224, batch 64 (examples/resnet34-imagenet.sh:7)
    ("r34_stem", 64, 224, 224, 4, 64, 7, 8, 7, 2, 3),
    ("r34_l1", 64, 56, 56, 64, 64, 3, 3, 3, 1, 1),
    ("r34_l2", 64, 28, 28, 128, 128, 3, 3, 3, 1, 1),
    ("r34_l3", 64, 14, 14, 256, 256, 3, 3, 3, 1, 1),
    ("r34_l4", 64, 7, 7, 512, 512, 3, 3, 3, 1, 1),
    # config 5: DeNet-101 wide at 512x512, batch 16 (papers/dss/denet101.sh:13-19): bottleneck 1x1 pairs 64/256, 256/1024, 512/2048
    # channels, the strided projection into the last stage, the deepest up-sampling convolution (18 432-term sums), the 128x128
    # corner convolution and the first two head layers at M = 16 x 2304 RoIs
    ("d101_b1_1x1", 16, 128, 128, 64, 256, 1, 1, 1, 1, 0),
    ("d101_b1_1x1r", 16, 128, 128, 256, 64, 1, 1, 1, 1, 0),
    ("d101_l3_1x1", 16, 32, 32, 256, 1024, 1, 1, 1, 1, 0),
    ("d101_l3_1x1r", 16, 32, 32, 1024, 256, 1, 1, 1, 1, 0),
    ("d101_l4_1x1", 16, 16, 16, 512, 2048, 1, 1, 1, 1, 0),
    ("d101_l4_1x1r", 16, 16, 16, 2048, 512, 1, 1, 1, 1, 0),
    ("d101_l4_1x1s2", 16, 32, 32, 1024, 2048, 1, 1, 1, 2, 0),
    ("d101_up1_3x3", 16, 32, 32, 2048, 1024, 3, 3, 3, 1, 1),
    ("d101_dnc_1x1", 16, 128, 128, 256, 160, 1, 1, 1, 1, 0),
    ("d101_head1", 16, 48, 48, 6304, 2048, 1, 1, 1, 1, 0),
    ("d101_head2", 16, 48, 48, 2048, 1536, 1, 1, 1, 1, 0),
]

RESULTS = {}
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "conv_fullsize_parity.json")


def _metrics(got, ref):
    got = got.double()
    s = float(ref.abs().max())
    d = (got - ref).abs()
    mx = float(d.max()) / s
    rel = (d / (ref.abs() + float(ref.pow(2).mean().sqrt()))).reshape(-1)
    if rel.numel() > (1 << 24):                       # torch.quantile's input limit: a fixed stride sample
        rel = rel[:: rel.numel() // (1 << 24) + 1]
    p999 = float(torch.quantile(rel.float(), 0.999))
    return mx, p999


def _ref_fp64(x, w, dy, stride, pad, s_real, need_dx):
    """fp64 correlation in NHWC / KRSC terms (the kernels' own layout: filters already flipped by Param.to_dev_layout),
    computed in image chunks so that the fp64 copies stay small"""
    B = x.shape[0]
    w64 = w[:, :, :s_real, :].double().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    ys, dxs = [], []
    dw = torch.zeros_like(w64)
    step = max(1, min(B, (1 << 27) // max(1, x[0].numel() + dy[0].numel())))
    for i in range(0, B, step):
        xc = x[i:i + step].double().permute(0, 3, 1, 2).contiguous().requires_grad_(need_dx)
        y = Fn.conv2d(xc, w64, None, stride=stride, padding=pad)
        g = torch.autograd.grad(y, [w64] + ([xc] if need_dx else []), dy[i:i + step].double().permute(0, 3, 1, 2))
        dw += g[0]
        ys.append(y.detach().permute(0, 2, 3, 1))
        if need_dx:
            dxs.append(g[1].permute(0, 2, 3, 1))
    return torch.cat(ys), (torch.cat(dxs) if need_dx else None), dw.permute(0, 2, 3, 1)


@pytest.mark.parametrize("geom", GEOMS, ids=[g[0] for g in GEOMS])
def test_conv_passes_at_benchmark_geometry(hip, geom):
    from denet_amd import ops
    name, B, H, W, C, K, R, S, s_real, stride, pad = geom
    gen = torch.Generator(device="cpu").manual_seed(len(name) * 131 + C + K)
    x = torch.randn(B, H, W, C, generator=gen).cuda()
    if C == 4:
        x[..., 3] = 0                                             # the padded input channel is zero by construction
    w = (torch.randn(K, R, S, C, generator=gen) * (2.0 / (R * s_real * C)) ** 0.5).cuda()
    if S != s_real:
        w[:, :, s_real:, :] = 0                                   # padded taps carry zero weight
    OH = (H + 2 * pad - R) // stride + 1
    OW = (W + 2 * pad - s_real) // stride + 1
    dy = torch.randn(B, OH, OW, K, generator=gen).cuda()
    need_dx = C >= 32
    y_ref, dx_ref, dw_ref = _ref_fp64(x, w, dy, stride, pad, s_real, need_dx)

    saved = (ops.AUTOTUNE, dict(ops._WINO), set(ops._TUNED))
    res = {}
    try:
        g0 = ops.conv_geom(x.shape, w.shape, stride, pad, s_real)
        labels = [("direct", False), ("tuned", True)]
        if C == 64 and ops.conv_wino4t_ok(0, g0) and ops.conv_wino4t_ok(1, g0):
            labels.append(("fused4", True))        # what the committed file runs on the 64-channel stage: csrc/wino4t.hip (algorithm 44)
        for label, tuned in labels:
            ops.AUTOTUNE = tuned
            ops._WINO.clear()
            ops._TUNED.clear()
            if label == "fused4":
                ops._WINO[(0, g0)] = ops._WINO[(1, g0)] = ops.FUSED4
            # "tuned": the algorithm of ops.static_policy (fused 64-channel / F(4x4) / F(2x2) wherever the geometry allows one - the
            # widest kernel coverage; nothing is timed) on the launch configurations of the committed file
            for rep in range(2):                                   # first call: decides and sets buffers up; second: the fixed choice alone
                y = ops.conv_fwd(x, w, stride=stride, pad=pad, s_real=s_real)
                dw = ops.conv_wgrad(x, dy, tuple(w.shape), stride=stride, pad=pad, s_real=s_real)
                dx = ops.conv_dgrad(dy, w, tuple(x.shape), stride=stride, pad=pad, s_real=s_real) if need_dx else None
            g = ops.conv_geom(x.shape, w.shape, stride, pad, s_real)
            algo = {m: ops._WINO.get((i, g), 0) for i, m in enumerate(("fwd", "dgrad", "wgrad"))}
            ent = {"fwd": _metrics(y, y_ref), "wgrad": _metrics(dw[:, :, :s_real, :], dw_ref), "algo": algo}
            if need_dx:
                ent["dgrad"] = _metrics(dx, dx_ref)
            if S != s_real:
                assert float(dw[:, :, s_real:, :].abs().max()) == 0.0
            # the same forward pass with the batch-norm column sums written by its epilogue (what ConvLayer.forward runs in
            # training): identical output, sums against fp64
            cache = {"train": True}
            y2 = ops.conv_fwd(x, w, stride=stride, pad=pad, s_real=s_real, cache=cache, bn_stats=True)
            st = cache.get("bn_stats")
            assert torch.equal(y2, y)
            if st is not None:
                part = st[0][:st[1] * 2 * K].view(st[1], 2, K).sum(0)
                yd = y_ref.reshape(-1, K)
                ent["bn_sums"] = (float((part[0] - yd.sum(0)).abs().max() / (yd.abs().sum(0).max() + 1e-30)),
                                  float((part[1] - (yd * yd).sum(0)).abs().max() / (yd * yd).sum(0).max()))
                assert ent["bn_sums"][0] <= 1e-5 and ent["bn_sums"][1] <= 1e-5, (name, label, ent["bn_sums"])
            res[label] = ent
    finally:
        ops.AUTOTUNE = saved[0]
        ops._WINO.clear()
        ops._WINO.update(saved[1])
        ops._TUNED.clear()
        ops._TUNED.update(saved[2])
    RESULTS[name] = res
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(RESULTS, f, indent=1)
    for label, ent in res.items():
        for p in ("fwd", "dgrad", "wgrad"):
            if p not in ent:
                continue
            mx, p999 = ent[p]
            wino = ent["algo"][p]
            # direct kernels: fp32 FMA chains of <= 524 288 terms; Winograd F(4x4): transform constants up to 8 amplify the
            # rounding of the 36-point products (Lavin & Gray report ~1e-5 for F(4x4) in fp32)
            bound = {0: 1e-5, 2: 2e-5, 4: 8e-5, 22: 1e-5, 44: 8e-5}[wino]     # 22: ops.FUSED2, F(2x2) in one kernel; 44: ops.FUSED4, F(4x4) in one kernel
            assert mx <= bound, "%s %s %s (winograd tile %d): max-norm error %.2e > %.0e" % (name, label, p, wino, mx, bound)
            assert p999 <= 3 * bound, "%s %s %s: p99.9 element-wise error %.2e" % (name, label, p, p999)
            assert mx <= 1e-3 and p999 <= 1e-3                      # the north-star budget itself
