"""CPU tests of the oracle (oracle/): pins it against the reference's own known answers and against an
independent second implementation (torch CPU ops + autograd), as SURVEY.md §8(c) prescribes."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as Fn

from oracle import layers as L
from oracle import model as OM

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _corner_map(case):
    H, W, Cn = case["H"], case["W"], case["Cn"]
    P = np.full((1, Cn, H, W), case["default_p"], np.float64)
    for c in case["corners"]:
        P[0, c["type"], c["y"], c["x"]] = c["p"]
    pos = np.log(P).astype(np.float32)
    neg = np.log(1 - P).astype(np.float32)
    return np.ascontiguousarray(np.stack([neg, pos], axis=1))


def test_build_samples_known_answers():
    """SURVEY.md §8(c) known answers of the compiled reference build_samples"""
    kat = json.load(open(os.path.join(GOLDEN, "build_samples_kat.json")))
    for case in kat["cases"]:
        pr = _corner_map(case)
        lists = OM.oracle_build_samples(pr, case["corner_threshold"], case["sample_num"], case["max_corners"],
                                        case["local_max"])
        got = lists[0]
        if "expect" in case:
            assert len(got) == len(case["expect"]), case["name"]
            for g, e in zip(got, case["expect"]):
                assert np.float32(g[0]) == np.float32(e["pr"]), (case["name"], g[0])
                assert tuple(g[1]) == tuple(e["box"]), case["name"]
        else:
            assert sorted(tuple(g[1]) for g in got) == sorted(tuple(b) for b in case["expect_boxes"]), case["name"]


def _random_corner_map(rng, B, Cn, H, W, mu, quantise=0):
    """log-probability maps [B,2,Cn,H,W] from random logits; quantise > 0 rounds the logits to that many levels so that
    exactly equal scores (tie groups) occur"""
    z = rng.normal(mu, 2.0, (B, Cn, H, W))
    if quantise:
        z = np.round(z * quantise) / quantise
    p = 1.0 / (1.0 + np.exp(-z))
    return np.ascontiguousarray(np.stack([np.log1p(-p), np.log(p)], axis=1).astype(np.float32))


def _assert_same_ranking(cpp_rows, cpp_boxes, naive_full, count, tag):
    """cpp: the C++ oracle's output rows (pr, box fp32) + integer boxes; naive_full: the naive implementation's ranked list
    WITHOUT the final cut. Scores must agree bit for bit position by position; boxes are compared per tie group (the
    reference's std::partial_sort leaves the order inside a group, and the members of a group cut by the top-K limit,
    unspecified)."""
    from oracle import build_samples_naive as NV
    want = naive_full[:count]
    assert len(cpp_rows) == len(want), (tag, len(cpp_rows), len(want))
    for i, (row, w) in enumerate(zip(cpp_rows, want)):
        assert np.float32(row[0]) == w[0], (tag, i, row[0], w[0])
    full_groups = NV.tie_groups(naive_full)
    pos = 0
    stats = {"multi": 0, "cut": 0}         # tie groups with > 1 member inside the list / a group straddling the top-K cut
    for score, members in full_groups:
        if pos >= len(cpp_rows):
            break
        n = min(len(members), len(cpp_rows) - pos)
        stats["multi"] += int(len(members) > 1)
        stats["cut"] += int(n < len(members))
        got = {tuple(int(v) for v in cpp_boxes[pos + k]) for k in range(n)}
        assert len(got) == n, (tag, "duplicate boxes inside a tie group")
        if n == len(members):
            assert got == members, (tag, pos, score)
        else:
            assert got <= members, (tag, pos, score)        # the group straddles the cut: any n members
        for k in range(n):
            x0, y0, x1, y1 = (int(v) for v in cpp_boxes[pos + k])
            b = cpp_rows[pos + k][1:5]
            H, W = _assert_same_ranking.hw
            assert tuple(np.float32(v) for v in b) == (np.float32(x0 / W), np.float32(y0 / H), np.float32((x1 + 1) / W),
                                                      np.float32((y1 + 1) / H)), (tag, pos + k)
        pos += n
    return stats


def test_build_samples_cpp_vs_independent_naive_implementation():
    """oracle/build_samples.cc (the checker of the GPU RoI proposal) against oracle/build_samples_naive.py, a
    second implementation written separately from denet_sparse.cc:271-557 (dict de-dup, sorted + explicit tie groups):
    several hundred random maps - sparse, dense, > max_corners per type (truncation re-orders the corner lists,
    :526-530), local_max 0..2 (:474-487), C = 5 (:374-466), quantised maps with exact score ties."""
    from oracle import build_samples_naive as NV
    rng = np.random.RandomState(20260928)
    n_cases = 0
    tally = {k: {False: 0, True: 0} for k in ("images", "images_with_tie_groups", "images_with_cut_tie_group")}
    for case in range(260):
        Cn = 5 if case % 5 == 4 else 4
        H = int(rng.randint(6, 21))
        W = int(rng.randint(6, 21))
        B = 1 + case % 2
        mu = [-3.5, -2.5, -1.5, -0.5][case % 4]
        quant = 2 if case % 7 == 3 else 0
        local_max = case % 3
        max_corners = [1024, 1024, 12, 5][(case // 3) % 4]
        sample_num = [24, 6, 3][(case // 2) % 3]
        thr = [0.01, 0.05, 0.3][(case // 5) % 3]
        pr = _random_corner_map(rng, B, Cn, H, W, mu, quant)
        if quant and max_corners < 1024:
            max_corners = 1024          # ties at the corner truncation cut leave the kept SET unspecified in the reference
        out, box, absd, cnt = OM.oracle_build_samples_raw(pr, thr, sample_num, max_corners, local_max, 1.0)
        _assert_same_ranking.hw = (H, W)
        for b in range(B):
            corners = [NV.find_corners(pr, b, ci, NV._logf(np.float32(thr)), max_corners, local_max) for ci in range(Cn)]
            full = NV.rank(NV.search_corners(pr, b, corners))
            st = _assert_same_ranking(out[b, :cnt[b]], box[b, :cnt[b]], full, sample_num * sample_num,
                                      (case, b, Cn, H, W, mu, quant, local_max, max_corners, sample_num, thr))
            n_cases += 1
            tally["images_with_tie_groups"][bool(quant)] += int(st["multi"] > 0)
            tally["images_with_cut_tie_group"][bool(quant)] += int(st["cut"] > 0)
            tally["images"][bool(quant)] += 1
    assert n_cases >= 380
    # How far "RoI lists bit-identical to the reference" reaches: inside a group of exactly equal fp32
    # scores the reference's std::partial_sort leaves the order - and, for a group cut by the top-K limit, the membership -
    # unspecified, so such groups are compared as sets above. Continuous random maps: distinct |pr_f - pr_t| still collide
    # in fp32 now and then, and the cut hits such a group only rarely; quantised maps (built to tie) do both often.
    print("tie-group tally (continuous maps / quantised maps):", {k: (v[False], v[True]) for k, v in tally.items()})
    assert tally["images_with_cut_tie_group"][True] > 0, "the quantised maps no longer exercise a tie group at the cut"


def test_build_samples_clustering_cpp_vs_naive():
    """apply_cluster (denet_sparse.cc:165-242) of the C++ oracle against the naive implementation: continuous random maps
    (no score ties, so the result is fully specified): identical sample sequences"""
    from oracle import build_samples_naive as NV
    rng = np.random.RandomState(77)
    ran = 0
    for case in range(40):
        H = W = int(rng.randint(10, 19))
        sample_num = [3, 4, 5][case % 3]
        cthr = [0.3, 0.5, 0.7, 0.9][case % 4]
        pr = _random_corner_map(rng, 1, 4, H, W, -1.5)
        out, box, absd, cnt = OM.oracle_build_samples_raw(pr, 0.05, sample_num, 1024, 0, cthr)
        want = NV.build_samples(pr, 0.05, sample_num, 1024, 0, cthr)[0]
        plain = NV.build_samples(pr, 0.05, sample_num, 1024, 0, 1.0)[0]
        scores = [float(w[0]) for w in want]
        if len(set(scores)) != len(scores):
            continue
        ran += want != plain
        assert cnt[0] == len(want), (case, cnt[0], len(want))
        for i, w in enumerate(want):
            assert np.float32(out[0, i, 0]) == w[0], (case, i)
            assert tuple(int(v) for v in box[0, i]) == w[1], (case, i)
    assert ran >= 20         # clustering really changed the selection in most cases


def test_build_samples_hand_derived_known_answers():
    """Known answers derived BY HAND from denet_sparse.cc (no implementation involved in the expectation): both the C++
    oracle and the naive implementation must reproduce them."""
    from oracle import build_samples_naive as NV
    lo, hi = 1e-4, 0.9

    def run(corners, H=8, W=8, Cn=4, thr=0.01, sample_num=24, max_corners=1024, local_max=0):
        case = {"H": H, "W": W, "Cn": Cn, "default_p": lo, "corners": corners}
        pr = _corner_map(case)
        out, box, absd, cnt = OM.oracle_build_samples_raw(pr, thr, sample_num, max_corners, local_max, 1.0)
        cpp = [tuple(int(v) for v in box[0, i]) for i in range(cnt[0])]
        nv = [s[1] for s in NV.build_samples(pr, thr, sample_num, max_corners, local_max, 1.0)[0]]
        return cpp, nv, pr

    def c(t, x, y, p=hi):
        return {"type": t, "x": x, "y": y, "p": p}

    # (1) duplicate through both generators: TL(1,1) x BR(5,6) (:333-352) and TR(x=5,y=1) x BL(x=1,y=6) (:358-373) describe
    # the same box (1,1,5,6); the 64-bit key (:311-318) is already in the map when the second loop reaches it (:367) ->
    # exactly one sample
    cpp, nv, _ = run([c(0, 1, 1), c(3, 5, 6), c(1, 5, 1), c(2, 1, 6)])
    assert cpp == nv == [(1, 1, 5, 6)]

    # (2) degenerate pairs are skipped (:343 `x1 <= x0 || y1 <= y0`): BR in the same column / the same row / above-left
    cpp, nv, _ = run([c(0, 3, 3), c(3, 3, 6), c(3, 6, 3), c(3, 1, 1)])
    assert cpp == nv == []

    # (3) ranking (:78, :547): score = 1/(1+exp|pr_f - pr_t|), pr_f - pr_t = sum over the 4 cells of log(1-p) - log(p).
    # The TR / BL cells of these boxes carry the default p = 1e-4 (+9.21 each), the TL cell p = .9 (-2.197); a BR cell at
    # p = .9 adds -2.197 (total 14.03), at p = .5 adds 0 (total 16.22). Smaller |.| = larger score: the box through the
    # p = .9 BR corner (6,6) ranks first although it is generated second (BR inner loop in raster order: (4,4) before (6,6))
    cpp, nv, pr = run([c(0, 1, 1), c(3, 4, 4, 0.5), c(3, 6, 6, hi)])
    assert cpp == nv == [(1, 1, 6, 6), (1, 1, 4, 4)]

    # (4) generation order on exact ties (:333-352: TL outer in raster order y-major, BR inner): four boxes whose corner
    # cells all carry p = hi and whose other two cells (TR, BL planes) carry the default -> four identical scores. The
    # naive implementation keeps generation order on ties (stable sort): TL(1,1)xBR(5,5), TL(1,1)xBR(6,6), TL(2,2)xBR(5,5),
    # TL(2,2)xBR(6,6); the C++ oracle may permute them (std::partial_sort) but must return the same set
    cpp, nv, _ = run([c(0, 1, 1), c(0, 2, 2), c(3, 5, 5), c(3, 6, 6)])
    assert nv == [(1, 1, 5, 5), (1, 1, 6, 6), (2, 2, 5, 5), (2, 2, 6, 6)]
    assert sorted(cpp) == sorted(nv)

    # (5) corner truncation (:526-530): three TL candidates, max_corners = 2 -> the two with the highest log-probability
    # survive ((1,1) p=.9 and (3,1) p=.8; (2,1) p=.5 is dropped), so only two boxes are produced
    cpp, nv, _ = run([c(0, 1, 1, 0.9), c(0, 2, 1, 0.5), c(0, 3, 1, 0.8), c(3, 6, 6, 0.9)], max_corners=2)
    assert sorted(cpp) == sorted(nv) == [(1, 1, 6, 6), (3, 1, 6, 6)]

    # (6) top-K cut (:547-549): sample_num = 1 keeps the single best box of case (3)
    cpp, nv, _ = run([c(0, 1, 1), c(3, 4, 4, 0.5), c(3, 6, 6, hi)], sample_num=1)
    assert cpp == nv == [(1, 1, 6, 6)]

    # (7) local maximum window (:474-487) is [y-l, y+l) x [x-l, x+l) after clipping the upper bound to size-1: with l = 1 a
    # stronger neighbour at (x-1, y-1) suppresses the corner, a stronger neighbour at (x+1, y+1) does not
    cpp, nv, _ = run([c(0, 2, 2, 0.5), c(0, 1, 1, 0.9), c(3, 6, 6)], local_max=1)
    assert sorted(cpp) == sorted(nv) == [(1, 1, 6, 6)]                      # (2,2) suppressed by (1,1)
    cpp, nv, _ = run([c(0, 2, 2, 0.5), c(0, 3, 3, 0.9), c(3, 6, 6)], local_max=1)
    assert sorted(cpp) == sorted(nv) == [(2, 2, 6, 6), (3, 3, 6, 6)]        # (3,3) is outside (2,2)'s half-open window

    # (8) threshold is strict on the log-probability (:510 `log_pr > threshold`, threshold = logf(corner_threshold) :503): a
    # corner whose fp32 log-probability equals logf(thr) exactly is NOT a corner
    thr = 0.25
    case = {"H": 8, "W": 8, "Cn": 4, "default_p": lo, "corners": [c(0, 1, 1, 0.9), c(3, 6, 6, 0.9), c(3, 5, 5, 0.9)]}
    pr = _corner_map(case)
    pr[0, 1, 3, 5, 5] = NV._logf(np.float32(thr))
    out, box, absd, cnt = OM.oracle_build_samples_raw(pr, thr, 24, 1024, 0, 1.0)
    nv = [s[1] for s in NV.build_samples(pr, thr, 24, 1024, 0, 1.0)[0]]
    assert [tuple(int(v) for v in box[0, i]) for i in range(cnt[0])] == nv == [(1, 1, 6, 6)]

    # (9) the score of case (1), from the formula alone (fp32 running sums in the order TL,TR,BL,BR :276-294; expf; double
    # division :306): SURVEY 8c recorded 3.6007157e-07 from the compiled reference for P = 0.9 / 0.8 at the two corners
    pr = _corner_map({"H": 8, "W": 8, "Cn": 4, "default_p": 1e-4, "corners": [c(0, 1, 1, 0.9), c(3, 5, 6, 0.8)]})
    got = NV.build_samples(pr, 0.01, 24)[0]
    assert len(got) == 1 and got[0][0] == np.float32(3.6007157e-07) and got[0][2] == (0.125, 0.125, 0.75, 0.875)


def test_bn_known_answer():
    """the reference's own BN test block (denet/layer/batch_norm.py:131-154)"""
    np.random.seed(1002)
    eps = 1e-4
    x = np.random.uniform(0.0, 1.0, (64, 128, 32, 32)).astype(np.float32)
    C = 128
    y, mean, invstd = L.bn_train(x, np.ones(C, np.float32), np.zeros(C, np.float32), 1e-5)
    rm, rs = L.bn_running_update(np.zeros(C, np.float32), np.ones(C, np.float32), mean, invstd, 0.9)
    assert abs(y.mean()) < eps and abs(y.std() - 1.0) < eps
    assert abs(rm.mean() - x.mean() * 0.1) < eps
    assert abs(rs.mean() - 1.24641) < eps


def _t(a, grad=False):
    return torch.tensor(np.asarray(a, dtype=np.float64), requires_grad=grad)


@pytest.mark.parametrize("k,stride,pad", [(3, 1, 1), (3, 2, 1), (1, 2, 0), (7, 2, 3), (4, 1, 2)])
def test_conv_vs_torch(k, stride, pad):
    rng = np.random.RandomState(k * 10 + stride)
    x = rng.randn(2, 5, 12, 12).astype(np.float32)
    w = rng.randn(6, 5, k, k).astype(np.float32)
    b = rng.randn(6).astype(np.float32)
    tx, tw, tb = _t(x, True), _t(w, True), _t(b, True)
    ty = Fn.conv2d(tx, torch.flip(tw, [2, 3]), tb, stride=stride, padding=pad)   # true convolution
    y = L.conv2d(x, w, b, stride, pad)
    np.testing.assert_allclose(y, ty.detach().numpy(), rtol=1e-4, atol=1e-4)
    dy = rng.randn(*y.shape).astype(np.float32)
    ty.backward(_t(dy))
    dx, dw, db = L.conv2d_grad(x, w, dy, stride, pad)
    np.testing.assert_allclose(dx, tx.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(dw, tw.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(db, tb.grad.numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("k,stride", [(3, 1), (3, 2), (4, 2), (1, 1), (5, 3)])
def test_deconv_vs_torch(k, stride):
    """`DC` = gradient of the true convolution w.r.t. its input (denet/layer/deconvolution.py:54-67): second
    implementation = torch conv_transpose2d on the flipped, axis-swapped filters, output_padding = stride - 1"""
    rng = np.random.RandomState(k * 10 + stride)
    pad = k // 2
    x = rng.randn(2, 5, 6, 7).astype(np.float32)
    w = rng.randn(4, 5, k, k).astype(np.float32)           # omega (C_out, C_in, kh, kw)
    b = rng.randn(4).astype(np.float32)
    tx, tw, tb = _t(x, True), _t(w, True), _t(b, True)
    wt = torch.flip(tw.permute(1, 0, 2, 3), [2, 3])        # correlation filters of F, (C_in, C_out, kh, kw)
    ty = Fn.conv_transpose2d(tx, wt, tb, stride=stride, padding=pad, output_padding=stride - 1)
    y = L.deconv2d(x, w, b, stride, pad)
    assert y.shape == (2, 4, 6 * stride - 2 * pad + k - 1, 7 * stride - 2 * pad + k - 1)
    np.testing.assert_allclose(y, ty.detach().numpy(), rtol=1e-4, atol=1e-4)
    dy = rng.randn(*y.shape).astype(np.float32)
    ty.backward(_t(dy))
    dx, dw, db = L.deconv2d_grad(x, w, dy, stride, pad)
    np.testing.assert_allclose(dx, tx.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(dw, tw.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(db, tb.grad.numpy(), rtol=1e-4, atol=1e-4)


def test_border_crop_dropout_restatements():
    """`B` (border.py:30-33), `CM` (crop_mirror.py:26-56) and `D` (dropout.py:20-24): shapes, adjoints and the
    statistics of the counter-based generator that replaces the (third-party, unpinned) Theano MRG stream"""
    rng = np.random.RandomState(2)
    x = rng.randn(3, 2, 5, 6).astype(np.float32)
    y = L.border(x, (1, 2, 3, 0))                         # (left, right, top, bottom)
    assert y.shape == (3, 2, 8, 9) and np.array_equal(y[:, :, 3:8, 1:7], x) and y.sum() == pytest.approx(x.sum(), rel=1e-5)
    assert np.array_equal(L.border_grad(y, (1, 2, 3, 0)), x)

    # crop-mirror: test mode = centre crop; train mode = a permutation of a window, adjoint identity <Ax,y> = <x,A'y>
    g0 = L.crop_mirror_geom(3, 5, 6, 3, 4, 0.5, 0.5, False, 123)
    assert g0 == [(1, 1, False, False)] * 3
    assert np.array_equal(L.crop_mirror(x, (3, 4), g0), x[:, :, 1:4, 1:5])
    seen = set()
    for seed in range(200):
        g = L.crop_mirror_geom(3, 5, 6, 3, 4, 0.5, 0.25, True, L.layer_seed(7, 1, seed))
        for r0, c0, flip, mirror in g:
            assert 0 <= r0 <= 2 and 0 <= c0 <= 2
            seen.add((r0, c0, flip, mirror))
        ycm = L.crop_mirror(x, (3, 4), g)
        dy = rng.randn(*ycm.shape).astype(np.float32)
        assert np.vdot(ycm, dy) == pytest.approx(np.vdot(x, L.crop_mirror_grad(dy, x.shape, g)), rel=1e-4, abs=1e-4)
    assert len(seen) == 3 * 3 * 2 * 2                      # every offset / flip / mirror combination occurs
    one = L.crop_mirror(x[:1], (3, 4), [(2, 1, True, True)])
    assert np.array_equal(one[0], x[0, :, 2:5, 1:5][:, ::-1, ::-1])

    # dropout: keep probability 1 - rate, scale 1/(1 - rate), different per (layer, iteration), reproducible
    m = L.dropout_mask((8, 16, 32, 32), 0.3, L.layer_seed(5, 2, 0))
    assert set(np.unique(m).tolist()) == {0.0, float(np.float32(1.0 / (1.0 - float(np.float32(0.3)))))}
    assert abs((m > 0).mean() - 0.7) < 5e-3
    assert abs(m.mean() - 1.0) < 1e-2
    assert np.array_equal(m, L.dropout_mask((8, 16, 32, 32), 0.3, L.layer_seed(5, 2, 0)))
    assert (m != L.dropout_mask((8, 16, 32, 32), 0.3, L.layer_seed(5, 2, 1))).mean() > 0.3
    assert (m != L.dropout_mask((8, 16, 32, 32), 0.3, L.layer_seed(5, 3, 0))).mean() > 0.3
    assert abs(np.corrcoef(m[:, :, :, :-1].ravel(), m[:, :, :, 1:].ravel())[0, 1]) < 5e-3   # neighbours independent
    assert L.u24(0, np.array([0], dtype=np.uint64))[0] == (0 >> 40)      # mix64(0) == 0: the finaliser's fixed point
    assert int(L.mix64(1, np.array([0], dtype=np.uint64))[0]) == 0x5692161D100B05E5        # splitmix64 known answer


def test_bn_grad_and_test_mode_vs_torch():
    rng = np.random.RandomState(3)
    x = (rng.randn(4, 6, 5, 5) * 2 + 1).astype(np.float32)
    g, b = rng.rand(6).astype(np.float32) + 0.5, rng.randn(6).astype(np.float32)
    tx, tg, tb = _t(x, True), _t(g, True), _t(b, True)
    ty = Fn.batch_norm(tx, None, None, tg, tb, training=True, eps=1e-5)
    y, mean, invstd = L.bn_train(x, g, b, 1e-5)
    np.testing.assert_allclose(y, ty.detach().numpy(), rtol=1e-5, atol=1e-5)
    dy = rng.randn(*x.shape).astype(np.float32)
    ty.backward(_t(dy))
    dx, dg, db = L.bn_grad(x, dy, g, mean, invstd)
    np.testing.assert_allclose(dx, tx.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dg, tg.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(db, tb.grad.numpy(), rtol=1e-4, atol=1e-5)
    # test mode: eps is applied twice (batch_norm.py:50-52)
    rm, rs = rng.randn(6).astype(np.float32), (rng.rand(6) + 0.5).astype(np.float32)
    yt = L.bn_test(x, g, b, rm, rs, 1e-5)
    var = (1.0 / rs.astype(np.float64)) ** 2
    ref = (x - rm[None, :, None, None]) / np.sqrt(var + 1e-5)[None, :, None, None] * g[None, :, None, None] + b[None, :, None, None]
    np.testing.assert_allclose(yt, ref, rtol=1e-5, atol=1e-5)


def test_pools_vs_torch():
    rng = np.random.RandomState(4)
    x = rng.randn(2, 3, 11, 13).astype(np.float32)
    for k, s, p in [(3, 2, 1), (2, 2, 0)]:
        tx = _t(x, True)
        ty = Fn.max_pool2d(tx, k, s, p)
        y, arg = L.pool_max(x, k, s, p)
        np.testing.assert_array_equal(y, ty.detach().numpy().astype(np.float32))
        dy = rng.randn(*y.shape).astype(np.float32)
        ty.backward(_t(dy))
        np.testing.assert_allclose(L.pool_max_grad(dy, arg, x.shape, k, s, p), tx.grad.numpy(), rtol=1e-6, atol=1e-6)
        tx = _t(x, True)
        ty = Fn.avg_pool2d(tx, k, s, p, count_include_pad=True)
        np.testing.assert_allclose(L.pool_avg(x, k, s, p), ty.detach().numpy(), rtol=1e-5, atol=1e-6)
        ty.backward(_t(dy))
        np.testing.assert_allclose(L.pool_avg_grad(dy, x.shape, k, s, p), tx.grad.numpy(), rtol=1e-5, atol=1e-6)
    # pool-inv: the reference's own differential design (pool_inv.py:43-88): op == double repeat
    rng = np.random.RandomState(1)
    x = rng.uniform(-5, 5, (4, 64, 4, 4)).astype(np.float32)
    tx = _t(x, True)
    ty = tx.repeat_interleave(2, 2).repeat_interleave(2, 3)
    np.testing.assert_array_equal(L.pool_inv(x, (2, 2)), ty.detach().numpy().astype(np.float32))
    ty.sum().backward()
    np.testing.assert_allclose(L.pool_inv_grad(np.ones((4, 64, 8, 8), np.float32), (2, 2)), tx.grad.numpy())


def test_corner_cost_grad_vs_autograd():
    rng = np.random.RandomState(5)
    x = (rng.randn(2, 4, 6, 6) * 2).astype(np.float32)
    t = rng.rand(2, 2, 4, 6, 6).astype(np.float32) / 100
    tx = _t(x, True)
    lp = torch.log_softmax(torch.stack([tx, -tx], 1), 1)
    cost = 100.0 * (-(_t(t) * lp).sum(dim=(1, 2, 3, 4)).mean() / np.log(2))
    cost.backward()
    pr = L.corner_pr(x)
    np.testing.assert_allclose(pr, lp.detach().numpy(), rtol=1e-5, atol=1e-6)
    c, g = L.corner_cost(t, pr, 100.0)
    assert abs(c - float(cost)) < 1e-4 * abs(float(cost))
    np.testing.assert_allclose(g, tx.grad.numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("rule", [0, 1])
def test_sparse_sample_vs_gather(rule):
    """the reference's own differential design (denet_sparse.py:222-285): op vs explicit gather, fwd and grad"""
    import random
    rng = np.random.RandomState(1)
    random.seed(1)
    B, Fc, H, W, sn, gs = 2, 8, 16, 16, 5, 7
    x = rng.uniform(-5, 5, (B, Fc, H, W)).astype(np.float32)
    bbox = np.zeros((B, sn, sn, 4), np.float32)
    for b in range(B):
        for idx in range(sn * sn):
            j, i = idx // sn, idx % sn
            bbox[b, j, i, 0] = random.uniform(0.0, 1.0)
            bbox[b, j, i, 1] = random.uniform(0.0, 1.0)
            bbox[b, j, i, 2] = random.uniform(bbox[b, j, i, 0], 1.0)
            bbox[b, j, i, 3] = random.uniform(bbox[b, j, i, 1], 1.0)
    out, taps = L.sparse_sample(x, bbox, gs, rule)
    ys, xs = taps
    assert out.shape == (B, gs * gs * Fc + 2, sn, sn)
    # scalar restatement of one RoI
    for (b, j, i) in [(0, 0, 0), (1, 3, 2), (1, 4, 4)]:
        m = b * sn * sn + j * sn + i
        x0, y0, x1, y1 = bbox[b, j, i]
        for yi in range(gs):
            for xi in range(gs):
                v = out[b, (yi * gs + xi) * Fc:(yi * gs + xi + 1) * Fc, j, i]
                np.testing.assert_array_equal(v, x[b, :, ys[m, yi], xs[m, xi]])
        assert out[b, gs * gs * Fc, j, i] == np.float32(y1 - y0) and out[b, gs * gs * Fc + 1, j, i] == np.float32(x1 - x0)
    # gradient of sum(output) w.r.t. fmap == tap counts
    g = L.sparse_sample_grad(np.ones_like(out), taps, x.shape, gs)
    counts = np.zeros((B, H, W))
    for m in range(B * sn * sn):
        for yi in range(gs):
            for xi in range(gs):
                counts[m // (sn * sn), ys[m, yi], xs[m, xi]] += 1
    np.testing.assert_allclose(g, np.broadcast_to(counts[:, None], g.shape))


def test_tap_rules_differ_only_at_half_cells():
    """(i*w)/(gs-1) round-half-even vs i*w*(1/(gs-1)) lroundf: lattice boxes produce exact .5 taps"""
    bb = np.array([[1 / 64, 1 / 64, 4 / 64, 4 / 64]], np.float32)   # w = 3 cells: tap 1 -> 1.5
    y0, x0, _, _ = L.sparse_taps(bb, 7, 64, 64, 0)
    y1, x1, _, _ = L.sparse_taps(bb, 7, 64, 64, 1)
    assert x0[0, 1] == 2 and x1[0, 1] == 2          # 1.5 -> 2 under both (even / away)
    assert x0[0, 3] == 2 and x1[0, 3] == 3          # 2.5 -> 2 (half to even) vs 3 (half away)


@pytest.mark.parametrize("bounded", [False, True])
def test_detect_cost_grad_vs_autograd(bounded):
    rng = np.random.RandomState(6)
    B, sn, C = 2, 3, 5
    s0 = C + 1
    out = rng.randn(B, s0 + 4, sn, sn).astype(np.float32)
    det_t = rng.rand(B, s0, sn, sn).astype(np.float32)
    det_t /= det_t.sum(1, keepdims=True) * sn * sn
    valid = (rng.rand(B, sn, sn) > 0.4).astype(np.float32) / (sn * sn)
    box = rng.rand(B, sn, sn, 4).astype(np.float32)
    box[..., 2:] = box[..., :2] + 0.05 + 0.5 * box[..., 2:]
    tb = rng.rand(B, sn, sn, 4).astype(np.float32)
    tb[..., 2:] = tb[..., :2] + 0.05 + 0.5 * tb[..., 2:]

    def cxcywh(bx):
        return np.stack([0.5 * (bx[..., 0] + bx[..., 2]), 0.5 * (bx[..., 1] + bx[..., 3]), bx[..., 2] - bx[..., 0],
                         bx[..., 3] - bx[..., 1]], axis=1)

    reg_t = np.concatenate([cxcywh(tb), cxcywh(box)], axis=1).astype(np.float32)
    dc, bc, g = L.detect_cost(out, det_t, valid, reg_t, box, s0, 1.5, 2.0, bounded)
    to = _t(out, True)
    lp = torch.log_softmax(to[:, :s0], 1)
    det_err = -(_t(det_t) * lp).sum(1) / np.log(s0)
    reg = to[:, s0:]
    bt = _t(reg_t)
    if not bounded:
        t = torch.stack([(bt[:, 0] - bt[:, 4]) / bt[:, 6], (bt[:, 1] - bt[:, 5]) / bt[:, 7], torch.log(bt[:, 2] / bt[:, 6]),
                         torch.log(bt[:, 3] / bt[:, 7])], 1)
        d = t - reg
    else:
        sb = _t(box)
        scx, scy = 0.5 * (sb[..., 0] + sb[..., 2]), 0.5 * (sb[..., 1] + sb[..., 3])
        sw, sh = sb[..., 2] - sb[..., 0], sb[..., 3] - sb[..., 1]
        pcx, pcy = reg[:, 0] * sw + scx, reg[:, 1] * sh + scy
        pw, ph = torch.exp(reg[:, 2]) * sw, torch.exp(reg[:, 3]) * sh
        e = 0.001
        dx, dy = bt[:, 0] - pcx, bt[:, 1] - pcy
        cx = torch.where(dx >= 0, 2 * dx / (bt[:, 2] + dx + e), -2 * dx / (bt[:, 2] - dx + e))
        cy = torch.where(dy >= 0, 2 * dy / (bt[:, 3] + dy + e), -2 * dy / (bt[:, 3] - dy + e))
        cw = 1.0 - torch.minimum(bt[:, 2] / (pw + e), pw / (bt[:, 2] + e))
        ch = 1.0 - torch.minimum(bt[:, 3] / (ph + e), ph / (bt[:, 3] + e))
        d = torch.stack([cx, cy, cw, ch], 1)
    sl1 = torch.where(d.abs() < 1, 0.5 * d * d, d.abs() - 0.5)
    bbox_err = 2.0 * _t(valid) * sl1.sum(1)
    c_det, c_bbox = 1.5 * det_err.sum() / B, 2.0 * bbox_err.sum() / B
    (c_det + c_bbox).backward()
    assert abs(dc - float(c_det)) < 1e-5 and abs(bc - float(c_bbox)) < 1e-5
    np.testing.assert_allclose(g, to.grad.numpy(), rtol=1e-4, atol=1e-7)


def test_regression_cost_vs_autograd():
    rng = np.random.RandomState(7)
    x = rng.randn(4, 10, 1, 1).astype(np.float32)
    cls = np.array([1, 3, 3, 9])
    tx = _t(x, True)
    cost = -torch.log_softmax(tx.reshape(4, 10), 1)[torch.arange(4), torch.tensor(cls)].mean()
    cost.backward()
    c, g = L.regression_cost(x, cls)
    assert abs(c - float(cost)) < 1e-6
    np.testing.assert_allclose(g, tx.grad.numpy(), rtol=1e-5, atol=1e-7)


def test_nms_oracle_semantics():
    """denet_detect.cc:73-97: an instance is dropped iff a strictly better one overlaps it by more than the threshold"""
    import ctypes
    lib = OM.oracle_lib()
    B, C1, sn = 1, 3, 2
    det = np.full((B, C1, sn, sn), -10.0, np.float32)
    det[0, 0] = [[-0.1, -0.2], [-0.3, -5.0]]
    fit = det.copy()
    bbox = np.array([[[[0, 0, .5, .5], [0.05, 0, .55, .5]], [[.6, .6, 1, 1], [0, 0, 1, 1]]]], np.float32)
    num = np.array([4], np.int32)
    out = np.zeros((B, 16, 6), np.float32)
    cnt = np.zeros(B, np.int32)
    f = lib.oracle_build_detections_nms
    f.argtypes = [ctypes.c_float, ctypes.c_float, ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_void_p] * 2
    f(0.05, 0.5, 0, det.ctypes.data, fit.ctypes.data, bbox.ctypes.data, num.ctypes.data, B, C1, sn, 16, out.ctypes.data, cnt.ctypes.data)
    # candidates above log(0.05): three; the second overlaps the first (IoU 0.82) and scores lower -> dropped
    assert cnt[0] == 2
    np.testing.assert_allclose(out[0, 0, 0], np.exp(-0.1), rtol=1e-6)
    np.testing.assert_allclose(out[0, 1, 2:], [.6, .6, 1, 1])
