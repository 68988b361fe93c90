"""ORACLE - TEST INFRASTRUCTURE ONLY (nothing under denet_amd/ may import this file).

A SECOND, deliberately naive statement of the reference's RoI proposal (`build_samples`), written independently of
oracle/build_samples.cc straight from /root/reference/denet/layer/denet_sparse.cc so that a misreading shared by the
C++ oracle and the GPU kernels (both written from one reading) cannot pass unnoticed: tests/test_oracle.py runs the two
against each other on a few hundred random maps. Pure Python: lists, a dict for the de-duplication, `sorted` with
explicit tie groups. Every step cites the lines it restates.

Differences in *form* from the C++ oracle (on purpose): corners and boxes are tuples, ranking is a stable sort on the
score followed by explicit grouping of equal scores (the reference ranks with std::partial_sort on the score alone,
denet_sparse.cc:78,547: the order inside a group of equal scores, and which members of a group straddling a cut survive,
is unspecified there - callers of this module compare tie groups as sets), fp32 arithmetic is spelled out with
numpy.float32 scalars and glibc's expf / logf are called through ctypes (the compiled reference imports exactly those
two symbols from libm).
"""
import ctypes
import ctypes.util
import math

import numpy

_f32 = numpy.float32
_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.expf.restype = ctypes.c_float
_libm.expf.argtypes = [ctypes.c_float]
_libm.logf.restype = ctypes.c_float
_libm.logf.argtypes = [ctypes.c_float]


def _expf(v):
    return _f32(_libm.expf(float(v)))


def _logf(v):
    return _f32(_libm.logf(float(v)))


def find_corners(pr, b, ci, threshold, max_corners, local_max):
    """denet_sparse.cc:506-531. Returns [(x, y, logpr)].
    :507-508  raster scan, y outer / x inner;
    :510-511  keep a cell iff logpr > threshold (strict);
    :514      with local_max > 0 drop it iff logpr < max over the window of :474-487, whose bounds are
              [max(0,y-l), min(H-1,y+l)) x [max(0,x-l), min(W-1,x+l)) - upper bounds EXCLUSIVE, initial value -100000;
    :526-530  more than max_corners: std::partial_sort by logpr descending, keep max_corners - the kept corners are then
              in descending-logpr order, no longer in raster order (this changes the order pairs are generated in)."""
    H, W = pr.shape[3], pr.shape[4]
    out = []
    for y in range(H):
        for x in range(W):
            lp = pr[b, 1, ci, y, x]
            if not lp > threshold:
                continue
            if local_max > 0:
                m = _f32(-100000.0)
                for yy in range(max(0, y - local_max), min(H - 1, y + local_max)):
                    for xx in range(max(0, x - local_max), min(W - 1, x + local_max)):
                        m = max(m, pr[b, 1, ci, yy, xx])
                if lp < m:
                    continue
            out.append((x, y, lp))
    if len(out) > max_corners:
        out = sorted(out, key=lambda c: -float(c[2]))[:max_corners]       # stable: equal logpr stay in raster order
    return out


def score(pr, b, x0, y0, x1, y1):
    """denet_sparse.cc:271-308. fp32 running sums in the order TL(y0,x0), TR(y0,x1), BL(y1,x0), BR(y1,x1) over plane k of
    class 0 ("false", :276-284) and class 1 ("true", :286-294); C == 5 adds the centre plane at ((y0+y1)/2, (x0+x1)/2)
    with integer division (:297-304); pr = 1/(1+exp(|pr_f - pr_t|)) - expf on the fp32 difference, the rest in double,
    rounded to fp32 when stored in SampleType (:306-307, :53-55); the box is (x0/W, y0/H, (x1+1)/W, (y1+1)/H) in double,
    rounded to fp32 (:307)."""
    Cn, H, W = pr.shape[2], pr.shape[3], pr.shape[4]
    cells = [(0, y0, x0), (1, y0, x1), (2, y1, x0), (3, y1, x1)]
    if Cn == 5:
        cells.append((4, (y0 + y1) // 2, (x0 + x1) // 2))
    pf, pt = _f32(0.0), _f32(0.0)
    for k, y, x in cells:
        pf = _f32(pf + pr[b, 0, k, y, x])
    for k, y, x in cells:
        pt = _f32(pt + pr[b, 1, k, y, x])
    absd = _f32(abs(_f32(pf - pt)))
    p = _f32(1.0 / (1.0 + float(_expf(absd))))
    box = (_f32(x0 / W), _f32(y0 / H), _f32((x1 + 1) / W), _f32((y1 + 1) / H))
    return p, box


def search_corners(pr, b, corner_list):
    """denet_sparse.cc:321-471. Returns the candidates in GENERATION order: [(pr, (x0,y0,x1,y1) cells, (fx0,fy0,fx1,fy1))].
    :333-352  for every TL (list order) x every BR (list order): skip unless x1 > x0 and y1 > y0 (:343); 64-bit key
              (x0<<48)|(y0<<32)|(x1<<16)|y1 (:311-318); first seen wins (:346-350);
    :358-373  the same for TR x BL with x1,y0 from TR and x0,y1 from BL;
    :376-466  C == 5: for every centre, in turn TL, TR, BL, BR mirrored through the centre; skipped unless the mirrored
              corner is on the map and the box non-degenerate (:392, :410, :428, :446)."""
    H, W = pr.shape[3], pr.shape[4]
    seen = {}
    out = []

    def consider(x0, y0, x1, y1):
        key = (x0 << 48) | (y0 << 32) | (x1 << 16) | y1
        if key in seen:
            return
        seen[key] = True
        p, box = score(pr, b, x0, y0, x1, y1)
        out.append((p, (x0, y0, x1, y1), box))

    TL, TR, BL, BR = corner_list[0], corner_list[1], corner_list[2], corner_list[3]
    for (x0, y0, _) in TL:
        for (x1, y1, _) in BR:
            if x1 <= x0 or y1 <= y0:
                continue
            consider(x0, y0, x1, y1)
    for (x1, y0, _) in TR:
        for (x0, y1, _) in BL:
            if x1 <= x0 or y1 <= y0:
                continue
            consider(x0, y0, x1, y1)
    if len(corner_list) == 5:
        def ok(x0, y0, x1, y1):
            return not (x0 < 0 or y0 < 0 or x1 >= W or y1 >= H or x1 <= x0 or y1 <= y0)
        for (cx, cy, _) in corner_list[4]:
            for (x0, y0, _) in TL:
                x1, y1 = x0 + 2 * (cx - x0), y0 + 2 * (cy - y0)
                if ok(x0, y0, x1, y1):
                    consider(x0, y0, x1, y1)
            for (x1, y0, _) in TR:
                x0, y1 = x1 - 2 * (x1 - cx), y0 + 2 * (cy - y0)
                if ok(x0, y0, x1, y1):
                    consider(x0, y0, x1, y1)
            for (x0, y1, _) in BL:
                x1, y0 = x0 + 2 * (cx - x0), y1 - 2 * (y1 - cy)
                if ok(x0, y0, x1, y1):
                    consider(x0, y0, x1, y1)
            for (x1, y1, _) in BR:
                x0, y0 = x1 - 2 * (x1 - cx), y1 - 2 * (y1 - cy)
                if ok(x0, y0, x1, y1):
                    consider(x0, y0, x1, y1)
    return out


def _overlap(a, b):
    """SampleType::overlap, denet_sparse.cc:86-91 (fp32)"""
    dx = max(_f32(0.0), _f32(min(a[2], b[2]) - max(a[0], b[0])))
    dy = max(_f32(0.0), _f32(min(a[3], b[3]) - max(a[1], b[1])))
    return _f32(dx * dy)


def _area(a):
    return _f32(_f32(a[2] - a[0]) * _f32(a[3] - a[1]))


def _iou(a, b):
    """SampleType::overlap_iou, denet_sparse.cc:97-101 (fp32)"""
    ai = _overlap(a, b)
    au = _f32(_f32(_area(a) + _area(b)) - ai)
    with numpy.errstate(divide="ignore", invalid="ignore"):
        return _f32(ai / au)


def rank(samples):
    """stable sort by score descending (the reference: std::partial_sort with operator< = pr descending, :78)"""
    return sorted(samples, key=lambda s: -float(s[0]))


def apply_cluster(samples, threshold, input_num, output_num):
    """denet_sparse.cc:165-242. `samples` in generation order; returns the clustered list.
    :172-175  more than input_num: keep the input_num best (ranked);
    :179-209  in that order every sample joins the clusters that contain a member with IoU > threshold (after a cheap
              test against the cluster's bounding box, ClusterType::overlap :134-146): it is added to the LAST
              overlapping cluster in list order (:190-194) and the other overlapping clusters are merged into it (:197-202,
              merge keeps the member vectors in the order target-first); no overlap: a new cluster at the END (:206);
    :212-222  more than output_num clusters: std::list::sort (stable) by member count descending, keep output_num;
    :226      ratio = (output_num - #clusters) / (#samples - #clusters) in double;
    :229-235  every cluster contributes its 1 + floor(size * ratio) best members."""
    if len(samples) > input_num:
        samples = rank(samples)[:input_num]
    clusters = []          # each: {"bbox": [pr, x0, y0, x1, y1], "sv": [list, list, ...]}
    for s in samples:
        sb = s[2]
        hits = []
        for c in clusters:
            cb = c["bbox"]
            if _overlap(sb, cb[1:]) == 0:
                continue
            if any(_iou(sb, m[2]) > threshold for v in c["sv"] for m in v):
                hits.append(c)
        if hits:
            tgt = hits.pop()
            tgt["sv"][0].append(s)
            _bounds(tgt["bbox"], s[0], sb)
            for c in hits:
                _bounds(tgt["bbox"], c["bbox"][0], c["bbox"][1:])
                tgt["sv"].extend(c["sv"])
                clusters = [k for k in clusters if k is not c]
        else:
            clusters.append({"bbox": [s[0], sb[0], sb[1], sb[2], sb[3]], "sv": [[s]]})
    if len(clusters) > output_num:
        clusters = sorted(clusters, key=lambda c: -sum(len(v) for v in c["sv"]))[:output_num]
    ratio = (output_num - len(clusters)) / (len(samples) - len(clusters))
    out = []
    for c in clusters:
        members = [m for v in c["sv"] for m in v]
        n = 1 + int(math.floor(len(members) * ratio))
        out += rank(members)[:n]
    return out


def _bounds(bbox, p, box):
    """SampleType::update_bounds, denet_sparse.cc:80-86"""
    bbox[0] = max(p, bbox[0])
    bbox[1] = min(box[0], bbox[1])
    bbox[2] = min(box[1], bbox[2])
    bbox[3] = max(box[2], bbox[3])
    bbox[4] = max(box[3], bbox[4])


def build_samples(corner_pr, corner_threshold, sample_num, max_corners=1024, local_max=0, cluster_threshold=1.0):
    """run_build_samples for every image, denet_sparse.cc:489-557. corner_pr: fp32 [B,2,C,H,W] log-probabilities.
    Returns per image the ranked list [(pr, (x0,y0,x1,y1) cells, (fx0,fy0,fx1,fy1) fp32)] of at most sample_num^2
    candidates, equal scores in generation order.
    :503      threshold = logf(corner_threshold);
    :541-542  clustering iff #candidates > sample_num^2 and cluster_threshold < 1.0, with 10 sample_num^2 inputs (:497);
    :547-549  rank, keep sample_num^2."""
    pr = numpy.ascontiguousarray(corner_pr, dtype=numpy.float32)
    B, two, Cn, H, W = pr.shape
    assert two == 2 and Cn in (4, 5)
    count = sample_num * sample_num
    thr = _logf(_f32(corner_threshold))
    result = []
    for b in range(B):
        corners = [find_corners(pr, b, ci, thr, max_corners, local_max) for ci in range(Cn)]
        samples = search_corners(pr, b, corners)
        if len(samples) > count and cluster_threshold < 1.0:
            samples = apply_cluster(samples, _f32(cluster_threshold), 10 * count, count)
        result.append(rank(samples)[:count])
    return result


def tie_groups(ranked):
    """[(score, set of cell boxes)] - maximal runs of equal scores of a ranked list"""
    groups = []
    for p, cells, _ in ranked:
        if groups and groups[-1][0] == p:
            groups[-1][1].add(cells)
        else:
            groups.append((p, {cells}))
    return groups
