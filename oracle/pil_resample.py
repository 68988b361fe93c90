"""TEST INFRASTRUCTURE ONLY (like everything under oracle/: never imported by denet_amd/): numpy restatement of Pillow's convolution resampling, the third-party
arithmetic behind the reference's `augment.scale` (denet/dataset/augment.py:21-47 -> PIL.Image.thumbnail / resize).

Pillow is not part of /root/reference; the version in this image is 12.2.0 and the algorithm restated here is its
src/libImaging/Resample.c: `precompute_coeffs` (filter support scaled by max(scale, 1); per output pixel the taps
[xmin, xmax) around centre = in0 + (xx + 0.5) * scale, weights filter((x + xmin - centre + 0.5) / filterscale)
normalised to sum 1), `normalize_coeffs_8bpc` (22-bit fixed point, round half away from zero), and the two passes
`ImagingResampleHorizontal_8bpc` / `...Vertical_8bpc` (accumulator starts at 1 << 21, result clip8(acc >> 22)),
horizontal first, each pass stored as u8. `thumbnail_size` restates Image.thumbnail's aspect-preserving size
(src/PIL/Image.py). Pinned against Pillow itself in tests/test_image_render.py (Pillow is installed wherever the tests
run), so the parity of this file is PINNED."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
LANCZOS, BILINEAR, BICUBIC = 1, 2, 3       # PIL.Image.Resampling values


def _sinc(x):
    return 1.0 if x == 0.0 else math.sin(x * math.pi) / (x * math.pi)


def _lanczos(x):
    return _sinc(x) * _sinc(x / 3) if -3.0 <= x < 3.0 else 0.0


def _bilinear(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


FILTERS = {LANCZOS: (_lanczos, 3.0), BILINEAR: (_bilinear, 1.0), BICUBIC: (_bicubic, 2.0)}


def coeffs(in_size, in0, in1, out_size, flt):
    """-> bounds int32 [out][2] (first tap, tap count), kk int32 [out][ksize] fixed-point taps"""
    f, support0 = FILTERS[int(flt)]
    scale = filterscale = (in1 - in0) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = support0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [f((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x, v in enumerate(w):
            if ww != 0.0:
                v = v / ww
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(cur, out_n, b, kk, axis):
    shape = list(cur.shape)
    shape[axis] = out_n
    out = np.empty(shape, dtype=np.uint8)
    for o in range(out_n):
        lo, n = int(b[o, 0]), int(b[o, 1])
        k = kk[o, :n].astype(np.int64)
        if axis == 1:
            acc = (cur[:, lo:lo + n, :].astype(np.int64) * k[None, :, None]).sum(axis=1)
            out[:, o, :] = np.clip((acc + (1 << (PRECISION_BITS - 1))) >> PRECISION_BITS, 0, 255)
        else:
            acc = (cur[lo:lo + n, :, :].astype(np.int64) * k[:, None, None]).sum(axis=0)
            out[o, :, :] = np.clip((acc + (1 << (PRECISION_BITS - 1))) >> PRECISION_BITS, 0, 255)
    return out


def reduce(a, fx, fy):
    """Image.reduce((fx, fy)) (src/libImaging/Reduce.c, 8-bit): out = ((sum + n//2) * (2**24 // n)) >> 24 over the block's
    n pixels, partial blocks at the right / bottom edge averaged over what they have; size = ceil(in / f)"""
    H, W, _ = a.shape
    oh, ow = (H + fy - 1) // fy, (W + fx - 1) // fx
    out = np.zeros((oh, ow, 3), np.uint8)
    for y in range(oh):
        for x in range(ow):
            blk = a[y * fy:min((y + 1) * fy, H), x * fx:min((x + 1) * fx, W)].astype(np.int64)
            n = blk.shape[0] * blk.shape[1]
            out[y, x] = ((blk.sum(axis=(0, 1)) + n // 2) * ((1 << 24) // n)) >> 24
    return out


def resize(a, out_w, out_h, flt, reducing_gap=None):
    """a: uint8 (H, W, 3) -> uint8 (out_h, out_w, 3): Image.resize((out_w, out_h), flt, reducing_gap=...) with the default
    box. With a reducing gap g, a shrink by int(in / out / g) > 1 per axis is first done by reduce(); the convolution then
    maps the fractional box (0, 0, W / fx, H / fy) - passed to the C code as float32 - to the output"""
    H, W, _ = a.shape
    cur = a
    bw, bh = float(W), float(H)
    if reducing_gap is not None:
        fx = int(W / out_w / reducing_gap) or 1
        fy = int(H / out_h / reducing_gap) or 1
        if fx > 1 or fy > 1:
            cur = reduce(a, fx, fy)
            bw, bh = float(np.float32(W / fx)), float(np.float32(H / fy))
    h2, w2, _ = cur.shape
    if out_w != w2 or bw != w2:
        b, kk = coeffs(w2, 0.0, bw, out_w, flt)
        cur = _pass(cur, out_w, b, kk, 1)
    if out_h != h2 or bh != h2:
        b, kk = coeffs(h2, 0.0, bh, out_h, flt)
        cur = _pass(cur, out_h, b, kk, 0)
    return cur


def thumbnail(a, s, flt):
    """Image.thumbnail((s, s), flt): aspect-preserving shrink with reducing_gap = 2.0"""
    t = thumbnail_size(a.shape[1], a.shape[0], s)
    if t is None or t == (a.shape[1], a.shape[0]):
        return a
    return resize(a, t[0], t[1], flt, reducing_gap=2.0)


def thumbnail_size(w, h, s):
    """size after Image.thumbnail((s, s)): None when the image already fits"""
    def round_aspect(number, key):
        return max(min(math.floor(number), math.ceil(number), key=key), 1)

    x = y = int(math.floor(s))
    if x >= w and y >= h:
        return None
    aspect = w / h
    if x / y >= aspect:
        x = round_aspect(y * aspect, key=lambda n: abs(aspect - n / y))
    else:
        y = round_aspect(x / aspect, key=lambda n: 0 if n == 0 else abs(aspect - x / n))
    return x, y
