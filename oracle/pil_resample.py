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


def resize(a, out_w, out_h, flt):
    """a: uint8 (H, W, 3) -> uint8 (out_h, out_w, 3), Image.resize((out_w, out_h), flt) with the default box"""
    H, W, _ = a.shape
    cur = a
    if out_w != W:
        b, kk = coeffs(W, 0, W, out_w, flt)
        cur = _pass(cur, out_w, b, kk, 1)
    if out_h != H:
        b, kk = coeffs(H, 0, H, out_h, flt)
        cur = _pass(cur, out_h, b, kk, 0)
    return cur


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
