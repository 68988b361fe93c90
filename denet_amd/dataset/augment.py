"""Host-side image augmentation of the data pipeline. Same functions, arguments, return tuples and
call-for-call use of the stdlib / numpy random streams as denet/dataset/augment.py, so a seeded run draws the same
crops as the reference: scale :21-47, add_border :51-61, center_crop :64-72, random_crop :75-83, multi_crop(_mirror)
:86-104, lenet_crop :107-135, ssd_crop :168-225, denet_crop :228-268, photometric :271-285, colorspace :288-293.

Every geometric augmentation is split into a PLAN (the random decisions: which window of the bordered image, which
scale) and a RENDER (PIL crop + resample of that window), so the plan can be computed for a whole batch up front
and the pixels rendered wherever it is cheapest. Pixel work uses the same Pillow calls as the reference, therefore
identical pixels under the same Pillow version. `Image.ANTIALIAS` of the reference is Pillow's LANCZOS filter (the
alias was removed in Pillow 10)."""
import math
import random

import numpy
from PIL import Image

from .. import common

LANCZOS = Image.LANCZOS


def image_to_array(im):
    """PIL image -> float32 (C, H, W) in [0, 1]"""
    if im.mode != "RGB":
        im = im.convert("RGB")
    return numpy.ascontiguousarray((numpy.array(im, dtype=numpy.float32) / 255.0).transpose(2, 0, 1))


def _scaled_size(src, size, scale_mode):
    w, h = src
    if scale_mode == "warp":
        return (size, size)
    if scale_mode not in ("small", "large"):
        raise Exception("Unknown scale mode")
    # "small": the shorter side becomes `size`; "large": the longer side does (the other keeps the aspect, rounded up)
    width_is_ref = (w < h) if scale_mode == "small" else (w > h)
    if width_is_ref:
        return (size, int(math.ceil(size * h / w)))
    return (int(math.ceil(size * w / h)), size)


def scale(im, size, scale_mode="small", interp_mode=LANCZOS):
    """returns (image, scale_x, scale_y). Like the reference this shrinks `im` IN PLACE with thumbnail() first when
    both sides are larger than the target (augment.py:37-39) and resizes to the exact target afterwards"""
    old = im.size
    new = _scaled_size(old, size, scale_mode)
    if im.size[0] > new[0] and im.size[1] > new[1]:
        s = max(new)
        im.thumbnail((s, s), interp_mode)
    out = im if im.size == new else im.resize(new, interp_mode)
    assert out.size == new, "Scaling Error! " + str(im.size) + " != " + str(new)
    return out, new[0] / old[0], new[1] / old[1]


def add_border(im, size):
    """centre `im` on black so that both sides are >= size; returns (image, -pad_x, -pad_y)"""
    if im.size[0] >= size and im.size[1] >= size:
        return im, 0, 0
    full = (max(im.size[0], size), max(im.size[1], size))
    px = int((full[0] - im.size[0]) // 2)
    py = int((full[1] - im.size[1]) // 2)
    canvas = Image.new("RGB", full)
    canvas.paste(im, box=(px, py, px + im.size[0], py + im.size[1]))
    return canvas.copy(), -px, -py


def center_crop(im, size):
    canvas, bx, by = add_border(im, size)
    dx = math.ceil((canvas.size[0] - size) / 2)
    dy = math.ceil((canvas.size[1] - size) / 2)
    return canvas.crop((dx, dy, dx + size, dy + size)), bx + dx, by + dy


def random_crop(im, size):
    canvas, bx, by = add_border(im, size)
    dx = random.randint(0, canvas.size[0] - size)
    dy = random.randint(0, canvas.size[1] - size)
    return canvas.crop((dx, dy, dx + size, dy + size)), bx + dx, by + dy


def multi_crop(im, size):
    """centre + four corner crops (test-time 10-crop, first half)"""
    center, cx, cy = center_crop(im, size)
    w, h = im.size
    corners = [(0, 0), (w - size, 0), (0, h - size), (w - size, h - size)]
    crops = [center] + [im.crop((x, y, x + size, y + size)) for x, y in corners]
    return crops, [cx] + [x for x, _ in corners], [cy] + [y for _, y in corners]


def multi_crop_mirror(im, size):
    crops, ox, oy = multi_crop(im, size)
    crops = crops + [c.transpose(Image.FLIP_LEFT_RIGHT) for c in crops]
    return crops, ox + ox, oy + oy, [False] * 5 + [True] * 5


# ---- plans: the random decisions only -------------------------------------------------------------------------
def plan_lenet_crop(im_size, size, area_min=0.08, aspect_factor=3 / 4, max_trials=10):
    """(x0, y0, x1, y1) window or None (augment.py:109-124)"""
    area = im_size[0] * im_size[1]
    for _ in range(max_trials):
        target_area = random.uniform(area_min, 1.0) * area
        aspect_ratio = random.uniform(aspect_factor, 1.0 / aspect_factor)
        w = int(math.sqrt(target_area * aspect_ratio))
        h = int(math.sqrt(target_area / aspect_ratio))
        if random.random() < 0.5:
            w, h = h, w
        if w <= im_size[0] and h <= im_size[1]:
            x0 = random.randint(0, im_size[0] - w)
            y0 = random.randint(0, im_size[1] - h)
            return (x0, y0, x0 + w, y0 + h)
    return None


def _window_geometry(win, size, border_x, border_y):
    x0, y0, x1, y1 = win
    sx, sy = size / (x1 - x0), size / (y1 - y0)
    return sx, sy, (border_x + x0) * sx, (border_y + y0) * sy


def _norm_box(bbox, sx, sy, ox, oy, size):
    return ((bbox[0] * sx - ox) / size, (bbox[1] * sy - oy) / size, (bbox[2] * sx - ox) / size, (bbox[3] * sy - oy) / size)


def plan_denet_crop(side, border_x, border_y, size, bboxs, area_min=0.08, aspect_factor=1, max_trials=10):
    """window of the square-bordered image that keeps >= 50 % of some object on screen, or None (augment.py:233-262)"""
    for _ in range(max_trials):
        target_area = random.uniform(area_min, 1.0) * side * side
        aspect_ratio = pow(aspect_factor, random.uniform(-1.0, 1.0))
        w = int(math.sqrt(target_area * aspect_ratio))
        h = int(math.sqrt(target_area / aspect_ratio))
        if w > side or h > side:
            continue
        x0 = random.randint(0, side - w)
        y0 = random.randint(0, side - h)
        win = (x0, y0, x0 + w, y0 + h)
        sx, sy, ox, oy = _window_geometry(win, size, border_x, border_y)
        for bbox in bboxs:
            if common.overlap_rel(_norm_box(bbox, sx, sy, ox, oy, size)) >= 0.5:
                return win
    return None


def plan_ssd_crop(im_size, side, border_x, border_y, size, bboxs):
    """SSD-style candidate windows, one per minimum-overlap level that found a match, then a uniform pick
    (augment.py:173-218). Note the reference samples the window inside the UNbordered image extent"""
    crops = [(0, 0, side, side)]
    for min_jaccard_overlap in [0.0, 0.1, 0.3, 0.5, 0.7, 0.9]:
        for _ in range(50):
            s = random.uniform(0.3, 1.0)
            w = int(s * im_size[0])
            h = int(s * im_size[1])
            x0 = random.randint(0, im_size[0] - w)
            y0 = random.randint(0, im_size[1] - h)
            win = (x0, y0, x0 + w, y0 + h)
            sx, sy, ox, oy = _window_geometry(win, size, border_x, border_y)
            if any(common.overlap_iou(_norm_box(b, sx, sy, ox, oy, size)) > min_jaccard_overlap for b in bboxs):
                crops.append(win)
                break
    return random.choice(crops)


# ---- plan + render ----------------------------------------------------------------------------------------------
def lenet_crop(im, size, area_min=0.08, aspect_factor=3 / 4, max_trials=10, scale_mode="small"):
    win = plan_lenet_crop(im.size, size, area_min, aspect_factor, max_trials)
    if win is not None:
        sx, sy = size / (win[2] - win[0]), size / (win[3] - win[1])
        out = im.crop(win).resize((size, size), Image.BICUBIC)
        return out, sx, sy, win[0] * sx, win[1] * sy
    print("warning: using lenet crop fallback")
    im, sx, sy = scale(im, size, scale_mode)
    im, ox, oy = center_crop(im, size)
    return im, sx, sy, ox, oy


def resnet_crop(im, size, *args):
    # the reference's resnet_crop cannot run: its caller passes four arguments to a two-argument function
    # (image_loader.py:56) and the body reads an undefined name (augment.py:164)
    raise TypeError("resnet_crop is unusable in the reference (augment.py:141-165); use crop_mode lenet / denet / ssd")


def ssd_crop(im, size, bboxs):
    side = max(im.size)
    canvas, bx, by = add_border(im, side)
    win = plan_ssd_crop(im.size, side, bx, by, size, bboxs)
    sx, sy, ox, oy = _window_geometry(win, size, bx, by)
    interp_mode = random.choice([Image.NEAREST, Image.BILINEAR, Image.BICUBIC, LANCZOS])
    out, _, _ = scale(canvas.crop(win), size, scale_mode="warp", interp_mode=interp_mode)
    return out, sx, sy, ox, oy


def denet_crop(im, size, bboxs, area_min=0.08, aspect_factor=1, max_trials=10, interp_mode=LANCZOS):
    side = max(im.size)
    canvas, bx, by = add_border(im, side)
    win = plan_denet_crop(side, bx, by, size, bboxs, area_min, aspect_factor, max_trials)
    if win is not None:
        sx, sy, ox, oy = _window_geometry(win, size, bx, by)
        out, _, _ = scale(canvas.crop(win), size, scale_mode="warp", interp_mode=interp_mode)
        return out, sx, sy, ox, oy
    # no window kept an object on screen: the whole bordered image, smaller side scaled to `size`
    out, sx, sy = scale(canvas, size, interp_mode=interp_mode)
    return out, sx, sy, bx * sx, by * sy


# ---- colour -------------------------------------------------------------------------------------------------------
_GREY = (0.299, 0.587, 0.114)


def _grey(im_x):
    return _GREY[0] * im_x[0, :, :] + _GREY[1] * im_x[1, :, :] + _GREY[2] * im_x[2, :, :]


def photometric(im_x, v=0.4):
    """brightness / contrast / saturation jitter in a random order, each with alpha ~ U(1-v, 1+v) (augment.py:271-285)"""
    assert type(im_x) is numpy.ndarray
    for op in random.sample(["contrast", "brightness", "saturation"], 3):
        alpha = random.uniform(1.0 - v, 1.0 + v)
        if op == "brightness":
            im_x = im_x * alpha
        elif op == "contrast":
            im_x = im_x * alpha + (1.0 - alpha) * numpy.mean(_grey(im_x))
        else:
            im_x = im_x * alpha + (1.0 - alpha) * _grey(im_x)[None, :, :]
    return im_x


def colorspace(im_x, rgb_eigen_val, rgb_eigen_vec, v=0.1):
    """PCA lighting noise (Krizhevsky), in place (augment.py:288-293)"""
    assert type(im_x) is numpy.ndarray
    aug = numpy.random.normal(0, v, 3) * rgb_eigen_val
    im_x += numpy.dot(rgb_eigen_vec, aug.T)[:, None, None]
    return im_x
