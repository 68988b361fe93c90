"""Augmentation PLANS: everything `load_sample_proc` decides about one image - the random window, the resampling
stages, the colour jitter factors, the mirror flag, the ground truth in crop coordinates - without touching a pixel.

A plan is a list of geometric steps plus colour parameters; it is executed either by Pillow on the host
(`render_pil`, identical to denet_amd/dataset/image_loader.load_sample_proc, which is itself pinned against the
reference) or on the GPU (denet_amd/dataset/device_render.py -> csrc/image.hip). The random streams are consumed
call for call like denet/dataset/image_loader.py:44-105 and the augment functions it calls, so a seeded run gives the
same crops on either executor.

steps:
  ("crop", x0, y0, x1, y1, px, py, cw, ch)   window of a black cw x ch canvas with the image pasted at (px, py)
  ("thumbnail", s, filter)                    Image.thumbnail((s, s), filter)
  ("resize", w, h, filter)                    Image.resize((w, h), filter)
"""
import math
import random

import numpy
from PIL import Image

from . import augment
from .image_loader import _sample_meta

NAME_OP = {"brightness": 0, "contrast": 1, "saturation": 2}


def thumbnail_size(w, h, s):
    """size after Image.thumbnail((s, s)) (Pillow's aspect-preserving rule), None if the image already fits"""
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


def scale_steps(size_in, size, scale_mode="small", interp_mode=augment.LANCZOS):
    """steps of augment.scale -> (steps, (w, h) after, scale_x, scale_y)"""
    new = augment._scaled_size(size_in, size, scale_mode)
    steps, cur = [], tuple(size_in)
    if cur[0] > new[0] and cur[1] > new[1]:
        s = max(new)
        t = thumbnail_size(cur[0], cur[1], s)
        if t is not None and t != cur:
            steps.append(("thumbnail", s, interp_mode))
            cur = t
    if cur != new:
        steps.append(("resize", new[0], new[1], interp_mode))
    return steps, new, new[0] / size_in[0], new[1] / size_in[1]


def border_geometry(size_in, size):
    """add_border: (canvas w, canvas h, paste x, paste y)"""
    w, h = size_in
    if w >= size and h >= size:
        return w, h, 0, 0
    cw, ch = max(w, size), max(h, size)
    return cw, ch, int((cw - w) // 2), int((ch - h) // 2)


def _crop_step(win, canvas):
    cw, ch, px, py = canvas
    return ("crop", int(win[0]), int(win[1]), int(win[2]), int(win[3]), px, py, cw, ch)


def plan_train_view(im_size, a, bboxs):
    """steps + (scale_x, scale_y, offset_x, offset_y) of one training view; same draws as image_loader._train_view"""
    mode = a.get("cropMode", "default")
    crop = a["crop"]
    if mode == "denet":
        side = max(im_size)
        canvas = border_geometry(im_size, side)
        bx, by = -canvas[2], -canvas[3]
        win = augment.plan_denet_crop(side, bx, by, crop, bboxs, a.get("areaMin", 0.08), a.get("aspectFactor", 3 / 4),
                                      a.get("maxTrials", 10))
        if win is not None:
            sx, sy, ox, oy = augment._window_geometry(win, crop, bx, by)
            steps, _, _, _ = scale_steps((win[2] - win[0], win[3] - win[1]), crop, "warp")
            return [_crop_step(win, canvas)] + steps, sx, sy, ox, oy
        steps, _, sx, sy = scale_steps((canvas[0], canvas[1]), crop, "small")
        return [_crop_step((0, 0, canvas[0], canvas[1]), canvas)] + steps, sx, sy, bx * sx, by * sy
    if mode == "lenet":
        win = augment.plan_lenet_crop(im_size, crop, a.get("areaMin", 0.08), a.get("aspectFactor", 3 / 4),
                                      a.get("maxTrials", 10))
        if win is not None:
            sx, sy = crop / (win[2] - win[0]), crop / (win[3] - win[1])
            return ([_crop_step(win, (im_size[0], im_size[1], 0, 0)), ("resize", crop, crop, Image.BICUBIC)],
                    sx, sy, win[0] * sx, win[1] * sy)
        print("warning: using lenet crop fallback")
        steps, new, sx, sy = scale_steps(im_size, crop, a.get("scaleMode", "small"))
        c, ox, oy = center_crop_step(new, crop)
        return steps + [c], sx, sy, ox, oy
    if mode in ("default", "center"):
        steps, new, sx, sy = scale_steps(im_size, a["scale"], a.get("scaleMode", "small"))
        c, ox, oy = (random_crop_step if mode == "default" else center_crop_step)(new, crop)
        return steps + [c], sx, sy, ox, oy
    if mode == "ssd":
        # augment.ssd_crop draw for draw: the candidate windows, the pick, then one of four resampling filters
        side = max(im_size)
        canvas = border_geometry(im_size, side)
        bx, by = -canvas[2], -canvas[3]
        win = augment.plan_ssd_crop(im_size, side, bx, by, crop, bboxs)
        sx, sy, ox, oy = augment._window_geometry(win, crop, bx, by)
        interp_mode = random.choice([Image.NEAREST, Image.BILINEAR, Image.BICUBIC, augment.LANCZOS])
        steps, _, _, _ = scale_steps((win[2] - win[0], win[3] - win[1]), crop, "warp", interp_mode)
        return [_crop_step(win, canvas)] + steps, sx, sy, ox, oy
    raise Exception("crop mode '%s' has no plan (resnet is unusable in the reference)" % mode)


def center_crop_step(size_in, size):
    canvas = border_geometry(size_in, size)
    dx = math.ceil((canvas[0] - size) / 2)
    dy = math.ceil((canvas[1] - size) / 2)
    return _crop_step((dx, dy, dx + size, dy + size), canvas), -canvas[2] + dx, -canvas[3] + dy


def random_crop_step(size_in, size):
    canvas = border_geometry(size_in, size)
    dx = random.randint(0, canvas[0] - size)
    dy = random.randint(0, canvas[1] - size)
    return _crop_step((dx, dy, dx + size, dy + size), canvas), -canvas[2] + dx, -canvas[3] + dy


def plan_sample(args, im_size=None):
    """the plan of one image (single view): same argument dictionary and random draws as load_sample_proc.
    Returns {"fname", "steps", "photo": [(op id, alpha)], "noise": 3 doubles | None, "mean_std": 6 floats | None,
    "mirror", "meta"}; `im_size` may be given when the image has been opened already"""
    image = args["image"]
    image_bboxs = image.get("bboxs", [])
    crop = args["crop"]
    seed = args.get("seed", None)
    random.seed(seed)
    numpy.random.seed(seed)
    if im_size is None:
        with Image.open(image["fname"]) as im:
            im_size = im.size
    photo, noise, mirrored = [], None, False
    if args["isTraining"]:
        steps, sx, sy, ox, oy = plan_train_view(im_size, args, [bb for _, bb in image_bboxs])
        if args.get("augmentPhoto", False):
            for name in random.sample(["contrast", "brightness", "saturation"], 3):
                photo.append((NAME_OP[name], random.uniform(1.0 - 0.4, 1.0 + 0.4)))
        if args.get("augmentColor", False):
            aug = numpy.random.normal(0, 0.1, 3) * numpy.array(args["rgbEigenVal"], dtype=numpy.float32)
            noise = numpy.dot(numpy.array(args["rgbEigenVec"], dtype=numpy.float32), aug.T)
        mirrored = bool(args.get("augmentMirror", False) and random.random() >= 0.5)
    else:
        if args.get("multicrop", False):
            raise Exception("multicrop has ten views per image: use plan_views")
        steps, new, sx, sy = scale_steps(im_size, args["scale"], args.get("scaleMode", "small"))
        c, ox, oy = center_crop_step(new, crop)
        steps = steps + [c]
    mean_std = None
    if args.get("subtractMean", False):
        mean_std = [float(v) for v in args["rgbMean"]] + [float(v) for v in args["rgbStd"]]
    meta = _sample_meta(image, image_bboxs, crop, sx, sy, ox, oy, mirrored, im_size, args.get("checkOnscreen", 0.0),
                        args.get("checkCenter", False))
    return {"fname": image["fname"], "steps": steps, "photo": photo, "noise": noise, "mean_std": mean_std,
            "mirror": mirrored, "meta": meta}


def plan_views(args, im_size=None):
    """the plans of ALL views of one image, in load_sample_proc's order: one (plan_sample), or - test-time `multicrop`,
    augment.multi_crop_mirror - ten: centre and four corner crops of the scaled image, then the same five mirrored. Every
    view is a complete plan (scaling steps + its crop window), so either executor renders it like any other."""
    if args["isTraining"] or not args.get("multicrop", False):
        return [plan_sample(args, im_size)]
    image = args["image"]
    image_bboxs = image.get("bboxs", [])
    crop = args["crop"]
    seed = args.get("seed", None)
    random.seed(seed)
    numpy.random.seed(seed)
    if im_size is None:
        with Image.open(image["fname"]) as im:
            im_size = im.size
    steps, new, sx, sy = scale_steps(im_size, args["scale"], args.get("scaleMode", "small"))
    centre, cx, cy = center_crop_step(new, crop)
    w, h = new
    plain = (new[0], new[1], 0, 0)            # the corner crops are taken from the scaled image itself (no border canvas)
    corners = [(0, 0), (w - crop, 0), (0, h - crop), (w - crop, h - crop)]
    windows = [(centre, cx, cy)] + [(_crop_step((x, y, x + crop, y + crop), plain), x, y) for x, y in corners]
    mean_std = None
    if args.get("subtractMean", False):
        mean_std = [float(v) for v in args["rgbMean"]] + [float(v) for v in args["rgbStd"]]
    plans = []
    for mirrored in (False, True):
        for c, ox, oy in windows:
            meta = _sample_meta(image, image_bboxs, crop, sx, sy, ox, oy, mirrored, im_size, args.get("checkOnscreen", 0.0),
                                args.get("checkCenter", False))
            plans.append({"fname": image["fname"], "steps": steps + [c], "photo": [], "noise": None, "mean_std": mean_std,
                          "mirror": mirrored, "meta": meta})
    return plans


# ---- host executor (Pillow): the reference's pixel path ---------------------------------------------------------------
def render_steps_pil(im, steps):
    for st in steps:
        if st[0] == "crop":
            _, x0, y0, x1, y1, px, py, cw, ch = st
            if (cw, ch) != im.size or px or py:
                canvas = Image.new("RGB", (cw, ch))
                canvas.paste(im, box=(px, py, px + im.size[0], py + im.size[1]))
                im = canvas.copy()
            im = im.crop((x0, y0, x1, y1))
        elif st[0] == "thumbnail":
            im = im.copy() if st is steps[0] else im
            im.thumbnail((st[1], st[1]), st[2])
        elif st[0] == "resize":
            im = im.resize((st[1], st[2]), st[3])
        else:
            raise Exception("unknown plan step " + str(st[0]))
    return im


def render_pil(plan):
    """-> float32 (3, crop, crop) exactly like load_sample_proc's single view"""
    im = render_steps_pil(Image.open(plan["fname"]), plan["steps"])
    im_x = augment.image_to_array(im)
    for op, alpha in plan["photo"]:
        if op == 0:
            im_x = im_x * alpha
        elif op == 1:
            im_x = im_x * alpha + (1.0 - alpha) * numpy.mean(augment._grey(im_x))
        else:
            im_x = im_x * alpha + (1.0 - alpha) * augment._grey(im_x)[None, :, :]
    if plan["noise"] is not None:
        im_x += plan["noise"][:, None, None]
    if plan["mirror"]:
        im_x = im_x[:, :, ::-1]
    if plan["mean_std"] is not None:
        ms = numpy.array(plan["mean_std"], dtype=numpy.float32)
        im_x = (im_x - ms[:3, None, None]) / ms[3:, None, None]
    return im_x
