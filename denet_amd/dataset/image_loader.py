"""Per-image loading + augmentation and its process pool. Mirrors denet/dataset/image_loader.py:
`load_sample_proc(args)` (:11-140; same argument keys, same seeding of both random streams, same meta dictionary)
and `ImageLoader` (:142-206; same format_params keys and defaults, one seed drawn per image from the parent's
stdlib stream so that a run is reproducible for any pool size)."""
import multiprocessing as mp
import os
import random

import numpy
from PIL import Image

from .. import common
from . import augment


def _train_view(im, a, bboxs):
    """training-time scale / crop augmentation -> (image, scale_x, scale_y, offset_x, offset_y)"""
    mode = a.get("cropMode", "default")
    crop = a["crop"]
    if mode == "resnet":
        return augment.resnet_crop(im, crop, a["scale"], 480)
    if mode == "lenet":
        return augment.lenet_crop(im, crop, a.get("areaMin", 0.08), a.get("aspectFactor", 3 / 4), a.get("maxTrials", 10),
                                  a.get("scaleMode", "small"))
    if mode == "denet":
        return augment.denet_crop(im, crop, bboxs, a.get("areaMin", 0.08), a.get("aspectFactor", 3 / 4),
                                  a.get("maxTrials", 10))
    if mode == "ssd":
        return augment.ssd_crop(im, crop, bboxs)
    if mode in ("default", "center"):
        im, sx, sy = augment.scale(im, a["scale"], a.get("scaleMode", "small"))
        im, ox, oy = (augment.random_crop if mode == "default" else augment.center_crop)(im, crop)
        return im, sx, sy, ox, oy
    raise Exception("Unknown crop mode:", mode)


def _sample_meta(image, image_bboxs, crop, sx, sy, ox, oy, mirrored, im_size, check_onscreen, check_center):
    """ground truth of one view in normalised crop coordinates (image_loader.py:108-137)"""
    bboxs, classes = [], []
    for cls, bb in image_bboxs:
        x0, y0 = (bb[0] * sx - ox) / crop, (bb[1] * sy - oy) / crop
        x1, y1 = (bb[2] * sx - ox) / crop, (bb[3] * sy - oy) / crop
        if mirrored:
            x0, x1 = 1.0 - x1, 1.0 - x0
        cx, cy = (x0 + x1) * 0.5, (y0 + y1) * 0.5
        clipped = (common.clip(x0, 0, 1), common.clip(y0, 0, 1), common.clip(x1, 0, 1), common.clip(y1, 0, 1))
        on_screen = common.overlap_rel((x0, y0, x1, y1)) >= check_onscreen
        if on_screen or (check_center and 0.0 <= cx <= 1.0 and 0.0 <= cy <= 1.0):
            bboxs.append(clipped)
            classes.append(cls)
    meta = {"class": classes, "bbox": bboxs, "scale": (sx, sy), "offset": (ox, oy), "mirror": mirrored,
            "image_size": im_size, "image": image}
    if image.get("class", None) is not None:
        meta["image_class"] = image["class"]
    return meta


def load_sample_proc(args):
    """one image -> list of (basename, float32 (3, crop, crop), meta); 1 view, or 10 for test-time multicrop"""
    image = args["image"]
    image_bboxs = image.get("bboxs", [])
    crop = args["crop"]
    subtract_mean = args.get("subtractMean", False)
    if subtract_mean:
        rgb_mean = numpy.array(args["rgbMean"], dtype=numpy.float32)
        rgb_std = numpy.array(args["rgbStd"], dtype=numpy.float32)

    seed = args.get("seed", None)
    random.seed(seed)
    numpy.random.seed(seed)

    im = Image.open(image["fname"])
    im_size = im.size
    if args["isTraining"]:
        im, sx, sy, ox, oy = _train_view(im, args, [bb for _, bb in image_bboxs])
        im_x = augment.image_to_array(im)
        if args.get("augmentPhoto", False):
            im_x = augment.photometric(im_x)
        if args.get("augmentColor", False):
            im_x = augment.colorspace(im_x, numpy.array(args["rgbEigenVal"], dtype=numpy.float32),
                                      numpy.array(args["rgbEigenVec"], dtype=numpy.float32))
        mirrored = bool(args.get("augmentMirror", False) and random.random() >= 0.5)
        if mirrored:
            im_x = im_x[:, :, ::-1]
        views = [(im_x, sx, sy, ox, oy, mirrored)]
    else:
        im, sx, sy = augment.scale(im, args["scale"], args.get("scaleMode", "small"))
        if args.get("multicrop", False):
            crops, oxs, oys, mirrors = augment.multi_crop_mirror(im, crop)
            views = [(augment.image_to_array(c), sx, sy, x, y, m) for c, x, y, m in zip(crops, oxs, oys, mirrors)]
        else:
            im, ox, oy = augment.center_crop(im, crop)
            views = [(augment.image_to_array(im), sx, sy, ox, oy, False)]

    data = []
    for im_x, sx, sy, ox, oy, mirrored in views:
        if subtract_mean:
            im_x = (im_x - rgb_mean[:, None, None]) / rgb_std[:, None, None]
        meta = _sample_meta(image, image_bboxs, crop, sx, sy, ox, oy, mirrored, im_size,
                            args.get("checkOnscreen", 0.0), args.get("checkCenter", False))
        data.append((os.path.basename(image["fname"]), im_x, meta))
    return data


class ImageLoader:
    # format_params key -> (attribute, default); `scale` defaults to the crop size (image_loader.py:145-160)
    PARAMS = (("crop", 224), ("multicrop", False), ("crop_mode", "default"), ("max_trials", 10), ("scale", None),
              ("scale_mode", "small"), ("area_min", 0.08), ("aspect_factor", 0.75), ("subtract_mean", False),
              ("augment_color", False), ("augment_photo", False), ("check_onscreen", 0.5), ("check_center", False))

    def __init__(self, thread_num, is_training, format_params={}):
        for key, default in self.PARAMS:
            setattr(self, key, format_params.get(key, default))
        if self.scale is None:
            self.scale = self.crop
        self.augment_mirror = True
        self.rgb_mean = numpy.zeros(3, dtype=numpy.float32)
        self.rgb_std = numpy.zeros(3, dtype=numpy.float32)
        self.rgb_eigen_val = numpy.zeros(3, dtype=numpy.float32)
        self.rgb_eigen_vec = numpy.zeros((3, 3), dtype=numpy.float32)
        self.is_training = is_training
        self.thread_num = thread_num
        # spawned workers (the reference forks, image_loader.py:172): forking a process that has initialised the GPU
        # runtime and carries its threads deadlocks sooner or later; the workers only import this light module
        self.procs = mp.get_context("spawn").Pool(self.thread_num) if self.thread_num > 1 else None

    def close(self):
        if self.procs is not None:
            self.procs.terminate()
            self.procs.join()
            self.procs = None

    def __str__(self):
        r = "thread_num: %i, is_training: %i, subtract_mean: %i, scale: %i, scale mode: %s, " % (
            self.thread_num, self.is_training, self.subtract_mean, self.scale, self.scale_mode)
        r += "crop: %i, crop_mode: %s, multicrop: %i, onscreen: %.1f, center: %i, " % (
            self.crop, self.crop_mode, self.multicrop, self.check_onscreen, self.check_center)
        r += "area: (%.2f,1.0), aspect: (%.2f,%.2f), max_trials: %i, " % (
            self.area_min, self.aspect_factor, 1.0 / self.aspect_factor, self.max_trials)
        r += "augment - mirror: %i, color: %i, photo: %i" % (self.augment_mirror, self.augment_color, self.augment_photo)
        return r

    def make_args(self, image):
        """the argument dictionary of load_sample_proc; draws the sample's seed from the parent's random stream"""
        return {"image": image, "isTraining": self.is_training, "multicrop": self.multicrop,
                "checkOnscreen": self.check_onscreen, "checkCenter": self.check_center, "scale": self.scale,
                "scaleMode": self.scale_mode, "crop": self.crop, "cropMode": self.crop_mode,
                "subtractMean": self.subtract_mean, "maxTrials": self.max_trials, "areaMin": self.area_min,
                "aspectFactor": self.aspect_factor, "rgbMean": self.rgb_mean.tolist(), "rgbStd": self.rgb_std.tolist(),
                "rgbEigenVec": self.rgb_eigen_vec.tolist(), "rgbEigenVal": self.rgb_eigen_val.tolist(),
                "augmentMirror": self.augment_mirror, "augmentColor": self.augment_color,
                "augmentPhoto": self.augment_photo, "seed": random.randint(0, 1000000)}

    def load(self, images):
        args_list = [self.make_args(image) for image in images]
        if self.procs is None:       # one worker: no pool, same result (each sample re-seeds both streams)
            state, np_state = random.getstate(), numpy.random.get_state()
            results = [load_sample_proc(a) for a in args_list]
            random.setstate(state)
            numpy.random.set_state(np_state)
        else:
            results = self.procs.imap(load_sample_proc, args_list)
        return sum(results, [])
