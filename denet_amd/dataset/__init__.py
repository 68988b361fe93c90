"""Dataset layer of the data pipeline: the container the training / prediction drivers consume.
Mirrors denet/dataset/__init__.py — `DatasetExportThread` (:14-41, loads + exports the next subset while the GPU
trains on the current one), `DatasetAbstract` (:43-366; `data` is a list of (fname, image-or-array, meta), same
stdlib-random use in shuffle() and in the padding of export()) and the `load()` dispatcher (:369-388).
Host-only Python, like the reference: nothing here is on the measured hot path."""
import math
import random
import threading

import numpy
from PIL import Image

from .. import common


class DatasetExportThread(threading.Thread):
    """background load_from_subset() + export() of one subset"""

    def __init__(self, model, data, subset, batch_size, training):
        threading.Thread.__init__(self)
        self.model, self.data, self.subset = model, data, subset
        self.batch_size, self.training = batch_size, training
        self.data_export = None
        self.start()

    def run(self):
        timer = common.Timer()
        self.data.load_from_subset(self.subset)
        timer.mark()
        self.data_export = self.data.export(self.batch_size)
        timer.mark()

    def wait(self):
        self.join()

    def get_export(self):
        return self.data_export

    def get_labels(self):
        return self.data.get_labels()


class DatasetAbstract(object):
    def __init__(self):
        self.data = []                 # [(fname, PIL image | (C,H,W) ndarray, meta)]
        self.src_prefix = ""
        self.class_labels = {}
        self.subset_num = 1
        self.subset_index = -1
        self.subset_total_size = 0
        self.subset_size = 0
        self.thread_num = 1
        self.partial_mode = "ignore"
        self.sample_mode = "default"

    _COPIED = ("src_prefix", "class_labels", "subset_num", "subset_index", "subset_total_size", "subset_size",
               "thread_num", "partial_mode")

    def copy(self, copy_data=True):
        r = type(self)()
        for k in self._COPIED:
            setattr(r, k, getattr(self, k))
        if copy_data and len(self.data) > 0:
            kind = self.get_data_type()
            dup = (lambda d: d.copy()) if kind == "image" else numpy.copy
            r.data = [(fname, dup(d), meta.copy()) for fname, d, meta in self.data]
        return r

    def load(self, src_prefix, data_format, is_training=False, thread_num=1, class_labels=None):
        raise NotImplementedError()

    def load_from_subset(self, index):
        pass

    def get_subset_size(self, subset=0):
        if subset == (self.subset_num - 1):
            return self.subset_total_size % self.subset_size
        return self.subset_size

    def __len__(self):
        return len(self.data)

    def get_total_size(self):
        return self.subset_total_size

    def get_class_num(self):
        return len(self.class_labels)

    def get_labels(self):
        return [meta["image_class"] for _, _, meta in self.data]

    def get_metas(self):
        return [meta for _, _, meta in self.data]

    def get_data_type(self):
        if len(self.data) > 0:
            if isinstance(self.data[0][1], Image.Image):
                return "image"
            if type(self.data[0][1]) is numpy.ndarray:
                return "array"
        raise Exception("Cannot get data type!")

    def get_data_shape(self):
        if len(self.data) == 0:
            raise Exception("Cannot get data shape! Please override get_data_shape() in Dataset class.")
        d = self.data[0][1]
        if self.get_data_type() == "image":
            return (3 if d.mode == "RGB" else 1, d.size[0], d.size[1])
        return d.shape

    def split_folds(self, nfolds):
        folds = [self.copy(False) for _ in range(nfolds)]
        for i, d in enumerate(self.data):
            folds[i % nfolds].data.append(d)
        return folds

    def concatenate(self, data):
        r = self.copy(True)
        r.data += data.data
        return r

    def shuffle(self, mode="random"):
        if mode != "random":
            raise Exception("Unknown shuffle mode:", mode)
        random.shuffle(self.data)

    def set_image_mode(self, mode):
        assert self.get_data_type() == "image"
        self.data = [(fname, im.convert(mode, dither=None), meta) for fname, im, meta in self.data]

    def augment_mirror(self):
        if self.get_data_type() == "image":
            self.data += [(fname, im.transpose(Image.FLIP_LEFT_RIGHT), meta) for fname, im, meta in self.data]
        else:
            self.data += [(fname, d[:, :, ::-1], meta) for fname, d, meta in self.data]

    def set_data(self, data):
        self.data = [(fname, d, meta) for fname, d, meta in data
                     if not (self.partial_mode == "ignore" and meta.get("partial", True))]

    def export(self, batch_size=1, dtype=numpy.float32):
        """-> (float (size, C, H, W), metas, real sample count); the last batch is padded with samples drawn by
        random.randint, one draw per padding slot in order (denet/dataset/__init__.py:349-366)"""
        n = len(self.data)
        size = batch_size * math.ceil(n / batch_size)
        shape = self.get_data_shape()
        data_x = numpy.zeros((size, shape[0], shape[1], shape[2]), dtype=dtype)
        is_image = self.get_data_type() == "image"
        metas = []
        for i in range(size):
            _, d, meta = self.data[i if i < n else random.randint(0, n - 1)]
            if is_image:      # (H, W, C) / 255 -> (C, H, W)
                d = (numpy.array(d, dtype=dtype) / 255.0).transpose(2, 0, 1)
            data_x[i, ...] = d[...]
            metas.append(meta)
        return (data_x, metas, n)


def load(src_prefix, data_format, is_training=False, thread_num=1, class_labels=None):
    """pick the dataset class from the format string: "imagenet,...", "mscoco,...", "voc,...", "npy" / "npz", or a
    file extension for a directory of class folders"""
    from .basic import DatasetFromArray, DatasetFromDir
    from .imagenet import DatasetImagenet
    from .mscoco import DatasetMSCOCO
    from .pascal_voc import DatasetPascalVOC
    if "imagenet" in data_format:
        data = DatasetImagenet()
    elif "mscoco" in data_format:
        data = DatasetMSCOCO()
    elif "voc" in data_format:
        data = DatasetPascalVOC()
    elif data_format in ("npy", "npz"):
        data = DatasetFromArray()
    else:
        data = DatasetFromDir()
    data.load(src_prefix, data_format, is_training, thread_num, class_labels)
    return data
