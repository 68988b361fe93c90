"""Directory-of-class-folders and array datasets. Mirrors denet/dataset/basic.py (DatasetFromDir :13-55,
DatasetFromArray :57-78)."""
import fnmatch
import os

import numpy
from PIL import Image

from . import DatasetAbstract


class DatasetFromDir(DatasetAbstract):
    @staticmethod
    def find_class_labels(src_dir):
        labels = dict()
        for c in os.listdir(src_dir):
            if os.path.isdir(os.path.join(src_dir, c)) and c not in labels:
                labels[c] = len(labels)
        return labels

    @staticmethod
    def find_paths(directory, pattern):
        paths = [os.path.join(root, name) for root, _, files in os.walk(directory, topdown=False, followlinks=True)
                 for name in files]
        return sorted(p for p in paths if fnmatch.fnmatch(p, pattern))

    def load(self, input_dir, ext, is_training=False, thread_num=1, class_labels=None):
        self.class_labels = class_labels if class_labels is not None else DatasetFromDir.find_class_labels(input_dir)
        for c in os.listdir(input_dir):
            cls = self.class_labels[c]
            for f in DatasetFromDir.find_paths(os.path.join(input_dir, c), "*." + ext):
                self.data.append((f.replace(input_dir, ""), Image.open(f).copy(), {"image_class": cls, "partial": False}))
        self.data.sort(key=lambda d: d[2]["image_class"])       # stable: keeps the directory order within a class


class DatasetFromArray(DatasetAbstract):
    def load(self, src_prefix, ext, is_training=False, thread_num=1, class_labels=None):
        data_fname = os.path.join(src_prefix, "_data.npy")
        data = numpy.load(data_fname)
        labels = numpy.load(os.path.join(src_prefix, "_labels.npy"))
        if class_labels is None:
            self.class_labels = {}
            for i in range(int(labels.min()), int(labels.max()) + 1):
                self.class_labels[str(i)] = len(self.class_labels)
        else:
            self.class_labels = class_labels
        # (the reference reads an undefined `data_fname` here, basic.py:76; the file name is what it stands for)
        self.data = [(data_fname, numpy.array(data[i], dtype=numpy.float32, copy=True),
                      {"class": self.class_labels[str(int(labels[i]))], "partial": False}) for i in range(data.shape[0])]
