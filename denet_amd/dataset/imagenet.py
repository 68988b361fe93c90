"""ImageNet (classification + localisation boxes) dataset. Mirrors denet/dataset/imagenet.py (DatasetImagenet
:17-180): class folders under `input_dir`, optional `class_labels.txt` / `image_list.json` next to it, per-image
XML boxes from the sibling `bbox/` directory, subsets loaded through the ImageLoader with the ImageNet colour
statistics; `get_localization_error` is the top-5 localisation error."""
import math
import os
import random
import sys
import xml.etree.ElementTree as xml

import numpy

from .. import common
from . import DatasetAbstract
from .image_loader import ImageLoader


class DatasetImagenet(DatasetAbstract):
    def copy(self, copy_data=True):
        r = super().copy(copy_data)
        r.images = self.images
        r.image_loader = self.image_loader
        return r

    def get_data_shape(self):
        return (3, self.image_loader.crop, self.image_loader.crop)

    def shuffle(self, mode="random"):
        random.shuffle(self.images)

    def load_from_subset(self, subset):
        if self.subset_index == subset:
            return
        lo = subset * self.subset_size
        hi = min((subset + 1) * self.subset_size, self.subset_total_size)
        self.data = self.image_loader.load(self.images[lo:hi])
        self.subset_index = subset

    @staticmethod
    def _scan(input_dir):
        from .basic import DatasetFromDir
        bbox_dir = os.path.join(os.path.dirname(input_dir), "bbox")
        if not os.path.isdir(bbox_dir):
            raise Exception("ERROR: cannot find bbox dir:" + bbox_dir)
        images = []
        for c in sorted(os.listdir(input_dir)):
            for fname in DatasetFromDir.find_paths(os.path.join(input_dir, c), "*.JPEG"):
                obj_fname = os.path.join(bbox_dir, c, os.path.splitext(os.path.basename(fname))[0] + ".xml")
                bboxs = []
                if os.path.isfile(obj_fname):
                    for obj in xml.parse(obj_fname).getroot().iter("object"):
                        box = obj.find("bndbox")
                        bboxs.append({"x0": int(box.find("xmin").text), "x1": int(box.find("xmax").text),
                                      "y0": int(box.find("ymin").text), "y1": int(box.find("ymax").text)})
                images.append({"fname": fname, "bboxs": bboxs})
        return images

    def load(self, input_dir, data_format, is_training=False, thread_num=1, class_labels=None):
        from .basic import DatasetFromDir
        self.input_dir = input_dir[:-1] if input_dir[-1] == "/" else input_dir
        self.data_format = data_format
        self.thread_num = thread_num
        self.class_labels = class_labels
        labels_fname = os.path.join(os.path.dirname(self.input_dir), "class_labels.txt")
        if os.path.isfile(labels_fname) and self.class_labels is None:
            self.class_labels = {}
            with open(labels_fname, "r") as f:
                for line in f.readlines():
                    tokens = line.rstrip("\n").split(" ")
                    self.class_labels[tokens[1]] = int(tokens[0])
        elif self.class_labels is None:
            self.class_labels = DatasetFromDir.find_class_labels(input_dir)

        list_fname = os.path.join(input_dir, "image_list.json")
        if os.path.isfile(list_fname):
            json_data = common.json_from_file(list_fname)
            if json_data.get("version", 0) < 1:
                self.images = [{"fname": fname, "bboxs": []} for fname in json_data["images"]]
            else:
                self.images = json_data["images"]
        else:
            self.images = self._scan(input_dir)
            try:
                common.json_to_file(list_fname, {"images": self.images, "version": 1})
            except Exception:
                pass       # a read-only dataset directory just means the scan is repeated next time

        for image in self.images:
            cls = self.class_labels[os.path.basename(os.path.dirname(image["fname"]))]
            image["class"] = cls
            image["bboxs"] = [(cls, (bb["x0"], bb["y0"], bb["x1"], bb["y1"])) for bb in image["bboxs"]]

        format_params = common.get_params_dict(",".join(data_format.split(",")[1:]))
        self.image_loader = ImageLoader(thread_num, is_training, format_params)
        self.image_loader.rgb_mean = numpy.array([0.485, 0.456, 0.406], dtype=numpy.float32)
        self.image_loader.rgb_std = numpy.array([0.229, 0.224, 0.225], dtype=numpy.float32)
        self.image_loader.rgb_eigen_val = numpy.array([0.2175, 0.0188, 0.0045], dtype=numpy.float32)
        self.image_loader.rgb_eigen_vec = numpy.array([[-0.5675, 0.7192, 0.4009], [-0.5808, -0.0045, -0.8140],
                                                       [-0.5836, -0.6948, 0.4203]], dtype=numpy.float32)
        self.subset_size = format_params.get("images_per_subset", 10000)
        self.use_null_class = format_params.get("null", False)
        self.subset_num = format_params.get("subset_num", sys.maxsize)
        self.bbox_only = format_params.get("bbox_only", False)
        if self.image_loader.is_training and self.bbox_only:
            self.images = [image for image in self.images if len(image["bboxs"]) > 0]
        if self.use_null_class and "null" not in self.class_labels:
            self.class_labels["null"] = len(self.class_labels)
        self.subset_index = -1
        self.subset_total_size = len(self.images)
        self.subset_num = min(self.subset_num, int(math.ceil(self.subset_total_size / self.subset_size)))

    @staticmethod
    def get_localization_error(detections):
        """share of images (in %) without a class-matching detection of IoU > 0.5 among the five first detections
        after the ascending sort by score the reference applies (imagenet.py:160-180; ground-truth boxes are read
        from meta["bbox"], the key the loader writes)"""
        error = 0
        for d in detections:
            meta, dets = d["meta"], d["detections"]
            dets.sort(key=lambda t: t[0])
            hit = any(cls_a == cls_b and common.overlap_iou(bbox_a, bbox_b) > 0.5
                      for _, cls_a, bbox_a in dets[:5] for cls_b, bbox_b in zip(meta["class"], meta["bbox"]))
            error += 0 if hit else 1
        return 100.0 * error / len(detections)
