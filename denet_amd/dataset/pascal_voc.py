"""Pascal VOC detection dataset, result writer and the VOC-2007 11-point AP. Mirrors denet/dataset/pascal_voc.py
(DatasetPascalVOC :13-265): the format string selects image sets ("voc,2007-trainval,2012-trainval,crop=512,..."),
annotations come from the per-image XML (boxes made 0-based, :101-106), `export_detections` writes the
comp4_det_test_<class>.txt files and `get_precision` evaluates AP per class (returned here as well as logged)."""
import math
import os
import random
import sys
import xml.etree.ElementTree as xml

import numpy

from .. import common
from . import DatasetAbstract
from .image_loader import ImageLoader

VOC_CLASSES = ("aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "diningtable",
               "dog", "horse", "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor")


class DatasetPascalVOC(DatasetAbstract):
    def get_data_shape(self):
        return (3, self.output_size, self.output_size)

    def copy(self, copy_data=True):
        r = super().copy(copy_data)
        r.images = self.images
        r.image_loader = self.image_loader
        r.output_size = self.output_size
        return r

    def shuffle(self, mode="random"):
        random.shuffle(self.images)

    def load_from_subset(self, subset):
        lo = subset * self.subset_size
        hi = min((subset + 1) * self.subset_size, self.subset_total_size)
        self.data = self.image_loader.load(self.images[lo:hi])
        self.subset_index = subset

    @staticmethod
    def _read_objects(obj_fname, class_labels):
        bboxs, difficult = [], []
        for obj in xml.parse(obj_fname).getroot().iter("object"):
            box = obj.find("bndbox")
            # VOC pixel coordinates are 1-based
            xyxy = tuple(int(box.find(k).text) - 1 for k in ("xmin", "ymin", "xmax", "ymax"))
            bboxs.append((class_labels[obj.find("name").text], xyxy))
            difficult.append(bool(int(obj.find("difficult").text) > 0))
        return bboxs, difficult

    def load(self, input_dir, data_format, is_training=False, thread_num=1, class_labels=None):
        self.thread_num = thread_num
        format_params = common.get_params_dict(",".join(data_format.split(",")[1:]))
        self.class_labels = {name: i for i, name in enumerate(VOC_CLASSES)}

        files = []
        for year in ("2007", "2012"):
            keys = [s for s in format_params.keys() if str(s).startswith(year)]
            key = keys[0] if len(keys) > 0 else ""
            for image_set in ("train", "val", "test"):     # "2007-trainval" selects train and val
                if image_set in key:
                    with open(os.path.join(input_dir, "VOC%s/ImageSets/Main/%s.txt" % (year, image_set)), "r") as f:
                        files += [os.path.join(input_dir, "VOC%s/JPEGImages/%s.jpg" % (year, line.rstrip()))
                                  for line in f.readlines()]

        self.images = []
        for fname in files:
            anno_dir = os.path.join(os.path.dirname(os.path.dirname(fname)), "Annotations")
            obj_fname = os.path.join(anno_dir, os.path.splitext(os.path.basename(fname))[0] + ".xml")
            bboxs, difficult = [], []
            if os.path.isfile(obj_fname):
                bboxs, difficult = self._read_objects(obj_fname, self.class_labels)
            elif is_training:
                raise Exception("Could not find annotations for training data!")
            self.images.append({"fname": fname, "bboxs": bboxs, "difficult": difficult})
        self.images.sort(key=lambda im: im["fname"])

        self.image_loader = ImageLoader(thread_num, is_training, format_params)
        # ImageNet statistics (natural images), pascal_voc.py:120-125
        self.image_loader.rgb_mean = numpy.array([0.485, 0.456, 0.406], dtype=numpy.float32)
        self.image_loader.rgb_std = numpy.array([0.229, 0.224, 0.225], dtype=numpy.float32)
        self.image_loader.rgb_eigen_val = numpy.array([0.2175, 0.0188, 0.0045], dtype=numpy.float32)
        self.image_loader.rgb_eigen_vec = numpy.array([[-0.5675, 0.7192, 0.4009], [-0.5808, -0.0045, -0.8140],
                                                       [-0.5836, -0.6948, 0.4203]], dtype=numpy.float32)
        self.output_size = self.image_loader.crop
        self.subset_size = min(format_params.get("images_per_subset", 10000), len(self.images))
        self.subset_total_size = len(self.images)
        self.subset_num = min(format_params.get("subset_num", sys.maxsize),
                              int(math.ceil(self.subset_total_size / self.subset_size)))
        self.subset_index = -1

    @staticmethod
    def export_detections(output_dir, detections, width, height, class_labels_inv):
        """comp4_det_test_<class>.txt: "<image id> <score> <x0> <y0> <x1> <y1>", 1-based integer pixels (:136-160)"""
        per_class = {}
        for r in detections:
            meta = r["meta"]
            image_id = os.path.splitext(os.path.basename(meta["image"]["fname"]))[0]
            sx, sy = meta["scale"]
            ox, oy = meta["offset"]
            image_width, image_height = meta["image_size"]
            for pr, cls, bbox in r["detections"]:
                x0 = max(min(int((bbox[0] * width + ox) / sx) + 1, image_width), 1)
                y0 = max(min(int((bbox[1] * height + oy) / sy) + 1, image_height), 1)
                x1 = max(min(int((bbox[2] * width + ox) / sx) + 1, image_width), 1)
                y1 = max(min(int((bbox[3] * height + oy) / sy) + 1, image_height), 1)
                per_class.setdefault(cls, []).append((image_id, pr, x0, y0, x1, y1))
        for cls, rows in per_class.items():
            with open(os.path.join(output_dir, "comp4_det_test_%s.txt" % class_labels_inv[cls]), "w") as f:
                for row in rows:
                    f.write("%s %0.6f %.6f %.6f %.6f %.6f\n" % row)

    @staticmethod
    def get_precision(detections, overlap_threshold=0.5):
        """VOC-2007 11-point interpolated AP per class over all images (:163-265) -> (mean AP, [AP per class]).
        A detection is matched to the ground-truth box of its image with the highest IoU; matches to `difficult`
        boxes count neither way, a second match to the same box is a false positive."""
        gts_cls = [[] for _ in VOC_CLASSES]
        dts_cls = [[] for _ in VOC_CLASSES]
        for image_id, r in enumerate(detections):
            for pr, cls, bbox in r["detections"]:
                dts_cls[cls].append((image_id, pr, bbox))
            meta = r["meta"]
            for cls, bbox, difficult in zip(meta["class"], meta["bbox"], meta["image"]["difficult"]):
                gts_cls[cls].append((image_id, difficult, bbox))

        aps = []
        for gts, dts in zip(gts_cls, dts_cls):
            non_difficult = sum(1 for _, diff, _ in gts if not diff)
            dts.sort(key=lambda d: -d[1])
            tp = numpy.zeros((len(dts),), dtype=numpy.int64)
            fp = numpy.zeros((len(dts),), dtype=numpy.int64)
            found = set()
            for d, (image_id, _, bbox) in enumerate(dts):
                best, best_i = 0, 0
                for gi, (g_image, _, g_bbox) in enumerate(gts):
                    if g_image == image_id:
                        iou = common.overlap_iou(bbox, g_bbox)
                        if iou > best:
                            best, best_i = iou, gi
                if best >= overlap_threshold:
                    if not gts[best_i][1]:
                        if best_i in found:
                            fp[d] = 1
                        else:
                            found.add(best_i)
                            tp[d] = 1
                else:
                    fp[d] = 1
            tp, fp = numpy.cumsum(tp), numpy.cumsum(fp)
            with numpy.errstate(divide="ignore", invalid="ignore"):
                recall = tp / non_difficult
                prec = tp / (tp + fp)
            ap = 0
            for t in numpy.linspace(0.0, 1.0, 11):
                n = (recall >= t)
                ap += (prec[n].max() if n.any() else 0.0) / 11
            aps.append(float(ap))
        return sum(aps) / len(VOC_CLASSES), aps
