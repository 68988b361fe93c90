"""MSCOCO detection dataset + result writer. Mirrors denet/dataset/mscoco.py (DatasetMSCOCO :14-169): the format
string selects the annotation files ("mscoco,2014-train,2014-val,crop=512,crop_mode=denet,..."), class indices are
assigned in order of first appearance of the category names, boxes become (x0, y0, x1, y1) in pixels, images are
loaded subset by subset through the ImageLoader, and `export_detections` writes the COCO results JSON."""
import json
import math
import os
import random
import sys

from .. import common
from . import DatasetAbstract
from .image_loader import ImageLoader

DATA_TYPES = (("2014-train", "train2014"), ("2014-val", "val2014"), ("2014-test", "test2014"),
              ("2015-test", "test2015"), ("2015-test-dev", "test-dev2015"))


class DatasetMSCOCO(DatasetAbstract):
    def get_data_shape(self):
        return (3, self.output_size, self.output_size)

    def copy(self, copy_data=True):
        r = super().copy(copy_data)
        r.images = self.images
        r.image_loader = self.image_loader
        return r

    def shuffle(self, mode="random"):
        random.shuffle(self.images)

    def load_from_subset(self, subset):
        if self.subset_index == subset:
            return
        lo = subset * self.subset_size
        hi = min((subset + 1) * self.subset_size, self.subset_total_size)
        self.data = self.image_loader.load(self.images[lo:hi])
        self.subset_index = subset

    def load(self, input_dir, data_format, is_training=False, thread_num=1, class_labels=None):
        self.data = []
        self.thread_num = thread_num
        format_params = common.get_params_dict(",".join(data_format.split(",")[1:]))
        self.data_types = [name for key, name in DATA_TYPES if format_params.get(key, False)]
        if len(self.data_types) == 0:
            raise Exception("please specify mscoco subset")

        self.images = []
        self.class_labels = {}
        self.categories = None
        for data_type in self.data_types:
            kind = "image_info" if "test" in data_type else "instances"
            json_data = common.json_from_file(os.path.join(input_dir, "annotations/%s_%s.json" % (kind, data_type)))
            categories = {}
            for cat in json_data["categories"]:
                categories[cat["id"]] = cat["name"]
                if cat["name"] not in self.class_labels:
                    self.class_labels[cat["name"]] = len(self.class_labels)
            assert (self.categories is None) or (self.categories == categories)
            self.categories = categories

            bboxs = {}
            for ann in json_data.get("annotations", []):
                x, y, w, h = ann["bbox"]
                cls = self.class_labels[self.categories[ann["category_id"]]]
                bboxs.setdefault(ann["image_id"], []).append((cls, (x, y, x + w, y + h)))

            folder = "test2015" if data_type == "test-dev2015" else data_type
            for image in json_data["images"]:
                self.images.append({"fname": os.path.join(input_dir, folder, image["file_name"]),
                                    "bboxs": bboxs.get(image["id"], []), "id": image["id"]})

        self.image_loader = ImageLoader(thread_num, is_training, format_params)
        self.output_size = self.image_loader.crop
        self.images_per_subset = format_params.get("images_per_subset", 10000)
        self.subset_total_size = len(self.images)       # counted before the bbox_only filter, as in the reference
        self.subset_num = min(format_params.get("subset_num", sys.maxsize),
                              int(math.ceil(self.subset_total_size / self.images_per_subset)))
        self.subset_index = -1
        self.subset_size = self.images_per_subset
        self.bbox_only = format_params.get("bbox_only", False)
        if self.image_loader.is_training and self.bbox_only:
            self.images = [image for image in self.images if len(image["bboxs"]) > 0]

    def export_detections(self, output_fname, detection_list):
        """COCO results file: boxes back in original-image pixels, 1-based, [x, y, w, h] rounded to 0.1 (:140-169)"""
        cat_of_label = {self.class_labels[name]: cat_id for cat_id, name in self.categories.items()}
        results = []
        for d in detection_list:
            meta = d["meta"]
            sx, sy = meta["scale"]
            ox, oy = meta["offset"]
            width, height = meta["image_size"]
            dets = d["detections"]
            dets.sort(key=lambda t: -t[0])
            for pr, cls, bbox in dets:
                x0 = max(min((bbox[0] * self.output_size + ox) / sx + 1, width), 1)
                y0 = max(min((bbox[1] * self.output_size + oy) / sy + 1, height), 1)
                x1 = max(min((bbox[2] * self.output_size + ox) / sx + 1, width), 1)
                y1 = max(min((bbox[3] * self.output_size + oy) / sy + 1, height), 1)
                results.append({"image_id": meta["image"]["id"], "category_id": cat_of_label[cls],
                                "bbox": [round(x0, 1), round(y0, 1), round(x1 - x0, 1), round(y1 - y0, 1)],
                                "score": round(pr, 6)})
        with open(output_fname, "w") as f:
            json.dump(results, f)
