"""Scenarios of the data pipeline, written once and run against EITHER package:
  * `make_dataset_fixtures.py` runs them on the reference's `denet.dataset` (imported in the build container) and
    stores the results in dataset_fixtures.json;
  * `tests/test_dataset.py` runs them on `denet_amd.dataset` and compares.
Everything here is this repository's own test code: it only CALLS the package it is handed. The synthetic dataset
(images, COCO JSON, VOC / ImageNet XML) is generated deterministically into a scratch directory."""
import hashlib
import json
import os
import random

import numpy
from PIL import Image


def _sha(a):
    return hashlib.sha1(numpy.ascontiguousarray(a).tobytes()).hexdigest()


def _img_digest(im):
    if isinstance(im, Image.Image):
        a = numpy.array(im.convert("RGB"), dtype=numpy.uint8)
        return {"size": list(im.size), "sha1": _sha(a), "mean": float(a.mean())}
    a = numpy.asarray(im)
    return {"shape": list(a.shape), "dtype": str(a.dtype), "sha1": _sha(a), "mean": float(a.astype(numpy.float64).mean()),
            "abs": float(numpy.abs(a.astype(numpy.float64)).mean())}


def synth_image(seed, w, h):
    """smooth random blobs + noise: resampling filters give distinguishable results on it"""
    rng = numpy.random.RandomState(seed)
    yy, xx = numpy.mgrid[0:h, 0:w]
    a = numpy.zeros((h, w, 3))
    for _ in range(6):
        cx, cy, r = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(4, max(w, h) / 3)
        col = rng.uniform(0, 255, 3)
        a += numpy.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * r * r))[:, :, None] * col
    a += rng.uniform(0, 40, (h, w, 3))
    return Image.fromarray(numpy.clip(a, 0, 255).astype(numpy.uint8), "RGB")


def synth_boxes(seed, w, h, n, classes):
    rng = numpy.random.RandomState(seed)
    out = []
    for _ in range(n):
        bw, bh = rng.randint(4, max(5, w // 2)), rng.randint(4, max(5, h // 2))
        x0, y0 = rng.randint(0, w - bw), rng.randint(0, h - bh)
        out.append((int(rng.randint(0, classes)), (int(x0), int(y0), int(x0 + bw), int(y0 + bh))))
    return out


SIZES = [(97, 64), (64, 97), (80, 80), (150, 40), (33, 120), (128, 96)]


def build_dataset(root):
    """writes a tiny COCO-, VOC- and ImageNet-shaped tree under `root`; PNG data (lossless) under the expected names"""
    os.makedirs(root, exist_ok=True)
    # ---- COCO
    cats = [{"id": 7, "name": "cat"}, {"id": 3, "name": "dog"}, {"id": 11, "name": "bird"}]
    for split, n0 in (("train2014", 0), ("val2014", 100)):
        os.makedirs(os.path.join(root, "coco", split), exist_ok=True)
        os.makedirs(os.path.join(root, "coco", "annotations"), exist_ok=True)
        images, anns = [], []
        for i, (w, h) in enumerate(SIZES):
            name = "COCO_%s_%06d.jpg" % (split, n0 + i)
            synth_image(n0 + i, w, h).save(os.path.join(root, "coco", split, name), format="PNG")
            images.append({"id": 1000 + n0 + i, "file_name": name, "width": w, "height": h})
            if i != 3:                       # one image without objects
                for cls, (x0, y0, x1, y1) in synth_boxes(n0 + i, w, h, 1 + i % 3, 3):
                    anns.append({"image_id": 1000 + n0 + i, "category_id": cats[cls]["id"],
                                 "bbox": [x0 + 0.5, y0 + 0.25, x1 - x0, y1 - y0]})
        with open(os.path.join(root, "coco", "annotations", "instances_%s.json" % split), "w") as f:
            json.dump({"categories": cats, "images": images, "annotations": anns}, f)
    # ---- VOC
    voc_names = ["aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "diningtable",
                 "dog", "horse", "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor"]
    for year, n0 in (("VOC2007", 200), ("VOC2012", 300)):
        base = os.path.join(root, "voc", year)
        for d in ("JPEGImages", "Annotations", "ImageSets/Main"):
            os.makedirs(os.path.join(base, d), exist_ok=True)
        ids = []
        for i, (w, h) in enumerate(SIZES):
            iid = "%06d" % (n0 + i)
            ids.append(iid)
            synth_image(n0 + i, w, h).save(os.path.join(base, "JPEGImages", iid + ".jpg"), format="PNG")
            objs = ""
            for j, (cls, (x0, y0, x1, y1)) in enumerate(synth_boxes(n0 + i, w, h, 1 + i % 4, 20)):
                objs += ("<object><name>%s</name><difficult>%d</difficult><bndbox><xmin>%d</xmin><ymin>%d</ymin>"
                         "<xmax>%d</xmax><ymax>%d</ymax></bndbox></object>" % (voc_names[cls], int((i + j) % 5 == 0),
                                                                                x0 + 1, y0 + 1, x1 + 1, y1 + 1))
            with open(os.path.join(base, "Annotations", iid + ".xml"), "w") as f:
                f.write("<annotation><size><width>%d</width><height>%d</height></size>%s</annotation>" % (w, h, objs))
        with open(os.path.join(base, "ImageSets/Main/train.txt"), "w") as f:
            f.write("\n".join(ids[:3]) + "\n")
        with open(os.path.join(base, "ImageSets/Main/val.txt"), "w") as f:
            f.write("\n".join(ids[3:5]) + "\n")
        with open(os.path.join(base, "ImageSets/Main/test.txt"), "w") as f:
            f.write("\n".join(ids[5:]) + "\n")
    # ---- ImageNet: <root>/imagenet/train/<wnid>/*.JPEG, <root>/imagenet/bbox/<wnid>/*.xml
    for ci, wnid in enumerate(("n01", "n02", "n03")):
        os.makedirs(os.path.join(root, "imagenet", "train", wnid), exist_ok=True)
        os.makedirs(os.path.join(root, "imagenet", "bbox", wnid), exist_ok=True)
        for i in range(2):
            w, h = SIZES[(ci * 2 + i) % len(SIZES)]
            name = "%s_%d" % (wnid, i)
            synth_image(400 + ci * 2 + i, w, h).save(os.path.join(root, "imagenet", "train", wnid, name + ".JPEG"), format="PNG")
            if i == 0:
                (_, (x0, y0, x1, y1)), = synth_boxes(400 + ci, w, h, 1, 1)
                with open(os.path.join(root, "imagenet", "bbox", wnid, name + ".xml"), "w") as f:
                    f.write("<annotation><size><width>%d</width><height>%d</height></size><object><bndbox><xmin>%d</xmin>"
                            "<ymin>%d</ymin><xmax>%d</xmax><ymax>%d</ymax></bndbox></object></annotation>" % (w, h, x0, y0, x1, y1))
    # ---- directory of class folders
    for c in ("a", "b"):
        os.makedirs(os.path.join(root, "dir", c), exist_ok=True)
        for i in range(2):
            synth_image(500 + i + 10 * (c == "b"), 24, 24).save(os.path.join(root, "dir", c, "im%d.png" % i))


def prepare_imagenet_cache(root, name):
    """a private copy of the ImageNet tree with the `image_list.json` cache and `class_labels.txt` the loader reads"""
    import shutil
    import xml.etree.ElementTree as ET
    src, dst = os.path.join(root, "imagenet", "train"), os.path.join(root, "imagenet", "train_%s" % name)
    shutil.copytree(src, dst)
    images = []
    for wnid in sorted(os.listdir(dst)):
        for f in sorted(os.listdir(os.path.join(dst, wnid))):
            xml_fname = os.path.join(root, "imagenet", "bbox", wnid, os.path.splitext(f)[0] + ".xml")
            bboxs = []
            if os.path.isfile(xml_fname):
                for obj in ET.parse(xml_fname).getroot().iter("object"):
                    b = obj.find("bndbox")
                    bboxs.append({"x0": int(b.find("xmin").text), "x1": int(b.find("xmax").text),
                                  "y0": int(b.find("ymin").text), "y1": int(b.find("ymax").text)})
            images.append({"fname": os.path.join(dst, wnid, f), "bboxs": bboxs})
    with open(os.path.join(dst, "image_list.json"), "w") as f:
        json.dump({"images": images, "version": 1}, f)
    with open(os.path.join(root, "imagenet", "class_labels.txt"), "w") as f:
        f.write("0 n01\n1 n02\n2 n03\n")
    return images


def _geom(t):
    return [float(v) for v in t]


def _meta_digest(meta, root):
    m = {k: v for k, v in meta.items() if k != "image"}
    m["bbox"] = [list(map(float, b)) for b in m["bbox"]]
    m["scale"], m["offset"] = _geom(m["scale"]), _geom(m["offset"])
    m["image_size"] = list(m["image_size"])
    m["image_fname"] = os.path.relpath(meta["image"]["fname"], root)
    return m


def _images_digest(images, root):
    out = []
    for im in images:
        d = {k: v for k, v in im.items() if k != "fname"}
        d["fname"] = os.path.relpath(im["fname"], root)
        d["bboxs"] = [[int(c), [float(v) for v in bb]] for c, bb in im["bboxs"]]
        out.append(d)
    return out


def run(pkg, root):
    """pkg: namespace with augment, image_loader, mscoco, pascal_voc, imagenet, basic, base (the package itself)"""
    A, IL = pkg.augment, pkg.image_loader
    out = {}
    imgs = [synth_image(i, w, h) for i, (w, h) in enumerate(SIZES)]
    lanczos = getattr(A, "LANCZOS", None) or Image.LANCZOS

    # ---- augment: deterministic helpers
    r = []
    for im in imgs:
        for size, mode in ((48, "small"), (48, "large"), (56, "warp"), (200, "small"), (16, "large")):
            o, sx, sy = A.scale(im.copy(), size, mode)
            r.append({"img": _img_digest(o), "s": [sx, sy]})
        for size in (40, 100, 130):
            o, x, y = A.add_border(im.copy(), size)
            r.append({"img": _img_digest(o), "o": [x, y]})
            o, x, y = A.center_crop(im.copy(), size)
            r.append({"img": _img_digest(o), "o": [x, y]})
        big = A.scale(im.copy(), 72, "small")[0]
        crops, ox, oy, mirror = A.multi_crop_mirror(big, 64)
        r.append({"imgs": [_img_digest(c) for c in crops], "ox": list(ox), "oy": list(oy), "mirror": list(mirror)})
    out["augment_fixed"] = r

    # ---- augment: random crops, seeded
    r = []
    for k, im in enumerate(imgs):
        w, h = im.size
        bboxs = [bb for _, bb in synth_boxes(k, w, h, 3, 5)]
        for seed in range(4):
            random.seed(1000 * k + seed)
            o, x, y = A.random_crop(im.copy(), 50)
            e = {"random_crop": {"img": _img_digest(o), "o": [x, y]}}
            o, sx, sy, x, y = A.lenet_crop(im.copy(), 32)
            e["lenet"] = {"img": _img_digest(o), "g": _geom((sx, sy, x, y))}
            o, sx, sy, x, y = A.lenet_crop(im.copy(), 32, max_trials=0)
            e["lenet_fallback"] = {"img": _img_digest(o), "g": _geom((sx, sy, x, y))}
            o, sx, sy, x, y = A.ssd_crop(im.copy(), 40, bboxs)
            e["ssd"] = {"img": _img_digest(o), "g": _geom((sx, sy, x, y))}
            o, sx, sy, x, y = A.denet_crop(im.copy(), 48, bboxs)
            e["denet"] = {"img": _img_digest(o), "g": _geom((sx, sy, x, y))}
            o, sx, sy, x, y = A.denet_crop(im.copy(), 48, bboxs, 0.08, 0.75, 10)
            e["denet_aspect"] = {"img": _img_digest(o), "g": _geom((sx, sy, x, y))}
            o, sx, sy, x, y = A.denet_crop(im.copy(), 48, [], 0.08, 1, 3)
            e["denet_fallback"] = {"img": _img_digest(o), "g": _geom((sx, sy, x, y))}
            e["state"] = random.random()          # both sides must have consumed the same number of draws
            r.append(e)
    out["augment_random"] = r

    # ---- colour
    r = []
    ev = numpy.array([0.2175, 0.0188, 0.0045], dtype=numpy.float32)
    evec = numpy.array([[-0.5675, 0.7192, 0.4009], [-0.5808, -0.0045, -0.8140], [-0.5836, -0.6948, 0.4203]], dtype=numpy.float32)
    for seed in range(6):
        x = A.image_to_array(imgs[seed % len(imgs)])
        random.seed(seed)
        numpy.random.seed(seed)
        p = A.photometric(x.copy())
        c = A.colorspace(x.copy(), ev, evec)
        r.append({"array": _img_digest(x), "photometric": _img_digest(p), "colorspace": _img_digest(c),
                  "values": [float(v) for v in numpy.asarray(p)[:, 3, 5]] + [float(v) for v in c[:, 3, 5]]})
    out["colour"] = r

    # ---- load_sample_proc
    coco = pkg.mscoco.DatasetMSCOCO()
    coco.load(os.path.join(root, "coco"), "mscoco,2014-train,2014-val,crop=48,crop_mode=denet,images_per_subset=5,check_center",
              True, 1)
    out["coco_images"] = _images_digest(coco.images, root)
    out["coco_labels"] = coco.class_labels
    out["coco_categories"] = {str(k): v for k, v in coco.categories.items()}
    out["coco_subsets"] = [coco.subset_num, coco.subset_size, coco.subset_total_size, coco.output_size]
    r = []
    base = {"isTraining": True, "scale": 56, "crop": 48, "rgbMean": [0.485, 0.456, 0.406], "rgbStd": [0.229, 0.224, 0.225],
            "rgbEigenVal": ev.tolist(), "rgbEigenVec": evec.tolist()}
    variants = [
        {"cropMode": "denet", "augmentMirror": True, "checkOnscreen": 0.5},
        {"cropMode": "denet", "augmentMirror": True, "augmentPhoto": True, "augmentColor": True, "subtractMean": True,
         "checkOnscreen": 0.5, "checkCenter": True, "aspectFactor": 0.75},
        {"cropMode": "default", "augmentMirror": True, "scaleMode": "small"},
        {"cropMode": "center", "scaleMode": "large"},
        {"cropMode": "lenet", "augmentMirror": True, "areaMin": 0.2},
        {"cropMode": "ssd", "augmentMirror": True, "checkOnscreen": 0.3},
        {"isTraining": False},
        {"isTraining": False, "multicrop": True, "subtractMean": True},
    ]
    for vi, var in enumerate(variants):
        for ii, image in enumerate(coco.images[:8]):
            args = dict(base)
            args.update(var)
            args.update({"image": image, "seed": 17 * vi + ii})
            data = IL.load_sample_proc(args)
            r.append([{"fname": f, "x": _img_digest(x), "meta": _meta_digest(m, root)} for f, x, m in data])
    out["load_sample_proc"] = r

    # ---- ImageLoader / dataset subsets: the parent's stream hands out the per-image seeds
    random.seed(99)
    coco.shuffle()
    coco.load_from_subset(0)
    first = [{"fname": f, "x": _img_digest(x), "meta": _meta_digest(m, root)} for f, x, m in coco.data]
    coco.load_from_subset(1)
    second = [{"fname": f, "x": _img_digest(x), "meta": _meta_digest(m, root)} for f, x, m in coco.data]
    data_x, data_m, n = coco.export(4)
    out["coco_epoch"] = {"subset0": first, "subset1": second, "export": _img_digest(data_x), "n": n,
                         "export_metas": [_meta_digest(m, root) for m in data_m], "loader_str": str(coco.image_loader),
                         "state": random.random()}

    # ---- COCO results writer
    rng = numpy.random.RandomState(5)
    dets = []
    for _, _, m in coco.data:
        dl = []
        for _ in range(3):
            b = numpy.sort(rng.uniform(-0.1, 1.1, (2, 2)), axis=0).T.reshape(-1)
            dl.append((float(rng.uniform(0, 1)), int(rng.randint(0, 3)), tuple(float(v) for v in b)))
        dets.append({"meta": m, "detections": dl})
    fname = os.path.join(root, "coco_results_%s.json" % pkg.name)
    coco.export_detections(fname, dets)
    out["coco_results"] = json.load(open(fname))

    # ---- VOC
    voc = pkg.pascal_voc.DatasetPascalVOC()
    voc.load(os.path.join(root, "voc"), "voc,2007-trainval,2012-test,crop=40,crop_mode=center,scale=44", False, 1)
    out["voc_images"] = _images_digest(voc.images, root)
    out["voc_subsets"] = [voc.subset_num, voc.subset_size, voc.subset_total_size, voc.output_size]
    random.seed(3)
    voc.load_from_subset(0)
    out["voc_data"] = [{"fname": f, "x": _img_digest(x), "meta": _meta_digest(m, root)} for f, x, m in voc.data]
    # detections: jittered ground truth + noise, so that AP is neither 0 nor 1
    rng = numpy.random.RandomState(8)
    vdets = []
    for _, _, m in voc.data:
        dl = []
        for cls, bb in zip(m["class"], m["bbox"]):
            if rng.uniform() < 0.8:
                j = rng.normal(0, 0.03, 4)
                dl.append((float(rng.uniform(0.3, 1)), int(cls), tuple(float(v + d) for v, d in zip(bb, j))))
            if rng.uniform() < 0.5:
                dl.append((float(rng.uniform(0.0, 0.6)), int(cls), tuple(float(v) for v in bb)))      # duplicate
        for _ in range(2):
            b = numpy.sort(rng.uniform(0, 1, (2, 2)), axis=0).T.reshape(-1)
            dl.append((float(rng.uniform(0, 0.7)), int(rng.randint(0, 20)), tuple(float(v) for v in b)))
        vdets.append({"meta": m, "detections": dl})
    out["voc_precision"] = pkg.voc_precision(vdets)
    # a larger evaluation set built directly as detection records (no images involved): 60 images, all 20 classes
    rng = numpy.random.RandomState(21)
    big = []
    for _ in range(60):
        classes, boxes, diff, dl = [], [], [], []
        for _ in range(rng.randint(0, 5)):
            b = numpy.sort(rng.uniform(0, 1, (2, 2)), axis=0).T.reshape(-1)
            classes.append(int(rng.randint(0, 20)))
            boxes.append(tuple(float(v) for v in b))
            diff.append(bool(rng.uniform() < 0.2))
            for _ in range(rng.randint(0, 3)):
                j = rng.normal(0, 0.04, 4)
                dl.append((float(rng.uniform(0, 1)), classes[-1] if rng.uniform() < 0.85 else int(rng.randint(0, 20)),
                           tuple(float(v + d) for v, d in zip(b, j))))
        for _ in range(rng.randint(0, 4)):
            b = numpy.sort(rng.uniform(0, 1, (2, 2)), axis=0).T.reshape(-1)
            dl.append((float(rng.uniform(0, 0.8)), int(rng.randint(0, 20)), tuple(float(v) for v in b)))
        big.append({"meta": {"class": classes, "bbox": boxes, "image": {"difficult": diff}}, "detections": dl})
    out["voc_precision_big"] = pkg.voc_precision(big)
    odir = os.path.join(root, "voc_results_%s" % pkg.name)
    os.makedirs(odir, exist_ok=True)
    inv = {v: k for k, v in voc.class_labels.items()}
    pkg.pascal_voc.DatasetPascalVOC.export_detections(odir, vdets, 40, 40, inv)
    out["voc_results"] = {f: open(os.path.join(odir, f)).read() for f in sorted(os.listdir(odir))}

    # ---- ImageNet + directory datasets
    inet = pkg.imagenet.DatasetImagenet()
    inet.load(os.path.join(root, "imagenet", "train_%s" % pkg.name), "imagenet,crop=32,scale=36,images_per_subset=4", True, 1)
    out["imagenet_images"] = _images_digest(inet.images, os.path.join(root, "imagenet", "train_%s" % pkg.name))
    out["imagenet_labels"] = inet.class_labels
    out["imagenet_subsets"] = [inet.subset_num, inet.subset_size, inet.subset_total_size]
    random.seed(4)
    inet.load_from_subset(1)
    out["imagenet_data"] = [{"fname": f, "x": _img_digest(x), "meta": _meta_digest(m, os.path.join(root, "imagenet", "train_%s" % pkg.name))}
                            for f, x, m in inet.data]
    dd = pkg.basic.DatasetFromDir()
    dd.load(os.path.join(root, "dir") + "/", "png", False, 1, {"a": 0, "b": 1})
    random.seed(6)
    dd.shuffle()
    x, m, n = dd.export(3)
    out["dir"] = {"fnames": [f for f, _, _ in dd.data], "labels": dd.get_labels(), "shape": list(dd.get_data_shape()),
                  "export": _img_digest(x), "metas": m, "n": n}
    return out
