"""CPU tests of the data pipeline (denet_amd/dataset) against fixtures produced by running the SAME
scenarios (tests/golden/dataset_scenarios.py) on the reference's `denet.dataset` package
(tests/golden/make_dataset_fixtures.py, build container only). Geometry, metas, random-stream consumption, file
lists and writer outputs must be identical; image content is compared by digest when the Pillow version is the one
the fixtures were made with (the pixel work is Pillow's on both sides) and by mean otherwise."""
import json
import math
import os
import sys
import types

import PIL
import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLDEN)
import dataset_scenarios as S  # noqa: E402

import denet_amd.dataset as D  # noqa: E402
from denet_amd.dataset import augment, basic, image_loader, imagenet, mscoco, pascal_voc  # noqa: E402


def _voc_precision(dets):
    mean, aps = pascal_voc.DatasetPascalVOC.get_precision(dets)
    return {"mean_ap_4dp": "%.4f" % mean, "ap_4dp": ["%.4f" % a for a in aps]}


PKG = types.SimpleNamespace(name="build", base=D, augment=augment, image_loader=image_loader, mscoco=mscoco,
                            pascal_voc=pascal_voc, imagenet=imagenet, basic=basic, voc_precision=_voc_precision)


def _same(a, b, path, same_pillow):
    if isinstance(b, dict):
        assert isinstance(a, dict) and set(a.keys()) == set(b.keys()), (path, sorted(a.keys()), sorted(b.keys()))
        for k in b:
            if k == "sha1" and not same_pillow:
                continue
            if k in ("mean", "abs") and not same_pillow:
                assert abs(a[k] - b[k]) <= 2e-2 * max(1.0, abs(b[k])), (path, k, a[k], b[k])
                continue
            _same(a[k], b[k], path + "/" + str(k), same_pillow)
    elif isinstance(b, (list, tuple)):
        assert isinstance(a, (list, tuple)) and len(a) == len(b), (path, len(a), len(b))
        for i, (x, y) in enumerate(zip(a, b)):
            _same(x, y, path + "[%d]" % i, same_pillow)
    elif isinstance(b, float):
        assert (math.isnan(a) and math.isnan(b)) or a == b, (path, a, b)
    else:
        assert a == b, (path, a, b)


@pytest.fixture(scope="module")
def result(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("dataset"))
    S.build_dataset(root)
    cache = S.prepare_imagenet_cache(root, "build")
    out = json.loads(json.dumps(S.run(PKG, root)))       # tuples -> lists, like the stored fixture
    return root, cache, out


def test_pipeline_matches_imported_reference(result):
    _, _, out = result
    fix = json.load(open(os.path.join(GOLDEN, "dataset_fixtures.json")))
    same_pillow = fix.pop("_pillow") == PIL.__version__
    assert set(out.keys()) == set(fix.keys())
    for k in sorted(fix):
        _same(out[k], fix[k], k, same_pillow)


def test_imagenet_directory_scan_equals_cache(result):
    """the scan branch (imagenet.py:74-103) cannot run in the reference (`os.listdir().sort()` is None); here it must
    find exactly what the image_list.json cache of the scenario holds, and write that cache"""
    root, cache, _ = result
    import shutil
    src = os.path.join(root, "imagenet", "train")
    dst = os.path.join(root, "imagenet", "train_scan")
    shutil.copytree(src, dst)
    ds = imagenet.DatasetImagenet()
    ds.load(dst, "imagenet,crop=32", False, 1)
    got = [(os.path.relpath(im["fname"], dst), [tuple(bb) for _, bb in im["bboxs"]], im["class"]) for im in ds.images]
    ref = [(os.path.relpath(im["fname"], os.path.join(root, "imagenet", "train_build")),
            [(b["x0"], b["y0"], b["x1"], b["y1"]) for b in im["bboxs"]],
            int(os.path.basename(os.path.dirname(im["fname"]))[1:]) - 1) for im in cache]
    assert got == ref
    assert os.path.isfile(os.path.join(dst, "image_list.json"))
    # localisation error: one image hit by its own box, one missed
    m0 = {"class": [1], "bbox": [(0.1, 0.1, 0.6, 0.6)]}
    dets = [{"meta": m0, "detections": [(0.9, 1, (0.12, 0.1, 0.6, 0.62))]},
            {"meta": m0, "detections": [(0.9, 2, (0.1, 0.1, 0.6, 0.6)), (0.5, 1, (0.5, 0.5, 0.9, 0.9))]}]
    assert imagenet.DatasetImagenet.get_localization_error(dets) == 50.0


def test_loader_pool_and_inline_paths_agree(result):
    """per-image seeds come from the parent's stream, so the worker count does not change the data"""
    import random
    root, _, _ = result
    out = []
    for threads in (1, 2):
        ds = D.load(os.path.join(root, "coco"), "mscoco,2014-val,crop=40,crop_mode=denet,augment_photo", True, threads)
        random.seed(12)
        ds.shuffle()
        ds.load_from_subset(0)
        x, metas, n = ds.export(4)
        out.append((x, [(m["scale"], m["offset"], m["mirror"], m["bbox"]) for m in metas], n, random.random()))
        ds.image_loader.close()
    assert out[0][1] == out[1][1] and out[0][2] == out[1][2] == 6 and out[0][3] == out[1][3]
    assert np.array_equal(out[0][0], out[1][0]) and out[0][0].shape == (8, 3, 40, 40)


def test_export_thread_and_abstract_helpers(result):
    root, _, _ = result
    ds = D.load(os.path.join(root, "dir") + "/", "png", False, 1, {"a": 0, "b": 1})
    assert isinstance(ds, basic.DatasetFromDir) and len(ds) == 4 and ds.get_data_type() == "image"
    t = D.DatasetExportThread(None, ds, 0, 2, True)
    t.wait()
    x, metas, n = t.get_export()
    assert x.shape == (4, 3, 24, 24) and n == 4 and t.get_labels() == [0, 0, 1, 1]
    assert float(x.max()) <= 1.0 and float(x.min()) >= 0.0
    folds = ds.split_folds(2)
    assert [len(f) for f in folds] == [2, 2]
    both = folds[0].concatenate(folds[1])
    assert len(both) == 4
    ds.augment_mirror()
    assert len(ds) == 8
    a = np.array(ds.data[0][1])
    assert np.array_equal(np.array(ds.data[4][1]), a[:, ::-1])
    ds.set_data([("f", ds.data[0][1], {"partial": True}), ("g", ds.data[1][1], {"partial": False})])
    assert [f for f, _, _ in ds.data] == ["g"]
    with pytest.raises(TypeError):
        augment.resnet_crop(ds.data[0][1], 16)
