"""Rendering of augmentation plans: Pillow's resampling restated (oracle/pil_resample.py) and pinned against Pillow
itself, the native coefficient tables, plan == load_sample_proc on the host, and (GPU) the device renderer against the
host path - bit-exact u8 resampling, bit-exact fp32 batch except under the `contrast` jitter (1e-6)."""
import ctypes
import os
import random
import sys

import numpy as np
import pytest
from PIL import Image

from oracle import pil_resample as PR

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLDEN)
import dataset_scenarios as S  # noqa: E402

from denet_amd import lib as dlib  # noqa: E402
from denet_amd.dataset import image_loader as IL, mscoco, plan as P  # noqa: E402

PIL_FILTER = {PR.LANCZOS: Image.LANCZOS, PR.BILINEAR: Image.BILINEAR, PR.BICUBIC: Image.BICUBIC}
EV = [0.2175, 0.0188, 0.0045]
EVEC = [[-0.5675, 0.7192, 0.4009], [-0.5808, -0.0045, -0.8140], [-0.5836, -0.6948, 0.4203]]
BASE = {"isTraining": True, "scale": 56, "crop": 48, "rgbMean": [0.485, 0.456, 0.406], "rgbStd": [0.229, 0.224, 0.225],
        "rgbEigenVal": EV, "rgbEigenVec": EVEC}
VARIANTS = [
    {"cropMode": "denet", "augmentMirror": True, "checkOnscreen": 0.5},
    {"cropMode": "denet", "augmentMirror": True, "augmentPhoto": True, "augmentColor": True, "subtractMean": True,
     "checkOnscreen": 0.5, "checkCenter": True, "aspectFactor": 0.75},
    {"cropMode": "denet", "aspectFactor": 1, "maxTrials": 0},                  # fallback: whole bordered image
    {"cropMode": "denet", "augmentPhoto": True, "crop": 96},                   # up-sampling crops
    {"cropMode": "default", "augmentMirror": True, "scaleMode": "small"},
    {"cropMode": "center", "scaleMode": "large"},
    {"cropMode": "default", "scale": 30, "crop": 48},                          # border after scaling
    {"cropMode": "lenet", "augmentMirror": True, "areaMin": 0.2, "augmentColor": True},
    {"cropMode": "lenet", "maxTrials": 0},
    {"isTraining": False},
    {"isTraining": False, "scale": 30, "subtractMean": True},
    {"cropMode": "ssd", "augmentMirror": True, "checkOnscreen": 0.5},          # one of NEAREST / BILINEAR / BICUBIC / LANCZOS per image
    {"cropMode": "ssd", "augmentPhoto": True, "crop": 96, "subtractMean": True},
    {"cropMode": "ssd", "crop": 20},                                           # strong shrinks (NEAREST: no reduce() pre-pass)
]


def test_oracle_resample_equals_pillow():
    """the restated Resample.c arithmetic against Pillow: every filter, up- and down-sampling, odd sizes"""
    rng = np.random.RandomState(0)
    for trial in range(45):
        W, H = rng.randint(5, 200), rng.randint(5, 200)
        ow, oh = rng.randint(4, 200), rng.randint(4, 200)
        if trial % 5 == 0:
            ow = W                                                              # one axis untouched
        a = rng.randint(0, 256, (H, W, 3)).astype(np.uint8)
        flt = [PR.LANCZOS, PR.BILINEAR, PR.BICUBIC][trial % 3]
        ref = np.array(Image.fromarray(a, "RGB").resize((ow, oh), PIL_FILTER[flt]))
        assert np.array_equal(PR.resize(a, ow, oh, flt), ref), (trial, W, H, ow, oh, flt)
    # Image.thumbnail: aspect-preserving size, and for shrinks >= 4x the reduce() pre-pass + fractional-box convolution
    for W, H, s in [(295, 230, 160), (120, 97, 64), (64, 97, 60), (80, 80, 50), (150, 40, 30), (33, 120, 33), (50, 50, 64),
                    (347, 222, 48), (500, 375, 40), (97, 411, 30), (640, 480, 17), (256, 256, 31)]:
        a = rng.randint(0, 256, (H, W, 3)).astype(np.uint8)
        im = Image.fromarray(a, "RGB")
        t = PR.thumbnail_size(W, H, s)
        assert t == P.thumbnail_size(W, H, s)
        im.thumbnail((s, s), Image.LANCZOS)
        assert im.size == (t if t is not None else (W, H))
        assert np.array_equal(PR.thumbnail(a, s, PR.LANCZOS), np.array(im)), (W, H, s)
    for trial in range(40):
        W, H = rng.randint(3, 90), rng.randint(3, 90)
        fx, fy = int(rng.randint(1, 8)), int(rng.randint(1, 8))
        a = rng.randint(0, 256, (H, W, 3)).astype(np.uint8)
        assert np.array_equal(PR.reduce(a, fx, fy), np.array(Image.fromarray(a, "RGB").reduce((fx, fy)))), (W, H, fx, fy)


def test_native_coefficient_tables_equal_oracle():
    L = dlib.load()
    rng = np.random.RandomState(1)
    for trial in range(60):
        n_in, n_out = int(rng.randint(3, 700)), int(rng.randint(2, 600))
        flt = [PR.LANCZOS, PR.BILINEAR, PR.BICUBIC][trial % 3]
        b_ref, k_ref = PR.coeffs(n_in, 0, n_in, n_out, flt)
        cap = n_out * k_ref.shape[1]
        bounds = np.empty(2 * n_out, dtype=np.int32)
        kk = np.empty(cap, dtype=np.int32)
        ks = L.denet_host_resample_coeffs(n_in, 0.0, float(n_in), n_out, flt, bounds.ctypes.data_as(ctypes.c_void_p),
                                          kk.ctypes.data_as(ctypes.c_void_p), cap)
        assert ks == k_ref.shape[1]
        assert np.array_equal(bounds.reshape(-1, 2), b_ref) and np.array_equal(kk.reshape(n_out, ks), k_ref)
    # filter 4 = NEAREST: a one-tap index table, against Pillow's own resize (ImagingScaleAffine's accumulated double steps)
    for trial in range(80):
        n_in, n_out = int(rng.randint(1, 700)), int(rng.randint(1, 600))
        bounds = np.empty(2 * n_out, dtype=np.int32)
        kk = np.empty(n_out, dtype=np.int32)
        assert L.denet_host_resample_coeffs(n_in, 0.0, float(n_in), n_out, 4, bounds.ctypes.data_as(ctypes.c_void_p),
                                            kk.ctypes.data_as(ctypes.c_void_p), n_out) == 1
        b2 = bounds.reshape(-1, 2)
        assert np.all(b2[:, 1] == 1) and np.all(kk == 1 << 22)
        row = np.arange(n_in, dtype=np.int64)
        a = np.stack([row % 256, (row // 256) % 256, np.zeros_like(row)], -1).astype(np.uint8)[None]       # pixel value = its index
        ref = np.array(Image.fromarray(a, "RGB").resize((n_out, 1), Image.NEAREST))[0]
        assert np.array_equal(a[0][b2[:, 0]], ref), (n_in, n_out)
        col = np.array(Image.fromarray(a.transpose(1, 0, 2).copy(), "RGB").resize((1, n_out), Image.NEAREST))[:, 0]
        assert np.array_equal(a[0][b2[:, 0]], col), (n_in, n_out)
    # argument validation
    assert L.denet_host_resample_coeffs(10, 0.0, 10.0, 5, 0, bounds.ctypes.data_as(ctypes.c_void_p),
                                        kk.ctypes.data_as(ctypes.c_void_p), 4) == -1000


@pytest.fixture(scope="module")
def coco(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("render"))
    S.build_dataset(root)
    ds = mscoco.DatasetMSCOCO()
    ds.load(os.path.join(root, "coco"), "mscoco,2014-train,2014-val,crop=48,crop_mode=denet", True, 1)
    return ds


def _args(var, image, seed):
    a = dict(BASE)
    a.update(var)
    a.update({"image": image, "seed": seed})
    return a


def test_plan_with_pillow_equals_load_sample_proc(coco, capsys):
    """the plan layer draws the same random numbers and renders the same pixels as the (reference-pinned) loader"""
    n = 0
    for vi, var in enumerate(VARIANTS):
        for ii, image in enumerate(coco.images):
            args = _args(var, image, 31 * vi + 7 * ii)
            (f, x, m), = IL.load_sample_proc(args)
            s1, n1 = random.random(), np.random.random()
            pl = P.plan_sample(args)
            s2, n2 = random.random(), np.random.random()
            assert (s1, n1) == (s2, n2), "random streams diverge"
            assert m == pl["meta"]
            assert np.array_equal(x, P.render_pil(pl)), (vi, ii)
            n += 1
    assert n == len(VARIANTS) * 12
    # the ssd variants really drew all four filters
    used = set()
    for vi, var in enumerate(VARIANTS):
        if var.get("cropMode") == "ssd":
            for ii, image in enumerate(coco.images):
                used |= {int(st[-1]) for st in P.plan_sample(_args(var, image, 31 * vi + 7 * ii))["steps"] if st[0] != "crop"}
    assert used == {int(Image.NEAREST), int(Image.BILINEAR), int(Image.BICUBIC), int(Image.LANCZOS)}, used
    with pytest.raises(Exception):
        P.plan_sample(_args({"cropMode": "resnet"}, coco.images[0], 0))


MULTICROP = [{"isTraining": False, "multicrop": True, "scale": 56, "crop": 48, "subtractMean": True},
             {"isTraining": False, "multicrop": True, "scale": 40, "crop": 48},            # corner windows reach outside the scaled image
             {"isTraining": False, "multicrop": True, "scale": 64, "crop": 32, "scaleMode": "large"}]


def test_multicrop_plans_equal_load_sample_proc(coco):
    """test-time 10-crop (augment.multi_crop_mirror): ten plans per image = the loader's ten views, metas and pixels"""
    for vi, var in enumerate(MULTICROP):
        for ii, image in enumerate(coco.images[:6]):
            args = _args(var, image, 5 * vi + ii)
            views = IL.load_sample_proc(args)
            plans = P.plan_views(args)
            assert len(views) == len(plans) == 10
            for (f, x, m), pl in zip(views, plans):
                assert m == pl["meta"]
                assert np.array_equal(x, P.render_pil(pl)), (vi, ii)
    with pytest.raises(Exception):
        P.plan_sample(_args(MULTICROP[0], coco.images[0], 0))


@pytest.mark.gpu
def test_device_render_multicrop_equals_host_path(hip, coco):
    from denet_amd import ops
    from denet_amd.dataset.device_render import DeviceImageLoader
    for vi, var in enumerate(MULTICROP):
        fmt = {"crop": var["crop"], "multicrop": True, "scale": var["scale"], "scale_mode": var.get("scaleMode", "small"),
               "subtract_mean": var.get("subtractMean", False)}
        loader = DeviceImageLoader(2, False, fmt)
        loader.params.rgb_mean = np.array(BASE["rgbMean"], np.float32)      # a bare loader has zero statistics (division by 0)
        loader.params.rgb_std = np.array(BASE["rgbStd"], np.float32)
        try:
            images = coco.images[:5]
            random.seed(3 + vi)
            x, metas = loader.load_batch(images)
            assert tuple(x.shape) == (50, var["crop"], var["crop"], 4) and len(metas) == 50
            got = ops.nhwc_to_nchw(x, 3).cpu().numpy()
            random.seed(3 + vi)
            k = 0
            for image in images:
                for f, ref, m in IL.load_sample_proc(loader.params.make_args(image)):
                    assert m == metas[k]
                    assert np.array_equal(got[k], ref), (vi, k, float(np.abs(got[k] - ref).max()))
                    k += 1
        finally:
            loader.close()


@pytest.mark.gpu
def test_device_render_equals_host_path(hip, coco):
    import torch
    from denet_amd import ops
    from denet_amd.dataset.device_render import DeviceRenderer
    worst = 0.0
    for vi, var in enumerate(VARIANTS):
        crop = var.get("crop", BASE["crop"])
        r = DeviceRenderer(crop, cp=4)
        plans = [P.plan_sample(_args(var, image, 31 * vi + 7 * ii)) for ii, image in enumerate(coco.images)]
        out = r.render_batch(plans)
        assert tuple(out.shape) == (len(plans), crop, crop, 4)
        got = ops.nhwc_to_nchw(out, 3).cpu().numpy()
        assert float(out[..., 3].abs().max()) == 0.0
        for b, pl in enumerate(plans):
            ref = P.render_pil(pl)
            if any(op == 1 for op, _ in pl["photo"]):       # contrast: grey mean from exact sums vs numpy's fp32 mean
                err = float(np.abs(got[b] - ref).max())
                worst = max(worst, err)
                assert err <= 2e-6 * max(1.0, float(np.abs(ref).max())), (vi, b, err)
            else:
                assert np.array_equal(got[b], ref), (vi, b, float(np.abs(got[b] - ref).max()))


@pytest.mark.gpu
def test_device_render_large_shrink_uses_reduce(hip, tmp_path):
    """photo-sized sources (ImageNet-like, 1900x1400 and 1333x2000) scaled to 256 then centre-cropped to 224, and a 4.6x
    shrink of a denet window: Image.thumbnail's reduce() box pre-pass + fractional-box convolution, bit-exact"""
    from denet_amd import ops
    from denet_amd.dataset.device_render import DeviceRenderer
    images = []
    for i, (w, h) in enumerate([(1900, 1400), (1333, 2000), (1024, 1024)]):
        f = str(tmp_path / ("big%d.png" % i))
        S.synth_image(60 + i, w, h).save(f)
        images.append({"fname": f, "bboxs": S.synth_boxes(60 + i, w, h, 4, 10), "id": i, "class": i})
    for var in ({"isTraining": False, "scale": 256, "crop": 224, "subtractMean": True},
                {"cropMode": "default", "scale": 256, "crop": 224, "augmentMirror": True},
                {"cropMode": "denet", "crop": 128, "areaMin": 0.6, "aspectFactor": 1}):
        plans = [P.plan_sample(_args(var, im, 9 + k)) for k, im in enumerate(images)]
        assert any(st[0] == "thumbnail" for pl in plans for st in pl["steps"])
        out = DeviceRenderer(var["crop"]).render_batch(plans)
        got = ops.nhwc_to_nchw(out, 3).cpu().numpy()
        for b, pl in enumerate(plans):
            assert np.array_equal(got[b], P.render_pil(pl)), (var, b)


@pytest.mark.gpu
def test_device_render_headline_geometry(hip, tmp_path):
    """MSCOCO-sized images (640x480 and 480x640, JPEG) to the 512x512 training batch: u8-exact against Pillow for the
    denet crop (Lanczos, thumbnail + resize, bordered canvas), then through a training step of the network input path"""
    from denet_amd import ops
    from denet_amd.dataset.device_render import DeviceRenderer
    images = []
    for i, (w, h) in enumerate([(640, 480), (480, 640), (500, 375), (640, 427)]):
        f = str(tmp_path / ("im%d.jpg" % i))
        S.synth_image(40 + i, w, h).save(f, format="JPEG", quality=92)
        images.append({"fname": f, "bboxs": S.synth_boxes(40 + i, w, h, 5, 80), "id": i})
    var = {"cropMode": "denet", "crop": 512, "augmentMirror": True, "checkOnscreen": 0.5, "aspectFactor": 0.75}
    plans = [P.plan_sample(_args(var, images[i % 4], 100 + i)) for i in range(8)]
    r = DeviceRenderer(512, cp=4)
    out = r.render_batch(plans)
    got = ops.nhwc_to_nchw(out, 3).cpu().numpy()
    for b, pl in enumerate(plans):
        assert np.array_equal(got[b], P.render_pil(pl)), b
    out2 = r.render_batch(plans)                      # buffers reused: same result
    assert bool((out == out2).all())


@pytest.mark.gpu
def test_device_loader_feeds_training_like_host_loader(hip, coco):
    """DeviceImageLoader: same seeds, metas and pixels as ImageLoader under the same parent random state, and a training
    step consumes the device batch directly with the same cost as the host batch"""
    import torch
    from denet_amd import ops
    from denet_amd.dataset.device_render import DeviceImageLoader
    from denet_amd.model import zoo
    fp = {"crop": 128, "crop_mode": "denet", "check_center": True, "augment_photo": False}
    images = [im for im in coco.images if len(im["bboxs"]) > 0][:4]
    random.seed(77)
    host = IL.ImageLoader(1, True, fp).load(images)
    after_host = random.random()
    random.seed(77)
    dev = DeviceImageLoader(2, True, fp)
    x_dev, metas = dev.load_batch(images)
    assert random.random() == after_host, "the two loaders consume the parent stream differently"
    assert [m for _, _, m in host] == metas
    x_host = np.stack([x for _, x, _ in host])
    assert np.array_equal(ops.nhwc_to_nchw(x_dev, 3).cpu().numpy(), x_host)
    B = 4
    model = zoo.denet34(B, "skip", 128, class_num=3, seed=1)
    model.build_train_func("nesterov")
    state = (model.P.clone(), model.M.clone(), model.S.clone())
    random.seed(5)
    c_host, _ = model.train_step(x_host, metas, 0, 0, 0.01, [0.9], 1e-4)
    model.P.copy_(state[0]); model.M.copy_(state[1]); model.S.copy_(state[2])
    random.seed(5)
    c_dev, _ = model.train_step(x_dev, metas, 0, 0, 0.01, [0.9], 1e-4)
    assert c_host == c_dev and np.isfinite(c_dev)


@pytest.mark.gpu
def test_device_epoch_equals_host_epoch(hip, coco):
    """a whole epoch: host loader + DatasetAbstract.export + train_epoch against DeviceImageLoader.iterate +
    train_epoch_device (next batch decoded / rendered on a side stream while the current one trains): identical
    batches, random-stream use, costs and final parameters"""
    import torch
    from denet_amd.dataset.device_render import DeviceImageLoader
    from denet_amd.model import zoo
    fmt = {"crop": 128, "crop_mode": "denet", "check_center": True, "augment_photo": False}
    images = [im for im in coco.images if len(im["bboxs"]) > 0]          # 10 images, batch 4 -> padded last batch
    B = 4
    model = zoo.denet34(B, "skip", 128, class_num=3, seed=1)
    model.build_train_func("nesterov")
    state = (model.P.clone(), model.M.clone(), model.S.clone())

    class HostSet(mscoco.DatasetMSCOCO):
        pass
    random.seed(21)
    host = HostSet()
    host.output_size = 128
    host.data = IL.ImageLoader(1, True, fmt).load(images)
    c_host = model.train_epoch(host, 0, 0.01, [0.9], 1e-4)
    after_host = random.random()
    p_host = model.P.clone()

    model.P.copy_(state[0]); model.M.copy_(state[1]); model.S.copy_(state[2])
    model.iteration = 0
    random.seed(21)
    c_dev = model.train_epoch_device(DeviceImageLoader(2, True, fmt), images, 0, 0.01, [0.9], 1e-4)
    assert random.random() == after_host
    assert c_dev == c_host and np.isfinite(c_dev)
    assert torch.equal(model.P, p_host)


@pytest.mark.gpu
def test_device_loader_process_decode(hip, coco):
    """decode="process": workers write decoded images into shared memory; same batch as the thread mode"""
    from denet_amd.dataset.device_render import DeviceImageLoader
    fp = {"crop": 64, "crop_mode": "denet", "augment_photo": True}
    images = coco.images[:6]
    random.seed(3)
    a = DeviceImageLoader(2, True, fp)
    xa, ma = a.load_batch(images)
    random.seed(3)
    b = DeviceImageLoader(2, True, fp, decode="process")
    try:
        xb, mb = b.load_batch(images)
        xb2, _ = b.load_batch(images[:3])          # second staging slot
    finally:
        b.close()
    assert ma == mb and bool((xa == xb).all()) and tuple(xb2.shape) == (3, 64, 64, 4)
