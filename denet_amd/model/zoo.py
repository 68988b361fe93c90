"""Benchmark configurations of BASELINE.json, assembled exactly the way the reference's recipe scripts do.

  resnet34        examples/resnet34-imagenet.sh:7   (config 2: backbone, 224x224)
  denet34_skip    papers/dss/denet34.sh:13-15 (skip MODEL_DESC), :87-88 (model-modify sequence): ResNet-34 ->
                  --convert-bn-relu --class-num --image-size 512 512 --layer-remove 3
                  --layer-insert 11:SKIPSRC.X[0] 18:SKIPSRC.X[1] -> --layer-append <skip desc>   (config 3/4)
  cifar3          README.md:52 three-layer CNN (config 1; `P.A` needs an explicit size in this code version)
plus the synthetic MSCOCO-shaped batches of SURVEY.md §8(d).
"""
import numpy

from . import model_cnn, modify

RESNET34_DESC = "C.B[64,7,2] BN A P[3,2,1] nRSN.O[3,64,3] nRSN.O[4,128,3,2] nRSN.O[6,256,3,2] nRSN.O[3,512,3,2] P.A[7] R.TB"
DENET34_SKIP_DESC = ("PI[2] C[256,3] SKIP[1] BNA PI[2] C[128,3] SKIP[0] BNA DNC[96,100] DNS[7,24,0.01,0.1] "
                     "C[1536,1] BNA C.B[1024,1] BNA C.B[768,1] BNA C.B[512,1] BNA DND[0.5,1,1]")
DENET34_STD_DESC = ("PI[2] C.B[256,3] BNA PI[2] C.B[128,3] BNA DNC[96,100] DNS[7,24,0.01,0.1] C.B[1536,1] BNA "
                    "C.B[1024,1] BNA C.B[768,1] BNA C.B[512,1] BNA DND[0.5,1,1]")
# ResNet-101 (bottleneck blocks 3/4/23/3): the desc string is not in the reference tree (the recipe downloads
# models/imagenet/resnet101.mdl.gz); it follows from the nRSN grammar (resnet.py:124-131: num, filters, size, stride,
# bottleneck) and is consistent with the layer indices denet101.sh:84-90 inserts at (7, 12, 24, 37 after --layer-remove 3)
RESNET101_DESC = ("C.B[64,7,2] BN A P[3,2,1] nRSN.O[3,256,3,1,64] nRSN.O[4,512,3,2,128] nRSN.O[23,1024,3,2,256] "
                  "nRSN.O[3,2048,3,2,512] P.A[7] R.TB")
DENET101_WIDE_DESC = ("PI[2] C[1024,3] SKIP[2] BNA PI[2] C[512,3] SKIP[1] BNA PI[2] C[256,3] SKIP[0] BNA SPLIT DNC[128,200] "
                      "DNS[7,48,0.01,0.1] C.B[2048,1] BNA C.B[1536,1] BNA C.B[1024,1] BNA C.B[768,1] BNA DND[0.5,1,1]")
DENET101_SKIP_DESC = ("PI[2] C.B[384,3] SKIP[1] BNA PI[2] C.B[192,3] SKIP[0] BNA DNC[128,50] DNS[7,24,0.01,0.1] C.B[2048,1] BNA "
                      "C.B[1536,1] BNA C.B[1024,1] BNA C.B[768,1] BNA DND[0.5,1,1]")
CIFAR3_DESC = "C[128,3] BN A P[2] C[256,3] BN A P[2] C[512,3] BN A P.A[8] R"


def resnet34(batch_size, image=224, class_num=1000, seed=1):
    numpy.random.seed(seed)
    m = model_cnn.ModelCNN()
    m.batch_size = batch_size
    m.class_num = class_num
    m.build(RESNET34_DESC, (3, image, image), "relu", "half", ["he-backward"])
    return m


def denet34(batch_size, variant="skip", image=512, class_num=80, seed=1, head_desc=None):
    """DeNet-34 <variant> built through the same surgery as papers/dss/denet34.sh:83-96"""
    numpy.random.seed(seed)
    m = model_cnn.ModelCNN()
    m.batch_size = batch_size
    m.class_num = 1000
    m.build(RESNET34_DESC, (3, 224, 224), "relu", "half", ["he-backward"])
    m = modify.modify_bn(m, 1, 0.9, 1e-5)
    m = modify.convert_bn_relu(m)
    m = modify.layer_remove(m, 3)      # the reference applies all flag edits before ONE reload (modify.py:153-159)
    m = modify.set_class_num(m, class_num)
    m = modify.set_image_size(m, image, image)
    if variant == "skip":
        m = modify.layer_insert(m, ["11:SKIPSRC.X[0]", "18:SKIPSRC.X[1]"])
        desc = DENET34_SKIP_DESC
    elif variant == "std":
        desc = DENET34_STD_DESC
    else:
        raise Exception("unknown DeNet-34 variant " + variant)
    m = modify.layer_append(m, head_desc or desc)
    return m


def denet101(batch_size, variant="wide", image=512, class_num=80, seed=1, head_desc=None):
    """DeNet-101 <variant> (BASELINE config 5 is `wide`, B=16): the surgery of papers/dss/denet101.sh:82-90,
    head descs :11-21. `wide` taps three scales (256@128^2 via a plain SKIPSRC, 512@64^2, 1024@32^2) and proposes
    48x48 = 2304 RoIs per image."""
    numpy.random.seed(seed)
    m = model_cnn.ModelCNN()
    m.batch_size = batch_size
    m.class_num = 1000
    m.build(RESNET101_DESC, (3, 224, 224), "relu", "half", ["he-backward"])
    m = modify.modify_bn(m, 1, 0.9, 1e-5)
    m = modify.convert_bn_relu(m)
    m = modify.layer_remove(m, 3)
    m = modify.set_class_num(m, class_num)
    m = modify.set_image_size(m, image, image)
    if variant == "wide":
        m = modify.layer_insert(m, ["7:SKIPSRC[0]", "12:SKIPSRC.X[1]", "24:SPLIT", "37:SKIPSRC.X[2]"])
        desc = DENET101_WIDE_DESC
    elif variant == "skip":
        m = modify.layer_insert(m, ["11:SKIPSRC.X[0]", "18:SKIPSRC.X[1]"])
        desc = DENET101_SKIP_DESC
    else:
        raise Exception("unknown DeNet-101 variant " + variant)
    m = modify.layer_append(m, head_desc or desc)
    return m


def cifar3(batch_size=32, class_num=10, seed=1):
    numpy.random.seed(seed)
    m = model_cnn.ModelCNN()
    m.batch_size = batch_size
    m.class_num = class_num
    m.build(CIFAR3_DESC, (3, 32, 32), "relu", "half", ["he-backward"])
    return m


def warm_corner_head(model, bias=7.5, std=0.3, seed=3):
    """SURVEY.md section 8(d) "warm" corner regime: the DNC corner rows (zero weights / bias +5 as initialised,
    denet_corner.py:41-47, i.e. P(corner) = 4.5e-5 < threshold: no detector RoIs) get random weights and a bias such that
    roughly 1 % of the cells of every corner type fire - the RoI proposal (denet_sparse.cc:337-373, the reference's dominant
    host cost) then has a few hundred corners per type and image to pair"""
    rng = numpy.random.RandomState(seed)
    dnc = [l for l in model.layers if l.type_name == "denet-corner"][0]
    conv = dnc.layers[-1]
    w = conv.omega.get_value().copy()
    w[:dnc.corner_num] = rng.normal(0, std, w[:dnc.corner_num].shape)
    conv.omega.set_value(w)
    b = conv.beta.get_value().copy()
    b[:dnc.corner_num] = bias
    conv.beta.set_value(b)
    return model


def synthetic_batch(batch_size, image=512, class_num=80, seed=1, image_class=False):
    """x ~ U(0,1) (B,3,H,W) float32 NCHW (the value range after /255,
    denet/dataset/__init__.py:359); per image n ~ clip(Poisson(7),1,30) boxes, centre ~ U(0.1,0.9)^2,
    w,h ~ U(0.05,0.6) clipped to [0,1]; class ~ U{0..class_num-1}. Meta keys as image_loader.py:134-136."""
    rs = numpy.random.RandomState(seed)
    x = rs.uniform(0.0, 1.0, (batch_size, 3, image, image)).astype(numpy.float32)
    metas = []
    for b in range(batch_size):
        n = int(numpy.clip(rs.poisson(7), 1, 30))
        bboxs, classes = [], []
        for _ in range(n):
            cx, cy = rs.uniform(0.1, 0.9), rs.uniform(0.1, 0.9)
            w, h = rs.uniform(0.05, 0.6), rs.uniform(0.05, 0.6)
            x0, y0 = max(0.0, cx - 0.5 * w), max(0.0, cy - 0.5 * h)
            x1, y1 = min(1.0, cx + 0.5 * w), min(1.0, cy + 0.5 * h)
            bboxs.append((float(x0), float(y0), float(x1), float(y1)))
            classes.append(int(rs.randint(0, class_num)))
        meta = {"bbox": bboxs, "class": classes, "image_class": int(rs.randint(0, class_num))}
        metas.append(meta)
    return x, metas
