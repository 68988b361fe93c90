"""Generates tests/golden/dataset_fixtures.json by running tests/golden/dataset_scenarios.py against the REFERENCE's
`denet.dataset` package, imported from /root/reference in the build container (it needs only Pillow / numpy):

    python tests/golden/make_dataset_fixtures.py        # never runs on the GPU box

Environment notes (no reference file is modified or copied):
  * the reference was written for Pillow < 10 and names the Lanczos filter `Image.ANTIALIAS`; that alias no longer
    exists, so it is re-created on the imported PIL module before the reference is imported;
  * the reference's `get_precision` only logs its result: the numbers are captured from its log calls;
  * `os.listdir(...).sort()` in the reference's ImageNet scan returns None (imagenet.py:82), so that branch cannot run;
    the scenario therefore supplies the `image_list.json` cache the reference reads instead (imagenet.py:64-72).
Only inputs and the reference's outputs are written."""
import json
import os
import re
import sys
import tempfile
import types

import PIL
from PIL import Image

Image.ANTIALIAS = Image.LANCZOS
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import denet.common.logging as ref_logging  # noqa: E402
import denet.dataset as ref_dataset  # noqa: E402
import denet.dataset.augment as ref_augment  # noqa: E402
import denet.dataset.basic as ref_basic  # noqa: E402
import denet.dataset.image_loader as ref_loader  # noqa: E402
import denet.dataset.imagenet as ref_imagenet  # noqa: E402
import denet.dataset.mscoco as ref_mscoco  # noqa: E402
import denet.dataset.pascal_voc as ref_voc  # noqa: E402

import dataset_scenarios as S  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def voc_precision(dets):
    lines = []
    keep = (ref_logging.info, ref_logging.warning)
    ref_logging.info = lambda *a: lines.append(" ".join(str(x) for x in a))
    ref_logging.warning = lambda *a: None
    try:
        ref_voc.DatasetPascalVOC.get_precision(dets)
    finally:
        ref_logging.info, ref_logging.warning = keep
    # printed with %.4f; "nan" when a class's best detection only matched `difficult` boxes (0/0 precision)
    aps = [re.search(r"AP: (\S+)", ln).group(1) for ln in lines if " - AP: " in ln]
    mean = [re.search(r"Mean AP: (\S+)", ln).group(1) for ln in lines if "Mean AP" in ln][0]
    return {"mean_ap_4dp": mean, "ap_4dp": aps}


pkg = types.SimpleNamespace(name="ref", base=ref_dataset, augment=ref_augment, image_loader=ref_loader,
                            mscoco=ref_mscoco, pascal_voc=ref_voc, imagenet=ref_imagenet, basic=ref_basic,
                            voc_precision=voc_precision)

with tempfile.TemporaryDirectory() as root:
    S.build_dataset(root)
    S.prepare_imagenet_cache(root, "ref")
    out = S.run(pkg, root)
out["_pillow"] = PIL.__version__
with open(os.path.join(HERE, "dataset_fixtures.json"), "w") as f:
    json.dump(out, f, indent=0, sort_keys=True)
print("wrote dataset_fixtures.json (%d top-level entries), Pillow %s" % (len(out), PIL.__version__))
