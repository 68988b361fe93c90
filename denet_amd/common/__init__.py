"""Host-side helpers mirroring denet/common/__init__.py of the reference (Timer :16-46, find_layers :65-86,
overlap / overlap_iou :91-109, ndarray_unpack :125-133, convert_num :142-149)."""
import time

import numpy

from .json_util import json_from_file, json_to_file  # noqa: F401 (denet.common re-exports them)


class Timer:
    """wall-clock marks, same phase bookkeeping as the reference Timer (denet/common/__init__.py:16-46)"""

    def __init__(self):
        self.reset()

    def reset(self):
        self.times = [time.time()]

    def mark(self):
        self.times.append(time.time())

    def current(self):
        return time.time() - self.times[0]

    def current_ms(self):
        return 1000.0 * self.current()

    def delta(self, index):
        return self.times[index + 1] - self.times[index]

    def delta_ms(self, index):
        return 1000.0 * self.delta(index)

    def deltas(self):
        return [self.delta(i) for i in range(len(self.times) - 1)]

    def deltas_ms(self):
        return [1000.0 * d for d in self.deltas()]


def find_layers(layers, layer_names, warn_missing=False):
    """first layer of each requested type_name (denet/common/__init__.py:65-86)"""
    single = isinstance(layer_names, str)
    names = [layer_names] if single else list(layer_names)
    found = [None] * len(names)
    for layer in layers:
        for i, name in enumerate(names):
            if found[i] is None and layer.type_name == name:
                found[i] = layer
    if warn_missing:
        missed = [names[i] for i, f in enumerate(found) if f is None]
        if missed:
            raise Exception("Could not find layers of name: ", missed)
    return found[0] if len(names) == 1 else found


def overlap(bbox0, bbox1=(0, 0, 1, 1)):
    dx = max(0, min(bbox0[2], bbox1[2]) - max(bbox0[0], bbox1[0]))
    dy = max(0, min(bbox0[3], bbox1[3]) - max(bbox0[1], bbox1[1]))
    return dx * dy


def overlap_rel(bbox0, bbox1=(0, 0, 1, 1)):
    """share of bbox0 that lies inside bbox1 (denet/common/__init__.py:97-102)"""
    a = (bbox0[2] - bbox0[0]) * (bbox0[3] - bbox0[1])
    return overlap(bbox0, bbox1) / a if a > 0 else 0.0


def overlap_iou(bbox0, bbox1=(0, 0, 1, 1)):
    a0 = (bbox0[2] - bbox0[0]) * (bbox0[3] - bbox0[1])
    a1 = (bbox1[2] - bbox1[0]) * (bbox1[3] - bbox1[1])
    ai = overlap(bbox0, bbox1)
    return ai / (a0 + a1 - ai)


def clip(x, x_min=None, x_max=None):
    """denet/common/__init__.py:112-118"""
    if x_min is None:
        return min(x, x_max)
    if x_max is None:
        return max(x, x_min)
    return min(x_max, max(x, x_min))


def ndarray_unpack(v, shapes):
    index = 0
    r = []
    for shape in shapes:
        size = int(numpy.prod(shape))
        r.append(v[index:(index + size)].reshape(shape))
        index += size
    return r


def convert_num(s):
    """int -> float -> str, the model-desc argument conversion (denet/common/__init__.py:142-149)"""
    try:
        return int(s)
    except ValueError:
        try:
            return float(s)
        except ValueError:
            return s


def get_params_dict(params):
    """"a=1,b=0.5,flag" -> {"a": 1, "b": 0.5, "flag": True} (denet/common/__init__.py:200-208; predict.py:168)"""
    out = {}
    for item in params.split(","):
        name, _, value = item.partition("=")
        out[name] = convert_num(value) if "=" in item else True
    return out


def get_overlap_iou(obj_bboxs, sample_bboxs):
    """fp32 IoU matrix objects x samples, the arithmetic of the compiled Theano function in
    denet/common/theano_util.py:38-59 (inputs down-cast to float32, all operations in float32)."""
    if len(obj_bboxs) == 0 or len(sample_bboxs) == 0:
        return None
    x = numpy.array(obj_bboxs, dtype=numpy.float32)
    y = numpy.array(sample_bboxs, dtype=numpy.float32)
    x_area = (x[:, 2] - x[:, 0]) * (x[:, 3] - x[:, 1])
    y_area = (y[:, 2] - y[:, 0]) * (y[:, 3] - y[:, 1])
    zero = numpy.float32(0)
    dx = numpy.maximum(numpy.minimum(x[:, None, 2], y[None, :, 2]) - numpy.maximum(x[:, None, 0], y[None, :, 0]), zero)
    dy = numpy.maximum(numpy.minimum(x[:, None, 3], y[None, :, 3]) - numpy.maximum(x[:, None, 1], y[None, :, 1]), zero)
    inter = dx * dy
    union = (x_area[:, None] + y_area[None, :]) - inter
    return inter / union
