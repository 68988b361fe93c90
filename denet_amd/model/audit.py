"""Which kernels a training step runs, layer by layer - the audit behind bench.py's `kernels_used` and the per-layer
assertions of the whole-network parity tests (tests/test_parity_gpu.py).

The implementation of a convolution pass is a measured choice (ops._WINO: direct implicit GEMM / Winograd F(2x2) / F(4x4) /
the fused 64-channel kernels; C side: launch configuration, fused product + output-transform kernel, dedicated
filter-gradient kernel). `KernelAudit` wraps ConvLayer.forward / backward for the duration of a step, lets the C side note the
instantiation of every matrix-kernel launch (ops.LaunchTrace: no events, no effect on streams or timing) and attributes the
launches to (layer, pass). Reference operator: denet/layer/convolution.py:80-83 and its gradients, model_cnn.py:318.
"""
import os

from .. import ops
from ..layer.convolution import ConvLayer
from .model_cnn import walk_layers


def conv_layers(model):
    """[(name, layer)] of every convolution layer (nested ones included), in the order of the layer list; the name is
    `<top-level index>.<type>[.<position inside the top-level layer>]`, e.g. `12.resnet.3`"""
    out = []
    for i, top in enumerate(model.layers):
        k = 0
        for l in walk_layers([top]):
            if isinstance(l, ConvLayer):
                out.append(("%d.%s%s" % (i, top.type_name, "" if l is top else ".%d" % k), l))
                k += 1
    return out


def layer_geometry(layer):
    """(fwd geometry tuple of ops.conv_geom, human-readable string) of a ConvLayer on its physical NHWC tensors"""
    n, _, h, w = layer.input_shape
    g = ops.conv_geom((n, h, w, layer.cp), layer.omega.dev_shape, layer.stride[0], layer.pad, layer.filter_shape[3])
    fs = layer.filter_shape
    return g, "%dx%d %d->%d %dx%d/%d" % (h, w, fs[1], fs[0], fs[2], fs[3], layer.stride[0])


def decisions_cover(model):
    """the (pass, geometry) pairs of `model` for which ops._WINO holds NO decision - a step over them would measure the
    implementations first (and which one wins may vary run to run). Empty for the benchmark configurations once
    denet_amd/tuned/gfx950.json is loaded."""
    ops._load_tuned_once()
    missing = []
    for name, l in conv_layers(model):
        g, txt = layer_geometry(l)
        three = g[5] == 3 and g[6] == 3 and g[7] == 3 and g[8] == 1 and g[9] == 1 and g[3] >= 32
        if not three:
            continue                       # only the 3x3 stride-1 layers have alternatives
        need_dx = getattr(l.input, "requires_grad", True)
        for mode in (0, 1, 2):
            if mode == 1 and not need_dx:
                continue
            if (mode, g) not in ops._WINO:
                missing.append((name, txt, ("fwd", "dgrad", "wgrad")[mode]))
    return missing


class KernelAudit:
    """with KernelAudit(model) as a: model.train_step(...)   ->   a.table: [{layer, geometry, fwd: [...], bwd: [...]}]"""

    def __init__(self, model):
        self.model = model
        self.names = {id(l): (name, layer_geometry(l)[1]) for name, l in conv_layers(model)}
        self.table = None

    def __enter__(self):
        self.trace = ops.LaunchTrace()
        self.trace.__enter__()
        trace, names = self.trace, self.names
        self._saved = (ConvLayer.forward, ConvLayer.forward_folded, ConvLayer.backward)
        f0, ff0, b0 = self._saved

        def fwd(layer, *a, **k):
            trace.mark(None)
            r = f0(layer, *a, **k)
            trace.mark((id(layer), "fwd"))
            return r

        def fwdf(layer, *a, **k):
            trace.mark(None)
            r = ff0(layer, *a, **k)
            trace.mark((id(layer), "fwd"))
            return r

        def bwd(layer, *a, **k):
            trace.mark(None)
            r = b0(layer, *a, **k)
            trace.mark((id(layer), "bwd"))
            return r

        ConvLayer.forward, ConvLayer.forward_folded, ConvLayer.backward = fwd, fwdf, bwd
        return self

    def __exit__(self, *a):
        ConvLayer.forward, ConvLayer.forward_folded, ConvLayer.backward = self._saved
        self.trace.__exit__(*a)
        by = self.trace.by_label()
        self.table = []
        for name, l in conv_layers(self.model):
            self.table.append({"layer": name, "geometry": self.names[id(l)][1], "fwd": by.get((id(l), "fwd"), []),
                               "bwd": by.get((id(l), "bwd"), [])})
        self.other = by.get(None, [])
        return False

    def summary(self):
        """{geometry: {"layers": n, "fwd": [...], "bwd": [...]}} - layers of one geometry run the same kernels (asserted)"""
        out = {}
        for r in self.table:
            e = out.setdefault(r["geometry"], {"layers": 0, "fwd": r["fwd"], "bwd": r["bwd"]})
            e["layers"] += 1
            if e["fwd"] != r["fwd"] or e["bwd"] != r["bwd"]:
                e.setdefault("variants", []).append({"layer": r["layer"], "fwd": r["fwd"], "bwd": r["bwd"]})
        return out


def active_switches():
    """the DENET_* environment switches set in this process (denet_amd/switches.py lists all of them with their defaults; the
    product default is none set)"""
    from .. import switches
    return switches.active()
