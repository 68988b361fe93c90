"""`P` layer — max / average_inc_pad pooling. Mirrors denet/layer/pool.py (PoolLayer :10-67): cuDNN semantics
(max: padded taps are -inf; average_inc_pad: divisor k*k). Only ignore_border=True (the cuDNN path)."""
import math

from . import AbstractLayer, Act
from .. import ops


class PoolLayer(AbstractLayer):
    type_name = "pool"

    def __init__(self, layers, size=(2, 2), stride=None, pad=(0, 0), mode="max", ignore_border=True, json_param={}):
        super().__init__(layer_index=len(layers))
        self.input = layers[-1].output
        self.input_shape = layers[-1].output_shape
        self.size = tuple(json_param.get("size", size))
        self.pad = tuple(json_param.get("pad", pad))
        self.ignore_border = json_param.get("ignoreBorder", ignore_border)
        self.mode = json_param.get("mode", mode)
        self.stride = json_param.get("stride", stride)
        if self.stride is None:
            self.stride = self.size
        self.stride = tuple(self.stride)
        if self.size[0] is None:
            raise Exception("P layer needs an explicit size (pool.py:51: params.get(0) has no default)")
        if not self.ignore_border:
            raise NotImplementedError("ignore_border=False (non-cuDNN pooling) is outside the hot path")
        assert self.size[0] == self.size[1] and self.stride[0] == self.stride[1] and self.pad[0] == self.pad[1]
        assert self.mode in ("max", "average_inc_pad"), self.mode
        h = int(math.floor((self.input_shape[2] + 2 * self.pad[0] - self.size[0]) / self.stride[0])) + 1
        w = int(math.floor((self.input_shape[3] + 2 * self.pad[1] - self.size[1]) / self.stride[1])) + 1
        self.output_shape = (self.input_shape[0], self.input_shape[1], h, w)
        self.output = Act(self.output_shape, self.input.cp, "pool%i" % self.layer_index)
        self._arg = None

    @staticmethod
    def parse_desc(layers, name, tags, params):
        if name != "P":
            return False
        size = (params.get(0), params.get(0))
        stride = (params.get(1, size[0]), params.get(1, size[0]))
        pad = (params.get(2, 0), params.get(2, 0))
        mode = "average_inc_pad" if "A" in tags else "max"
        ignore_border = bool(not "B" in tags)
        layers.append(PoolLayer(layers, size, stride, pad, ignore_border=ignore_border, mode=mode))
        return True

    def export_json(self):
        json = super().export_json()
        json.update({"mode": self.mode, "size": self.size, "stride": self.stride, "pad": self.pad,
                     "ignoreBorder": self.ignore_border})
        return json

    def forward(self, ctx):
        k, s, p = self.size[0], self.stride[0], self.pad[0]
        if ctx is not None and getattr(self, "_fused_in", None) is ctx:
            return                               # the BN + ReLU layer in front has already written output and argmax (this pass)
        if self.mode == "max":
            self.output.data, self._arg = ops.maxpool_fwd(self.input.data, k, s, p)
        else:
            self.output.data = ops.avgpool_fwd(self.input.data, k, s, p)

    def backward(self, ctx):
        k, s, p = self.size[0], self.stride[0], self.pad[0]
        if ctx is not None and getattr(self, "_fused_in", None) is ctx:
            self._fused_in = None                # the BN + ReLU layer in front gathers this layer's gradient itself
            return
        shape = tuple(self.input.data.shape)
        if self.mode == "max":
            self.input.add_grad(ops.maxpool_bwd(self.output.grad, self._arg, shape, k, s, p))
        else:
            self.input.add_grad(ops.avgpool_bwd(self.output.grad, shape, k, s, p))
