"""`R` — final convolution + soft-max classification cost. Mirrors denet/layer/regression.py (RegressionLayer
:10-98, parse_desc :53-63 appends the conv then this layer). Cost: -mean(log_softmax(x)[b, image_class]) (:97-98)."""
import math

import numpy

from . import AbstractLayer, Act
from .convolution import ConvLayer
from .. import ops


class RegressionLayer(AbstractLayer):
    type_name = "regression"

    def __init__(self, layers, use_center=True, valid=[], json_param={}):
        super().__init__(layer_index=len(layers))
        self.input = layers[-1].output
        self.input_shape = layers[-1].output_shape
        if use_center:
            valid = [(0, self.input_shape[-2] // 2, self.input_shape[-1] // 2)]
        self.valid = json_param.get("valid", valid)
        if len(self.valid) > 0 or self.input_shape[2] != 1 or self.input_shape[3] != 1:
            raise NotImplementedError("regression over a spatial map / centre pixel is outside the hot path "
                                      "(use `R` after a full-size pooling, as the shipped recipes do)")
        self.log_pr_shape = self.input_shape
        self.output_shape = (self.input_shape[0], self.input_shape[1])
        self.output = Act(self.output_shape, self.input.cp, "regression")
        self.class_num = self.input_shape[1]
        self._target = None

    @staticmethod
    def parse_desc(layers, name, tags, params):
        if name != "R":
            return False
        use_bias = bool("B" in tags)
        use_center = bool("C" in tags)
        filter_shape = (params["classNum"], layers[-1].output_shape[1], params.get(0, layers[-1].output_shape[2]),
                        params.get(0, layers[-1].output_shape[3]))
        layers.append(ConvLayer(layers, filter_shape, (1, 1), use_bias, "valid", params["wb"]))
        layers.append(RegressionLayer(layers, use_center))
        return True

    def export_json(self):
        json = super().export_json()
        json.update({"valid": self.valid})
        return json

    def get_target(self, model, samples, metas):
        yt_index = [numpy.ravel_multi_index((b, metas[b]["image_class"]), self.output_shape) for b in range(len(metas))]
        return numpy.array(yt_index, dtype=numpy.int64), numpy.array([], dtype=numpy.float32)

    def cost(self, yt_index, yt_value):
        return True

    def forward(self, ctx):
        # probabilities are only materialised on request (predict); training needs the logits only
        self.output.data = self.input.data

    def set_target(self, ctx, yt_index, yt_value):
        import torch
        B, C = self.output_shape
        t = numpy.zeros((B, C), dtype=numpy.float32)
        t.reshape(-1)[numpy.asarray(yt_index)] = 1.0
        self._target = torch.from_numpy(t).cuda()

    def loss_backward(self, ctx, cost_out, want_grad=True):
        """cost = -mean_b log_softmax(x)[b, cls]; evaluated with the detection-cost kernel (one wave per row):
        cost_factor = ln(C) cancels its 1/ln(C) normalisation, batch = B gives the mean."""
        B, C = self.output_shape
        logits = self.input.data.view(B, self.input.cp)
        dl = ops.empty(B, self.input.cp) if want_grad else None
        ops.detect_loss(logits, self._target, None, None, None, dl, cost_out, B, C, 0, math.log(C), 0.0)
        if want_grad:
            self.input.grad = dl.view(self.input.data.shape)

    def backward(self, ctx):
        pass
