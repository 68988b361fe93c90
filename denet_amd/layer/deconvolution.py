"""`DC` layer — transposed convolution. Mirrors denet/layer/deconvolution.py (DeconvLayer :11-107): the output is
`conv2d_grad_wrt_inputs(input, omega^T, ...)` (:63-65), i.e. the data gradient of the TRUE convolution
F: (B, C_out, h, w) -> (B, C_in, H, W) whose filters are omega with its first two axes swapped, border `half`
only, h = H*s - 2(k//2) + k - 1 (:54-60). Bias is ON unless the `B` tag is given (:79 - the opposite of `C`).

Nothing new runs on the device: forward is the implicit-GEMM data-gradient kernel of csrc/igemm.hip applied to
the layer input, the input gradient is F's forward kernel applied to the output gradient, and the filter gradient
is F's weight-gradient kernel with the roles of activation and gradient swapped."""
import math

import numpy

from . import AbstractLayer, Act, Param, round_up
from .. import ops


class DeconvParam(Param):
    """omega in the reference layout (C_out, C_in, kh, kw); on the device the filters of F as correlation taps,
    w_dev[ci, r, s, co] = omega[co, ci, R-1-r, S-1-s], zero padded to [Cin_p][R][S][Cout_p]"""

    def to_dev_layout(self, v=None):
        v = self.value if v is None else numpy.asarray(v, dtype=numpy.float32)
        out = numpy.zeros(self.dev_shape, dtype=numpy.float32)
        O, I, R, S = v.shape
        out[:I, :R, :S, :O] = v[:, :, ::-1, ::-1].transpose(1, 2, 3, 0)
        return out

    def from_dev_layout(self, d):
        d = numpy.asarray(d, dtype=numpy.float32).reshape(self.dev_shape)
        O, I, R, S = self.value.shape
        return numpy.ascontiguousarray(d[:I, :R, :S, :O].transpose(3, 0, 1, 2)[:, :, ::-1, ::-1])


class DeconvLayer(AbstractLayer):
    type_name = "deconv"

    def __init__(self, layers, filter_shape=None, filter_stride=(1, 1), use_bias=True, border_mode="valid",
                 wb="he-backward", json_param={}):
        super().__init__(layer_index=len(layers))
        self.input = layers[-1].output
        self.input_shape = layers[-1].output_shape
        self.border_mode = json_param.get("border", border_mode)
        self.filter_shape = tuple(json_param.get("shape", filter_shape))
        self.stride = tuple(json_param.get("stride", filter_stride))
        self.use_bias = json_param.get("useBias", use_bias)
        self.size = (self.filter_shape[2], self.filter_shape[3])

        fs = self.filter_shape
        # deconvolution.py:27-37
        if type(wb) is float or type(wb) is int:
            self.w_bound = float(wb)
        elif "he-forward" in wb:
            self.w_bound = math.sqrt(2.0 / (fs[2] * fs[3] * fs[1]))
        elif "he-backward" in wb:
            self.w_bound = math.sqrt(2.0 / (fs[2] * fs[3] * fs[0]))
        elif "xavier-forward" in wb:
            self.w_bound = math.sqrt(1.0 / (fs[2] * fs[3] * fs[1]))
        elif "xavier-backward" in wb:
            self.w_bound = math.sqrt(1.0 / (fs[2] * fs[3] * fs[0]))
        else:
            raise Exception("Unknown weight initialisation: " + str(wb))
        # same numpy.random call sequence as the reference (deconvolution.py:40-46)
        if self.w_bound > 0:
            if type(wb) is str and "uniform" in wb:
                w = numpy.random.uniform(-self.w_bound, self.w_bound, size=fs)
            else:
                w = numpy.random.normal(0.0, self.w_bound, size=fs)
        else:
            w = numpy.zeros(shape=fs)

        assert fs[1] == self.input_shape[1], "filter channels %i != input channels %i" % (fs[1], self.input_shape[1])
        assert self.stride[0] == self.stride[1], "only square strides are supported"
        assert fs[2] == fs[3], "only square filters are supported"
        if self.border_mode != "half":
            raise Exception("Unknown border mode: " + str(self.border_mode))   # deconvolution.py:61
        self.pad = fs[2] // 2
        h = self.input_shape[2] * self.stride[0] - 2 * self.pad + fs[2] - 1
        wd = self.input_shape[3] * self.stride[1] - 2 * self.pad + fs[3] - 1
        self.output_shape = (self.input_shape[0], fs[0], h, wd)

        self.cin_p = self.input.cp
        assert self.cin_p % 32 == 0, "DC needs a 32-channel-padded input (not the raw image)"
        self.cout_p = round_up(fs[0], 32)
        self.omega = DeconvParam(w, "deconv omega", "deconv", (self.cin_p, fs[2], fs[3], self.cout_p), s_real=fs[3])
        if self.use_bias:
            self.beta = Param(numpy.zeros((fs[0],)), "deconv beta", "vector", (self.cout_p,))
        self.output = Act(self.output_shape, self.cout_p, "deconv%i" % self.layer_index)

    @staticmethod
    def parse_desc(layers, name, tags, params):
        if name != "DC":
            return False
        use_bias = bool("B" not in tags)
        if bool("X" in tags):
            filter_shape = (params.get(0), layers[-1].output_shape[1], params.get(1), params.get(2))
            filter_stride = (params.get(3, 1), params.get(4, 1))
        else:
            filter_shape = (params.get(0), layers[-1].output_shape[1], params.get(1, 1), params.get(1, 1))
            filter_stride = (params.get(2, 1), params.get(2, 1))
        layers.append(DeconvLayer(layers, filter_shape, filter_stride, use_bias, params["borderMode"], params["wb"]))
        return True

    def weights(self):
        return [self.omega]

    def biases(self):
        return [self.beta] if self.use_bias else []

    def all_params(self):
        return [self.omega] + ([self.beta] if self.use_bias else [])

    def import_json(self, json_param):
        super().import_json(json_param)
        if self.use_bias:
            self.beta.set_value(numpy.asarray(json_param["bias"], dtype=numpy.float32))
        self.omega.set_value(numpy.asarray(json_param["weight"], dtype=numpy.float32))

    def export_json(self):
        json = super().export_json()
        json.update({"shape": self.filter_shape,
                     "stride": self.stride,
                     "border": self.border_mode,
                     "useBias": self.use_bias,
                     "bias": self.beta.get_value() if self.use_bias else None,
                     "weight": self.omega.get_value()})
        return json

    # ---- execution: F has C = C_out (its input channels) and K = C_in (its output channels) ----
    def _w(self):
        return self.omega.dev.view(self.omega.dev_shape)

    def _logical(self):
        return (self.filter_shape[0], self.filter_shape[1])

    def forward(self, ctx):
        x = self.input.data
        N = x.shape[0]
        out_shape = (N, self.output_shape[2], self.output_shape[3], self.cout_p)
        y = ops.conv_dgrad(x, self._w(), out_shape, stride=self.stride[0], pad=self.pad, s_real=self.filter_shape[3],
                           logical=self._logical())
        if self.use_bias:
            ops.add_bias(y, self.beta.dev, out=y)
        self.output.data = y

    def backward(self, ctx):
        dy = self.output.grad
        x = self.input.data
        st, pad, sr = self.stride[0], self.pad, self.filter_shape[3]
        if self.omega.grad is not None:
            with ops.wgrad_stream():
                ops.conv_wgrad(dy, x, self.omega.dev_shape, stride=st, pad=pad, s_real=sr,
                               out=self.omega.grad.view(self.omega.dev_shape), logical=self._logical())
                if self.use_bias:
                    ops.colsum(dy.view(-1, self.cout_p), out=self.beta.grad)
        if getattr(self.input, "requires_grad", True):
            self.input.add_grad(ops.conv_fwd(dy, self._w(), stride=st, pad=pad, s_real=sr, logical=self._logical()))
