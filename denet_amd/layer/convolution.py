"""`C` layer — 2-D convolution. Mirrors denet/layer/convolution.py (ConvLayer :10-136, parse_desc :99-112,
export/import_json :114-136). The reference lowers to Theano `conv2d` (a TRUE convolution: flipped filters,
convolution.py:80-83) and cuDNN; here forward / data-gradient / weight-gradient are the MFMA implicit-GEMM
kernels of csrc/igemm.hip, the filters being stored flipped in KRSC layout on the device (layer.Param)."""
import math

import numpy

from . import AbstractLayer, Act, Param, round_up
from .. import ops


def conv_padding(border_mode, k):
    """symmetric zero padding implied by a Theano border mode for a k-wide filter"""
    if border_mode == "valid":
        return 0
    if border_mode == "full":
        return k - 1
    if border_mode == "half":
        return k // 2
    if border_mode == "same":
        if k % 2 == 0:
            raise NotImplementedError("border_mode 'same' with an even filter needs asymmetric padding")
        return (k - 1) // 2
    if isinstance(border_mode, (tuple, list)):
        assert border_mode[0] == border_mode[1], "only symmetric padding is supported"
        return int(border_mode[0])
    if isinstance(border_mode, int):  # includes False == 0 (denet_corner.py:39)
        return int(border_mode)
    raise Exception("Unknown border mode: " + str(border_mode))


class ConvLayer(AbstractLayer):
    type_name = "conv"

    def __init__(self, layers, filter_shape=None, filter_stride=(1, 1), use_bias=False, border_mode="half",
                 wb="he-backward", json_param={}):
        super().__init__(layer_index=len(layers))
        self.input = layers[-1].output
        self.input_shape = layers[-1].output_shape

        self.border_mode = json_param.get("border", border_mode)
        if isinstance(self.border_mode, list):
            self.border_mode = tuple(self.border_mode)
        self.filter_shape = tuple(json_param.get("shape", filter_shape))
        self.stride = tuple(json_param.get("stride", filter_stride))
        self.use_bias = json_param.get("useBias", use_bias)
        self.size = (self.filter_shape[2], self.filter_shape[3])
        self.enabled = json_param.get("enabled", True)

        fs = self.filter_shape
        # weight initialisation bound (convolution.py:31-40)
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

        # same numpy.random call sequence as the reference (convolution.py:42-48)
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
        self.pad = conv_padding(self.border_mode, fs[2])
        if self.border_mode == "same":
            assert self.stride == (1, 1)

        # device geometry: physical channels, padded taps for the small-C first layer
        self.cp = self.input.cp
        self.kp = round_up(fs[0], 32)
        self.s_pad = fs[3] if self.cp >= 32 else round_up(fs[3], 32 // self.cp)
        self.omega = Param(w, "conv omega", "conv", (self.kp, fs[2], self.s_pad, self.cp), s_real=fs[3])
        if self.use_bias:
            self.beta = Param(numpy.zeros((fs[0],)), "conv beta", "vector", (self.kp,))

        # output shape (convolution.py:55-74)
        oh = int(math.ceil((self.input_shape[-2] + 2 * self.pad - fs[2] + 1) / self.stride[0]))
        ow = int(math.ceil((self.input_shape[-1] + 2 * self.pad - fs[3] + 1) / self.stride[1]))
        self.output_shape = (self.input_shape[0], fs[0], oh, ow)
        self.output = Act(self.output_shape, self.kp, "conv%i" % self.layer_index)

    @staticmethod
    def parse_desc(layers, name, tags, params):
        if name != "C":
            return False
        use_bias = bool("B" in tags)
        if bool("X" in tags):
            filter_shape = (params.get(0), layers[-1].output_shape[1], params.get(1), params.get(2))
            filter_stride = (params.get(3, 1), params.get(4, 1))
        else:
            filter_shape = (params.get(0), layers[-1].output_shape[1], params.get(1, 1), params.get(1, 1))
            filter_stride = (params.get(2, 1), params.get(2, 1))
        layers.append(ConvLayer(layers, filter_shape, filter_stride, use_bias, params["borderMode"], params["wb"]))
        return True

    def weights(self):
        return super().weights() + ([self.omega] if self.enabled else [])

    def biases(self):
        return super().biases() + ([self.beta] if self.use_bias and self.enabled else [])

    def all_params(self):
        """every array of the layer, trained or frozen (device packing)"""
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
                     "enabled": self.enabled,
                     "useBias": self.use_bias,
                     "bias": self.beta.get_value() if self.use_bias else None,
                     "weight": self.omega.get_value()})
        return json

    # ---- execution ----
    def _w(self):
        return self.omega.dev.view(self.omega.dev_shape)

    def _logical(self):
        return (self.filter_shape[1], self.filter_shape[0])

    def _cache(self):
        """per-layer state of the Winograd passes (chosen tiles, transformed filters, kept input transform)"""
        c = self.__dict__.get("_wino_cache")
        if c is None:
            c = self.__dict__["_wino_cache"] = {}
        return c

    def forward(self, ctx, add=None):
        from . import get_train
        # the input may be the output of a batch norm whose pointwise pass is still pending (ops.BnLink): a Winograd pass runs it
        # inside its input transform; whatever path conv_fwd takes, the activation exists afterwards and is stored back
        link = self.input.take_pending_data()
        cache = self._cache()
        cache["train"] = bool(get_train()) and self.enabled and self.omega.grad is not None
        # a batch norm directly behind this layer (it flags its input Act) gets its statistics from this pass's epilogue
        want_stats = bool(get_train()) and getattr(self.output, "want_stats", False)
        cache["bn_final"] = None
        self._planar_x = None
        if isinstance(link, ops.NchwLink):
            # the network input, still in the reference's planar layout: the first layer's kernels read it as it is (forward and,
            # in backward(), the filter gradient); anybody else who asks the Act for .data gets the NHWC tensor made then
            self.input.set_pending_data(link)
            if add is None and self.use_bias and ops.conv_stem_ok(link.x, self.omega.dev_shape, self.stride[0], self.pad,
                                                                   self.filter_shape[3]):
                self.output.data = ops.conv_stem_fwd(link.x, self._w(), self.beta.dev, cache, want_stats, logical=self._logical())
                self.output.stats = cache.pop("bn_stats", None) if want_stats else None
                self._planar_x = link.x
                return
            link = None
        up = None
        if isinstance(link, ops.UpLink):
            up, link = link, None                 # the pool-inverse layer in front has not written its output (ops.UpLink)
        x = self.input.data if (link is None and up is None) else None
        skip = getattr(self, "skip_behind", None)
        if skip is not None and add is None and get_train() and ctx is not None:
            # the SKIP layer behind adds its tap to this layer's output: here, in the epilogue (ModelCNN.build_train_func links the
            # two); the sum is the SKIP layer's output, and what a batch norm behind THAT wants to know about it is measured here
            want_skip_stats = getattr(skip.output, "want_stats", False)
            sbn = getattr(skip.output, "stats_bn", None) if want_skip_stats else None
            cache["bn_final"] = sbn.stats_final(skip.output) if sbn is not None else None
            y = ops.conv_fwd(x, self._w(), bias=None, add=skip.y.data, stride=self.stride[0], pad=self.pad,
                             s_real=self.filter_shape[3], logical=self._logical(), cache=cache, bn_stats=want_skip_stats, link=link,
                             up=up)
            if link is not None:
                self.input.data = link.materialise()
            self._settle_up(up)
            self.output.data = None              # the convolution's own output is never written
            self.output.stats = None
            skip.output.data = y
            skip.output.stats = cache.pop("bn_stats", None) if want_skip_stats else None
            skip._fused_in = ctx
            return
        sbn = getattr(self.output, "stats_bn", None) if (want_stats and ctx is not None) else None
        cache["bn_final"] = sbn.stats_final(self.output) if sbn is not None else None
        self.output.data = ops.conv_fwd(x, self._w(), bias=self.beta.dev if self.use_bias else None, add=add,
                                        stride=self.stride[0], pad=self.pad, s_real=self.filter_shape[3],
                                        logical=self._logical(), cache=cache, bn_stats=want_stats, link=link, up=up)
        if link is not None:
            self.input.data = link.materialise()
        self._settle_up(up)
        self.output.stats = cache.pop("bn_stats", None) if want_stats else None

    def _settle_up(self, up):
        """after a forward pass on an un-written up-sampled input: the tensor if the pass had to make it, else still pending (the
        filter gradient asks for it on its own stream, anybody else through Act.data)"""
        if up is None:
            return
        if up.result is not None:
            self.input.data = up.result
        else:
            self.input.set_pending_data(up)

    def forward_folded(self, ctx, bn, add=None, relu=False, out_act=None):
        """inference only: this convolution with the batch-norm layer behind it folded into its filters (recomputed
        when the weights change), residual `add` and ReLU in the epilogue; writes the batch norm's output"""
        cache = self._cache()
        cache["train"] = False
        ent = cache.get("fold")
        if ent is None or ent[0] != ops.WEIGHTS_VERSION or ent[1] is not bn:
            w_f, b_f = ops.bn_fold(self._w(), self.beta.dev if self.use_bias else None, bn.omega.dev, bn.beta.dev,
                                   bn.mean.dev, bn.stdinv.dev, bn.eps)
            ent = cache["fold"] = (ops.WEIGHTS_VERSION, bn, w_f, b_f)
        link = self.input.take_pending_data()
        if isinstance(link, ops.NchwLink):
            self.input.set_pending_data(link)          # (see forward: the first layer reads the planar batch)
            if add is None and ops.conv_stem_ok(link.x, self.omega.dev_shape, self.stride[0], self.pad, self.filter_shape[3]):
                y = ops.conv_stem_fwd(link.x, ent[2], ent[3], cache, False, logical=self._logical(), relu=relu)
                self.output.data = y
                (out_act if out_act is not None else bn.output).data = y
                return y
        elif link is not None:
            self.input.data = link.materialise()
        y = ops.conv_fwd(self.input.data, ent[2], bias=ent[3], add=add, stride=self.stride[0], pad=self.pad,
                         s_real=self.filter_shape[3], logical=self._logical(), cache=cache, relu=relu)
        self.output.data = y
        (out_act if out_act is not None else bn.output).data = y
        return y

    def backward(self, ctx):
        # (the first layer on the planar network input has no data gradient and reads the image as it is)
        planar = getattr(self, "_planar_x", None) if not getattr(self.input, "requires_grad", True) else None
        # an input that is the un-written output of a pool-inverse layer (ops.UpLink, see forward): the data gradient needs its shape
        # only, the filter gradient makes the tensor on ITS stream (it has the time) unless the transformed input was kept
        up = self.input._pending_data if isinstance(self.input._pending_data, ops.UpLink) and self.input._data is None else None
        if up is not None and self.output._pending_grad is not None:
            up = None                             # (a linked batch-norm gradient: the one-kernel form reads x itself)
        x = self.input.data if (planar is None and up is None) else None
        x_shape = tuple(up.shape) if up is not None else (tuple(x.shape) if x is not None else None)
        st, pad, sr = self.stride[0], self.pad, self.filter_shape[3]
        # the batch norm (of this training step) whose output is this layer's input: the data-gradient pass below writes the
        # gradient of that output, and a Winograd pass can leave that batch norm's two backward reductions behind (ops.BnSums)
        sums = None
        bn = self.input.bn_producer
        if (ops.BWD_SUMS and bn is not None and self._cache().get("train") and getattr(self.input, "requires_grad", True)
                and not getattr(self, "not_last_writer", False)):
            sums = bn.sums_request(self.input)
        link = self.output.take_pending_grad()
        if link is not None:
            # the gradient of the output is the pending pointwise pass of a batch norm's backward (ops.BnLink): a 3x3 layer whose
            # data- and filter-gradient passes are Winograd passes forms it inside ONE transform kernel that feeds both
            fs = self.filter_shape
            if (self.enabled and self.omega.grad is not None and getattr(self.input, "requires_grad", True) and not self.use_bias
                    and fs[2] == 3 and fs[3] == 3 and st == 1 and self.stride[1] == 1 and pad == 1):
                dx = ops.conv_backward_linked(link, x, self._w(), self.omega.dev_shape, self.input.grad,
                                              self.omega.grad.view(self.omega.dev_shape), self._cache(), stride=st, pad=pad,
                                              s_real=sr, logical=self._logical(), sums=sums)
                if dx is not None:
                    self.input.grad = dx
                    self.input.grad_sums = sums
                    return
            self.output.grad = link.materialise()
        dy = self.output.grad
        need_dx = getattr(self.input, "requires_grad", True)
        # two MFMA-bound GEMMs of hundreds of GFLOP gain nothing from sharing the chip (measured: the 4736 -> 1536 head layer's
        # pair takes 4.9 ms side by side, 2.1 + 2.2 ms one after the other) and the data gradient is the one the backward sweep
        # waits for: it goes first, the filter gradient follows on the second stream beside the HBM-bound passes that come next
        first = need_dx and sr == 1 and st == 1 and 2e-9 * dy.numel() * x_shape[-1] >= ops.DGRAD_FIRST_GFLOP
        if first:
            self.input.grad = ops.conv_dgrad(dy, self._w(), x_shape, add=self.input.grad, stride=st, pad=pad,
                                             s_real=sr, logical=self._logical(), cache=self._cache(), sums=sums)
            self.input.grad_sums = sums
        if self.enabled and self.omega.grad is not None:
            # the first layer of the network has no data gradient: its filter gradient is the tail of the backward sweep on the
            # second stream, and the bias column sums (a pass over the largest tensor of the network) run beside it on the
            # compute stream, which has nothing left to do, instead of behind it
            tail = not getattr(self.input, "requires_grad", True)
            with ops.wgrad_stream():        # independent of the data-gradient chain below
                if planar is not None:
                    ops.conv_stem_wgrad(planar, dy, self.omega.dev_shape, self.omega.grad.view(self.omega.dev_shape),
                                        logical=self._logical())
                else:
                    if up is not None:
                        x = self.input.data          # (written here, on the filter-gradient stream)
                    ops.conv_wgrad(x, dy, self.omega.dev_shape, stride=st, pad=pad, s_real=sr,
                                   out=self.omega.grad.view(self.omega.dev_shape), logical=self._logical(),
                                   cache=self._cache())
                if self.use_bias and not tail:
                    ops.colsum(dy.view(-1, self.kp), out=self.beta.grad)
            if self.use_bias and tail:
                ops.colsum(dy.view(-1, self.kp), out=self.beta.grad)
        if need_dx and not first:
            self.input.grad = ops.conv_dgrad(dy, self._w(), x_shape, add=self.input.grad, stride=st, pad=pad,
                                             s_real=sr, logical=self._logical(), cache=self._cache(), sums=sums)
            self.input.grad_sums = sums
