"""Layer operator surface of the DeNet hot path: same protocol as denet/layer/__init__.py of the reference
(AbstractLayer :64-143, globals :6-28, import_json :31-60) with the Theano symbolic variables replaced by
static activation handles (`Act`) that the executor in denet_amd/model/model_cnn.py fills with NHWC device
buffers, and `theano.shared` parameters replaced by `Param` (host value in the reference's layout + a view
into the model's flat device buffers in the kernel layout).
"""
import numpy

# ---- global training state (reference: layer_train_enable / layer_train_epoch / layer_train_it) ----------
_state = {"train": True, "epoch": 0, "iteration": 0, "rng_seed": 0}


def get_train():
    return _state["train"]


def set_train(v):
    _state["train"] = bool(v)


def get_epoch():
    return _state["epoch"]


def set_epoch(v):
    _state["epoch"] = int(v)


def get_iteration():
    return _state["iteration"]


def set_iteration(v):
    _state["iteration"] = v


def set_rng_seed(v):
    """reference: seeds the MRG stream of the `D` / `CM` layers (layer/__init__.py:14-16); here: the base key of
    their counter-based generator"""
    _state["rng_seed"] = int(v)


def get_rng_seed():
    return _state["rng_seed"]


def round_up(v, m):
    return ((int(v) + m - 1) // m) * m


class Act:
    """Static activation handle: what `layer.output` is in this build.

    shape  logical shape in the reference's convention, (N, C, H, W) or (N, C)
    cp     physical channel count of the NHWC device buffer (C rounded up to 32; 4 for the image input)
    data   torch tensor [N, H, W, cp] (or [N, cp]) once the forward pass reached it
    grad   gradient w.r.t. data during the backward pass, None if nothing flowed yet
    """

    def __init__(self, shape, cp=None, name=""):
        self.shape = tuple(int(s) for s in shape)
        self.cp = int(cp) if cp is not None else round_up(self.shape[1], 32)
        self.name = name
        self._data = None
        self._grad = None
        self._pending_data = None       # ops.BnLink: the tensor is the output of a batch norm whose pointwise pass has not run
        self._pending_grad = None       # ops.BnLink: the gradient is what a batch norm's backward pointwise pass would write
        self.bn_producer = None         # the batch-norm layer whose forward pass (training) wrote this tensor
        self.grad_sums = None           # ops.BnSums left by the data-gradient pass that wrote .grad LAST (cleared by any later write)

    # `data` / `grad` may be PENDING: a batch-norm layer has reduced its statistics but left the pointwise pass to the consumer
    # (a Winograd convolution evaluates it inside its input transform, ops.conv_fwd / conv_backward_linked). Any other reader
    # simply takes .data / .grad, which runs the pointwise kernel then - the same kernel the batch norm would have launched.
    @property
    def data(self):
        if self._data is None and self._pending_data is not None:
            import torch
            link, self._pending_data = self._pending_data, None
            self._data = link.materialise()
            # written on whatever stream the first reader runs on (a filter-gradient pass materialises an up-sampled tensor on the
            # second stream): a later reader on another stream is ordered behind that write
            self._data_stream = torch.cuda.current_stream()
            self._data_event = torch.cuda.Event()
            self._data_event.record(self._data_stream)
        elif getattr(self, "_data_event", None) is not None:
            import torch
            cur = torch.cuda.current_stream()
            if cur != self._data_stream:
                cur.wait_event(self._data_event)
                self._data.record_stream(cur)
        return self._data

    @data.setter
    def data(self, v):
        self._data = v
        self._pending_data = None
        self._data_event = None

    @property
    def grad(self):
        if self._pending_grad is not None:
            link, self._pending_grad = self._pending_grad, None
            g = link.materialise()
            self._grad = g if self._grad is None else self._add(self._grad, g)
        return self._grad

    @grad.setter
    def grad(self, v):
        self._grad = v
        self._pending_grad = None
        self.grad_sums = None

    @staticmethod
    def _add(a, b):
        from .. import ops
        return ops.add(a, b)

    def set_pending_data(self, link):
        self._data = None
        self._pending_data = link
        self._data_event = None

    def take_pending_data(self):
        """the pending batch-norm link, or None; the taker must store the tensor it materialises into .data"""
        link, self._pending_data = self._pending_data, None
        return link

    def set_pending_grad(self, link):
        """only valid while nothing else has flowed into this tensor's gradient"""
        assert self._grad is None and self._pending_grad is None
        self._pending_grad = link
        self.grad_sums = None

    def take_pending_grad(self):
        link, self._pending_grad = self._pending_grad, None
        return link

    def phys_shape(self):
        if len(self.shape) == 4:
            return (self.shape[0], self.shape[2], self.shape[3], self.cp)
        return (self.shape[0], self.cp)

    def add_grad(self, g):
        """accumulate a gradient contribution (fan-out points: residual taps, skip sources)"""
        if self.grad is None:
            self.grad = g
        else:
            from .. import ops
            self.grad = ops.add(self.grad, g)

    def clear(self):
        self.data = None
        self.grad = None
        self.stats = None


class Param:
    """A learnable (or running-statistic) array.

    `value` is the host copy in the REFERENCE layout (conv: OIHW true-convolution filters, vectors: [C]).
    After ModelCNN packs the model, `dev` / `grad` / `mom` are views into the flat device buffers holding the
    kernel layout (conv: [Kp][R][Sp][Cp] flipped taps, zero padded) and get_value()/set_value() convert.
    """

    def __init__(self, value, name="", kind="vector", dev_shape=None, s_real=None):
        self.value = numpy.ascontiguousarray(value, dtype=numpy.float32)
        self.name = name
        self.kind = kind  # "conv" | "vector"
        self.dev_shape = tuple(dev_shape) if dev_shape is not None else tuple(self.value.shape)
        self.s_real = s_real
        self.dev = None
        self.grad = None
        self.mom = None

    @property
    def dev_size(self):
        return int(numpy.prod(self.dev_shape))

    def to_dev_layout(self, v=None):
        v = self.value if v is None else numpy.asarray(v, dtype=numpy.float32)
        out = numpy.zeros(self.dev_shape, dtype=numpy.float32)
        if self.kind == "conv":
            K, C, R, S = v.shape
            # true convolution -> correlation taps: w_dev[k, r, s, c] = omega[k, c, R-1-r, S-1-s]
            out[:K, :R, :S, :C] = v[:, :, ::-1, ::-1].transpose(0, 2, 3, 1)
        else:
            out[: v.shape[0]] = v
        return out

    def from_dev_layout(self, d):
        d = numpy.asarray(d, dtype=numpy.float32).reshape(self.dev_shape)
        if self.kind == "conv":
            K, C, R, S = self.value.shape
            return numpy.ascontiguousarray(d[:K, :R, :S, :C].transpose(0, 3, 1, 2)[:, :, ::-1, ::-1])
        return numpy.ascontiguousarray(d[: self.value.shape[0]])

    def get_value(self, borrow=False):
        if self.dev is not None:
            self.value = self.from_dev_layout(self.dev.detach().cpu().numpy())
        return self.value

    def set_value(self, v, borrow=False):
        v = numpy.ascontiguousarray(v, dtype=numpy.float32)
        assert v.shape == self.value.shape, (self.name, v.shape, self.value.shape)
        self.value = v
        if self.dev is not None:
            import torch
            from .. import ops
            self.dev.copy_(torch.from_numpy(self.to_dev_layout()).reshape(self.dev.shape))
            ops.bump_weights_version()

    def get_grad(self):
        """gradient in the reference layout (tests)"""
        return self.from_dev_layout(self.grad.detach().cpu().numpy())


def import_json(json_layers, x, x_shape, layer_range=None):
    """rebuild a layer list from the `.mdl.gz` layer dictionaries (denet/layer/__init__.py:31-60)"""
    if layer_range is None:
        start, end = 0, len(json_layers)
    elif type(layer_range) is tuple:
        start, end = layer_range[0], min(len(json_layers), layer_range[1])
    elif type(layer_range) is int:
        start, end = 0, min(len(json_layers), layer_range)
    else:
        raise Exception("Unknown layer range format:", layer_range)
    from .layer_types import layer_types
    layers = [InitialLayer(x, x_shape)]
    for layer_json in json_layers[start:end]:
        layer = None
        for layer_type in layer_types:
            if layer_json["type"] == layer_type.type_name:
                layer = layer_type(layers, json_param=layer_json)
                break
        assert layer is not None, "ERROR Unknown layer type: " + layer_json["type"]
        layer.import_json(layer_json)
        layers.append(layer)
    return layers


class AbstractLayer(object):
    type_name = "abstract"

    def __init__(self, layer_index, has_split=False):
        self.output = self.input = None
        self.output_shape = self.input_shape = None
        self.has_split = has_split
        self.layers = []
        self.layer_index = layer_index

    def __str__(self):
        param = {"int": [], "str": [], "float": [], "bool": [], "tuple": []}
        for k, v in self.__dict__.items():
            if k in ("has_split", "layer_index", "output_shape"):
                continue
            if type(v) is int:
                param["int"].append(k + ": %i" % v)
            elif type(v) is str:
                param["str"].append(k + ": " + v)
            elif type(v) is float:
                param["float"].append(k + ": %.3f" % v)
            elif type(v) is bool:
                param["bool"].append(k + ": %s" % v)
            elif type(v) is tuple:
                param["tuple"].append(k + ": " + str(v))

        def fmt(t):
            return (" " + " ".join(sorted(param[t]))) if param[t] else ""

        return "%i:" % self.layer_index + self.type_name + " - " + fmt("tuple") + fmt("str") + fmt("int") + fmt("float") + fmt("bool")

    # ---- reference protocol ----
    def weights(self):
        return sum([x.weights() for x in self.layers], [])

    def biases(self):
        return sum([x.biases() for x in self.layers], [])

    def updates(self, cost=None):
        """running statistics refreshed by the training step (BN mean / stdinv Params)"""
        return sum([x.updates(cost) for x in self.layers], [])

    def params(self):
        return self.weights() + self.biases()

    def cost(self, yt_index, yt_value):
        """reference: returns the symbolic cost; here: True if this layer contributes a cost term"""
        return None

    def get_target(self, model, samples, metas):
        return None

    def export_json(self):
        return {"type": type(self).type_name, "layers": [layer.export_json() for layer in self.layers]}

    def import_json(self, json_param):
        if "layers" in json_param:
            for i, json_layer in enumerate(json_param["layers"]):
                self.layers[i].import_json(json_layer)

    # ---- executor protocol (this build) ----
    def begin_step(self, metas):
        """called for every layer at the start of a training forward pass, before any kernel is queued: host work
        on the metas that needs no device result (default: nothing)"""

    def prepare_target(self, ctx, model, data_x, metas):
        """host-side target construction right before this layer's forward; layers with a faster internal
        representation override this, the default goes through the reference-format get_target()"""
        target = self.get_target(model, data_x, metas)
        if target is not None:
            self.set_target(ctx, target[0], target[1])

    def forward(self, ctx):
        """compute self.output.data from self.input.data (ctx: denet_amd.model.model_cnn.StepContext)"""
        raise NotImplementedError(type(self).__name__)

    def backward(self, ctx):
        """consume self.output.grad, accumulate into self.input.grad and the Param.grad views"""
        raise NotImplementedError(type(self).__name__)


class InitialLayer(AbstractLayer):
    type_name = "initial"

    def __init__(self, x, x_shape, json_param={}):
        super().__init__(layer_index=0)
        self.output = self.input = x
        self.output_shape = self.input_shape = tuple(x_shape)

    def forward(self, ctx):
        pass

    def backward(self, ctx):
        pass


class IdentityLayer(AbstractLayer):
    type_name = "identity"

    def __init__(self, layers, json_param={}):
        super().__init__(layer_index=len(layers))
        self.output = self.input = layers[-1].output
        self.output_shape = self.input_shape = layers[-1].output_shape

    @staticmethod
    def parse_desc(layers, name, tags, params):
        return False

    def forward(self, ctx):
        pass

    def backward(self, ctx):
        pass
