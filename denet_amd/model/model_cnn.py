"""ModelCNN — model-description parser, graph builder and trainer of the DeNet hot path.

Mirrors denet/model/model_cnn.py of the reference: build_layer / build (:122-157, the `TYPE.TAGS[a,b,..]`
mini-language), export_json / import_json (:159-203, `.mdl.gz` version 3), build_train_func (:205-405: cost
collection, L2 decay on weights only, sgd / torch==nesterov solvers, BN running-stat updates), train_step
(:407-445: get_target on every layer in order, then the device step; returns (cost, [per-layer costs])).

The Theano function compilation is replaced by a static executor: layers are run in order (forward), the cost
layers seed the gradients, layers are run in reverse (backward), gradients are (optionally) all-reduced over
RCCL and one fused solver kernel updates the flat parameter buffer. Every arithmetic step is a HIP kernel
reached through the C-ABI (denet_amd/ops.py); torch provides device memory, streams and the process group.
"""
import getpass
import math
import random

import os
import numpy

from .. import common
from .. import layer as layer_mod
from ..common import json_util, logging
from ..layer import Act, InitialLayer, round_up
from ..layer.layer_types import layer_types

SOLVER_MODES = {"sgd": 0, "torch": 1, "nesterov": 1, "adam": 2}


def load_from_json(json_obj, batch_size=32, layer_range=None):
    model = ModelCNN()
    model.batch_size = batch_size
    model.import_json(json_obj, layer_range)
    return model


def load_from_file(fname, batch_size=32, layer_range=None):
    model = load_from_json(json_util.json_from_gz(fname), batch_size, layer_range)
    model.fname = fname
    return model


def save_to_file(model, fname, compresslevel=9):
    json_util.json_to_gz(fname, model.export_json(), compresslevel)


def initialize(args, data_shape, class_labels, class_num):
    """model from CLI arguments (model_cnn.py:46-77)"""
    if args.model is None:
        model = ModelCNN()
        model.batch_size = args.batch_size
        model.class_labels = class_labels
        model.class_num = class_num
        try:
            n = int(args.border_mode)
            border_mode = (n, n)
        except ValueError:
            border_mode = args.border_mode
        model.build(args.model_desc, data_shape, args.activation, border_mode, list(args.weight_init))
    else:
        model = load_from_file(args.model, args.batch_size)
        model.class_labels = class_labels
        model.class_num = class_num
        assert tuple(data_shape) == tuple(model.data_shape), "Mismatching data shapes in .mdl and data"
    model.skip_layer_updates = getattr(args, "skip_layer_updates", [])
    return model


class _collector_paused:
    """the step loop of an epoch runs with Python's cyclic collector paused (a full collection stalls the host thread for
    milliseconds in the middle of a 30 ms step whose device queue the host feeds); one collection at the end of the epoch.
    Reference-counted garbage (every tensor view of a step) is freed as always. bench.py times its steps the same way."""

    def __enter__(self):
        import gc
        self.was = gc.isenabled()
        gc.disable()

    def __exit__(self, *exc):
        import gc
        if self.was:
            gc.enable()
            gc.collect()
        return False


def walk_layers(layers):
    """every layer and nested sub-layer, depth first, each object once"""
    seen, out = set(), []

    def rec(l):
        if id(l) in seen:
            return
        seen.add(id(l))
        out.append(l)
        for s in getattr(l, "layers", []):
            rec(s)

    for l in layers:
        rec(l)
    return out


class StepContext:
    def __init__(self, model):
        self.model = model
        self.has_sparse = any(l.type_name == "denet-sparse" for l in model.layers)


class ModelCNN:
    def __init__(self):
        self.batch_size = 0
        self.iteration = 0
        self.class_labels = None
        self.data_shape = None
        self.class_num = 0
        self.rng_seed = random.randint(1, 9999)
        layer_mod.set_rng_seed(self.rng_seed)
        self.gradient_clip = 0.0
        self.skip_layer_updates = []
        self.bias_decay = False
        self.layers = []
        self.func = {}
        self.input = None
        self.dist = None           # denet_amd.multi.DataParallel or None
        self._packed = False
        self.timing = {}

    # ---------------------------------------------------------------- construction
    def get_input_shape(self):
        assert self.data_shape is not None, "Data shape hasn't been set!"
        return tuple([self.batch_size] + list(self.data_shape))

    def get_output_shape(self):
        return self.layers[-1].output_shape

    def get_parameter_num(self):
        n = 0
        for layer in self.layers:
            for param in layer.params():
                n += param.value.size
        return n

    def _make_input(self):
        shape = self.get_input_shape()
        cp = 4 if shape[1] <= 4 else round_up(shape[1], 32)
        act = Act(shape, cp, "input")
        act.requires_grad = False
        self.input = act
        return act

    def build_layer(self, layer_desc, layers, activation, border_mode, wb):
        p_start = layer_desc.find("[")
        p_end = layer_desc.find("]")
        layer_params = {"classNum": self.class_num, "activation": activation, "borderMode": border_mode, "wb": wb}
        if p_start > 0 and p_end > p_start:
            layer_type = layer_desc[:p_start]
            for i, p in enumerate(layer_desc[(p_start + 1):p_end].split(",")):
                layer_params[i] = common.convert_num(p)
        else:
            layer_type = layer_desc
        t_index = layer_type.find(".")
        if t_index > 0:
            layer_tags = layer_type[(t_index + 1):]
            layer_type = layer_type[:t_index]
        else:
            layer_tags = ""
        for layer in layer_types:
            if layer.parse_desc(layers, layer_type, layer_tags, layer_params):
                return
        raise Exception("Invalid layer - type: ", layer_type, "tags:", layer_tags, "params:", layer_params)

    def build(self, model_desc, data_shape, activation="relu", border_mode="valid", weight_init="he-forward"):
        if isinstance(model_desc, str):
            model_desc = model_desc.split()
        if isinstance(weight_init, str):
            weight_init = [weight_init]
        self.model_desc = " ".join(model_desc)
        self.data_shape = tuple(data_shape)
        self.layers = [InitialLayer(self._make_input(), self.get_input_shape())]
        for i, layer_desc in enumerate(model_desc):
            wb = weight_init[min(len(weight_init) - 1, i)]
            self.build_layer(layer_desc, self.layers, activation, border_mode, wb)
        self._packed = False

    def export_json(self):
        from time import gmtime, strftime
        json_layers = [self.layers[index].export_json() for index in range(1, len(self.layers))]
        json_obj = {"classifierType": "CNN",
                    "classLabels": self.class_labels,
                    "classNum": self.class_num,
                    "dataShape": self.data_shape,
                    "date": strftime("%Y-%m-%d %H:%M:%S", gmtime()),
                    "user": getpass.getuser()}
        json_obj.update({"version": 3, "layers": json_layers})
        return json_obj

    def import_json(self, json_obj, layer_range=None):
        self.func = {}
        if json_obj.get("version", 0) == 0:
            raise Exception("Old format model file detected, no compatibility!")
        self.class_labels = json_obj["classLabels"]
        if "imageSize" in json_obj and "imageMode" in json_obj:
            width, height = json_obj["imageSize"][0], json_obj["imageSize"][1]
            image_mode = json_obj.get("imageMode", "RGB")
            self.data_shape = ({"RGB": 3, "L": 1}[image_mode], width, height)
        elif "dataShape" in json_obj:
            self.data_shape = tuple(json_obj["dataShape"])
        else:
            assert False, "Bad mdl file, Cannot determine input data shape!"
        assert json_obj.get("imageBorder", 0) == 0
        self.class_num = json_obj.get("classNum", len(self.class_labels) if self.class_labels else 0)
        self.layers = layer_mod.import_json(json_obj["layers"], self._make_input(), self.get_input_shape(), layer_range)
        self._packed = False

    # ---------------------------------------------------------------- device state
    def pack_device(self):
        """allocate the flat device buffers (parameters, gradients, momentum, BN statistics) and hand each
        Param its views; layout = [weights (layer order) | biases (layer order) | frozen]"""
        import torch
        from .. import host_tuning
        host_tuning()
        weights, biases = [], []
        for layer in self.layers:
            weights += layer.weights()
            biases += layer.biases()
        trainable = set(id(p) for p in weights + biases)
        frozen, stats = [], []
        for l in walk_layers(self.layers):
            for p in getattr(l, "all_params", lambda: [])():
                if id(p) not in trainable and all(p is not q for q in frozen):
                    frozen.append(p)
        for layer in self.layers:
            stats += layer.updates(None)
        order = weights + biases + frozen

        def align(n):
            return round_up(n, 64)

        total = sum(align(p.dev_size) for p in order)
        self.P = torch.zeros(total, device="cuda")
        self.G = torch.zeros(total, device="cuda")
        self.M = torch.zeros(total, device="cuda")
        host = numpy.zeros(total, dtype=numpy.float32)
        off = 0
        self.param_ranges = {}
        for p in order:
            n = p.dev_size
            host[off:off + n] = p.to_dev_layout().reshape(-1)
            p.dev = self.P[off:off + n]
            if id(p) in trainable:
                p.grad = self.G[off:off + n]
                p.mom = self.M[off:off + n]
            self.param_ranges[id(p)] = (off, off + n)
            off += align(n)
        self.P.copy_(torch.from_numpy(host))
        self.n_weights = sum(align(p.dev_size) for p in weights)
        self.n_trainable = sum(align(p.dev_size) for p in weights + biases)
        # per top-level layer range inside the weights region (for bucketed all-reduce)
        self.layer_weight_range = []
        for layer in self.layers:
            ws = layer.weights()
            if ws:
                self.layer_weight_range.append((layer, self.param_ranges[id(ws[0])][0],
                                                align(self.param_ranges[id(ws[-1])][1])))
        stotal = sum(align(p.dev_size) for p in stats)
        self.S = torch.zeros(max(stotal, 1), device="cuda")
        shost = numpy.zeros(max(stotal, 1), dtype=numpy.float32)
        off = 0
        for p in stats:
            n = p.dev_size
            shost[off:off + n] = p.to_dev_layout().reshape(-1)
            p.dev = self.S[off:off + n]
            off += align(n)
        self.S.copy_(torch.from_numpy(shost))
        self.acts = []
        seen = set()
        for l in walk_layers(self.layers):
            for a in (l.input, l.output):
                if isinstance(a, Act) and id(a) not in seen:
                    seen.add(id(a))
                    self.acts.append(a)
        self.cost_buf = torch.zeros(16, device="cuda")
        self._packed = True
        from .. import ops
        ops.bump_weights_version()

    def build_train_func(self, solver_mode="sgd", cost_factors=[], use_acc_mode=False, skip_build=False):
        if solver_mode not in SOLVER_MODES:
            raise NotImplementedError("unknown solver '%s' (sgd, torch, nesterov, adam)" % solver_mode)
        if use_acc_mode and solver_mode == "adam":
            raise NotImplementedError("--use-acc-mode averages would-be updates; that equals one update with the mean "
                                      "gradient only for the linear solvers (sgd, torch, nesterov)")
        self.use_acc_mode = bool(use_acc_mode)
        self._acc = None
        self.solver_mode = solver_mode
        import torch
        if torch.cuda.is_available():
            from .. import ops
            ops.init_streams()
        self.cost_layers = []
        self.cost_layer_names = []
        for layer in self.layers:
            if layer.cost(None, None) is not None:
                self.cost_layers.append(layer)
                self.cost_layer_names.append(layer.type_name)
        self.cost_factors = [1.0] * len(self.cost_layers) if len(cost_factors) == 0 else [float(c) for c in cost_factors]
        assert len(self.cost_factors) == len(self.cost_layers), \
            "Different number of cost factors (%i) and cost layers (%i)" % (len(self.cost_factors), len(self.cost_layers))
        assert len(self.cost_layers) <= 8
        if self.gradient_clip > 0.0:
            raise NotImplementedError("gradient clipping is outside the hot path")
        self.use_split_mode = False   # split points are identities here (288 GB of HBM)
        # `BN A` pairs whose batch-norm output nobody else reads run as the fused BN + ReLU (ActivationLayer; DENET_BN_ACT_FUSE=0: apart)
        for l in walk_layers(self.layers):
            if l.type_name == "batchnorm" and getattr(l, "act_behind", None) is not None:
                l.act_fused = os.environ.get("DENET_BN_ACT_FUSE", "1") != "0" and self._consumers(l.output) == 1
        # a max pool that is the only reader of a BN + ReLU layer's output (the ResNet stem): the two run as one pass in training
        for a, b in zip(self.layers[:-1], self.layers[1:]):
            a.pool_behind = None
            if a.type_name == "batchnorm-relu" and b.type_name == "pool" and b.mode == "max" and b.input is a.output \
                    and self._consumers(a.output) == 1:
                a.pool_behind = b
        for a, act, b in zip(self.layers[:-2], self.layers[1:-1], self.layers[2:]):
            # the same for `BN A P` written as three layers: the batch norm writes the fused activation's output
            if a.type_name == "batchnorm" and getattr(a, "act_fused", False) and a.act_behind is act and b.type_name == "pool" \
                    and b.mode == "max" and b.input is act.output and self._consumers(act.output) == 1:
                a.pool_behind = b
        # a SKIP layer that adds its tap (same channel count: no projection) to the output of the convolution right in front of it
        # (the up-sampling path of the skip models, skip.py:81-86): the addition goes into that convolution's epilogue, and with it the statistics of the batch
        # norm behind the SKIP layer - no pass of its own over the sum (DENET_SKIP_FUSE=0: separate passes)
        for a, b in zip(self.layers[:-1], self.layers[1:]):
            a.skip_behind = None
            if (os.environ.get("DENET_SKIP_FUSE", "1") != "0" and a.type_name == "conv" and b.type_name == "skip"
                    and getattr(b, "combine_mode", None) == "proj-add" and len(getattr(b, "layers", [])) <= 1 and b.x is a.output
                    and self._consumers(a.output) == 1 and not a.use_bias):
                a.skip_behind = b
        if not skip_build:
            self.pack_device()
        self.func["train_step"] = self._device_step

    # ---------------------------------------------------------------- execution
    def _upload_input(self, data_x):
        import torch
        from .. import ops
        if isinstance(data_x, torch.Tensor) and data_x.is_cuda and tuple(data_x.shape) == self.input.phys_shape() \
                and tuple(data_x.shape) != self.get_input_shape():
            # a batch rendered on the device (denet_amd/dataset/device_render.py): already NHWC with padded channels
            self.input.data = data_x if data_x.is_contiguous() else data_x.contiguous()
            return
        if isinstance(data_x, torch.Tensor):
            x = data_x if data_x.is_cuda else data_x.cuda(non_blocking=True)
            if x.dtype != torch.float32:
                x = x.float()                      # the first layer's kernels take raw float pointers
        else:
            x = torch.from_numpy(numpy.ascontiguousarray(data_x, dtype=numpy.float32)).cuda(non_blocking=True)
        assert tuple(x.shape) == self.get_input_shape(), (tuple(x.shape), self.get_input_shape())
        # the NHWC copy is made when a layer asks for it: the first convolution's own kernels read the planar batch (ops.NchwLink)
        # A caller's device tensor is read IN PLACE, and last of all by the first layer's filter gradient on the second stream at
        # the very end of the backward sweep, after train_step has returned: the buffer must stay untouched until
        # `self.input_consumed` (an event recorded behind the solver, which waits for that stream) has completed.
        x = x.contiguous()
        if ops._WGRAD_STREAM is not None:
            x.record_stream(ops._WGRAD_STREAM)     # the caching allocator must not hand the block out while that stream reads it
        self.input.set_pending_data(ops.NchwLink(x, self.input.cp))

    def _consumers(self, act):
        """number of layers (nested ones included) that read `act` as their input or as a skip tap"""
        counts = self.__dict__.get("_consumer_counts")
        if counts is None:
            counts = self.__dict__["_consumer_counts"] = {}
            for l in walk_layers(self.layers):
                for a in {id(getattr(l, k)): getattr(l, k) for k in ("input", "x", "y", "skip") if getattr(l, k, None) is not None}.values():
                    if a is not getattr(l, "output", None) or l.type_name == "skip-src":
                        counts[id(a)] = counts.get(id(a), 0) + 1
        return counts.get(id(act), 0)

    def forward(self, data_x, data_m=None, train=True):
        """runs the layers in order; in training mode get_target of layer i is called right before its forward
        (so DNS sees the corner map of this very pass and DND sees the edited RoI list)"""
        layer_mod.set_train(train)
        if not self._packed:
            self.pack_device()
        for a in self.acts:
            a.grad = None
        from .. import ops
        step_begin = None
        if train:
            import torch
            step_begin = torch.cuda.Event()      # everything before this step (the solver update of the weights) is behind it
            step_begin.record()
        self._upload_input(data_x)
        ctx = StepContext(self)
        fold = (not train) and ops.INFER_FOLD
        skip_next = False
        for i, layer in enumerate(self.layers[1:]):
            if skip_next:          # a batch norm folded into the convolution in front of it (inference)
                skip_next = False
                continue
            if fold and layer.type_name == "conv" and layer.enabled and i + 2 < len(self.layers):
                nxt = self.layers[i + 2]
                if nxt.type_name in ("batchnorm", "batchnorm-relu") and nxt.enabled and nxt.input is layer.output \
                        and self._consumers(layer.output) == 1:
                    # folded only when the batch norm is the ONLY reader of the convolution's output: the folded pass
                    # writes normalised values into it, any other consumer (skip source, split, detection tail) needs the raw ones
                    act = nxt.act_behind if getattr(nxt, "act_fused", False) else None      # `BN A` as one pass: ActivationLayer
                    layer.forward_folded(ctx, nxt, relu=nxt.type_name == "batchnorm-relu" or act is not None,
                                         out_act=act.output if act is not None else None)
                    skip_next = True
                    continue
            if train and data_m is not None:
                layer.prepare_target(ctx, self, data_x, data_m)
            layer.forward(ctx)
            if train and i == 0:
                # filters of the Winograd passes: transformed for all layers on a side stream while the stem runs (the
                # ~60 launches are queued behind the first convolution so that the compute stream never waits for them)
                from .. import ops
                convs = getattr(self, "_conv_layers", None)
                if convs is None:
                    convs = self._conv_layers = [l for l in walk_layers(self.layers) if l.type_name == "conv" and
                                                 getattr(l, "enabled", True)]
                ops.wino_prefetch_filters([(l._cache(), l._w()) for l in convs], after=step_begin)
            if train and data_m is not None and i == 0:
                # host work that needs no device result (corner targets, ...): done while the first layer runs, so the
                # device is not left idle in front of it
                for other in self.layers[2:]:
                    other.begin_step(data_m)
        return ctx

    def backward(self, ctx):
        """cost layers seed the gradients, then the reverse sweep; all-reduce buckets are launched as soon as the
        last layer of a bucket has produced its weight gradient"""
        import torch
        from .. import ops
        for i, layer in enumerate(self.cost_layers):
            layer.loss_backward(ctx, self.cost_buf[2 * i:2 * i + 2])
            if self.cost_factors[i] != 1.0:
                g = layer.dconv if layer.type_name == "denet-corner" else (
                    layer.conv.output.grad if hasattr(layer, "conv") else layer.input.grad)
                ops._L().denet_scale(g.data_ptr(), g.numel(), float(self.cost_factors[i]), ops.stream_ptr())
        # the costs are final here (the loss kernels are queued): copy them to the host on a side stream now, so that
        # train_step can return them without waiting for the backward sweep and the solver - the host then prepares the
        # next step while the device finishes this one
        if getattr(self, "_cost_host", None) is None:
            self._cost_host = torch.empty(16, dtype=torch.float32).pin_memory()
            self._cost_stream = ops.side_stream(3)
        self._cost_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._cost_stream):
            self._cost_host.copy_(self.cost_buf, non_blocking=True)
            self._cost_ready = torch.cuda.Event()
            self._cost_ready.record(self._cost_stream)
        dist = self.dist
        if dist is not None:
            dist.begin_step(self)
        for layer in reversed(self.layers[1:]):
            layer.backward(ctx)
            if dist is not None:
                dist.layer_done(self, layer)
        ops.join_wgrad_stream()           # filter / bias gradients (second stream) before all-reduce tail and solver
        if dist is not None:
            dist.finish_step(self)

    # ---- --use-acc-mode (model_cnn.py:374-392): train_begin zeroes accumulators, every train_step ADDS the values the
    # update targets would take (parameters, momentum, BN running statistics - all computed from the state at train_begin)
    # instead of applying them, train_end installs their mean. For the linear solvers that is one update with the mean
    # gradient and the mean of the would-be running statistics, which is how it is computed here: gradients and would-be
    # statistics are accumulated, the solver runs once in train_end.
    def train_begin(self):
        import torch
        assert self.use_acc_mode, "build_train_func(use_acc_mode=True) first"
        if not self._packed:
            self.pack_device()
        self._acc = {"G": torch.zeros_like(self.G), "S": torch.zeros_like(self.S), "S0": self.S.clone(), "n": 0, "args": None}

    def train_end(self):
        from .. import ops
        acc = self._acc
        assert acc is not None and acc["n"] > 0, "train_end without accumulated steps"
        it, learn_rate, momentum, decay = acc["args"]
        n_decay = self.n_trainable if self.bias_decay else self.n_weights
        scale = 1.0 / acc["n"]
        ops.solver_step(self.P[:self.n_trainable], self.M[:self.n_trainable], acc["G"][:self.n_trainable], n_decay,
                        float(learn_rate), float(momentum[0]), it, float(decay), SOLVER_MODES[self.solver_mode], scale)
        self.S.copy_(acc["S"])
        ops.check(ops._L().denet_scale(self.S.data_ptr(), self.S.numel(), scale, ops.stream_ptr()), "scale")
        ops.bump_weights_version()
        self._acc = None

    def _device_step(self, epoch, it, learn_rate, momentum, decay, data_x, data_m, fetch_cost=True):
        from .. import ops
        acc = self._acc if getattr(self, "use_acc_mode", False) else None
        if getattr(self, "use_acc_mode", False) and acc is None:
            raise Exception("--use-acc-mode: call train_begin() before train_step()")
        if acc is not None:
            self.S.copy_(acc["S0"])            # every sub-step starts from the statistics at train_begin
        ctx = self.forward(data_x, data_m, train=True)
        self.backward(ctx)
        # host work a layer has left for later (the RoI list's bookkeeping after a short hand-off, DeNetSparseLayer._resolve_edit) is
        # done before the step returns at the latest: the caller's own draws from `random` must find the generator where the
        # reference's step would have left it
        for layer in walk_layers(self.layers):
            fin = getattr(layer, "_resolve_edit", None)
            if fin is not None:
                fin()
        scale = 1.0 / self.dist.world_size if self.dist is not None else 1.0
        n_decay = self.n_trainable if self.bias_decay else self.n_weights
        if acc is not None:
            ops.add(acc["G"], self.G, out=acc["G"])
            ops.add(acc["S"], self.S, out=acc["S"])
            acc["n"] += 1
            acc["args"] = (it, learn_rate, momentum, decay)
        elif self.solver_mode == "adam":
            import torch
            assert len(momentum) >= 2, "adam takes momentum = (beta1, beta2)"
            if getattr(self, "V", None) is None:
                self.V = torch.zeros_like(self.M)       # second-moment accumulators (model_cnn.py:299)
            ops.solver_adam(self.P[:self.n_trainable], self.M[:self.n_trainable], self.V[:self.n_trainable],
                            self.G[:self.n_trainable], n_decay, float(learn_rate), float(momentum[0]),
                            float(momentum[1]), it, float(decay), scale)
        else:
            ops.solver_step(self.P[:self.n_trainable], self.M[:self.n_trainable], self.G[:self.n_trainable], n_decay,
                            float(learn_rate), float(momentum[0]), it, float(decay), SOLVER_MODES[self.solver_mode],
                            scale)
        ops.bump_weights_version()       # parameters and BN running statistics moved: inference caches are stale
        import torch
        if getattr(self, "input_consumed", None) is None:
            self.input_consumed = torch.cuda.Event()
        self.input_consumed.record()     # behind the solver: every reader of this step's input batch (both streams) is done
        if not fetch_cost:
            return None
        self._cost_ready.synchronize()
        costs = self._cost_host[:2 * len(self.cost_layers)].numpy().reshape(-1, 2).copy()
        layer_costs = []
        for i, layer in enumerate(self.cost_layers):
            c = float(costs[i, 0]) + (float(costs[i, 1]) if layer.type_name == "denet-detect" else 0.0)
            layer_costs.append(c)
        total = sum(f * c for f, c in zip(self.cost_factors, layer_costs))
        self.last_cost_terms = costs
        return [total] + layer_costs

    def train_step(self, data_x, data_m, epoch, it, learning_rate, momentum, decay, fetch_cost=True):
        """same signature and return value as the reference (model_cnn.py:407-445)"""
        layer_mod.set_iteration(it)
        layer_mod.set_epoch(epoch)
        layer_mod.set_rng_seed(self.rng_seed)      # several models may live in one process
        momentum = numpy.array(momentum, dtype=numpy.float32).reshape(-1)
        costs = self.func["train_step"](epoch, it, learning_rate, momentum, decay, data_x, data_m, fetch_cost)
        if costs is None:
            return None, []
        return costs[0], costs[1:]

    def train_epoch(self, dataset, epoch, learning_rate, momentum=[0, 1, 0], decay=0.0, solver_mode="sgd"):
        dataset_x, dataset_m, dataset_size = dataset.export(self.batch_size)
        index_num = math.ceil(dataset_size / self.batch_size)
        with _collector_paused():
            return self._train_epoch_steps(dataset_x, dataset_m, index_num, epoch, learning_rate, momentum, decay)

    def _train_epoch_steps(self, dataset_x, dataset_m, index_num, epoch, learning_rate, momentum, decay):
        total_cost = 0
        for index in range(index_num):
            timer = common.Timer()
            data_x = dataset_x[index * self.batch_size:(index + 1) * self.batch_size]
            data_m = dataset_m[index * self.batch_size:(index + 1) * self.batch_size]
            cost, _ = self.train_step(data_x, data_m, epoch, self.iteration, learning_rate, momentum, decay)
            if math.isnan(cost):
                raise Exception("ERROR: Cost is NaN")
            # model_cnn.py:466. train_step returns once the costs are on the host; the device may still be in the backward
            # sweep, so "took" is the host's time per step (the steady-state step time once the queue is full)
            if logging.verbose_enabled():
                logging.verbose("Batch %i.%i - iteration: %i cost:" % (epoch, index * self.batch_size, self.iteration), cost,
                                "took: %i ms" % timer.current_ms())
            total_cost += cost
            self.iteration += 1
        return total_cost

    def train_epoch_device(self, loader, images, epoch, learning_rate, momentum=[0, 1, 0], decay=0.0):
        """train_epoch over batches rendered on the GPU by a denet_amd.dataset.device_render.DeviceImageLoader: same
        batches, metas and random-stream use as `train_epoch(dataset)` after `dataset.load_from_subset`"""
        total_cost = 0
        with _collector_paused():
            for data_x, data_m in loader.iterate(images, self.batch_size):
                cost, _ = self.train_step(data_x, data_m, epoch, self.iteration, learning_rate, momentum, decay)
                if math.isnan(cost):
                    raise Exception("ERROR: Cost is NaN")
                total_cost += cost
                self.iteration += 1
        return total_cost

    def predict_output(self, dataset):
        """last-layer output for every sample of the loaded subset, padding of the last batch cropped
        (reference model_cnn.py:484-508)"""
        dataset_x, _, dataset_size = dataset.export(self.batch_size)
        n = math.ceil(dataset_size / self.batch_size)
        pr = numpy.concatenate([self.predict_output_step(dataset_x[i * self.batch_size:(i + 1) * self.batch_size])
                                for i in range(n)], axis=0)
        return pr[:dataset_size]

    def predict_label(self, dataset):
        pr = self.predict_output(dataset)
        assert pr.ndim == 2
        return [int(numpy.argmax(pr[i, ...])) for i in range(pr.shape[0])]

    def predict_output_step(self, data_x):
        """inference forward (BN in test mode); returns the last layer's output as a numpy array in the reference's
        NCHW convention (class probabilities for a regression head)"""
        from .. import ops
        self.forward(data_x, None, train=False)
        last = self.layers[-1]
        out = last.output.data
        if last.type_name == "regression":
            # regression.py:44-50: output = exp(log_softmax(x)); B x C values, evaluated on the host
            B, C = last.output_shape
            logits = out.view(B, last.input.cp).cpu().numpy()[:, :C].astype(numpy.float32)
            xdev = logits - logits.max(axis=1, keepdims=True)
            return numpy.exp(xdev - numpy.log(numpy.exp(xdev).sum(axis=1, keepdims=True)))
        if out.dim() == 4:
            return ops.nhwc_to_nchw(out, last.output_shape[1]).cpu().numpy()
        return out.cpu().numpy()
