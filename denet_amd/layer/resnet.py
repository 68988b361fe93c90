"""`RSN` / `nRSN` — residual block. Mirrors denet/layer/resnet.py (ResnetLayer :13-167): original (post-activation)
or pre-activation design, basic or bottleneck body, optional 1x1 projection shortcut (+BN for "original").
The block is composed of the same sub-layers as the reference (so weights()/JSON line up); the executor fuses
the residual add + ReLU into the last normalisation kernel (resnet.py:109-113)."""
from . import AbstractLayer, Act, InitialLayer
from .activation import ActivationLayer
from .batch_norm import BatchNormLayer
from .batch_norm_relu import BatchNormReluLayer
from .convolution import ConvLayer


class ResnetLayer(AbstractLayer):
    type_name = "resnet"

    def __init__(self, layers, filter_shape=None, stride=(1, 1), bottleneck=0, activation="relu", version="original",
                 json_param={}):
        super().__init__(layer_index=len(layers))
        self.input = layers[-1].output
        self.input_shape = layers[-1].output_shape
        self.filter_shape = tuple(json_param.get("shape", filter_shape))
        self.stride = tuple(json_param.get("stride", stride))
        self.bottleneck = json_param.get("bottleneck", bottleneck)
        self.version = json_param.get("version", version)
        self.activation = json_param.get("activation", activation)
        self.bn_json_param = json_param.get("bnParam", {"enabled": json_param.get("enableBatchNorm", True)})

        fs = self.filter_shape
        if self.bottleneck > 0:
            self.size = (fs[2], fs[3])
            shape0 = (self.bottleneck, fs[1], 1, 1)
            shape1 = (self.bottleneck, self.bottleneck, fs[2], fs[3])
            shape2 = (fs[0], self.bottleneck, 1, 1)
        else:
            self.size = (fs[2] * 2 - 1, fs[3] * 2 - 1)
            shape0 = fs
            shape1 = (fs[0], fs[0], fs[2], fs[3])
            shape2 = None

        fused = ("bnrelu" in self.version) and self.activation == "relu"

        def add_bn_act(ls):
            if fused:
                ls.append(BatchNormReluLayer(ls, json_param=self.bn_json_param))
            else:
                ls.append(BatchNormLayer(ls, json_param=self.bn_json_param))
                ls.append(ActivationLayer(ls, self.activation))

        self.layers = [InitialLayer(self.input, self.input_shape)]
        if "pre-activation" in self.version:
            add_bn_act(self.layers)
        self.layers.append(ConvLayer(self.layers, filter_shape=shape0, filter_stride=self.stride, border_mode="half",
                                     use_bias=False))
        add_bn_act(self.layers)
        self.layers.append(ConvLayer(self.layers, filter_shape=shape1, border_mode="half", use_bias=False))
        if self.bottleneck > 0:
            add_bn_act(self.layers)
            self.layers.append(ConvLayer(self.layers, filter_shape=shape2, border_mode="half", use_bias=False))
        if "pre-activation" not in self.version:
            self.layers.append(BatchNormLayer(self.layers, json_param=self.bn_json_param))
        self.n_main = len(self.layers)

        y_shape = self.layers[-1].output_shape
        if self.input_shape != y_shape:
            if "pre-activation" in self.version:
                input_layers = self.layers[0:2]
            else:
                input_layers = [InitialLayer(self.input, self.input_shape)]
            self.layers.append(ConvLayer(input_layers, filter_shape=(y_shape[1], self.input_shape[1], 1, 1),
                                         filter_stride=self.stride, use_bias=False, border_mode="half"))
            if "original" in self.version:
                self.layers.append(BatchNormLayer(self.layers, json_param=self.bn_json_param))
        self.output_shape = y_shape
        self.output = Act(self.output_shape, self.layers[self.n_main - 1].output.cp, "rsn%i" % self.layer_index)

    @staticmethod
    def parse_desc(layers, name, tags, params):
        if name == "RSN":
            version = "original" if "O" in tags else "pre-activation"
            filter_shape = (params.get(0), layers[-1].output_shape[1], params.get(1), params.get(1))
            filter_stride = (params.get(2, 1), params.get(2, 1))
            layers.append(ResnetLayer(layers, filter_shape, filter_stride, params.get(3, 0), params["activation"], version))
            return True
        elif name == "nRSN":
            version = "original" if "O" in tags else "pre-activation"
            bottleneck = params.get(4, 0)
            for i in range(params.get(0)):
                filter_shape = (params.get(1), layers[-1].output_shape[1], params.get(2), params.get(2))
                filter_stride = (params.get(3, 1), params.get(3, 1)) if i == 0 else (1, 1)
                layers.append(ResnetLayer(layers, filter_shape, filter_stride, bottleneck, params["activation"], version))
            return True
        return False

    def updates(self, cost=None):
        return sum([layer.updates(cost) for layer in self.layers], [])

    def weights(self):
        return sum([layer.weights() for layer in self.layers], [])

    def biases(self):
        return sum([layer.biases() for layer in self.layers], [])

    def import_json(self, json_param):
        n = 0
        for json_layer in json_param["layers"]:
            if json_layer["type"] == "identity":
                continue
            assert json_layer["type"] == self.layers[n].type_name, (json_layer["type"], self.layers[n].type_name)
            self.layers[n].import_json(json_layer)
            n += 1

    def export_json(self):
        json = super().export_json()
        json.update({"shape": self.filter_shape, "stride": self.stride, "bottleneck": self.bottleneck,
                     "bnParam": self.bn_json_param, "activation": self.activation, "version": self.version})
        json.update({"layers": [layer.export_json() for layer in self.layers]})
        return json

    # ---- execution ----
    def _main(self):
        return self.layers[1:self.n_main]

    def _shortcut(self):
        return self.layers[self.n_main:]

    def _fold_plan(self):
        """inference: [(conv, bn, relu, output tensor or None)] of the main path and of the shortcut when every batch norm sits directly behind
        a convolution (the `original` blocks with fused BN-ReLU), else None"""
        if "pre-activation" in self.version or not self.bn_json_param.get("enabled", True):
            return None
        main, sc = self._main(), self._shortcut()

        def pairs(ls):
            out, i = [], 0
            while i < len(ls):
                if ls[i].type_name != "conv" or i + 1 >= len(ls) or ls[i + 1].type_name not in ("batchnorm", "batchnorm-relu") \
                        or not ls[i].enabled or not ls[i + 1].enabled:
                    return None
                bn = ls[i + 1]
                if getattr(bn, "act_fused", False) and i + 2 < len(ls) and ls[i + 2] is bn.act_behind:
                    out.append((ls[i], bn, True, bn.act_behind.output))       # `BN A`: the folded pass writes the activation's output
                    i += 3
                    continue
                out.append((ls[i], bn, bn.type_name == "batchnorm-relu", None))
                i += 2
            return out
        pm, ps = pairs(main), pairs(sc)
        if pm is None or ps is None or pm[-1][2]:
            return None
        return pm, ps

    def forward(self, ctx):
        from . import get_train
        from .. import ops
        plan = None if get_train() or not ops.INFER_FOLD else self.__dict__.setdefault("_plan", self._fold_plan())
        if plan:
            pm, ps = plan
            for conv, bn, relu, oa in pm[:-1]:
                conv.forward_folded(ctx, bn, relu=relu, out_act=oa)
            for conv, bn, relu, oa in ps:
                conv.forward_folded(ctx, bn, relu=relu, out_act=oa)
            res = ps[-1][1].output.data if ps else self.input.data
            relu = self.activation in ("relu", "relu-safe")
            assert relu or self.activation == "none", self.activation
            pm[-1][0].forward_folded(ctx, pm[-1][1], add=res, relu=relu, out_act=self.output)
            return
        main, sc = self._main(), self._shortcut()
        for l in main[:-1]:
            l.forward(ctx)
        for l in sc:
            l.forward(ctx)
        res = sc[-1].output.data if sc else self.input.data
        if "pre-activation" in self.version:
            # output = x + y : the add rides in the last convolution's epilogue
            main[-1].forward(ctx, add=res)
            self.output.data = main[-1].output.data
        else:
            relu = self.activation in ("relu", "relu-safe")
            assert relu or self.activation == "none", self.activation
            main[-1].forward(ctx, res=res, relu=relu, out_act=self.output)

    def backward(self, ctx):
        main, sc = self._main(), self._shortcut()
        if "pre-activation" in self.version:
            dres = self.output.grad
            main[-1].output.grad = self.output.grad
            main[-1].backward(ctx)
        else:
            dres = main[-1].backward(ctx, want_dres=True)
        done = None
        if len(main) >= 2 and main[-2].output._pending_grad is not None:
            # the last batch norm left its pointwise backward pass (which also writes dres) to the convolution in front of it
            # (ops.BnLink): that convolution runs first, so that dres exists before the shortcut branch reads it
            done = main[-2]
            done.backward(ctx)
        if sc:
            sc[-1].output.grad = dres
            for l in reversed(sc):
                # the main branch's first convolution adds its data gradient to the block input's AFTER this branch: backward sums
                # of the batch norm in front of the block left here would be thrown away (ConvLayer.backward)
                l.not_last_writer = True
                l.backward(ctx)
        else:
            self.input.add_grad(dres)
        for l in reversed(main[:-1]):
            if l is not done:
                l.backward(ctx)
