"""`SKIPSRC` / `SKIP` — skip connections. Mirrors denet/layer/skip.py (SkipSrcLayer :9-57, SkipLayer :59-115).
A `.X` source is a split point in the reference (a Theano memory workaround, model_cnn.py:241-280); with 288 GB of
HBM the build keeps the flag for the JSON surface and treats the tap as a plain pass-through."""
from . import AbstractLayer, Act, InitialLayer
from .convolution import ConvLayer
from .. import ops


class SkipSrcLayer(AbstractLayer):
    type_name = "skip-src"

    def __init__(self, layers, skip_index=0, split=False, json_param={}):
        super().__init__(layer_index=len(layers))
        self.skip_index = json_param.get("index", skip_index)
        self.has_split = json_param.get("split", split)
        self.input = layers[-1].output
        self.input_shape = layers[-1].output_shape
        self.output_shape = self.input_shape
        self.skip = self.output = self.input

    def export_json(self):
        j = super().export_json()
        j.update({"index": self.skip_index, "split": self.has_split})
        return j

    @staticmethod
    def parse_desc(layers, name, tags, params):
        if name != "SKIPSRC":
            return False
        layers.append(SkipSrcLayer(layers, params.get(0, 0), "X" in tags))
        return True

    def forward(self, ctx):
        pass

    def backward(self, ctx):
        pass


class SkipLayer(AbstractLayer):
    type_name = "skip"

    def __init__(self, layers, skip_index=0, combine_mode="proj-add", json_param={}):
        super().__init__(layer_index=len(layers))
        self.combine_mode = json_param.get("combineMode", combine_mode)
        self.skip_index = json_param.get("index", skip_index)
        self.skip_layer = None
        for layer in layers:
            if layer.type_name == "skip-src" and layer.skip_index == self.skip_index:
                self.skip_layer = layer
                break
        assert self.skip_layer is not None
        self.input = self.x = layers[-1].output
        self.input_shape = self.x_shape = layers[-1].output_shape
        self.y = self.skip_layer.skip
        self.y_shape = self.skip_layer.output_shape
        if self.combine_mode == "proj-add":
            self.output_shape = self.x_shape
            if self.y_shape[1] != self.x_shape[1]:
                self.layers = [InitialLayer(self.y, self.y_shape)]
                self.layers.append(ConvLayer(self.layers, filter_shape=(self.x_shape[1], self.y_shape[1], 1, 1)))
            self.output = Act(self.output_shape, self.x.cp, "skip%i" % self.layer_index)
        elif self.combine_mode == "concat":
            # skip.py:93-96 (only reachable through the JSON key "combineMode"): channels of x, then of the tap
            self.output_shape = (self.x_shape[0], self.x_shape[1] + self.y_shape[1], self.x_shape[2], self.x_shape[3])
            self.output = Act(self.output_shape, None, "skip%i" % self.layer_index)
        else:
            raise Exception("Unknown combine mode: %s" % self.combine_mode)

    def export_json(self):
        j = super().export_json()
        j.update({"index": self.skip_index, "combineMode": self.combine_mode})
        return j

    @staticmethod
    def parse_desc(layers, name, tags, params):
        if name != "SKIP":
            return False
        layers.append(SkipLayer(layers, params.get(0, 0)))
        return True

    def forward(self, ctx):
        if ctx is not None and getattr(self, "_fused_in", None) is ctx:
            self._fused_in = None        # the convolution in front has written x + tap (ConvLayer.forward, skip_behind)
            return
        if self.combine_mode == "concat":
            self.output.data = ops.concat_fwd(self.x.data, self.y.data, self.x_shape[1], self.y_shape[1], self.output.cp)
        elif len(self.layers) > 1:
            # 1x1 projection of the tap with the add fused in its epilogue
            self.layers[1].forward(ctx, add=self.x.data)
            self.output.data = self.layers[1].output.data
        else:
            self.output.data = ops.add(self.x.data, self.y.data)

    def backward(self, ctx):
        g = self.output.grad
        if self.combine_mode == "concat":
            gx, gy = ops.concat_bwd(g, self.x_shape[1], self.x.cp, self.y_shape[1], self.y.cp)
            self.x.add_grad(gx)
            self.y.add_grad(gy)
            return
        self.x.add_grad(g)
        if len(self.layers) > 1:
            self.layers[1].output.grad = g
            self.layers[1].backward(ctx)
        else:
            self.y.add_grad(g)
