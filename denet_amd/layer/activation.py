"""`A` layer — activation. Mirrors denet/layer/activation.py (ActivationLayer :10-56). Only the activations
the shipped recipes use run on the device: relu / relu-safe ((x+|x|)/2 == max(x,0) for finite x) and none."""
from . import AbstractLayer, Act
from .. import ops


class ActivationLayer(AbstractLayer):
    type_name = "activation"

    def __init__(self, layers, activation="relu", json_param={}):
        super().__init__(layer_index=len(layers))
        self.input = layers[-1].output
        self.input_shape = layers[-1].output_shape
        self.activation = json_param.get("activation", activation)
        if self.activation not in ("relu", "relu-safe", "none"):
            raise NotImplementedError("activation '%s' is outside the hot path of this build" % self.activation)
        self.output_shape = self.input_shape
        self.output = self.input if self.activation == "none" else Act(self.output_shape, self.input.cp,
                                                                        "act%i" % self.layer_index)
        # `BN A` written as two layers (the un-converted ResNets, examples/resnet34-imagenet.sh): when this layer is the only
        # reader of the batch norm's output, the batch norm runs its fused BN + ReLU passes straight into this layer's output
        # (ModelCNN.build_train_func sets bn.act_fused after counting the readers) - what --convert-bn-relu does to the model
        # file, without touching the layer list or the JSON
        prev = layers[-1]
        self.fused_into = None
        if self.activation in ("relu", "relu-safe") and getattr(prev, "type_name", None) == "batchnorm" and prev.enabled \
                and prev.output is self.input:
            self.fused_into = prev
            prev.act_behind = self

    @staticmethod
    def parse_desc(layers, name, tags, params):
        if name != "A":
            return False
        layers.append(ActivationLayer(layers, params["activation"]))
        return True

    def export_json(self):
        json = super().export_json()
        json.update({"activation": self.activation})
        return json

    def _fused(self):
        return self.fused_into is not None and getattr(self.fused_into, "act_fused", False)

    def forward(self, ctx):
        if self.activation != "none" and not self._fused():      # fused: the batch norm in front has written (or linked) the output
            self.output.data = ops.relu_fwd(self.input.data)

    def backward(self, ctx):
        if self.activation != "none" and not self._fused():      # fused: the batch norm's backward pass reads this output's gradient
            self.input.add_grad(ops.relu_bwd(self.output.data, self.output.grad))
