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

    def forward(self, ctx):
        if self.activation != "none":
            self.output.data = ops.relu_fwd(self.input.data)

    def backward(self, ctx):
        if self.activation != "none":
            self.input.add_grad(ops.relu_bwd(self.output.data, self.output.grad))
