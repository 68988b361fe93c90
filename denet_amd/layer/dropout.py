"""`D` layer — dropout. Mirrors denet/layer/dropout.py (DropoutLayer :9-40): while training
output = input * mask / (1 - rate) with mask ~ Bernoulli(1 - rate) per element, identity otherwise (:20-24).

The reference draws the mask from Theano's MRG_RandomStreams seeded by ModelCNN.rng_seed (layer/__init__.py:5-16,
model_cnn.py:91) — a third-party stream that cannot be reproduced here. The build's mask is a pure function of
(rng seed, layer index, iteration, logical element index) computed inside the kernel (csrc/augment.hip), so no mask
tensor is written or kept: the backward pass regenerates the same bits."""
from . import AbstractLayer, Act, get_iteration, get_rng_seed, get_train
from .. import ops


class DropoutLayer(AbstractLayer):
    type_name = "dropout"

    def __init__(self, layers, dropout_rate=1.0, json_param={}):
        super().__init__(layer_index=len(layers))
        self.input = layers[-1].output
        self.input_shape = layers[-1].output_shape
        self.dropout_rate = float(json_param.get("dropoutRate", dropout_rate))
        assert 0.0 <= self.dropout_rate < 1.0, "dropout rate must be in [0, 1)"
        self.output_shape = self.input_shape
        self.output = Act(self.output_shape, self.input.cp, "dropout%i" % self.layer_index)
        self.output.requires_grad = getattr(self.input, "requires_grad", True)
        self._seed = None

    @staticmethod
    def parse_desc(layers, name, tags, params):
        if name != "D":
            return False
        layers.append(DropoutLayer(layers, params.get(0, 0.5)))
        return True

    def export_json(self):
        json = super().export_json()
        json.update({"dropoutRate": self.dropout_rate})
        return json

    def step_seed(self):
        return ops.layer_seed(get_rng_seed(), self.layer_index, get_iteration())

    def forward(self, ctx):
        if get_train():
            self._seed = self.step_seed()
            self.output.data = ops.dropout(self.input.data, self.input_shape[1], self.dropout_rate, self._seed)
        else:
            self._seed = None
            self.output.data = self.input.data

    def backward(self, ctx):
        if not getattr(self.input, "requires_grad", True):
            return
        g = self.output.grad
        if self._seed is not None:
            g = ops.dropout(g, self.input_shape[1], self.dropout_rate, self._seed)
        self.input.add_grad(g)
