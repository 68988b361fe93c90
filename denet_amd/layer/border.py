"""`B` layer — zero border around the feature map. Mirrors denet/layer/border.py (BorderLayer :9-46): border =
(left, right, top, bottom), an int or 1-tuple means the same width on all four sides (:18-22); the output is a
zero tensor with the input written into its interior (:30-33). One pass of csrc/augment.hip each way."""
from . import AbstractLayer, Act
from .. import ops


class BorderLayer(AbstractLayer):
    type_name = "border"

    def __init__(self, layers, border=0, json_param={}):
        super().__init__(layer_index=len(layers))
        self.input = layers[-1].output
        self.input_shape = layers[-1].output_shape
        if type(border) is int:
            border = (border, border, border, border)
        elif len(border) == 1:
            border = (border[0], border[0], border[0], border[0])
        assert len(border) == 4
        self.border = tuple(int(b) for b in json_param.get("border", border))
        shape = list(self.input_shape)
        shape[-1] += self.border[0] + self.border[1]
        shape[-2] += self.border[2] + self.border[3]
        self.output_shape = tuple(shape)
        self.output = Act(self.output_shape, self.input.cp, "border%i" % self.layer_index)
        self.output.requires_grad = getattr(self.input, "requires_grad", True)

    @staticmethod
    def parse_desc(layers, name, tags, params):
        if name != "B":
            return False
        layers.append(BorderLayer(layers, params.get(0, 0)))
        return True

    def export_json(self):
        json = super().export_json()
        json.update({"border": self.border})
        return json

    def forward(self, ctx):
        self.output.data = ops.border_fwd(self.input.data, self.border)

    def backward(self, ctx):
        if getattr(self.input, "requires_grad", True):
            self.input.add_grad(ops.border_bwd(self.output.grad, self.border))
