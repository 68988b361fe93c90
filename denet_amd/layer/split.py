"""`SPLIT` — in the reference cuts the Theano graph into separately compiled forward/backward functions to save
device memory (denet/layer/split.py:7-46, model_cnn.py:241-280). Identity here."""
from . import AbstractLayer


class SplitLayer(AbstractLayer):
    type_name = "split"

    def __init__(self, layers, json_param={}):
        super().__init__(layer_index=len(layers))
        self.enabled = json_param.get("enabled", True)
        self.has_split = self.enabled
        self.output = self.input = layers[-1].output
        self.output_shape = self.input_shape = layers[-1].output_shape

    @staticmethod
    def parse_desc(layers, name, tags, params):
        if name != "SPLIT":
            return False
        layers.append(SplitLayer(layers))
        return True

    def export_json(self):
        j = super().export_json()
        j.update({"enabled": self.enabled})
        return j

    def forward(self, ctx):
        pass

    def backward(self, ctx):
        pass
