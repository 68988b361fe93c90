"""`BNA` layer — batch normalisation fused with ReLU. Mirrors denet/layer/batch_norm_relu.py
(BatchNormReluOp :15-57: cuDNN BN followed by an in-place (x+|x|)/2; grad masks dy by xn > 0 then cuDNN BN-grad;
BatchNormReluLayer :85-167). One normalise+ReLU kernel forward, one masked BN-gradient backward."""
from .batch_norm import BatchNormLayer


class BatchNormReluLayer(BatchNormLayer):
    type_name = "batchnorm-relu"
    fused_relu = True

    def __init__(self, layers, momentum=0.9, eps=1e-5, json_param={}):
        jp = dict(json_param)
        jp["enabled"] = True   # the fused layer has no `enabled` switch in the reference (:92-94)
        super().__init__(layers, momentum, eps, json_param=jp)
        assert self.eps >= 1e-5, "BatchNormReluOp requires epsilon >= 1e-5 (batch_norm_relu.py:23)"

    @staticmethod
    def parse_desc(layers, name, tags, params):
        if name != "BNA":
            return False
        layers.append(BatchNormReluLayer(layers, params.get(0, 0.9), params.get(1, 1e-5)))
        return True

    def params(self):
        return [self.omega, self.beta, self.mean, self.stdinv]

    def updates(self, cost=None):
        return [self.mean, self.stdinv]

    def biases(self):
        return [self.omega, self.beta]

    def export_json(self):
        json = {"type": type(self).type_name, "layers": []}
        json.update({"momentum": self.momentum,
                     "eps": self.eps,
                     "mean": self.mean.get_value(),
                     "std": self.stdinv.get_value(),
                     "gamma": self.omega.get_value(),
                     "bias": self.beta.get_value()})
        return json
