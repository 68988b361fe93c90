"""`PI` layer — nearest-neighbour up-sampling ("pool-inv"). Mirrors denet/layer/pool_inv.py (PoolInvLayer :10-41)
and the CUDA ops of denet/layer/pool_inv_op.py (k_pool_inv :38-63, k_pool_inv_grad :144-169)."""
from . import AbstractLayer, Act
from .. import ops


class PoolInvLayer(AbstractLayer):
    type_name = "pool-inv"

    def __init__(self, layers, size=(2, 2), json_param={}):
        super().__init__(layer_index=len(layers))
        self.input = layers[-1].output
        self.input_shape = layers[-1].output_shape
        self.size = tuple(json_param.get("size", size))
        # pool_inv.py:21: (N, C, size[1]*H, size[0]*W)
        self.output_shape = (self.input_shape[0], self.input_shape[1], self.size[1] * self.input_shape[2],
                             self.size[0] * self.input_shape[3])
        self.output = Act(self.output_shape, self.input.cp, "poolinv%i" % self.layer_index)

    @staticmethod
    def parse_desc(layers, name, tags, params):
        if name != "PI":
            return False
        layers.append(PoolInvLayer(layers, (params.get(0), params.get(0))))
        return True

    def export_json(self):
        json = super().export_json()
        json.update({"size": self.size})
        return json

    def forward(self, ctx):
        from . import get_train
        if ctx is not None and get_train() and self.size == (2, 2) and ops.UP_LINK:
            # training: the up-sampled tensor is written by whoever asks for it - a Winograd convolution behind this layer reads
            # the small one inside its input transform (ops.UpLink, ConvLayer.forward)
            self.output.set_pending_data(ops.UpLink(self.input.data))
            return
        self.output.data = ops.pool_inv_fwd(self.input.data, self.size[1], self.size[0])

    def backward(self, ctx):
        self.input.add_grad(ops.pool_inv_bwd(self.output.grad, self.size[1], self.size[0]))
