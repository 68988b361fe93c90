"""`DNC` — DeNet corner layer. Mirrors denet/layer/denet_corner.py (DeNetCornerLayer :17-134): a 1x1 convolution
producing `corner_num` corner logits + `sample_feat` sampling features (:39), corner rows zero-initialised with
bias +5 (:41-47), corner_pr = log_softmax([x,-x]) over a new axis (:50-53), host-side corner target rasteriser
(get_target :81-123) and the corner NLL cost (:126-134). The layer passes its input through."""
import numpy

from . import AbstractLayer, InitialLayer
from .convolution import ConvLayer
from .. import ops


class DeNetCornerLayer(AbstractLayer):
    type_name = "denet-corner"

    def __init__(self, layers, sample_feat=512, cost_factor=1, dropout=0.0, use_center=False, json_param={}):
        super().__init__(layer_index=len(layers))
        self.input = layers[-1].output
        self.input_shape = layers[-1].output_shape
        self.output = layers[-1].output
        self.output_shape = layers[-1].output_shape
        self.batch_size, self.features, self.height, self.width = self.input_shape

        self.sample_feat = json_param.get("sampleFeat", sample_feat)
        self.cost_factor = json_param.get("costFactor", cost_factor)
        self.use_center = json_param.get("useCenter", use_center)
        self.dropout = json_param.get("dropout", dropout)

        self.corner_num = 5 if self.use_center else 4
        self.layers = [InitialLayer(self.input, self.input_shape)]
        self.layers.append(ConvLayer(self.layers, (self.corner_num + self.sample_feat, self.features, 1, 1), (1, 1), True, False))

        conv = self.layers[-1]
        omega = conv.omega.get_value()
        omega[:self.corner_num, :, :, :] = 0.0
        conv.omega.set_value(omega)
        beta = conv.beta.get_value()
        beta[:self.corner_num] = 5.0
        conv.beta.set_value(beta)

        self.corner_shape = (self.batch_size, 2, self.corner_num, self.height, self.width)
        self.sample_shape = (self.batch_size, self.sample_feat, self.height, self.width)
        # device state of the current step
        self.corner_pr = None      # [B,2,Cn,H,W] log-probabilities (reference layout)
        self.dconv = None          # gradient w.r.t. the conv output, filled by the corner cost and the RoI gather
        self._target = None
        self.sample_shared = None  # inference: sampling features kept from get_samples (denet_sparse.py:122)

    @staticmethod
    def parse_desc(layers, name, tags, params):
        if name != "DNC":
            return False
        layers.append(DeNetCornerLayer(layers, params.get(0, 512), params.get(1, 1.0), params.get(2, 0.0), "C" in tags))
        return True

    def export_json(self):
        json = super().export_json()
        json.update({"sampleFeat": self.sample_feat, "useCenter": self.use_center, "costFactor": self.cost_factor,
                     "dropout": self.dropout})
        return json

    def get_target(self, model, samples, metas):
        """corner-target map of denet_corner.py:81-123, rasterised for all boxes of the batch at once: cell =
        round-half-to-even(coordinate * size) (numpy.rint == Python's round), far corner = max(near, cell - 1), a corner is
        written when both of its cells are on the map; plane 1 = corner, plane 0 = 1 - plane 1, everything divided by
        W * H * corner_num. Pinned against the reference's own method (tests/golden/make_layer_method_fixtures.py)."""
        B, _, cn, H, W = self.corner_shape
        pos = numpy.zeros((B, cn, H, W), dtype=numpy.float32)
        counts = [len(m["bbox"]) for m in metas]
        if sum(counts) > 0:
            box = numpy.concatenate([numpy.asarray(m["bbox"], dtype=numpy.float64).reshape(-1, 4) for m in metas], axis=0)
            img = numpy.repeat(numpy.arange(len(metas)), counts)
            x0 = numpy.rint(box[:, 0] * W).astype(numpy.int64)
            y0 = numpy.rint(box[:, 1] * H).astype(numpy.int64)
            x1 = numpy.maximum(x0, numpy.rint(box[:, 2] * W).astype(numpy.int64) - 1)
            y1 = numpy.maximum(y0, numpy.rint(box[:, 3] * H).astype(numpy.int64) - 1)
            points = [(x0, y0), (x1, y0), (x0, y1), (x1, y1)]                    # TL, TR, BL, BR
            if self.use_center:
                points.append((numpy.rint((box[:, 0] + box[:, 2]) * 0.5 * W).astype(numpy.int64),
                               numpy.rint((box[:, 1] + box[:, 3]) * 0.5 * H).astype(numpy.int64)))
            for kind, (px, py) in enumerate(points):
                on = (px >= 0) & (px < W) & (py >= 0) & (py < H)
                pos[img[on], kind, py[on], px[on]] = 1.0
        corner_pr = numpy.empty(self.corner_shape, dtype=numpy.float32)
        corner_pr[:, 1] = pos
        corner_pr[:, 0] = 1.0 - pos
        corner_pr /= W * H * cn
        if self.dropout > 0.0:
            # denet_corner.py:113-116: one Bernoulli(1 - dropout) mask per cell and corner type from numpy's global generator
            mask = numpy.random.binomial(1, 1.0 - self.dropout, (B, cn, H, W)).astype(numpy.float32)
            corner_pr *= mask[:, None, :, :, :] / (1.0 - self.dropout)
        return numpy.array([], dtype=numpy.int64), corner_pr.flatten()

    def cost(self, yt_index, yt_value):
        return True

    # ---- execution ----
    @property
    def conv(self):
        return self.layers[-1]

    def begin_step(self, metas):
        """the corner targets depend on the ground truth only: rasterised and uploaded (copy stream) while the first
        layers run, instead of in front of this layer where the device would wait for the host"""
        target = self.get_target(None, None, metas)
        self.set_target(None, target[0], target[1])
        self._target_metas = metas

    def prepare_target(self, ctx, model, data_x, metas):
        if getattr(self, "_target_metas", None) is metas:
            self._target_metas = None         # prepared by begin_step of this very step
            return
        super().prepare_target(ctx, model, data_x, metas)

    def set_target(self, ctx, yt_index, yt_value):
        import torch
        v = numpy.ascontiguousarray(yt_value, dtype=numpy.float32)
        if getattr(self, "_pinned", None) is None or self._pinned.numel() != v.size:
            self._pinned = torch.empty(v.size, dtype=torch.float32).pin_memory()
        ev = getattr(self, "_target_ev", None)
        if ev is not None:
            ev.synchronize()                   # the previous upload has finished reading the pinned buffer
        self._pinned.copy_(torch.from_numpy(v))
        self._target, self._target_ev = ops.upload_async(self._pinned)

    def forward(self, ctx):
        self.conv.forward(ctx)
        self.corner_pr = ops.corner_fwd(self.conv.output.data, self.corner_num)
        self.dconv = None

    def sample_map(self):
        """(tensor, channel offset, F): the sampling features live in channels [Cn, Cn+F) of the conv output"""
        return self.conv.output.data, self.corner_num, self.sample_feat

    def alloc_dconv(self, zero):
        import torch
        if self.dconv is None:
            shape = self.conv.output.data.shape
            self.dconv = torch.zeros(shape, device="cuda") if zero else ops.empty(*shape)
        return self.dconv

    def loss_backward(self, ctx, cost_out, want_grad=True):
        dconv = None
        if want_grad:
            # if no RoI gather follows, the sampling / padding channels carry no gradient: start from zeros
            dconv = self.alloc_dconv(zero=not ctx.has_sparse)
        ops.wait_upload(getattr(self, "_target_ev", None))
        ops.corner_loss(self.corner_pr, self._target, dconv, cost_out, float(self.cost_factor))

    def backward(self, ctx):
        if self.dconv is None:
            return
        self.conv.output.grad = self.dconv
        self.conv.backward(ctx)
