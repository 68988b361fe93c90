"""`CM` layer — per-image random crop, column mirror and row flip on the device. Mirrors
denet/layer/crop_mirror.py (CropMirrorLayer :10-70): while training each image independently gets
mirror ~ Bernoulli(mirror_pr) (reverses dim 3, :32-35), flip ~ Bernoulli(flip_pr) (reverses dim 2, :38-41) and a
crop offset uniform in [0, in - crop] per axis (:44-53); at test time no mirror / flip and the centre crop
((in - crop)//2, :47-48). As for `D`, the Theano MRG stream is replaced by the counter-based generator of
csrc/augment.hip keyed by (rng seed, layer index, iteration, image index)."""
from . import AbstractLayer, Act, get_iteration, get_rng_seed, get_train
from .. import ops


class CropMirrorLayer(AbstractLayer):
    type_name = "crop-mirror"

    def __init__(self, layers, crop_size=None, mirror_pr=0.0, flip_pr=0.0, json_param={}):
        super().__init__(layer_index=len(layers))
        self.input = layers[-1].output
        self.input_shape = layers[-1].output_shape
        self.crop_size = tuple(int(c) for c in json_param.get("crop", crop_size))
        self.mirror_pr = float(json_param.get("mirror", mirror_pr))
        self.flip_pr = float(json_param.get("flip", flip_pr))
        assert self.crop_size[0] <= self.input_shape[2] and self.crop_size[1] <= self.input_shape[3]
        self.output_shape = (self.input_shape[0], self.input_shape[1], self.crop_size[0], self.crop_size[1])
        self.output = Act(self.output_shape, self.input.cp, "cropmirror%i" % self.layer_index)
        self.output.requires_grad = getattr(self.input, "requires_grad", True)
        self._key = (False, 0)

    @staticmethod
    def parse_desc(layers, name, tags, params):
        if name != "CM":
            return False
        layers.append(CropMirrorLayer(layers, (params.get(0), params.get(0)), params.get(1, 0.0), params.get(2, 0.0)))
        return True

    def export_json(self):
        json = super().export_json()
        json.update({"crop": self.crop_size, "mirror": self.mirror_pr, "flip": self.flip_pr})
        return json

    def step_seed(self):
        return ops.layer_seed(get_rng_seed(), self.layer_index, get_iteration())

    def forward(self, ctx):
        self._key = (bool(get_train()), self.step_seed())
        self.output.data = ops.crop_mirror_fwd(self.input.data, self.crop_size, self.mirror_pr, self.flip_pr, *self._key)

    def backward(self, ctx):
        if getattr(self.input, "requires_grad", True):
            self.input.add_grad(ops.crop_mirror_bwd(self.output.grad, tuple(self.input.data.shape), self.mirror_pr,
                                                    self.flip_pr, *self._key))
