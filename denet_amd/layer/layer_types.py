"""Layer registry: the first class whose parse_desc accepts a token wins (denet/layer/layer_types.py:17-25)."""
from . import IdentityLayer, InitialLayer
from .convolution import ConvLayer
from .pool import PoolLayer
from .pool_inv import PoolInvLayer
from .batch_norm import BatchNormLayer
from .batch_norm_relu import BatchNormReluLayer
from .activation import ActivationLayer
from .resnet import ResnetLayer
from .regression import RegressionLayer
from .split import SplitLayer
from .skip import SkipLayer, SkipSrcLayer
from .deconvolution import DeconvLayer
from .dropout import DropoutLayer
from .crop_mirror import CropMirrorLayer
from .border import BorderLayer

# same order as the reference list
layer_types = [IdentityLayer, DropoutLayer, BorderLayer, ConvLayer, PoolLayer, PoolInvLayer, RegressionLayer,
               CropMirrorLayer, ActivationLayer, BatchNormLayer, BatchNormReluLayer, ResnetLayer, DeconvLayer,
               SplitLayer, SkipLayer, SkipSrcLayer]

from .denet_corner import DeNetCornerLayer
from .denet_sparse import DeNetSparseLayer
from .denet_detect import DeNetDetectLayer

layer_types += [DeNetCornerLayer, DeNetSparseLayer, DeNetDetectLayer]
