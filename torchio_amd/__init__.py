"""torchio_amd — MI355X-native (gfx950) engine for TorchIO's augmentation hot path.

Drop-in for the spatial + intensity augmentation path of TorchIO 2.0
(``Affine`` / ``ElasticDeformation`` / ``Spatial`` / ``Resample``, ``BiasField``,
``Blur``, ``Noise``, ``Gamma``, ``Compose``, ``Subject`` / ``SubjectsBatch``):
same class names, constructor arguments, RNG draw order, params dicts and
history/inverse behaviour; the compute under ``apply_transform`` runs as
hand-written HIP kernels behind the C ABI of ``include/tio_hip.h``.
"""
from .data import AffineMatrix
from .data import GridSampler
from .data import LabelSampler
from .data import PatchAggregator
from .data import PatchLocation
from .data import PatchSampler
from .data import Queue
from .data import UniformSampler
from .data import WeightedSampler
from .data import Image
from .data import ImagesBatch
from .data import LabelMap
from .data import ScalarImage
from .data import Subject
from .data import SubjectsBatch
from .ops import calibrate_draw_policy
from .ops import get_draw_policy
from .ops import get_noise_plan
from .ops import set_noise_plan
from .ops import get_resample_precision
from .ops import set_draw_policy
from .ops import get_stencil_precision
from .ops import set_resample_precision
from .ops import set_stencil_precision
from .transforms import Affine
from .transforms import Anisotropy
from .transforms import AppliedTransform
from .transforms import BiasField
from .transforms import Blur
from .transforms import Choice
from .loader import ImagesLoader
from .loader import SubjectsLoader
from .transforms import Compose
from .transforms import Crop
from .transforms import Pad
from .transforms import ElasticDeformation
from .transforms import Flip
from .transforms import Gamma
from .transforms import IntensityTransform
from .transforms import Motion
from .transforms import Noise
from .transforms import OneOf
from .transforms import SomeOf
from .transforms import Resample
from .transforms import Resize
from .transforms import Spatial
from .transforms import SpatialTransform
from .transforms import Transform
from .transforms import apply_inverse_transform
from .transforms import get_inverse_transform
from .transforms import get_noise_rng
from .transforms import set_noise_rng

__version__ = "0.1.0"

__all__ = [
    "Affine", "AffineMatrix", "Anisotropy", "AppliedTransform", "BiasField", "Blur", "Choice", "Compose", "Crop", "ElasticDeformation", "Flip",
    "Gamma", "GridSampler", "Image", "ImagesBatch", "ImagesLoader", "IntensityTransform", "LabelMap", "LabelSampler", "Motion", "Noise", "OneOf",
    "Pad", "PatchAggregator", "PatchLocation", "PatchSampler", "Queue", "Resample", "Resize", "ScalarImage", "SomeOf", "Spatial", "SpatialTransform", "Subject",
    "SubjectsBatch", "SubjectsLoader", "Transform", "UniformSampler", "WeightedSampler", "apply_inverse_transform", "calibrate_draw_policy", "get_draw_policy", "set_draw_policy", "get_noise_plan", "set_noise_plan", "get_inverse_transform", "get_noise_rng", "get_resample_precision", "get_stencil_precision", "set_noise_rng", "set_resample_precision", "set_stencil_precision",
]
