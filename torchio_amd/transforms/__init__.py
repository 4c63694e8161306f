from .anisotropy import Anisotropy
from .bias_field import BiasField
from .blur import Blur
from .compose import Compose
from .compose import OneOf
from .compose import SomeOf
from .flip import Flip
from .gamma import Gamma
from .inverse import apply_inverse_transform
from .inverse import get_inverse_transform
from .motion import Motion
from .noise import Noise
from .noise import get_noise_rng
from .noise import set_noise_rng
from .pad import Crop
from .pad import Pad
from .parameter_range import Choice
from .resize import Resize
from .spatial import Affine
from .spatial import ElasticDeformation
from .spatial import Resample
from .spatial import Spatial
from .transform import AppliedTransform
from .transform import IntensityTransform
from .transform import SpatialTransform
from .transform import Transform

__all__ = [
    "Affine", "Anisotropy", "AppliedTransform", "BiasField", "Blur", "Choice", "Compose", "Crop", "ElasticDeformation", "Flip", "Gamma",
    "IntensityTransform", "Motion", "Noise", "OneOf", "Pad", "Resample", "Resize", "SomeOf", "Spatial", "SpatialTransform", "Transform",
    "apply_inverse_transform", "get_inverse_transform", "get_noise_rng", "set_noise_rng",
]
