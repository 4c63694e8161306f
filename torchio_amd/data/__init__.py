from .affine import AffineMatrix
from .aggregator import PatchAggregator
from .batch import ImagesBatch
from .batch import SubjectsBatch
from .image import Image
from .image import LabelMap
from .image import ScalarImage
from .patch import PatchLocation
from .queue import Queue
from .sampler import GridSampler
from .sampler import LabelSampler
from .sampler import PatchSampler
from .sampler import UniformSampler
from .sampler import WeightedSampler
from .subject import Subject

__all__ = [
    "AffineMatrix", "GridSampler", "Image", "ImagesBatch", "LabelMap", "LabelSampler", "PatchAggregator", "PatchLocation",
    "PatchSampler", "Queue", "ScalarImage", "Subject", "SubjectsBatch", "UniformSampler", "WeightedSampler",
]
