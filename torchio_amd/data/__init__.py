from .affine import AffineMatrix
from .batch import ImagesBatch
from .batch import SubjectsBatch
from .image import Image
from .image import LabelMap
from .image import ScalarImage
from .subject import Subject

__all__ = ["AffineMatrix", "Image", "ImagesBatch", "LabelMap", "ScalarImage", "Subject", "SubjectsBatch"]
