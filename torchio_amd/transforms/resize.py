"""``Resize`` on the HIP engine (mirror of reference ``transforms/spatial/resize.py``).

``F.interpolate(data.float(), size=target, mode).to(dtype)`` as one streaming launch of
``tio_interpolate3d`` (legacy ``"nearest"`` for label maps, ``"trilinear"`` with
``align_corners=True`` otherwise); the field of view is kept, so each affine's voxel columns
are rescaled by ``old / new`` exactly like the reference (resize.py:77-81).
"""
from __future__ import annotations

from typing import Any

from .. import ops
from ..data.batch import SubjectsBatch
from ..data.image import LabelMap
from .transform import SpatialTransform


class Resize(SpatialTransform):
    """Resize images to a target spatial shape; the field of view is preserved (resize.py:14-55)."""

    def __init__(self, target_shape, *, image_interpolation: str = "linear", label_interpolation: str = "nearest", **kwargs: Any) -> None:
        super().__init__(**kwargs)
        if isinstance(target_shape, int):
            target_shape = (target_shape, target_shape, target_shape)
        self.target_shape = target_shape
        self.image_interpolation = image_interpolation
        self.label_interpolation = label_interpolation

    def make_params(self, batch: SubjectsBatch) -> dict[str, Any]:
        return {"target_shape": self.target_shape}

    def apply_transform(self, batch: SubjectsBatch, params: dict[str, Any]) -> SubjectsBatch:
        target = [int(s) for s in params["target_shape"]]
        engine = ops.engine()
        for img_batch in batch.images.values():
            is_label = issubclass(img_batch._image_class, LabelMap)
            mode = self.label_interpolation if is_label else self.image_interpolation
            old_shape = tuple(img_batch.data.shape[2:])
            img_batch.data = engine.interpolate3d(img_batch.data, target, "nearest" if mode == "nearest" else "linear")
            for affine in img_batch.affines:  # spacing changes to fit the new shape into the same field of view
                for axis in range(3):
                    affine._matrix[:3, axis] *= old_shape[axis] / target[axis]
        return batch
