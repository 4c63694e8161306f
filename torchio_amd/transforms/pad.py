"""``Pad`` and ``Crop`` (mirror of reference ``transforms/spatial/pad.py``, ``crop.py``, ``_padding.py``).

``Pad`` is one element-move launch (``tio_pad3d``: constant / reflect / replicate / circular as
``F.pad`` defines them; the ``mean`` / ``median`` / ``minimum`` modes are constant padding with one
value per batch element, computed with the reference's formulas on the device).  ``Crop`` is a
view, like in the reference.  Each is the other's inverse; both shift the affine origin.
"""
from __future__ import annotations

import math
import warnings
from typing import Any

import torch
from torch import Tensor

from .. import ops
from ..data.batch import SubjectsBatch
from .transform import SpatialTransform

_PADDING_MODES = ("constant", "reflect", "replicate", "circular", "mean", "median", "minimum")
_STATISTIC_MODES = ("mean", "median", "minimum")


def parse_padding_mode(padding_mode: str) -> str:
    if padding_mode not in _PADDING_MODES:
        raise ValueError(f"padding_mode must be one of {_PADDING_MODES}, got {padding_mode!r}")
    return padding_mode


def _six(values, what: str) -> tuple[int, int, int, int, int, int]:
    """int -> all sides; 3 values -> symmetric per axis; 6 values -> per side (pad.py:21-33)."""
    if isinstance(values, int):
        return (values,) * 6  # type: ignore[return-value]
    values = list(values)
    if len(values) == 3:
        i, j, k = values
        return (i, i, j, j, k, k)
    if len(values) == 6:
        return tuple(values)  # type: ignore[return-value]
    raise ValueError(f"{what} must have 1, 3, or 6 values, got {len(values)}")


def _quantile(values: Tensor, q: float) -> Tensor:
    """One quantile of a 1-D tensor through ``kthvalue`` with linear interpolation (_statistics.py:11-43)."""
    index = q * (values.numel() - 1)
    lower = math.floor(index)
    lower_value = torch.kthvalue(values, lower + 1).values
    if index == lower:
        return lower_value
    return lower_value.lerp(torch.kthvalue(values, lower + 2).values, index - lower)


def _padding_statistic(data: Tensor, padding_mode: str) -> Tensor:
    """One whole-volume value per batch element (_padding.py:34-59)."""
    flat = data.flatten(start_dim=1)
    if padding_mode == "minimum":
        return flat.amin(dim=1)
    if not torch.is_floating_point(data):
        warnings.warn(
            f'The constant value computed for padding mode "{padding_mode}" might be truncated in the output, as the data'
            " type of the input image is not float. Consider converting the image to a floating point type before applying"
            " this transform.",
            RuntimeWarning,
            stacklevel=4,
        )
    float_flat = flat if data.dtype in (torch.float32, torch.float64) else flat.float()
    if padding_mode == "mean":
        statistic = float_flat.mean(dim=1)
    else:
        statistic = torch.stack([_quantile(values, 0.5) for values in float_flat])
    return statistic.to(data.dtype)


def pad_tensor(data: Tensor, padding, padding_mode: str, fill: float) -> Tensor:
    """Pad a 4-D image tensor or a 5-D image batch (_padding.py:62-104)."""
    if data.ndim not in (4, 5):
        raise ValueError(f"Expected a 4D or 5D image tensor, got {data.ndim}D")
    batch = data.unsqueeze(0) if data.ndim == 4 else data
    engine = ops.engine()
    if padding_mode in _STATISTIC_MODES:
        padded = engine.pad3d(batch, padding, "constant", fill_per_element=_padding_statistic(batch, padding_mode))
    else:
        if padding_mode != "constant" and fill != 0:  # F.pad's own rule
            raise RuntimeError(f'Padding mode "{padding_mode}" doesn\'t take in value argument')
        padded = engine.pad3d(batch, padding, padding_mode, fill=fill)
    return padded[0] if data.ndim == 4 else padded


class Pad(SpatialTransform):
    """Add a border of voxels to each side of the volume (pad.py:36-122)."""

    def __init__(self, *, padding, padding_mode: str = "constant", fill: float = 0, **kwargs: Any) -> None:
        super().__init__(**kwargs)
        self.padding = _six(padding, "Padding")
        self.padding_mode = parse_padding_mode(padding_mode)
        self.fill = fill

    def make_params(self, batch: SubjectsBatch) -> dict[str, Any]:
        return {"padding": self.padding, "padding_mode": self.padding_mode, "fill": self.fill}

    def apply_transform(self, batch: SubjectsBatch, params: dict[str, Any]) -> SubjectsBatch:
        i0, i1, j0, j1, k0, k1 = params["padding"]
        for img_batch in self._get_images(batch).values():
            img_batch.data = pad_tensor(img_batch.data, (i0, i1, j0, j1, k0, k1), params["padding_mode"], params["fill"])
            for affine in img_batch.affines:  # the origin moves back by the leading pad
                affine._matrix[:3, 3] += affine.data[:3, :3] @ affine.data.new_tensor([-float(i0), -float(j0), -float(k0)])
        return batch

    @property
    def invertible(self) -> bool:
        return True

    def inverse(self, params: dict[str, Any]):
        return Crop(cropping=params["padding"], copy=False)


class Crop(SpatialTransform):
    """Remove a border of voxels from each side of the volume (crop.py:33-112)."""

    def __init__(self, *, cropping, **kwargs: Any) -> None:
        super().__init__(**kwargs)
        self.cropping = _six(cropping, "Cropping")

    def make_params(self, batch: SubjectsBatch) -> dict[str, Any]:
        return {"cropping": self.cropping}

    def apply_transform(self, batch: SubjectsBatch, params: dict[str, Any]) -> SubjectsBatch:
        i0, i1, j0, j1, k0, k1 = params["cropping"]
        for img_batch in self._get_images(batch).values():
            data = img_batch.data
            si, sj, sk = data.shape[-3:]
            img_batch.data = data[..., i0 : si - i1 or None, j0 : sj - j1 or None, k0 : sk - k1 or None]
            for affine in img_batch.affines:
                affine._matrix[:3, 3] += affine.data[:3, :3] @ affine.data.new_tensor([float(i0), float(j0), float(k0)])
        return batch

    @property
    def invertible(self) -> bool:
        return True

    def inverse(self, params: dict[str, Any]):
        return Pad(padding=params["cropping"], copy=False)
