"""``Anisotropy`` on the HIP engine (mirror of reference ``transforms/spatial/anisotropy.py``).

Down-sample along one axis with nearest neighbours and up-sample back.  Shared parameters:
two ``tio_interpolate3d`` launches (the reference's two ``F.interpolate`` calls,
anisotropy.py:372-392).  Per-element parameters: the reference composes both steps into
per-element source-index tables with integer arithmetic and gathers / blends along the axis
(anisotropy.py:128-310); the tables are built the same way on the host (a few hundred
integers) and one ``tio_axis_gather_lerp`` launch per degraded axis does the rest.
"""
from __future__ import annotations

from typing import Any

import torch
from torch import Tensor

from .. import ops
from ..data.batch import SubjectsBatch
from ..data.image import LabelMap
from .parameter_range import to_nonneg_range
from .transform import Transform


class Anisotropy(Transform):
    r"""Simulate an anisotropic acquisition (anisotropy.py:17-63).

    Args:
        axes: spatial axes eligible for down-sampling; one is drawn per application.
        downsampling: factor :math:`m \geq 1` (scalar, or ``(a, b)`` for :math:`\mathcal{U}(a, b)`).
        image_interpolation: interpolation used when up-sampling scalar images.
    """

    def __init__(self, *, axes: tuple[int, ...] = (0, 1, 2), downsampling=1.0, image_interpolation: str = "linear", **kwargs: Any) -> None:
        super().__init__(**kwargs)
        self.axes = axes
        self.downsampling = to_nonneg_range(downsampling)
        self.image_interpolation = image_interpolation
        _low, high = self.downsampling._ranges[0]
        if high < 1.0:
            raise ValueError(f"downsampling range upper bound must be >= 1, got {high}")
        self._warn_if_noop(is_noop=self.downsampling.is_constant(1.0), hint="downsampling=(1.5, 5)")

    @property
    def supports_per_instance_params(self) -> bool:
        return True

    @property
    def supports_per_instance_p(self) -> bool:
        return True

    def make_params(self, batch: SubjectsBatch) -> dict[str, Any]:
        """Axis (``torch.randint``) then factor, per element when batched (anisotropy.py:72-91)."""
        n = self._resolve_n(batch)
        if n is None:
            axis = self.axes[int(torch.randint(len(self.axes), (1,)).item())]
            return {"axis": axis, "factor": max(1.0, self.downsampling.sample_1d())}
        keep = self._keep_mask(batch, n)
        axes: list[int] = []
        factors: list[float] = []
        for index in range(n):
            if keep is not None and not keep[index]:
                axes.append(self.axes[0])
                factors.append(1.0)
                continue
            axes.append(self.axes[int(torch.randint(len(self.axes), (1,)).item())])
            factors.append(max(1.0, self.downsampling.sample_1d()))
        params = {"axis": axes, "factor": factors}
        self._tag_batched(params, batch, n, keep, ["axis", "factor"])
        return params

    def apply_transform(self, batch: SubjectsBatch, params: dict[str, Any]) -> SubjectsBatch:
        per_instance = self._is_per_instance_params(params)
        for img_batch in batch.images.values():
            mode = "nearest" if issubclass(img_batch._image_class, LabelMap) else self.image_interpolation
            if per_instance:
                img_batch.data = _simulate_anisotropy_per_instance(img_batch.data, axes=params["axis"], factors=params["factor"], mode=mode)
            elif params["factor"] > 1.0:
                img_batch.data = _simulate_anisotropy(img_batch.data, axis=params["axis"], factor=params["factor"], mode=mode)
        return batch


def _simulate_anisotropy(data: Tensor, *, axis: int, factor: float, mode: str) -> Tensor:
    """Nearest down-sampling to ``round(length / factor)`` then up-sampling back (anisotropy.py:355-392)."""
    original = list(data.shape[2:])
    down = list(original)
    down[axis] = max(1, round(original[axis] / factor))
    engine = ops.engine()
    # the reference converts to float32 first; nearest is an element move, so only the up-sampling
    # of non-float32 data needs the conversion (interpolate3d computes in float32 and casts back)
    low_resolution = engine.interpolate3d(data, down, "nearest")
    return engine.interpolate3d(low_resolution, original, "nearest" if mode == "nearest" else "linear")


def _simulate_anisotropy_per_instance(data: Tensor, *, axes: list[int], factors: list[float], mode: str) -> Tensor:
    """Every element with its own axis and factor (anisotropy.py:128-178); elements with factor <= 1 are untouched."""
    factors_tensor = torch.as_tensor(factors, dtype=torch.float64)
    axes_tensor = torch.as_tensor(axes, dtype=torch.long)
    active = factors_tensor > 1.0
    if not bool(active.any()):
        return data
    if bool(((axes_tensor[active] < 0) | (axes_tensor[active] > 2)).any()):
        raise ValueError(f"Anisotropy axis must be in {{0, 1, 2}}, got {sorted(set(axes))}")
    output = data
    engine = ops.engine()
    for axis in range(3):
        selected = active & (axes_tensor == axis)
        if not bool(selected.any()):
            continue
        length = data.shape[axis + 2]
        # tables for every element (rows of unselected elements are never read: they are copied)
        safe_factors = torch.where(selected, factors_tensor, torch.ones_like(factors_tensor))
        down_sizes = torch.round(length / safe_factors).clamp_min(1).to(torch.long)
        if mode == "nearest":
            lower, upper, weights = _nearest_source_indices(length, down_sizes), None, None
        else:
            lower, upper, weights = _linear_source_indices(length, down_sizes)
        # each element is degraded along ONE axis, so the passes never touch the same element twice
        output = engine.axis_gather_lerp(output, axis, lower, upper, weights, active=selected)
    return output


def _downsample_source_indices(length: int, down_sizes: Tensor, lowres_indices: Tensor) -> Tensor:
    """Low-resolution index -> source index of the nearest down-sampling (anisotropy.py:290-310)."""
    source = torch.div(lowres_indices * length, down_sizes[:, None], rounding_mode="floor")
    return source.clamp(max=length - 1)


def _nearest_source_indices(length: int, down_sizes: Tensor) -> Tensor:
    positions = torch.arange(length, dtype=torch.long)
    lowres = torch.div(positions * down_sizes[:, None], length, rounding_mode="floor")
    return _downsample_source_indices(length, down_sizes, lowres)


def _linear_source_indices(length: int, down_sizes: Tensor) -> tuple[Tensor, Tensor, Tensor]:
    """Lower / upper source indices and the upper weight of the align-corners up-sampling (anisotropy.py:246-287)."""
    positions = torch.arange(length, dtype=torch.float32)
    if length == 1:
        lowres_positions = torch.zeros(len(down_sizes), 1, dtype=torch.float32)
    else:
        scale = (down_sizes.to(torch.float32) - 1.0) / (length - 1)
        lowres_positions = positions * scale[:, None]
    lower_lowres = lowres_positions.floor().to(torch.long)
    upper_lowres = torch.minimum(lower_lowres + 1, down_sizes[:, None] - 1)
    weights = lowres_positions - lower_lowres.to(torch.float32)
    return (
        _downsample_source_indices(length, down_sizes, lower_lowres),
        _downsample_source_indices(length, down_sizes, upper_lowres),
        weights,
    )
