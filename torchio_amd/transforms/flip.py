"""``Flip`` on the HIP engine (mirror of reference ``transforms/spatial/flip.py``).

Reverse the voxel order along spatial axes: one element-move launch (``tio_flip3d``) per image,
also with per-element axes — the reference flips the whole batch once per axis and selects
with ``torch.where`` (flip.py:208-236).  Same constructor, anatomical axis labels resolved
through the image orientation, draw order (``torch.rand(3)`` per application / per element)
and history; ``Flip`` is its own inverse.
"""
from __future__ import annotations

from collections.abc import Sequence
from typing import Any

import torch

from .. import ops
from ..data.batch import ImagesBatch
from ..data.batch import SubjectsBatch
from .transform import SpatialTransform

# anatomical label (first letter) -> the pair of orientation codes of its axis
_LABEL_PAIRS = {"L": ("L", "R"), "R": ("L", "R"), "A": ("A", "P"), "P": ("A", "P"), "I": ("I", "S"), "S": ("I", "S")}


def _resolve_axes(axes, orientation=None) -> tuple[int, ...]:
    """Ints and anatomical strings (``'L'``, ``'Right'``, ``'AP'`` …) as a sorted tuple of axes in {0, 1, 2} (flip.py:28-73)."""
    if isinstance(axes, (int, str)):
        axes = (axes,)
    result: list[int] = []
    for axis in axes:
        if isinstance(axis, int):
            if axis not in (0, 1, 2):
                raise ValueError(f"Axis must be 0, 1, or 2; got {axis}")
            result.append(axis)
        elif isinstance(axis, str):
            letter = axis[0].upper()
            if letter not in _LABEL_PAIRS:
                raise ValueError(f"Unknown anatomical label {axis!r}. Use L, R, A, P, I, S or full names like 'Left', 'Right', etc.")
            if orientation is None:
                raise ValueError(f"Cannot resolve anatomical axis label {axis!r} without image orientation")
            for dim, code in enumerate(orientation):
                if code in _LABEL_PAIRS[letter]:
                    result.append(dim)
                    break
        else:
            raise TypeError(f"Axis must be int or str, got {type(axis).__name__}")
    return tuple(sorted(set(result)))


class Flip(SpatialTransform):
    """Reverse the order of elements along the given axes; ``flip_probability`` is a per-axis coin (flip.py:76-121)."""

    def __init__(self, *, axes=0, flip_probability: float = 1.0, **kwargs: Any) -> None:
        super().__init__(**kwargs)
        self.axes = axes
        if not 0 <= flip_probability <= 1:
            raise ValueError(f"flip_probability must be in [0, 1], got {flip_probability}")
        self.flip_probability = flip_probability

    @property
    def supports_per_instance_params(self) -> bool:
        return True

    @property
    def supports_per_instance_p(self) -> bool:
        return True

    def make_params(self, batch: SubjectsBatch) -> dict[str, Any]:
        images = self._get_images(batch)
        if not images:
            return {"axes": ()}
        first = next(iter(images.values()))
        n = self._resolve_n(batch)
        if n is None:
            orientation = first.affines[0].orientation if first.batch_size > 0 else None
            resolved = _resolve_axes(self.axes, orientation)
            flip_mask = torch.rand(3) < self.flip_probability
            return {"axes": tuple(a for a in resolved if flip_mask[a].item())}
        keep = self._keep_mask(batch, n)
        params = {"axes": self._sample_per_element_axes(n, first, keep)}
        self._tag_batched(params, batch, n, keep, ["axes"])
        return params

    def _sample_per_element_axes(self, n: int, first: ImagesBatch, keep) -> list[list[int]]:
        axes_list: list[list[int]] = []
        for index in range(n):
            if keep is not None and not keep[index]:
                axes_list.append([])
                continue
            resolved = _resolve_axes(self.axes, first.affines[index].orientation)  # each element may have its own orientation
            flip_mask = torch.rand(3) < self.flip_probability
            axes_list.append([a for a in resolved if flip_mask[a].item()])
        return axes_list

    def apply_transform(self, batch: SubjectsBatch, params: dict[str, Any]) -> SubjectsBatch:
        axes = params["axes"]
        if self._is_per_instance_params(params):
            for img_batch in self._get_images(batch).values():
                img_batch.data = _flip_per_element(img_batch.data, axes)
            return batch
        if not axes:
            return batch
        for img_batch in self._get_images(batch).values():
            img_batch.data = ops.engine().flip3d(img_batch.data, [int(a) for a in axes])
        return batch

    @property
    def invertible(self) -> bool:
        return True

    def inverse(self, params: dict[str, Any]):
        if self._is_per_instance_params(params):
            return _FlipInverse(axes_per_element=params["axes"], copy=False)
        return Flip(axes=params["axes"], copy=False)


def _flip_per_element(data, axes_per_element: list[list[int]]):
    """Every batch element flipped along its own axes (flip.py:208-236), one launch."""
    if not any(axes_per_element):
        return data
    flags = torch.tensor([[axis in axes for axis in range(3)] for axes in axes_per_element], dtype=torch.uint8)
    return ops.engine().flip3d(data, per_element=flags)


class _FlipInverse(SpatialTransform):
    """Inverse of a per-instance ``Flip`` for history replay (flip.py:239-262)."""

    def __init__(self, *, axes_per_element: list[list[int]], **kwargs: Any) -> None:
        super().__init__(**kwargs)
        self._axes_per_element = axes_per_element

    def make_params(self, batch: SubjectsBatch) -> dict[str, Any]:
        return {}

    def apply_transform(self, batch: SubjectsBatch, params: dict[str, Any]) -> SubjectsBatch:
        for img_batch in self._get_images(batch).values():
            img_batch.data = _flip_per_element(img_batch.data, self._axes_per_element)
        return batch
