"""Tensor-backed images (mirror of the in-memory part of reference ``data/image.py``).

The reference ``Image`` is a lazy, file-backed object (image.py:104); the
augmentation path only ever sees its ``(C, I, J, K)`` tensor and its affine, so
this mirror keeps exactly those.  File I/O is out of scope.
"""
from __future__ import annotations

import copy as _copy

import numpy as np
import torch
from torch import Tensor

from .affine import AffineMatrix


class Image:
    """A 4-D ``(C, I, J, K)`` tensor plus its voxel-to-world affine."""

    def __init__(self, data, *, affine=None, **metadata) -> None:
        if isinstance(data, np.ndarray):
            data = torch.as_tensor(data)
        if not isinstance(data, Tensor):
            raise TypeError(f"Image data must be a tensor or ndarray, got {type(data).__name__}")
        if data.ndim == 3:
            data = data.unsqueeze(0)
        if data.ndim != 4:
            raise ValueError(f"Image data must be 4D (C, I, J, K), got {data.ndim}D")
        self._data = data
        self._affine = affine if isinstance(affine, AffineMatrix) else AffineMatrix(affine)
        self.metadata = dict(metadata)
        self.applied_transforms: list = []

    # -- accessors -----------------------------------------------------------
    @property
    def data(self) -> Tensor:
        return self._data

    def set_data(self, value: Tensor) -> None:
        if value.ndim != 4:
            raise ValueError(f"Image data must be 4D, got {value.ndim}D")
        self._data = value

    @property
    def affine(self) -> AffineMatrix:
        return self._affine

    @affine.setter
    def affine(self, value) -> None:
        self._affine = value if isinstance(value, AffineMatrix) else AffineMatrix(value)

    @property
    def shape(self) -> tuple[int, int, int, int]:
        return tuple(self._data.shape)  # type: ignore[return-value]

    @property
    def spatial_shape(self) -> tuple[int, int, int]:
        return tuple(int(s) for s in self._data.shape[1:])  # type: ignore[return-value]

    @property
    def num_channels(self) -> int:
        return int(self._data.shape[0])

    @property
    def spacing(self) -> tuple[float, float, float]:
        return self._affine.spacing

    @property
    def device(self) -> torch.device:
        return self._data.device

    @property
    def dtype(self) -> torch.dtype:
        return self._data.dtype

    def __getitem__(self, item):
        """Metadata lookup (``str``) or a crop along ``(C, I, J, K)`` with the affine origin moved (image.py:832-899).

        Integers keep their axis (size 1), one ``Ellipsis`` expands to full slices and missing
        trailing indices are full slices (``normalize_index``, data/backends.py:52-106).  The
        result is a view of the same storage on the same device - what the patch samplers
        hand to a ``DataLoader`` or ``Queue``.
        """
        if isinstance(item, str):
            if item in self.metadata:
                return self.metadata[item]
            raise KeyError(f"{type(self).__name__} has no metadata key {item!r}")
        sc, si, sj, sk = _normalize_index(item)
        cropped = self._data[sc, si, sj, sk]
        matrix = self._affine.data.clone()
        starts = [axis.indices(size)[0] for axis, size in zip((si, sj, sk), self.shape[1:], strict=True)]
        matrix[:3, 3] += matrix[:3, :3] @ torch.tensor(starts, dtype=torch.float64, device=matrix.device)
        new = type(self)(cropped, affine=AffineMatrix(matrix), **_copy.deepcopy(self.metadata))
        new.applied_transforms = list(self.applied_transforms)
        return new

    def to(self, *args, **kwargs) -> "Image":
        self._data = self._data.to(*args, **kwargs)
        return self

    def numpy(self) -> np.ndarray:
        return self._data.detach().cpu().numpy()

    def __deepcopy__(self, memo):
        new = type(self)(self._data.clone(), affine=self._affine.clone(), **_copy.deepcopy(self.metadata, memo))
        new.applied_transforms = list(self.applied_transforms)
        return new

    def __repr__(self) -> str:
        return f"{type(self).__name__}(shape={self.shape}, dtype={self.dtype}, device={self.device})"


def _normalize_index(item, ndim: int = 4) -> tuple[slice, ...]:
    """An index as exactly ``ndim`` slices (data/backends.py:52-106)."""
    if isinstance(item, (int, slice)) or item is Ellipsis:
        items = (item,)
    elif isinstance(item, tuple):
        items = item
    else:
        raise TypeError(f"Index type {type(item).__name__} not understood")
    if sum(1 for entry in items if entry is Ellipsis) > 1:
        raise IndexError("an index can only have a single ellipsis ('...')")
    if any(entry is Ellipsis for entry in items):
        position = next(index for index, entry in enumerate(items) if entry is Ellipsis)
        fill = max(ndim - (len(items) - 1), 0)
        items = (*items[:position], *([slice(None)] * fill), *items[position + 1 :])
    if len(items) > ndim:
        raise IndexError(f"Too many indices: expected at most {ndim} (C, I, J, K), got {len(items)}")
    parsed = []
    for entry in items:
        if isinstance(entry, bool) or not isinstance(entry, (int, slice)):
            raise TypeError(f"Index type {type(entry).__name__} not understood")
        if isinstance(entry, int):  # keep the axis; slice(-1, 0) would be empty
            parsed.append(slice(entry, None) if entry == -1 else slice(entry, entry + 1))
        else:
            parsed.append(entry)
    parsed += [slice(None)] * (ndim - len(parsed))
    return tuple(parsed)


class ScalarImage(Image):
    """Intensity image: linear interpolation, touched by intensity transforms."""


class LabelMap(Image):
    """Label map: nearest-neighbour interpolation, skipped by intensity transforms."""
