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


class ScalarImage(Image):
    """Intensity image: linear interpolation, touched by intensity transforms."""


class LabelMap(Image):
    """Label map: nearest-neighbour interpolation, skipped by intensity transforms."""
