"""Batch containers (mirror of reference ``src/torchio/data/batch.py``).

``ImagesBatch`` is the data layout the engine consumes: one dense
``(B, C, I, J, K)`` tensor per named image plus one affine per sample
(batch.py:21-50).  Per-instance transform history is sliced per element on
``unbatch()`` exactly like batch.py:337-399.
"""
from __future__ import annotations

import copy as _copy
import dataclasses
from typing import Any

import torch
from torch import Tensor

from . import _lazy
from . import _pending
from .affine import AffineMatrix
from .image import Image
from .image import ScalarImage

#: bookkeeping keys written by ``Transform._tag_batched`` (batch.py:18)
_BATCH_META_KEYS = ("_batch_size", "_batched_keys", "_keep")


class _History:
    """Shared ``applied_transforms`` behaviour of both batch types."""

    applied_transforms: list

    def get_inverse_transform(self, *, warn: bool = True, ignore_intensity: bool = False):
        from ..transforms.inverse import get_inverse_transform  # noqa: PLC0415

        return get_inverse_transform(self.applied_transforms, warn=warn, ignore_intensity=ignore_intensity)

    def apply_inverse_transform(self, **kwargs):
        result = self.get_inverse_transform(**kwargs)(self)
        if hasattr(result, "applied_transforms"):
            result.applied_transforms = []
        return result

    def clear_history(self) -> None:
        self.applied_transforms = []


class ImagesBatch(_History):
    """5-D tensor ``(B, C, I, J, K)`` with one affine per sample."""

    def __init__(self, data: Tensor, affines: list[AffineMatrix], *, image_class: type[Image] = ScalarImage) -> None:
        if data.ndim != 5:
            raise ValueError(f"Expected 5D tensor (B, C, I, J, K), got {data.ndim}D")
        if len(affines) != data.shape[0]:
            raise ValueError(f"Expected {data.shape[0]} affines, got {len(affines)}")
        self._data = data
        self._pending = None  # deferred BiasField / Blur stages (data/_pending.py); None = tensor is final
        self._affines = affines
        self._image_class = image_class
        self.applied_transforms = []

    @classmethod
    def from_images(cls, images: list[Image]) -> "ImagesBatch":
        if not images:
            raise ValueError("Cannot create batch from empty list")
        data = torch.stack([image.data for image in images])
        return cls(data, [image.affine.clone() for image in images], image_class=type(images[0]))

    @property
    def data(self) -> Tensor:
        if self._pending is not None:  # somebody wants values: launch whatever was deferred
            self._flush()
        return self._data

    @data.setter
    def data(self, value: Tensor) -> None:
        if value.ndim != 5:
            raise ValueError(f"Expected 5D tensor, got {value.ndim}D")
        self._pending = None
        self._data = value

    def _flush(self, *, noise=None) -> None:
        """Launch the deferred stages (see data/_pending.py), optionally with a trailing noise stage."""
        pending, self._pending = self._pending, None
        if pending is None or pending.is_empty():
            if noise is not None:
                raise RuntimeError("noise can only be folded into a pending blur")
            return
        self._data = _pending.flush(self._data, pending, noise=noise)

    @property
    def affines(self) -> list[AffineMatrix]:
        return self._affines

    @property
    def batch_size(self) -> int:
        return int(self._data.shape[0])

    @property
    def device(self) -> torch.device:
        return self._data.device

    def to(self, *args, **kwargs) -> "ImagesBatch":
        self._data = self.data.to(*args, **kwargs)
        return self

    def __getitem__(self, index: int) -> Image:
        return self._image_class(self.data[index], affine=self._affines[index].clone())

    def __len__(self) -> int:
        return self.batch_size

    def unbatch(self) -> list[Image]:
        return [self[i] for i in range(self.batch_size)]

    def __deepcopy__(self, memo):
        scope = _lazy.active_scope()
        data = self.data if scope is not None else self.data.clone()  # (.data: finished values)
        new = ImagesBatch(data, [a.clone() for a in self._affines], image_class=self._image_class)
        if scope is not None:
            scope.borrow(new, data)  # cloned later unless a transform replaces it (see _lazy.py)
        new.applied_transforms = list(self.applied_transforms)
        return new

    def __repr__(self) -> str:
        b, c, i, j, k = self._data.shape
        return f"ImagesBatch({self._image_class.__name__}, batch_size={b}, shape=({c}, {i}, {j}, {k}))"


class SubjectsBatch(_History):
    """Dict of named ``ImagesBatch`` plus per-sample metadata lists."""

    def __init__(self, images: dict[str, ImagesBatch], *, metadata: dict[str, list[Any]] | None = None) -> None:
        self._images = images
        self._metadata = metadata or {}
        self.applied_transforms = []
        self._per_element_history: list[list[Any]] | None = None

    @classmethod
    def from_subjects(cls, subjects: list[Any]) -> "SubjectsBatch":
        if not subjects:
            raise ValueError("Cannot create batch from empty list")
        first = subjects[0]
        images = {
            name: ImagesBatch.from_images([subject.images[name] for subject in subjects])
            for name in first.images
        }
        metadata = {key: [subject.metadata[key] for subject in subjects] for key in first.metadata}
        return cls(images, metadata=metadata)

    def adopt_history(self, source: "SubjectsBatch", subjects: list[Any]) -> None:
        """Carry the history of *source* over after its subjects were unbatched, processed and re-stacked (batch.py:269-284)."""
        if source._per_element_history is not None:
            self.set_per_element_history([subject.applied_transforms for subject in subjects])
        else:
            self.applied_transforms = list(source.applied_transforms)

    def set_per_element_history(self, histories: list[list[Any]]) -> None:
        if len(histories) != self.batch_size:
            raise ValueError(f"Expected {self.batch_size} per-element histories, got {len(histories)}")
        self._per_element_history = [list(history) for history in histories]
        self.applied_transforms = []

    @property
    def batch_size(self) -> int:
        return next(iter(self._images.values())).batch_size

    @property
    def images(self) -> dict[str, ImagesBatch]:
        return self._images

    @property
    def metadata(self) -> dict[str, list[Any]]:
        return self._metadata

    @property
    def device(self) -> torch.device:
        return next(iter(self._images.values())).device

    def to(self, *args, **kwargs) -> "SubjectsBatch":
        for batch in self._images.values():
            batch.to(*args, **kwargs)
        return self

    def __getitem__(self, key: str) -> ImagesBatch:
        return self._images[key]

    def __getattr__(self, name: str) -> ImagesBatch:
        if name.startswith("_"):
            raise AttributeError(name)
        images = self.__dict__.get("_images", {})
        if name in images:
            return images[name]
        raise AttributeError(f"SubjectsBatch has no attribute {name!r}")

    def __len__(self) -> int:
        return self.batch_size

    def unbatch(self) -> list[Any]:
        """Split into ``Subject``s, slicing per-instance history per element (batch.py:239-264)."""
        from .subject import Subject  # noqa: PLC0415

        subjects = []
        for index in range(self.batch_size):
            entries: dict[str, Any] = {name: images[index] for name, images in self._images.items()}
            entries.update({key: values[index] for key, values in self._metadata.items()})
            subject = Subject(**entries)
            history = _slice_history(self.applied_transforms, index)
            if self._per_element_history is not None:
                history = list(self._per_element_history[index]) + history
            subject.applied_transforms = history
            subjects.append(subject)
        return subjects

    def clear_history(self) -> None:
        self.applied_transforms = []
        self._per_element_history = None

    def get_inverse_transform(self, **kwargs):
        if self._per_element_history is not None:
            raise RuntimeError(
                "This batch has per-element transform histories, so a single batch inverse is"
                " ambiguous. Call apply_inverse_transform() or unbatch() and invert each subject."
            )
        return super().get_inverse_transform(**kwargs)

    def apply_inverse_transform(self, **kwargs):
        if self._per_element_history is not None:
            inverted = [subject.apply_inverse_transform(**kwargs) for subject in self.unbatch()]
            return type(self).from_subjects(inverted)
        return super().apply_inverse_transform(**kwargs)

    def __deepcopy__(self, memo):
        new = SubjectsBatch(
            {name: _copy.deepcopy(images, memo) for name, images in self._images.items()},
            metadata=_copy.deepcopy(self._metadata, memo),
        )
        new.applied_transforms = list(self.applied_transforms)
        if self._per_element_history is not None:
            new._per_element_history = [list(h) for h in self._per_element_history]
        return new

    def __repr__(self) -> str:
        return f"SubjectsBatch(batch_size={self.batch_size}, images=[{', '.join(self._images)}])"


def _slice_params(params: dict[str, Any], index: int, batched_keys: list[str]) -> dict[str, Any]:
    """One element's view of a per-instance params dict (batch.py:337-363)."""
    return {
        key: (value[index] if key in batched_keys and isinstance(value, list) else value)
        for key, value in params.items()
        if key not in _BATCH_META_KEYS
    }


def _slice_history(history: list[Any], index: int) -> list[Any]:
    """Per-subject history for batch element *index* (batch.py:365-399)."""
    result = []
    for trace in history:
        params = getattr(trace, "params", None)
        if not isinstance(params, dict) or "_batched_keys" not in params:
            result.append(trace)
            continue
        size = params.get("_batch_size")
        if size is not None and not 0 <= index < size:
            raise IndexError(
                f"Cannot extract per-instance history for element {index}: the transform was"
                f" recorded for a batch of size {size}"
            )
        keep = params.get("_keep")
        if keep is not None and not keep[index]:
            continue
        result.append(dataclasses.replace(trace, params=_slice_params(params, index, params["_batched_keys"])))
    return result
