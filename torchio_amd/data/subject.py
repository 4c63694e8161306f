"""Subject container (mirror of the in-memory part of reference ``data/subject.py:25``)."""
from __future__ import annotations

import copy as _copy
from typing import Any

from .image import Image


class Subject:
    """Named images plus arbitrary metadata; carries the transform history."""

    def __init__(self, **entries: Any) -> None:
        object.__setattr__(self, "_images", {})
        object.__setattr__(self, "_metadata", {})
        self.applied_transforms: list = []
        for key, value in entries.items():
            if isinstance(value, Image):
                self._images[key] = value
            else:
                self._metadata[key] = value
        if not self._images:
            raise ValueError("A Subject needs at least one Image")

    @property
    def images(self) -> dict[str, Image]:
        return self._images

    @property
    def metadata(self) -> dict[str, Any]:
        return self._metadata

    def __getattr__(self, name: str):
        if name.startswith("_"):
            raise AttributeError(name)
        images = object.__getattribute__(self, "_images")
        if name in images:
            return images[name]
        metadata = object.__getattribute__(self, "_metadata")
        if name in metadata:
            return metadata[name]
        raise AttributeError(f"Subject has no attribute {name!r}")

    def __getitem__(self, key: str):
        if key in self._images:
            return self._images[key]
        return self._metadata[key]

    def __contains__(self, key: str) -> bool:
        return key in self._images or key in self._metadata

    def keys(self):
        return [*self._images, *self._metadata]

    @property
    def spatial_shape(self) -> tuple[int, int, int]:
        return next(iter(self._images.values())).spatial_shape

    @property
    def device(self):
        return next(iter(self._images.values())).device

    def to(self, *args, **kwargs) -> "Subject":
        for image in self._images.values():
            image.to(*args, **kwargs)
        return self

    def load(self) -> "Subject":
        return self

    # -- history -------------------------------------------------------------
    def get_inverse_transform(self, *, warn: bool = True, ignore_intensity: bool = False):
        from ..transforms.inverse import get_inverse_transform  # noqa: PLC0415

        return get_inverse_transform(self.applied_transforms, warn=warn, ignore_intensity=ignore_intensity)

    def apply_inverse_transform(self, **kwargs):
        result = self.get_inverse_transform(**kwargs)(self)
        result.applied_transforms = []
        return result

    def clear_history(self) -> None:
        self.applied_transforms = []

    def __deepcopy__(self, memo):
        entries = {k: _copy.deepcopy(v, memo) for k, v in self._images.items()}
        entries.update({k: _copy.deepcopy(v, memo) for k, v in self._metadata.items()})
        new = Subject(**entries)
        new.applied_transforms = list(self.applied_transforms)
        return new

    def __repr__(self) -> str:
        return f"Subject(images={list(self._images)}, metadata={list(self._metadata)})"
