"""``Compose`` (mirror of reference ``transforms/compose.py:38-93``).

A plain sequential loop: the input is deep-copied once, wrapped once, every
child runs with ``copy=False`` and appends its own history record.  ``Compose``
itself draws nothing from the RNG and records nothing (compose.py:84-93), which
is why ``Compose([Affine, ElasticDeformation])`` stays two separate resamplings
— a drop-in may not fuse them.
"""
from __future__ import annotations

import copy as _copy
from collections.abc import Mapping
from collections.abc import Sequence
from typing import Any

from .transform import Transform
from .transform import _wrap


class Compose(Transform):
    """Apply transforms one after the other."""

    def __init__(
        self,
        transforms: Sequence[Transform] | Mapping[str, Transform] | None = None,
        *,
        copy: bool = True,
        **kwargs: Any,
    ) -> None:
        super().__init__(copy=copy, **kwargs)
        if transforms is None:
            self.transforms: list[Transform] = []
        elif isinstance(transforms, Mapping):
            self.transforms = list(transforms.values())
        else:
            self.transforms = list(transforms)

    def _forward(self, data: Any) -> Any:
        batch, unwrap = _wrap(data)
        for transform in self.transforms:
            previous = transform.copy
            transform.copy = False
            try:
                batch = transform(batch)
            finally:
                transform.copy = previous
        return unwrap(batch)

    def __len__(self) -> int:
        return len(self.transforms)

    def __getitem__(self, index: int) -> Transform:
        return self.transforms[index]

    def __repr__(self) -> str:
        inner = ", ".join(repr(t) for t in self.transforms)
        return f"Compose([{inner}])"
