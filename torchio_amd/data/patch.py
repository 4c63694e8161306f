"""Patch location metadata (mirror of reference ``src/torchio/data/patch.py``)."""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class PatchLocation:
    """Where a patch sits in its volume: corner voxel ``index`` and spatial ``size`` (patch.py:10-23)."""

    index: tuple[int, int, int]
    size: tuple[int, int, int]
    subject_index: int | None = None

    @property
    def index_ini(self) -> tuple[int, int, int]:
        return self.index

    @property
    def index_fin(self) -> tuple[int, int, int]:
        return (self.index[0] + self.size[0], self.index[1] + self.size[1], self.index[2] + self.size[2])

    def to_slices(self) -> tuple[slice, slice, slice]:
        ini, fin = self.index_ini, self.index_fin
        return (slice(ini[0], fin[0]), slice(ini[1], fin[1]), slice(ini[2], fin[2]))

    def scaled(self, factor: tuple[float, float, float]) -> "PatchLocation":
        """Indices and size multiplied by ``factor`` and rounded half-to-even like the reference (patch.py:49-64)."""
        return PatchLocation(
            index=tuple(round(self.index[d] * factor[d]) for d in range(3)),  # type: ignore[arg-type]
            size=tuple(round(self.size[d] * factor[d]) for d in range(3)),  # type: ignore[arg-type]
            subject_index=self.subject_index,
        )
