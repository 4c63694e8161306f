"""Patch samplers for training and inference (mirror of reference ``src/torchio/data/sampler.py``).

Same classes, arguments, draw order on the global RNG and errors.  A patch is a crop
(``Image.__getitem__``: a view, affine origin moved) of whatever tensor the subject holds,
so a subject that lives on the GPU yields device-resident patches - the feeding side of the
hot path never goes through the host.  ``GridSampler`` pairs with
:class:`~torchio_amd.data.aggregator.PatchAggregator`.
"""
from __future__ import annotations

from collections.abc import Iterator
from typing import Any

import numpy as np
import torch
from torch import Tensor
from torch.utils.data import Dataset
from torch.utils.data import IterableDataset

from .patch import PatchLocation
from .subject import Subject


class PatchSampler:
    """Base class (sampler.py:24-72)."""

    def __init__(self, patch_size) -> None:
        if isinstance(patch_size, int):
            patch_size = (patch_size, patch_size, patch_size)
        self.patch_size: tuple[int, int, int] = tuple(patch_size)  # type: ignore[assignment]

    def __call__(self, subject: Subject, num_patches: int | None = None) -> Iterator[Subject]:
        raise NotImplementedError(f"{type(self).__name__} must implement __call__")

    def _extract_patch(self, subject: Subject, location: PatchLocation) -> Subject:
        si, sj, sk = location.to_slices()
        kwargs: dict[str, Any] = {name: image[:, si, sj, sk] for name, image in subject.images.items()}
        kwargs.update(subject.metadata)
        kwargs["patch_location"] = location
        return Subject(**kwargs)


class GridSampler(PatchSampler, Dataset):
    """Patches on a regular grid for dense inference: map-style dataset (sampler.py:75-190)."""

    def __init__(self, subject: Subject, patch_size, patch_overlap=0, padding_mode=None, fill: float = 0) -> None:
        super().__init__(patch_size)
        if isinstance(patch_overlap, int):
            patch_overlap = (patch_overlap, patch_overlap, patch_overlap)
        self.patch_overlap: tuple[int, int, int] = tuple(patch_overlap)  # type: ignore[assignment]
        self.padding_mode = padding_mode
        self.fill = fill
        self.subject = self._maybe_pad(subject)
        self.locations = self._compute_locations(self.subject.spatial_shape)

    def __len__(self) -> int:
        return len(self.locations)

    def __getitem__(self, index: int) -> Subject:
        return self._extract_patch(self.subject, self.locations[index])

    def _maybe_pad(self, subject: Subject) -> Subject:
        """Pad the volume by ``overlap // 2`` on each side before sampling (sampler.py:127-147)."""
        if self.padding_mode is None:
            return subject
        from ..transforms.pad import Pad  # noqa: PLC0415

        border = tuple(v // 2 for v in self.patch_overlap)
        padding = (border[0], border[0], border[1], border[1], border[2], border[2])
        return Pad(padding=padding, padding_mode=self.padding_mode, fill=self.fill, copy=False)(subject)

    def _compute_locations(self, spatial_shape) -> list[PatchLocation]:
        indices_per_axis: list[list[int]] = []
        for dim in range(3):
            size, patch, overlap = spatial_shape[dim], self.patch_size[dim], self.patch_overlap[dim]
            step = max(patch - overlap, 1)
            indices = list(range(0, size - patch + 1, step))
            if not indices or indices[-1] != size - patch:
                indices.append(max(size - patch, 0))
            indices_per_axis.append(indices)
        return [
            PatchLocation(index=(i, j, k), size=self.patch_size)
            for i in indices_per_axis[0]
            for j in indices_per_axis[1]
            for k in indices_per_axis[2]
        ]


class UniformSampler(PatchSampler, IterableDataset):
    """Random patches with uniform spatial probability (sampler.py:193-247)."""

    def __init__(self, subject: Subject, patch_size, num_patches: int | None = None) -> None:
        super().__init__(patch_size)
        self.subject = subject
        self.num_patches = num_patches

    def __call__(self, subject: Subject, num_patches: int | None = None) -> Iterator[Subject]:
        limit = num_patches or self.num_patches
        count = 0
        while limit is None or count < limit:
            location = PatchLocation(index=self._random_index(subject.spatial_shape), size=self.patch_size)
            yield self._extract_patch(subject, location)
            count += 1

    def __iter__(self) -> Iterator[Subject]:
        return self(self.subject, self.num_patches)

    def _random_index(self, spatial_shape) -> tuple[int, int, int]:
        def draw(d: int) -> int:  # one torch.randint per axis on the global CPU generator, i then j then k
            high = max(spatial_shape[d] - self.patch_size[d], 0) + 1
            return int(torch.randint(0, high, (1,)).item())

        return (draw(0), draw(1), draw(2))


class WeightedSampler(PatchSampler, IterableDataset):
    """Random patches whose centres follow a probability map image (sampler.py:250-310)."""

    def __init__(self, subject: Subject, patch_size, probability_map: str, num_patches: int | None = None) -> None:
        super().__init__(patch_size)
        self.subject = subject
        self.probability_map = probability_map
        self.num_patches = num_patches

    def __call__(self, subject: Subject, num_patches: int | None = None) -> Iterator[Subject]:
        prob_data = self._build_probability_map_for(subject)
        # the draw runs on the CPU generator like the reference's (a device-resident map is brought over
        # once per call; torch.multinomial on another device would consume a different generator)
        flat = prob_data.flatten().cpu()
        if flat.sum() == 0:
            raise RuntimeError(f"Probability map '{self.probability_map}' is all zeros")
        limit = num_patches or self.num_patches
        count = 0
        while limit is None or count < limit:
            flat_index = int(torch.multinomial(flat, 1).item())
            center = tuple(int(x) for x in np.unravel_index(flat_index, tuple(prob_data.shape)))
            index = _center_to_corner(center, subject.spatial_shape, self.patch_size)
            yield self._extract_patch(subject, PatchLocation(index=index, size=self.patch_size))
            count += 1

    def __iter__(self) -> Iterator[Subject]:
        return self(self.subject, self.num_patches)

    def _build_probability_map_for(self, subject: Subject) -> Tensor:
        prob_data = subject.images[self.probability_map].data[0].float()
        return _mask_borders(prob_data, subject.spatial_shape, self.patch_size)

    def _build_probability_map(self) -> Tensor:
        return self._build_probability_map_for(self.subject)


class LabelSampler(WeightedSampler):
    """Random patches centred on labelled voxels (sampler.py:313-366)."""

    def __init__(self, subject: Subject, patch_size, label_name: str, label_probabilities: dict | None = None,
                 num_patches: int | None = None) -> None:
        super().__init__(subject, patch_size, probability_map=label_name, num_patches=num_patches)
        self.label_name = label_name
        self.label_probabilities = label_probabilities

    def _build_probability_map_for(self, subject: Subject) -> Tensor:
        label_data = subject.images[self.label_name].data[0]
        if self.label_probabilities is not None:
            prob = torch.zeros_like(label_data, dtype=torch.float32)
            for label, weight in self.label_probabilities.items():
                prob[label_data == label] = weight
        else:
            prob = (label_data > 0).float()
        return _mask_borders(prob, subject.spatial_shape, self.patch_size)


def _mask_borders(prob: Tensor, spatial_shape, patch_size) -> Tensor:
    """Zero the probability where a patch centre cannot sit (sampler.py:373-392)."""
    prob = prob.clone()
    for d in range(3):
        half = patch_size[d] // 2
        if half > 0:
            low: list[slice] = [slice(None)] * 3
            low[d] = slice(0, half)
            prob[tuple(low)] = 0
        tail = spatial_shape[d] - half
        if tail < spatial_shape[d]:
            high: list[slice] = [slice(None)] * 3
            high[d] = slice(tail, None)
            prob[tuple(high)] = 0
    return prob


def _center_to_corner(center, spatial_shape, patch_size) -> tuple[int, int, int]:
    """Centre voxel -> patch corner, clamped into the volume (sampler.py:395-408)."""
    corner = []
    for d in range(3):
        value = max(0, center[d] - patch_size[d] // 2)
        corner.append(min(value, spatial_shape[d] - patch_size[d]))
    return (corner[0], corner[1], corner[2])
