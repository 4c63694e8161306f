"""Patch samplers: where patches go, and how they are cut out of a (possibly GPU-resident) subject.

Public contract = the reference's (``src/torchio/data/sampler.py``: ``PatchSampler``, ``GridSampler``,
``UniformSampler``, ``WeightedSampler``, ``LabelSampler`` with the same constructor arguments, the same
consumption of the global CPU generator and the same errors), implemented here around three small
pieces:

* :func:`axis_origins` / :func:`grid_placements` — the regular grid of a dense-inference sampler as pure
  functions of (extent, patch, overlap);
* :func:`cut` — a patch is four views (``Image.__getitem__``: no copy, affine origin moved) of
  whatever tensors the subject holds, so a subject that lives in HBM yields device-resident patches and the
  feeding side of the hot path never crosses PCIe;
* :class:`CentreLaw` — a discrete distribution over patch CENTRES (probability map or label weights) with
  the voxels whose patch would stick out of the volume masked by broadcasting three 1-D masks.

``GridSampler`` pairs with :class:`~torchio_amd.data.aggregator.PatchAggregator`.
"""
from __future__ import annotations

import itertools
from collections.abc import Iterator
from typing import Any

import torch
from torch import Tensor
from torch.utils.data import Dataset
from torch.utils.data import IterableDataset

from .patch import PatchLocation
from .subject import Subject

Triple = tuple[int, int, int]


def _triple(value) -> Triple:
    return (value, value, value) if isinstance(value, int) else tuple(value)  # type: ignore[return-value]


def axis_origins(extent: int, patch: int, overlap: int) -> list[int]:
    """First voxels of the patches along one axis: stride ``patch - overlap``, plus one flush with the far end."""
    last = max(extent - patch, 0)
    stride = max(patch - overlap, 1)
    origins = list(range(0, extent - patch + 1, stride))
    if not origins or origins[-1] != last:
        origins.append(last)
    return origins


def grid_placements(shape: Triple, patch: Triple, overlap: Triple) -> list[PatchLocation]:
    """Every patch of the regular grid, k fastest (the order the aggregator's batches arrive in)."""
    per_axis = [axis_origins(shape[d], patch[d], overlap[d]) for d in range(3)]
    return [PatchLocation(index=origin, size=patch) for origin in itertools.product(*per_axis)]


def cut(subject: Subject, where: PatchLocation) -> Subject:
    """The sub-volume *where* of every image of *subject* (views), metadata carried over, tagged with its location."""
    window = (slice(None), *where.to_slices())
    fields: dict[str, Any] = {name: image[window] for name, image in subject.images.items()}
    fields.update(subject.metadata)
    fields["patch_location"] = where
    return Subject(**fields)


def corner_of(centre: Triple, shape: Triple, patch: Triple) -> Triple:
    """Patch corner for a centre voxel, pushed back inside the volume where the patch would overhang."""
    return tuple(min(max(centre[d] - patch[d] // 2, 0), shape[d] - patch[d]) for d in range(3))  # type: ignore[return-value]


class CentreLaw:
    """Unnormalised probability of every voxel being a patch centre, as a flat CPU vector to draw from."""

    def __init__(self, weights: Tensor, patch: Triple) -> None:
        shape = tuple(weights.shape)
        allowed = torch.ones(shape, dtype=torch.bool, device=weights.device)
        for axis in range(3):
            half = patch[axis] // 2
            position = torch.arange(shape[axis], device=weights.device)
            inside = (position >= half) & (position < shape[axis] - half) if half > 0 else torch.ones_like(position, dtype=torch.bool)
            view = [1, 1, 1]
            view[axis] = shape[axis]
            allowed &= inside.view(view)
        self.shape: Triple = shape  # type: ignore[assignment]
        self.map = torch.where(allowed, weights, torch.zeros((), dtype=weights.dtype, device=weights.device))

    def flat_cpu(self) -> Tensor:
        # the draw must consume the global CPU generator like the reference's: a device-resident map comes over once
        return self.map.reshape(-1).cpu()

    def centre(self, flat_index: int) -> Triple:
        ij, k = divmod(flat_index, self.shape[2])
        i, j = divmod(ij, self.shape[1])
        return (i, j, k)


class PatchSampler:
    """Common part: the patch size and the cutting (reference sampler.py:24-72)."""

    def __init__(self, patch_size) -> None:
        self.patch_size: Triple = _triple(patch_size)

    def __call__(self, subject: Subject, num_patches: int | None = None) -> Iterator[Subject]:
        raise NotImplementedError(f"{type(self).__name__} must implement __call__")

    def _extract_patch(self, subject: Subject, location: PatchLocation) -> Subject:
        return cut(subject, location)


class GridSampler(PatchSampler, Dataset):
    """Dense inference: a map-style dataset of the grid's patches (reference sampler.py:75-190)."""

    def __init__(self, subject: Subject, patch_size, patch_overlap=0, padding_mode=None, fill: float = 0) -> None:
        super().__init__(patch_size)
        self.patch_overlap: Triple = _triple(patch_overlap)
        self.padding_mode = padding_mode
        self.fill = fill
        self.subject = subject if padding_mode is None else self._padded(subject)
        self.locations = grid_placements(tuple(self.subject.spatial_shape), self.patch_size, self.patch_overlap)

    def _padded(self, subject: Subject) -> Subject:
        """Half an overlap of context on every side, so that border patches are blended like inner ones."""
        from ..transforms.pad import Pad  # noqa: PLC0415

        padding = tuple(itertools.chain.from_iterable((o // 2, o // 2) for o in self.patch_overlap))
        return Pad(padding=padding, padding_mode=self.padding_mode, fill=self.fill, copy=False)(subject)

    def __len__(self) -> int:
        return len(self.locations)

    def __getitem__(self, index: int) -> Subject:
        return cut(self.subject, self.locations[index])


class _RandomSampler(PatchSampler, IterableDataset):
    """Iterable samplers: an endless (or ``num_patches`` long) stream of patches of one subject."""

    def __init__(self, subject: Subject, patch_size, num_patches: int | None = None) -> None:
        super().__init__(patch_size)
        self.subject = subject
        self.num_patches = num_patches

    def __iter__(self) -> Iterator[Subject]:
        return self(self.subject, self.num_patches)

    def __call__(self, subject: Subject, num_patches: int | None = None) -> Iterator[Subject]:
        corners = self._corners(subject)
        budget = num_patches or self.num_patches
        stream = corners if budget is None else itertools.islice(corners, budget)
        return (cut(subject, PatchLocation(index=corner, size=self.patch_size)) for corner in stream)

    def _corners(self, subject: Subject) -> Iterator[Triple]:
        raise NotImplementedError


class UniformSampler(_RandomSampler):
    """Corners uniform over the positions where the patch fits (reference sampler.py:193-247)."""

    def _corners(self, subject: Subject) -> Iterator[Triple]:
        shape = subject.spatial_shape
        spans = [max(shape[d] - self.patch_size[d], 0) + 1 for d in range(3)]
        while True:  # one scalar torch.randint per axis, i then j then k: the reference's consumption of the generator
            yield tuple(int(torch.randint(0, span, (1,)).item()) for span in spans)  # type: ignore[misc]


class WeightedSampler(_RandomSampler):
    """Centres drawn from a probability-map image of the subject (reference sampler.py:250-310)."""

    def __init__(self, subject: Subject, patch_size, probability_map: str, num_patches: int | None = None) -> None:
        super().__init__(subject, patch_size, num_patches)
        self.probability_map = probability_map

    def _weights(self, subject: Subject) -> Tensor:
        return subject.images[self.probability_map].data[0].float()

    def _law(self, subject: Subject) -> CentreLaw:
        return CentreLaw(self._weights(subject), self.patch_size)

    def _build_probability_map(self) -> Tensor:
        return self._law(self.subject).map

    def _corners(self, subject: Subject) -> Iterator[Triple]:
        law = self._law(subject)
        flat = law.flat_cpu()
        if flat.sum() == 0:
            raise RuntimeError(f"Probability map '{self.probability_map}' is all zeros")
        shape = tuple(subject.spatial_shape)
        while True:
            yield corner_of(law.centre(int(torch.multinomial(flat, 1).item())), shape, self.patch_size)


class LabelSampler(WeightedSampler):
    """Centres on labelled voxels, optionally weighted per label (reference sampler.py:313-366)."""

    def __init__(self, subject: Subject, patch_size, label_name: str, label_probabilities: dict | None = None,
                 num_patches: int | None = None) -> None:
        super().__init__(subject, patch_size, probability_map=label_name, num_patches=num_patches)
        self.label_name = label_name
        self.label_probabilities = label_probabilities

    def _weights(self, subject: Subject) -> Tensor:
        labels = subject.images[self.label_name].data[0]
        if self.label_probabilities is None:
            return (labels > 0).float()
        weights = torch.zeros(labels.shape, dtype=torch.float32, device=labels.device)
        for label, weight in self.label_probabilities.items():
            weights = torch.where(labels == label, torch.full((), float(weight), device=labels.device), weights)
        return weights
