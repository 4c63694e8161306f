"""``SubjectsLoader`` / ``ImagesLoader`` (mirror of reference ``loader.py``): ``DataLoader`` + batch collation.

The feeding side of the path: patches from the samplers (views of device-resident volumes) or whole
subjects are stacked into the ``SubjectsBatch`` / ``ImagesBatch`` the transforms and the model take
(loader.py:15-95).  Device tensors cannot cross a worker-process boundary cheaply, so device-resident
datasets are meant to run with ``num_workers=0`` (the ``Queue`` already overlaps loading with threads).
"""
from __future__ import annotations

from collections.abc import Sequence
from typing import Any

from torch.utils.data import DataLoader
from torch.utils.data import Dataset

from .data.batch import ImagesBatch
from .data.batch import SubjectsBatch


def collate_subjects(batch: Sequence[Any]) -> SubjectsBatch:
    """A list of ``Subject`` instances as one ``SubjectsBatch`` of stacked 5-D tensors (loader.py:15-25)."""
    return SubjectsBatch.from_subjects(list(batch))


def collate_images(batch: Sequence[Any]) -> ImagesBatch:
    """A list of ``Image`` instances as one ``ImagesBatch`` (loader.py:28-38)."""
    return ImagesBatch.from_images(list(batch))


def _refuse_collate_fn(kwargs: dict, name: str) -> None:
    if "collate_fn" in kwargs:
        raise ValueError(f"{name} sets collate_fn automatically; pass a plain DataLoader if you need a custom collate_fn")


class SubjectsLoader(DataLoader):
    """``DataLoader`` that yields ``SubjectsBatch`` instances (loader.py:41-65)."""

    def __init__(self, dataset: Dataset, **kwargs: Any) -> None:
        _refuse_collate_fn(kwargs, "SubjectsLoader")
        super().__init__(dataset, collate_fn=collate_subjects, **kwargs)


class ImagesLoader(DataLoader):
    """``DataLoader`` that yields ``ImagesBatch`` instances (loader.py:68-91)."""

    def __init__(self, dataset: Dataset, **kwargs: Any) -> None:
        _refuse_collate_fn(kwargs, "ImagesLoader")
        super().__init__(dataset, collate_fn=collate_images, **kwargs)


# aliases for radiology users (loader.py:93-95)
StudiesLoader = SubjectsLoader
collate_studies = collate_subjects
