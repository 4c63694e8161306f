"""Batch loaders: a ``torch.utils.data.DataLoader`` whose batches are the containers the transforms take.

``SubjectsLoader(dataset, **kw)`` collates ``Subject`` items into a :class:`~torchio_amd.data.batch.SubjectsBatch`
and ``ImagesLoader`` collates ``Image`` items into an :class:`~torchio_amd.data.batch.ImagesBatch` — the
reference's ``src/torchio/loader.py`` contract, including the refusal of a user ``collate_fn`` and the
``StudiesLoader`` / ``collate_studies`` aliases.  Both classes come out of one factory.

Device tensors do not cross a worker-process boundary cheaply: device-resident datasets are meant to run with
``num_workers=0`` (``Queue`` already overlaps loading with threads).
"""
from __future__ import annotations

from collections.abc import Callable
from collections.abc import Sequence
from typing import Any

from torch.utils.data import DataLoader
from torch.utils.data import Dataset

from .data.batch import ImagesBatch
from .data.batch import SubjectsBatch


def collate_subjects(batch: Sequence[Any]) -> SubjectsBatch:
    """``[Subject, ...]`` -> one ``SubjectsBatch`` (5-D tensors stacked per image name, metadata as lists)."""
    return SubjectsBatch.from_subjects(list(batch))


def collate_images(batch: Sequence[Any]) -> ImagesBatch:
    """``[Image, ...]`` -> one ``ImagesBatch``."""
    return ImagesBatch.from_images(list(batch))


def _loader_class(name: str, collate: Callable[[Sequence[Any]], Any], yields: str) -> type[DataLoader]:
    def __init__(self, dataset: Dataset, **kwargs: Any) -> None:
        if "collate_fn" in kwargs:
            raise ValueError(f"{name} sets collate_fn automatically; pass a plain DataLoader if you need a custom collate_fn")
        DataLoader.__init__(self, dataset, collate_fn=collate, **kwargs)

    doc = f"``DataLoader`` that yields ``{yields}`` instances; every other keyword goes to ``DataLoader`` unchanged."
    return type(name, (DataLoader,), {"__init__": __init__, "__doc__": doc, "__module__": __name__})


SubjectsLoader = _loader_class("SubjectsLoader", collate_subjects, "SubjectsBatch")
ImagesLoader = _loader_class("ImagesLoader", collate_images, "ImagesBatch")

# names radiology users know
StudiesLoader = SubjectsLoader
collate_studies = collate_subjects
