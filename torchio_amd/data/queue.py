"""Patch queue for patch-based training (mirror of reference ``src/torchio/data/queue.py``).

Same constructor, iteration order, shuffling (Python's ``random`` module, like the reference)
and properties.  Subjects that live on the GPU are transformed there (the transform runs the
HIP engine, re-entrant from the worker threads: the C ABI keeps no global mutable state and
its error text is thread-local) and the sampler hands out views of the transformed tensors,
so the buffer holds device-resident patches and nothing crosses PCIe on the way to the model.
"""
from __future__ import annotations

import random as _random
from collections import deque
from collections.abc import Iterator
from collections.abc import Sequence
from collections.abc import Sized
from concurrent.futures import Future
from concurrent.futures import ThreadPoolExecutor
from itertools import islice
from typing import Any

from torch.utils.data import IterableDataset
from torch.utils.data import Sampler

from .sampler import PatchSampler
from .subject import Subject


class Queue(IterableDataset):
    """Buffer of patches for stochastic patch-based training (queue.py:23-92).

    Args:
        subjects: subjects to sample patches from.
        patch_sampler: sampler called as ``patch_sampler(subject)``.
        max_length: patches held in the buffer before it is (shuffled and) drained.
        patches_per_volume: patches taken from each subject.
        num_workers: background threads that load / transform subjects (0 = synchronous).
        shuffle_subjects: shuffle the subject order at the start of each epoch.
        shuffle_patches: shuffle the buffer before it is drained.
        transform: applied to each subject before patch extraction.
        subject_sampler: a ``torch.utils.data.Sampler`` of subject indices (e.g.
            ``DistributedSampler``: the multi-GPU split of the feeding side); requires
            ``shuffle_subjects=False``.
    """

    def __init__(
        self,
        subjects: Sequence[Subject],
        patch_sampler: PatchSampler,
        max_length: int = 300,
        patches_per_volume: int = 10,
        num_workers: int = 0,
        shuffle_subjects: bool = True,
        shuffle_patches: bool = True,
        transform: Any | None = None,
        subject_sampler: Sampler | None = None,
    ) -> None:
        if subject_sampler is not None and shuffle_subjects:
            raise ValueError("shuffle_subjects must be False when subject_sampler is provided (the sampler controls the order)")
        self.subjects = subjects
        self.patch_sampler = patch_sampler
        self.max_length = max_length
        self.patches_per_volume = patches_per_volume
        self.num_workers = num_workers
        self.shuffle_subjects = shuffle_subjects
        self.shuffle_patches = shuffle_patches
        self.transform = transform
        self.subject_sampler = subject_sampler

    # -- iteration -------------------------------------------------------------------
    def __iter__(self) -> Iterator[Subject]:
        buffer: list[Subject] = []
        order = self._epoch_order()
        if self.num_workers > 0:
            yield from self._iterate_with_workers(order, buffer)
        else:
            for subject in order:
                buffer.extend(self._sample_patches(self._prepare(subject)))
                if len(buffer) >= self.max_length:
                    yield from self._drain(buffer)
            yield from self._drain(buffer)

    def _iterate_with_workers(self, order: Iterator[Subject], buffer: list[Subject]) -> Iterator[Subject]:
        """Subjects are prepared by a thread pool; finished ones are consumed in submission order (queue.py:114-145)."""
        with ThreadPoolExecutor(max_workers=self.num_workers) as pool:
            pending: deque[Future] = deque()
            for subject in order:
                pending.append(pool.submit(self._prepare, subject))
                while pending and pending[0].done():
                    buffer.extend(self._sample_patches(pending.popleft().result()))
                if len(buffer) >= self.max_length:
                    yield from self._drain(buffer)
            for future in pending:
                buffer.extend(self._sample_patches(future.result()))
        yield from self._drain(buffer)

    def _drain(self, buffer: list[Subject]) -> Iterator[Subject]:
        if self.shuffle_patches:
            _random.shuffle(buffer)
        while buffer:
            yield buffer.pop()

    def _prepare(self, subject: Subject) -> Subject:
        subject.load()
        if self.transform is not None:
            subject = self.transform(subject)
        return subject

    def _sample_patches(self, subject: Subject) -> list[Subject]:
        return list(islice(iter(self.patch_sampler(subject)), self.patches_per_volume))

    def _epoch_order(self) -> Iterator[Subject]:
        if self.subject_sampler is not None:
            return (self.subjects[index] for index in list(self.subject_sampler))
        subjects = list(self.subjects)
        if self.shuffle_subjects:
            _random.shuffle(subjects)
        return iter(subjects)

    # -- bookkeeping -----------------------------------------------------------------
    @property
    def num_subjects(self) -> int:
        if self.subject_sampler is not None:
            if not isinstance(self.subject_sampler, Sized):
                raise TypeError("subject_sampler must have a __len__ method")
            return len(self.subject_sampler)
        return len(self.subjects)

    @property
    def patches_per_epoch(self) -> int:
        return self.num_subjects * self.patches_per_volume

    @property
    def max_memory(self) -> int:
        """Upper bound of the buffer's footprint in bytes, float32 patches (queue.py:194-203)."""
        channels = sum(image.num_channels for image in self.subjects[0].images.values())
        voxels = 1
        for size in self.patch_sampler.patch_size:
            voxels *= size
        return 4 * channels * voxels * self.max_length

    @property
    def max_memory_pretty(self) -> str:
        value = float(self.max_memory)
        for unit in ("Bytes", "KiB", "MiB", "GiB", "TiB"):
            if value < 1024 or unit == "TiB":
                return f"{value:.0f} {unit}" if unit == "Bytes" else f"{value:.1f} {unit}"
            value /= 1024
        return f"{value:.1f} TiB"
