"""``Queue``: a shuffling buffer of patches between the subjects and the training loop.

Contract = the reference's ``src/torchio/data/queue.py`` (constructor arguments, epoch semantics, use of
Python's ``random`` for both shuffles, ``subject_sampler`` for the multi-GPU split, the size properties).
The implementation is a small pipeline of generators:

    subject order  ->  prepared subjects (load + transform, optionally on a thread pool, order kept)
                   ->  patches (``patches_per_volume`` from each)  ->  buffer of ``max_length``  ->  consumer

A subject that lives on the GPU is transformed there — the HIP engine is re-entrant from worker threads: no
global mutable state in the C ABI, thread-local error text — and the samplers hand out views of the
transformed tensors, so the buffer holds device-resident patches and nothing crosses PCIe on its way to the
model.
"""
from __future__ import annotations

import itertools
import random
from collections import deque
from collections.abc import Iterable
from collections.abc import Iterator
from collections.abc import Sequence
from collections.abc import Sized
from concurrent.futures import ThreadPoolExecutor
from typing import Any

from torch.utils.data import IterableDataset
from torch.utils.data import Sampler

from .sampler import PatchSampler
from .subject import Subject

_UNITS = ("Bytes", "KiB", "MiB", "GiB", "TiB")


def _in_order(pool: ThreadPoolExecutor, work, items: Iterable) -> Iterator:
    """``map(work, items)`` on *pool*, results in submission order, at most one item submitted ahead of demand
    per idle result (the consumer decides how fast the subject list is walked)."""
    waiting: deque = deque()
    for item in items:
        waiting.append(pool.submit(work, item))
        while waiting and waiting[0].done():
            yield waiting.popleft().result()
    while waiting:
        yield waiting.popleft().result()


class Queue(IterableDataset):
    """Buffer of patches for stochastic patch-based training.

    Args:
        subjects: subjects to sample patches from.
        patch_sampler: called as ``patch_sampler(subject)``; ``patches_per_volume`` patches are taken from it.
        max_length: patches collected before the buffer is (shuffled and) handed out.
        patches_per_volume: patches per subject.
        num_workers: threads that load / transform subjects ahead of the consumer (0 = inline).
        shuffle_subjects: new random subject order every epoch.
        shuffle_patches: shuffle each buffer before handing it out.
        transform: applied to every subject before its patches are cut.
        subject_sampler: a ``torch.utils.data.Sampler`` of subject indices (``DistributedSampler`` = the multi-GPU
            split of the feeding side, reference queue.py:48-50); excludes ``shuffle_subjects``.
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

    # ---- the pipeline ----------------------------------------------------------------------
    def _subject_order(self) -> list[Subject]:
        if self.subject_sampler is not None:
            return [self.subjects[index] for index in self.subject_sampler]
        order = list(self.subjects)
        if self.shuffle_subjects:
            random.shuffle(order)
        return order

    def _ready(self, subject: Subject) -> Subject:
        subject.load()
        return subject if self.transform is None else self.transform(subject)

    def _patches_of(self, subject: Subject) -> list[Subject]:
        return list(itertools.islice(self.patch_sampler(subject), self.patches_per_volume))

    def _hand_out(self, buffer: list[Subject]) -> Iterator[Subject]:
        if self.shuffle_patches:
            random.shuffle(buffer)
        while buffer:
            yield buffer.pop()

    def _stream(self, prepared: Iterable[Subject]) -> Iterator[Subject]:
        buffer: list[Subject] = []
        for subject in prepared:
            buffer += self._patches_of(subject)
            if len(buffer) >= self.max_length:
                yield from self._hand_out(buffer)
        yield from self._hand_out(buffer)

    def __iter__(self) -> Iterator[Subject]:
        order = self._subject_order()
        if self.num_workers <= 0:
            yield from self._stream(map(self._ready, order))
            return
        with ThreadPoolExecutor(max_workers=self.num_workers) as pool:
            yield from self._stream(_in_order(pool, self._ready, order))

    # ---- sizes -------------------------------------------------------------------------------
    @property
    def num_subjects(self) -> int:
        if self.subject_sampler is None:
            return len(self.subjects)
        if not isinstance(self.subject_sampler, Sized):
            raise TypeError("subject_sampler must have a __len__ method")
        return len(self.subject_sampler)

    @property
    def patches_per_epoch(self) -> int:
        return self.patches_per_volume * self.num_subjects

    @property
    def max_memory(self) -> int:
        """Bytes a full buffer of float32 patches takes (every image of the first subject counted)."""
        channels = sum(image.num_channels for image in self.subjects[0].images.values())
        si, sj, sk = self.patch_sampler.patch_size
        return self.max_length * channels * si * sj * sk * 4

    @property
    def max_memory_pretty(self) -> str:
        amount, unit = float(self.max_memory), 0
        while amount >= 1024 and unit < len(_UNITS) - 1:
            amount, unit = amount / 1024, unit + 1
        return f"{amount:.0f} {_UNITS[unit]}" if unit == 0 else f"{amount:.1f} {_UNITS[unit]}"
