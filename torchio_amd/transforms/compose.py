"""``Compose``, ``OneOf``, ``SomeOf`` (mirror of reference ``transforms/compose.py``).

A plain sequential loop: the input is deep-copied once, wrapped once, every
child runs with ``copy=False`` and appends its own history record.  ``Compose``
itself draws nothing from the RNG and records nothing (compose.py:84-93), which
is why ``Compose([Affine, ElasticDeformation])`` stays two separate resamplings
— a drop-in may not fuse them.
"""
from __future__ import annotations

import copy as _copy
import os
from collections.abc import Mapping
from collections.abc import Sequence
from typing import Any

import contextlib

import torch

from ..data.batch import SubjectsBatch
from .. import ops
from .transform import Transform
from .transform import _return_home
from .transform import _stage_on_engine_device
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
        home = _stage_on_engine_device(batch)
        if home is not None:  # host-resident data: ONE trip through the device for the whole container (transform.py)
            try:
                return unwrap(_return_home(self._forward(batch), home))
            finally:
                _return_home(batch, home)
        if self._may_draw_ahead():
            # Every child only reads the batch's geometry for its parameters and keeps that geometry: the gates and parameters of
            # ALL children are drawn first, in order — the global generator sees exactly the reference's sequence (gate 1,
            # parameters 1, gate 2, ...) — then the children apply in order.  What this buys: a later child's parameters are
            # known while the earlier ones are still being enqueued (Noise's seed: the plan of its generator stream, 0.6 - 0.9 ms
            # of host time in the reference-identical noise mode, is computed on a helper thread meanwhile).
            drawn = [transform._draw(batch) for transform in self.transforms]
            applying = [(transform, params) for transform, params in zip(self.transforms, drawn, strict=True) if params is not None]
            try:
                # (children whose preparation runs on a native THREAD go first — Noise: the plan of its generator's stream takes
                # 0.6 - 0.8 ms of wall time, and every other child's preparation, ~0.25 ms of host work, then runs beside it;
                # inside the try — a later child's `_prefetch` that raises must not leave the earlier ones' jobs running)
                for early in (True, False):
                    for transform, params in applying:
                        if bool(getattr(transform, "prefetch_is_threaded", False)) == early:
                            transform._prefetch(batch, params)
                for index, (transform, params) in enumerate(applying):
                    # the child after this one will ask for the minimum of what this one writes (default_pad_value="minimum"):
                    # a large resampling launch folds it into its stores (ops.expect_minimum_fill)
                    follower = applying[index + 1][0] if index + 1 < len(applying) else None
                    ops.expect_minimum_fill(follower is not None and getattr(follower, "asks_minimum_fill", False))
                    batch = transform._apply_drawn(batch, params)
            finally:
                ops.expect_minimum_fill(False)
                for transform, params in applying:  # (what a child prepared ahead and nobody collected — a child before it raised)
                    abandon = getattr(transform, "_abandon_prefetch", None)
                    if abandon is not None:
                        abandon(params)
            return unwrap(batch)
        for transform in self.transforms:
            # Children apply without copying: the container copied the input once (compose.py:18-35).  For a child whose
            # envelope is the stock one (no overridden `forward`, no module hooks) that is exactly `_forward(batch)`;
            # calling it directly skips nn.Module's call machinery and two `nn.Module.__setattr__` round trips per child.
            if type(transform).forward is Transform.forward and not (transform._forward_hooks or transform._forward_pre_hooks):
                batch = transform._forward(batch)
                continue
            previous = transform.copy
            transform.copy = False
            try:
                batch = transform(batch)
            finally:
                transform.copy = previous
        return unwrap(batch)

    def _may_draw_ahead(self) -> bool:
        if len(self.transforms) < 2 or os.environ.get("TIO_NO_DRAW_AHEAD", "") not in ("", "0"):
            return False
        for transform in self.transforms:
            stock = type(transform).forward is Transform.forward and type(transform)._forward is Transform._forward
            if not stock or transform._forward_hooks or transform._forward_pre_hooks or not transform.draws_ahead:
                return False
        return True

    def __len__(self) -> int:
        return len(self.transforms)

    def __getitem__(self, index: int) -> Transform:
        return self.transforms[index]

    def __repr__(self) -> str:
        inner = ", ".join(repr(t) for t in self.transforms)
        return f"Compose([{inner}])"


@contextlib.contextmanager
def _disabled_copy(transforms: Sequence[Transform]):
    """Children of a container apply without copying: the container copied the input once (compose.py:18-35)."""
    previous = [transform.copy for transform in transforms]
    for transform in transforms:
        transform.copy = False
    try:
        yield
    finally:
        for transform, value in zip(transforms, previous, strict=True):
            transform.copy = value


class OneOf(Transform):
    """Apply one of the given transforms, chosen at random (compose.py:101-181).

    ``transforms`` is a sequence (equal probabilities) or a ``dict`` transform -> relative weight.
    On a batch with ``per_instance=True`` (the default) every element draws its own gate and its
    own choice — global-RNG order per element: ``torch.rand(1)`` then ``torch.multinomial`` —
    which needs shape- and schema-preserving children so that the elements can be re-stacked.
    """

    def __init__(self, transforms: Sequence[Transform] | dict[Transform, float], **kwargs: Any) -> None:
        super().__init__(**kwargs)
        if isinstance(transforms, dict):
            self.transforms = list(transforms.keys())
            weights = list(transforms.values())
            total = sum(weights)
            self.weights = [weight / total for weight in weights]
        else:
            self.transforms = list(transforms)
            self.weights = [1.0 / len(self.transforms)] * len(self.transforms)

    def _forward(self, data: Any) -> Any:
        batch, unwrap = _wrap(data)
        home = _stage_on_engine_device(batch)
        if home is not None:  # host-resident data: ONE trip through the device for the whole container (transform.py)
            try:
                return unwrap(_return_home(self._forward(batch), home))
            finally:
                _return_home(batch, home)
        with _disabled_copy(self.transforms):
            if self.per_instance and batch.batch_size > 1:
                return unwrap(self._forward_per_element(batch))
            if torch.rand(1).item() >= self.p:
                return unwrap(batch)
            index = int(torch.multinomial(torch.tensor(self.weights), num_samples=1).item())
            return unwrap(self.transforms[index](batch))

    def _forward_per_element(self, batch: SubjectsBatch) -> SubjectsBatch:
        if self.p == 0:
            return batch
        weights = torch.tensor(self.weights)
        subjects, any_applied = [], False
        for subject in batch.unbatch():
            if torch.rand(1).item() < self.p:
                any_applied = True
                index = int(torch.multinomial(weights, num_samples=1).item())
                subject = _apply_to_element(subject, self.transforms[index])
            subjects.append(subject)
        return _rebatch_with_history(subjects, "OneOf") if any_applied else batch


class SomeOf(Transform):
    """Apply a random subset of the given transforms (compose.py:184-280).

    ``num_transforms`` is a count or a ``(min, max)`` range drawn with ``torch.randint``; the
    subset is ``torch.randperm(n)[:count]`` (or ``torch.randint`` indices with ``replace=True``).
    """

    def __init__(self, transforms: Sequence[Transform] | None = None, *, num_transforms: int | tuple[int, int] = 1,
                 replace: bool = False, **kwargs: Any) -> None:
        super().__init__(**kwargs)
        self.transforms = list(transforms) if transforms else []
        self.num_transforms = num_transforms
        self.replace = replace

    @property
    def _min_n(self) -> int:
        return self.num_transforms if isinstance(self.num_transforms, int) else self.num_transforms[0]

    @property
    def _max_n(self) -> int:
        return self.num_transforms if isinstance(self.num_transforms, int) else self.num_transforms[1]

    def _forward(self, data: Any) -> Any:
        batch, unwrap = _wrap(data)
        home = _stage_on_engine_device(batch)
        if home is not None:  # host-resident data: ONE trip through the device for the whole container (transform.py)
            try:
                return unwrap(_return_home(self._forward(batch), home))
            finally:
                _return_home(batch, home)
        with _disabled_copy(self.transforms):
            if self.per_instance and batch.batch_size > 1:
                return unwrap(self._forward_per_element(batch))
            if torch.rand(1).item() >= self.p:
                return unwrap(batch)
            return unwrap(self._apply_subset(batch))

    def _apply_subset(self, batch: SubjectsBatch) -> SubjectsBatch:
        count = int(torch.randint(self._min_n, self._max_n + 1, size=(1,)).item())
        available = len(self.transforms)
        if self.replace:
            indices = torch.randint(0, available, (count,))
        else:
            indices = torch.randperm(available)[: min(count, available)]
        for index in indices:
            batch = self.transforms[index](batch)
        return batch

    def _forward_per_element(self, batch: SubjectsBatch) -> SubjectsBatch:
        if self.p == 0:
            return batch
        subjects, any_applied = [], False
        for subject in batch.unbatch():
            if torch.rand(1).item() < self.p:
                any_applied = True
                subject = _apply_to_element(subject, self._apply_subset)
            subjects.append(subject)
        return _rebatch_with_history(subjects, "SomeOf") if any_applied else batch


def _apply_to_element(subject: Any, apply_fn: Any) -> Any:
    """One element as a one-element batch seeded with its own history (compose.py:283-303)."""
    element = SubjectsBatch.from_subjects([subject])
    element.applied_transforms = list(subject.applied_transforms)
    return apply_fn(element).unbatch()[0]


def _rebatch_with_history(subjects: list[Any], transform_name: str) -> SubjectsBatch:
    """Re-stack per-element results and freeze their distinct histories (compose.py:306-362)."""
    reference = {name: type(image) for name, image in subjects[0].images.items()}
    for subject in subjects[1:]:
        if {name: type(image) for name, image in subject.images.items()} != reference:
            raise RuntimeError(
                f"Per-instance {transform_name} produced batch elements with different image names or types, which cannot"
                f" be re-stacked. Use only schema-preserving transforms with per-instance {transform_name}, or pass"
                " per_instance=False."
            )
    try:
        batch = SubjectsBatch.from_subjects(subjects)
    except (RuntimeError, KeyError, ValueError) as error:
        raise RuntimeError(
            f"Per-instance {transform_name} produced batch elements with different shapes or schemas, which cannot be"
            f" re-stacked. Use only shape- and schema-preserving transforms with per-instance {transform_name}, or pass"
            " per_instance=False."
        ) from error
    batch.set_per_element_history([subject.applied_transforms for subject in subjects])
    return batch
