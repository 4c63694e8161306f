"""Undoing recorded transforms.

Every transform appends an ``AppliedTransform(name, params, include, exclude)`` to its output's history;
replaying that history backwards through each transform's ``inverse(params)`` gives the undo pipeline.
``get_inverse_transform`` / ``apply_inverse_transform`` have the reference's signature and warnings
(``transforms/inverse.py``); the history walk is written as a filter chain over (trace, class) pairs.
"""
from __future__ import annotations

import warnings
from collections.abc import Iterator

from .compose import Compose
from .transform import _TRANSFORM_REGISTRY
from .transform import AppliedTransform
from .transform import IntensityTransform


def _undo_steps(history: list[AppliedTransform], warn: bool, ignore_intensity: bool) -> Iterator:
    def note(message: str) -> None:
        if warn:
            warnings.warn(message, stacklevel=4)

    for trace in history[::-1]:
        kind = _TRANSFORM_REGISTRY.get(trace.name)
        if kind is None:
            note(f"Unknown transform {trace.name!r} in history, skipping")
        elif ignore_intensity and issubclass(kind, IntensityTransform):
            pass
        else:
            blank = kind.__new__(kind)  # `invertible` and `inverse` are functions of the recorded params only
            if blank.invertible:
                undo = blank.inverse(trace.params)
                undo.include, undo.exclude = trace.include, trace.exclude
                yield undo
            else:
                note(f"{trace.name} is not invertible, skipping")


def get_inverse_transform(history: list[AppliedTransform], *, warn: bool = True, ignore_intensity: bool = False) -> Compose:
    """The ``Compose`` that undoes *history*, newest step first; steps without an inverse are left out (with a warning)."""
    return Compose(list(_undo_steps(history, warn, ignore_intensity)))


def apply_inverse_transform(data, *, warn: bool = True, ignore_intensity: bool = False):
    """Undo everything recorded on *data* and clear its history; objects without a history pass through.

    A batch whose elements carry their own histories (per-instance ``OneOf`` / ``SomeOf``) inverts element by
    element through its own method.
    """
    history = getattr(data, "applied_transforms", None)
    if history is None:
        return data
    if getattr(data, "_per_element_history", None) is not None:
        return data.apply_inverse_transform(warn=warn, ignore_intensity=ignore_intensity)
    restored = get_inverse_transform(history, warn=warn, ignore_intensity=ignore_intensity)(data)
    if hasattr(restored, "applied_transforms"):
        restored.applied_transforms = []
    return restored
