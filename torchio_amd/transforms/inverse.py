"""History replay (mirror of reference ``transforms/inverse.py:15-61``)."""
from __future__ import annotations

import warnings

from .compose import Compose
from .transform import _TRANSFORM_REGISTRY
from .transform import AppliedTransform
from .transform import IntensityTransform


def get_inverse_transform(
    history: list[AppliedTransform], *, warn: bool = True, ignore_intensity: bool = False
) -> Compose:
    """A ``Compose`` undoing *history* (newest first); non-invertible steps are skipped."""
    steps = []
    for trace in reversed(history):
        cls = _TRANSFORM_REGISTRY.get(trace.name)
        if cls is None:
            if warn:
                warnings.warn(f"Unknown transform {trace.name!r} in history, skipping", stacklevel=2)
            continue
        if ignore_intensity and issubclass(cls, IntensityTransform):
            continue
        probe = object.__new__(cls)  # `invertible` / `inverse` never touch instance state
        if not probe.invertible:
            if warn:
                warnings.warn(f"{trace.name} is not invertible, skipping", stacklevel=2)
            continue
        step = probe.inverse(trace.params)
        step.include = trace.include
        step.exclude = trace.exclude
        steps.append(step)
    return Compose(steps)


def apply_inverse_transform(data, *, warn: bool = True, ignore_intensity: bool = False):
    """Undo every recorded transform of *data* (anything with ``applied_transforms``); inverse.py:64-100.

    Batches that carry per-element histories (per-instance ``OneOf`` / ``SomeOf``) invert each
    element with its own history through their own method.
    """
    if not hasattr(data, "applied_transforms"):
        return data
    if getattr(data, "_per_element_history", None) is not None:
        return data.apply_inverse_transform(warn=warn, ignore_intensity=ignore_intensity)
    inverse = get_inverse_transform(data.applied_transforms, warn=warn, ignore_intensity=ignore_intensity)
    result = inverse(data)
    if hasattr(result, "applied_transforms"):
        result.applied_transforms = []
    return result
