"""The reference-side binding, executable: patch the REAL ``torchio`` package so that its own classes run the
engine (INTEGRATION.md shows the same patch as a diff a maintainer would apply).

    import torchio
    from torchio_amd import reference_binding
    reference_binding.bind(torchio)          # tio.Affine()(cuda_subject) now launches tio_resample3d

Five seams are replaced:

    S1  torchio.transforms.spatial.spatial._apply_spatial_to_batch   -> tio_resample3d (+ tio_channel_min, tio_unique_labels)
    S2  torchio.transforms.intensity.blur._gaussian_smooth           -> tio_separable_conv3d
    S3  BiasField / _BiasFieldInverse .apply_transform                -> tio_bias_field_apply
    S4  Noise.apply_transform                                         -> tio_add_noise
    S5  Gamma / _GammaInverse .apply_transform                        -> tio_gamma_pow

Every replacement keeps the reference's original as the FALLBACK for what the engine does not take — CPU tensors
(the engine only reads device memory), autograd through an op that has no backward here (nearest-neighbour
resampling of a tensor that requires grad ...) — so
``tio.Affine()(cpu_subject)``, the reference's normal use, keeps working exactly as before.  Inputs that require
grad run on the engine like any other: trilinear resampling, bias field, blur, noise, gamma and flip have backward
passes (``ops.Engine``).  The replacements are this package's own seam functions called with the REFERENCE's
objects: containers, parameter dictionaries and helper methods are name-compatible by construction.
"""
from __future__ import annotations

import functools
import importlib
from typing import Any

import torch

from . import ops

_ORIGINALS: dict[tuple[Any, str], Any] = {}


def _with_fallback(ours, original, tensors_of):
    """``ours`` for device tensors, ``original`` (the reference's own code) for everything the engine does not take.

    The decision is made UP FRONT from the tensors of the call: a tensor that does not live where the engine computes
    (a host tensor, for the HIP engine) goes to the reference without touching the engine.  Of the engine path's errors
    only the two that MEAN "not taken" fall back — ``EngineError`` and ``NotImplementedError`` — and only while nothing has
    been written (the images of the batch are still the same tensor objects; after the first image was replaced falling
    back would apply the transform to that image twice: re-raised).  A ``TypeError`` / ``ValueError`` is a bug on the engine
    path (bad shapes, marshalling) or a genuine argument error and surfaces as such (swallowing them degraded
    silently to the slow host path).  The global RNG state is put back before the original runs, so that a fallback draws
    what the reference alone would have drawn.
    """
    @functools.wraps(original)
    def seam(*args, **kwargs):
        try:
            before = list(tensors_of(*args, **kwargs))
        except Exception:  # noqa: BLE001 - an unexpected call shape is the reference's business
            return original(*args, **kwargs)
        wanted = ops._ENGINE.device_type if ops._ENGINE is not None else "cuda"
        if any(getattr(t, "device", None) is None or t.device.type != wanted for t in before):
            return original(*args, **kwargs)  # the engine only reads device memory
        rng_state = torch.get_rng_state()
        try:
            return ours(*args, **kwargs)
        except (ops.EngineError, NotImplementedError):
            after = list(tensors_of(*args, **kwargs))
            if len(after) != len(before) or any(a is not b for a, b in zip(after, before)):
                raise
            torch.set_rng_state(rng_state)  # the engine path may have consumed draws
            return original(*args, **kwargs)

    seam.__tio_amd_original__ = original
    return seam


def _patch(owner, name: str, replacement) -> None:
    key = (owner, name)
    if key not in _ORIGINALS:
        _ORIGINALS[key] = getattr(owner, name)
    setattr(owner, name, replacement)


def _batch_tensors(transform, batch, params=None):
    return [image.data for image in transform._get_images(batch).values()]


def bind(torchio_module=None) -> None:
    """Install the engine under the reference's transform classes (idempotent)."""
    if torchio_module is None:
        torchio_module = importlib.import_module("torchio")
    base = torchio_module.__name__
    ref_spatial = importlib.import_module(f"{base}.transforms.spatial.spatial")
    ref_blur = importlib.import_module(f"{base}.transforms.intensity.blur")
    ref_bias = importlib.import_module(f"{base}.transforms.intensity.bias_field")
    ref_noise = importlib.import_module(f"{base}.transforms.intensity.noise")
    ref_gamma = importlib.import_module(f"{base}.transforms.intensity.gamma")
    from .transforms import bias_field as our_bias  # noqa: PLC0415
    from .transforms import blur as our_blur  # noqa: PLC0415
    from .transforms import gamma as our_gamma  # noqa: PLC0415
    from .transforms import noise as our_noise  # noqa: PLC0415
    from .transforms import spatial as our_spatial  # noqa: PLC0415

    # S1: one fused launch instead of the sampling grid + two grid_sample calls per image
    original = _ORIGINALS.get((ref_spatial, "_apply_spatial_to_batch"), ref_spatial._apply_spatial_to_batch)
    _patch(ref_spatial, "_apply_spatial_to_batch", _with_fallback(
        our_spatial._apply_spatial_to_batch, original,
        lambda **kw: [kw["batch"].images[name].data for name in kw["image_names"]],
    ))
    # S2: separable Gaussian
    original = _ORIGINALS.get((ref_blur, "_gaussian_smooth"), ref_blur._gaussian_smooth)
    _patch(ref_blur, "_gaussian_smooth", _with_fallback(our_blur._gaussian_smooth, original, lambda data, sigmas: [data]))
    # S3 - S5: the transform bodies (the reference instance is `self`: same attributes and helper methods)
    for ref_module, our_module, names in (
        (ref_bias, our_bias, ("BiasField", "_BiasFieldInverse")),
        (ref_noise, our_noise, ("Noise",)),
        (ref_gamma, our_gamma, ("Gamma", "_GammaInverse")),
    ):
        for name in names:
            ref_cls, our_cls = getattr(ref_module, name), getattr(our_module, name)
            original = _ORIGINALS.get((ref_cls, "apply_transform"), ref_cls.apply_transform)
            _patch(ref_cls, "apply_transform", _with_fallback(our_cls.apply_transform, original, _batch_tensors))


def unbind() -> None:
    """Put the reference's own functions back."""
    for (owner, name), original in _ORIGINALS.items():
        setattr(owner, name, original)
    _ORIGINALS.clear()
