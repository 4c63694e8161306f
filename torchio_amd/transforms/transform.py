"""Transform base class — the drop-in seam (mirror of reference ``transforms/transform.py``).

The protocol is the reference's documented two-method kernel interface
(transform.py:395-427): ``make_params(batch) -> dict`` does all the sampling on
the global CPU RNG, ``apply_transform(batch, params)`` is deterministic given
``params`` and is where this package calls the HIP engine instead of
``torch.nn.functional``.  ``forward`` reproduces transform.py:212-254: deepcopy,
wrap into a ``SubjectsBatch``, batch-wide p-gate (one ``torch.rand(1)`` draw
unless per-element gating is active), sample, apply, record history, unwrap.
"""
from __future__ import annotations

import copy as _copy
import inspect
import warnings
from dataclasses import dataclass
from dataclasses import field
from typing import Any

import numpy as np
import torch
from torch import Tensor
from torch import nn

from ..data._lazy import LazyCopyScope
from ..data.batch import ImagesBatch
from ..data.batch import SubjectsBatch
from ..data.image import Image
from ..data.image import ScalarImage
from ..data.subject import Subject

_DEFAULT_IMAGE = "tio_default_image"


@dataclass
class AppliedTransform:
    """History record: transform class name, sampled params and image scope."""

    name: str
    params: dict[str, Any] = field(default_factory=dict)
    include: list[str] | None = None
    exclude: list[str] | None = None


#: class name -> class, used by history replay (transform.py:47,132-134)
_TRANSFORM_REGISTRY: dict[str, type["Transform"]] = {}


def _all_gated_out(params: dict[str, Any]) -> bool:
    keep = params.get("_keep")
    return keep is not None and not any(keep)


class Transform(nn.Module):
    """Abstract base of every transform (see module docstring)."""

    def __init__(
        self,
        *,
        p: float = 1.0,
        copy: bool = True,
        per_instance: bool = True,
        include: list[str] | None = None,
        exclude: list[str] | None = None,
    ) -> None:
        super().__init__()
        if not 0 <= p <= 1:
            raise ValueError(f"Probability must be in [0, 1], got {p}")
        self.p = p
        self.copy = copy
        self.per_instance = per_instance
        self.include = include
        self.exclude = exclude

    def __init_subclass__(cls, **kwargs: Any) -> None:
        super().__init_subclass__(**kwargs)
        _TRANSFORM_REGISTRY[cls.__name__] = cls

    # -- the envelope ---------------------------------------------------------
    def forward(self, data: Any) -> Any:
        if self.copy and isinstance(data, (SubjectsBatch, ImagesBatch)):
            # device-resident batch: share the image tensors now, clone whatever no transform replaced
            with LazyCopyScope() as scope:
                data = _copy.deepcopy(data)
            try:
                return self._forward(data)
            finally:
                scope.materialise()
        if self.copy:
            data = _copy.deepcopy(data)
        return self._forward(data)

    def _forward(self, data: Any, *, _gated: bool = False) -> Any:
        batch, unwrap = _wrap(data)
        # the p-gate comes first: a transform that does not apply moves nothing (staging in front of the gate
        # paid a PCIe round trip for a skipped transform)
        if not _gated and not self._per_instance_p_active(batch) and torch.rand(1).item() >= self.p:
            return unwrap(batch)
        home = _stage_on_engine_device(batch)
        if home is not None:  # host-resident data on the HIP engine: through the device and back (see the helper)
            try:
                return unwrap(_return_home(self._forward(batch, _gated=True), home))
            finally:
                _return_home(batch, home)
        params = self.make_params(batch)
        result = unwrap(self._apply_drawn(batch, params))
        if isinstance(result, (Image, ImagesBatch)):
            result.applied_transforms = list(batch.applied_transforms)
        return result

    # -- the two halves of the envelope, for containers that draw ahead (Compose) ------------------------------------
    def _draw(self, batch: SubjectsBatch) -> dict[str, Any] | None:
        """The global-RNG half: the p-gate, then the parameters (`None`: gated out) — exactly what `_forward` draws."""
        if not self._per_instance_p_active(batch) and torch.rand(1).item() >= self.p:
            return None
        return self.make_params(batch)

    def _apply_drawn(self, batch: SubjectsBatch, params: dict[str, Any]) -> SubjectsBatch:
        """The data half: apply, then record (transform.py:229-248)."""
        batch = self.apply_transform(batch, params)
        if not _all_gated_out(params):
            batch.applied_transforms.append(
                AppliedTransform(
                    name=type(self).__name__,
                    params=params,
                    include=None if self.include is None else list(self.include),
                    exclude=None if self.exclude is None else list(self.exclude),
                )
            )
        return batch

    @property
    def draws_ahead(self) -> bool:
        """True when (a) `make_params` reads nothing of the batch that an earlier transform could change as long as that
        transform keeps the batch's geometry — no voxel values, only sizes, affines and image names — and (b) `apply_transform`
        itself keeps that geometry.  A `Compose` whose children all say so may draw every child's gate and parameters first,
        in order (the global generator sees the reference's sequence), and apply afterwards: what a later child needs for its
        launch (the plan of Noise's generator stream) can then be prepared while the earlier children are still being applied."""
        return False

    def _prefetch(self, batch: SubjectsBatch, params: dict[str, Any]) -> None:
        """Hook of the draw-ahead road: start whatever of `apply_transform(batch, params)` only needs the parameters."""

    def make_params(self, batch: SubjectsBatch) -> dict[str, Any]:
        return {}

    def apply_transform(self, batch: SubjectsBatch, params: dict[str, Any]) -> SubjectsBatch:
        raise NotImplementedError

    @property
    def invertible(self) -> bool:
        return False

    def inverse(self, params: dict[str, Any]) -> "Transform":
        raise NotImplementedError(f"{type(self).__name__} is not invertible")

    # -- per-instance machinery (transform.py:256-393) -----------------------
    @property
    def supports_per_instance_params(self) -> bool:
        return False

    @property
    def supports_per_instance_p(self) -> bool:
        return False

    def _per_instance_active(self, batch: SubjectsBatch) -> bool:
        return self.per_instance and self.supports_per_instance_params and batch.batch_size > 1

    def _per_instance_p_active(self, batch: SubjectsBatch) -> bool:
        return self.per_instance and self.supports_per_instance_p and batch.batch_size > 1 and 0.0 < self.p < 1.0

    def _resolve_n(self, batch: SubjectsBatch) -> int | None:
        return batch.batch_size if self._per_instance_active(batch) else None

    def _keep_mask(self, batch: SubjectsBatch, n: int | None) -> Tensor | None:
        if n is None or not self._per_instance_p_active(batch):
            return None
        return torch.rand(n) < self.p

    @staticmethod
    def _mask_identity(value, keep: Tensor | None, *, identity: float):
        if keep is None or not isinstance(value, Tensor):
            return value
        return torch.where(keep, value, torch.full_like(value, identity))

    @staticmethod
    def _serialize_param(value):
        return value.tolist() if isinstance(value, Tensor) else value

    @staticmethod
    def _is_per_instance_params(params: dict[str, Any]) -> bool:
        return "_batched_keys" in params

    def _tag_batched(self, params, batch, n, keep, batched_keys) -> None:
        if n is None:
            return
        params["_batch_size"] = batch.batch_size
        params["_batched_keys"] = list(batched_keys)
        if keep is not None:
            params["_keep"] = keep.tolist()

    # -- scope ---------------------------------------------------------------
    def _get_images(self, batch: SubjectsBatch) -> dict[str, ImagesBatch]:
        images = batch.images
        if self.include is not None:
            images = {k: v for k, v in images.items() if k in self.include}
        if self.exclude is not None:
            images = {k: v for k, v in images.items() if k not in self.exclude}
        return images

    # -- cosmetics -----------------------------------------------------------
    def _warn_if_noop(self, *, is_noop: bool, hint: str) -> None:
        if is_noop:
            warnings.warn(
                f"{type(self).__name__} is a no-op with the given parameters and will not change the data."
                f" Pass arguments to apply an effect (e.g. {hint}), or a range like (a, b) for random augmentation.",
                stacklevel=3,
            )

    def __repr__(self) -> str:
        from .parameter_range import _ParameterRange  # noqa: PLC0415

        parts = []
        for name, default in _init_defaults(type(self)).items():
            value = getattr(self, name, default)
            if isinstance(value, _ParameterRange):
                if value._original == default:
                    continue
            else:
                try:
                    if bool(value == default):
                        continue
                except (RuntimeError, ValueError):
                    pass
            parts.append(f"{name}={value!r}")
        return f"{type(self).__name__}({', '.join(parts)})"

    def __add__(self, other):
        if not isinstance(other, Transform):
            return NotImplemented
        from .compose import Compose  # noqa: PLC0415

        left = self.transforms if isinstance(self, Compose) else [self]
        right = other.transforms if isinstance(other, Compose) else [other]
        return Compose([*left, *right])


class SpatialTransform(Transform):
    """Transforms that modify geometry: apply to every image class."""


class IntensityTransform(Transform):
    """Transforms that modify intensities: ``ScalarImage`` batches only (transform.py:684-693)."""

    def _get_images(self, batch: SubjectsBatch) -> dict[str, ImagesBatch]:
        scalars = {k: v for k, v in batch.images.items() if v._image_class is ScalarImage}
        if self.include is not None:
            scalars = {k: v for k, v in scalars.items() if k in self.include}
        if self.exclude is not None:
            scalars = {k: v for k, v in scalars.items() if k not in self.exclude}
        return scalars


def _init_defaults(cls: type) -> dict[str, Any]:
    """``{name: default}`` of every named ``__init__`` parameter along the MRO."""
    found: dict[str, Any] = {}
    for klass in cls.__mro__:
        if klass in (object, nn.Module):
            break
        init = klass.__dict__.get("__init__")
        if init is None:
            continue
        for name, parameter in inspect.signature(init).parameters.items():
            if name == "self" or parameter.kind in (parameter.VAR_POSITIONAL, parameter.VAR_KEYWORD):
                continue
            found.setdefault(name, parameter.default)
    return found


# -- host-resident data ----------------------------------------------------------------------------------------------
def _stage_on_engine_device(batch: SubjectsBatch):
    """The reference transforms host tensors on the host (transform.py:212-254); this package computes on the GPU only.
    Host-resident subjects — what a reader hands over — are therefore STAGED: moved to the engine's device for the
    transform and back afterwards, so that ``tio.Affine()(cpu_subject)`` keeps working and returns host tensors.  Still
    no CPU compute path: without a GPU the engine raises as before.  (PCIe-bound: DESIGN.md section 5 has the rate.)

    Returns the device to go back to, or ``None`` when nothing has to move (data already on the engine's device, an
    engine that computes where the data lives — the CPU oracle of the tests —, or no GPU to stage on).
    """
    from .. import ops  # noqa: PLC0415

    images = batch.images
    if not images:
        return None
    on_host = [name for name, image in images.items() if getattr(image, "_data", None) is not None and image._data.device.type == "cpu"]
    if not on_host:
        return None
    engine = ops._ENGINE
    if engine is not None and engine.device_type != "cuda":
        return None
    if not torch.cuda.is_available():
        return None
    if not torch.cuda.is_initialized() and torch.utils.data.get_worker_info() is not None:
        # a DataLoader worker PROCESS: initialising the device runtime in every forked worker is never what a pipeline
        # wants — say so instead of doing it silently (transform on the main process, or hand over device tensors)
        raise ops.EngineError(
            "host-resident subject inside a DataLoader worker process: the HIP engine computes on the GPU; transform in the "
            "main process (Queue's worker THREADS are fine) or move the subject to the device first"
        )
    # every image is staged on its own and remembers its own home (a subject may mix host and device images)
    homes = {name: images[name]._data.device for name in on_host}
    device = torch.device("cuda", torch.cuda.current_device())
    for name in on_host:
        images[name].to(device)
    return homes


def _return_home(result, homes):
    if isinstance(result, SubjectsBatch):
        for name, device in homes.items():
            if name in result.images:
                result.images[name].to(device)
    elif isinstance(result, ImagesBatch):
        result.to(next(iter(homes.values())))
    return result


# -- input wrapping (transform.py:488-665): output type always matches input type ---
def _single(image: Image) -> SubjectsBatch:
    return SubjectsBatch.from_subjects([Subject(**{_DEFAULT_IMAGE: image})])


def _wrap(data: Any):
    """Wrap *data* into a ``SubjectsBatch`` and return ``(batch, unwrap)``."""
    if isinstance(data, SubjectsBatch):
        return data, lambda batch: batch
    if isinstance(data, ImagesBatch):
        return SubjectsBatch({_DEFAULT_IMAGE: data}), lambda batch: batch.images[_DEFAULT_IMAGE]
    if isinstance(data, Subject):
        return SubjectsBatch.from_subjects([data]), lambda batch: batch.unbatch()[0]
    if isinstance(data, dict):
        entries = {k: (ScalarImage(v) if isinstance(v, Tensor) else v) for k, v in data.items()}
        keys = [str(k) for k in data]

        def unwrap_dict(batch):
            subject = batch.unbatch()[0]
            out = {}
            for key in keys:
                entry = subject[key] if key in subject else None
                out[key] = entry.data if isinstance(entry, Image) else entry
            return out

        return SubjectsBatch.from_subjects([Subject(**entries)]), unwrap_dict
    if isinstance(data, Image):
        return _single(data), lambda batch: batch.unbatch()[0][_DEFAULT_IMAGE]
    if isinstance(data, Tensor):
        return _single(ScalarImage(data)), lambda batch: batch.unbatch()[0][_DEFAULT_IMAGE].data
    if isinstance(data, np.ndarray):
        tensor = torch.as_tensor(data.copy(), dtype=torch.float32)
        return _single(ScalarImage(tensor)), lambda batch: batch.unbatch()[0][_DEFAULT_IMAGE].data.cpu().numpy()
    raise TypeError(
        "Expected Subject, Image, Tensor, ndarray, dict, ImagesBatch, or SubjectsBatch,"
        f" got {type(data).__name__}"
    )
