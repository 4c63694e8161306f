"""Unified spatial transforms on the HIP engine.

Host-side mirror of reference ``src/torchio/transforms/spatial/spatial.py``:
``Spatial`` / ``Resample`` / ``Affine`` / ``ElasticDeformation`` with the same
constructor arguments, the same ``make_params`` draw order on the global CPU RNG
(spatial.py:382-558) and the same JSON-serialisable params dict
(spatial.py:455-513), so seeds, history, unbatching and inversion behave like the
reference.  What changes is everything under ``apply_transform``: instead of
materialising an ``(I, J, K, 3)`` grid and calling ``F.grid_sample`` twice
(spatial.py:1504-1731), ``_apply_spatial_to_batch`` hands the engine twelve
floats per element (and the control points) and every selected image is
resampled by ONE fused kernel launch (``tio_resample3d``).

The ``"label"`` partial-volume mode (spatial.py:1275-1389) is fused into the same
call for the usual case (single-channel label map, linear one-hot interpolation, no
antialias): the ``(B, L, I, J, K)`` one-hot tensor is never built.

Not implemented on the engine (raise, never fall back): interpolation orders
>= 2 (reference uses torch-interpol, spatial.py:1734-1761).
"""
from __future__ import annotations

import math
import warnings
from dataclasses import dataclass
from numbers import Number
from pathlib import Path
from typing import Any

import numpy as np
import torch
from torch import Tensor
from torch.distributions import Distribution

from .. import ops
from ..data.affine import AffineMatrix
from ..data.batch import ImagesBatch
from ..data.batch import SubjectsBatch
from ..data.image import Image
from ..data.image import LabelMap
from .blur import _stacked_gaussian_taps
from .parameter_range import Choice
from .parameter_range import ScalarDrawPlan
from .parameter_range import _ParameterRange
from ._lazy_params import LazyParams
from .transform import SpatialTransform

LABEL_INTERPOLATION = "label"
_ORDERS = {"nearest": 0, "linear": 1, "quadratic": 2, "cubic": 3, "fourth": 4, "fifth": 5, "sixth": 6, "seventh": 7}
_SUPPORTED_INTERPOLATIONS = (*_ORDERS, LABEL_INTERPOLATION)
_SUPPORTED_PAD_VALUES = ("minimum", "mean", "otsu")
_SPLINE_ORDER = 3


@dataclass
class _PerSampleGrids:
    """Per-element geometry for per-instance augmentation (spatial.py:49-65)."""

    affine_matrices: list[np.ndarray | None]
    control_points: list[Tensor | None]
    max_displacements: list[tuple[float, float, float] | None]


def _is_label_batch(img_batch) -> bool:
    """Label map or intensity image?  By class NAME along the MRO, so that the reference's own containers
    (``torchio.LabelMap`` under ``reference_binding``) are recognised like this package's."""
    kind = img_batch._image_class
    return isinstance(kind, type) and (issubclass(kind, LabelMap) or any(base.__name__ == "LabelMap" for base in kind.__mro__))


def _range_key(value):
    """Hashable identity of a parameter range by value (``_ParameterRange._axes``), for the draw-plan cache."""
    axes = getattr(value, "_axes", None)
    if axes is None:
        return ("raw", repr(value))
    try:
        return ("axes", tuple(tuple(a) if isinstance(a, (list, tuple)) else a for a in axes))
    except TypeError:
        return ("repr", repr(axes))


class Spatial(SpatialTransform):
    r"""Resampling, affine motion and elastic deformation in a single resampling pass.

    Arguments and semantics follow reference ``Spatial`` (spatial.py:158-369).
    """

    def __init__(
        self,
        *,
        target=None,
        scales=1.0,
        degrees=0.0,
        translation=0.0,
        isotropic: bool = False,
        center: str = "image",
        control_points=None,
        num_control_points: int | tuple[int, int, int] = 7,
        max_displacement=0.0,
        locked_borders: int = 2,
        affine_first: bool = True,
        image_interpolation: str | int = "linear",
        label_interpolation: str | int = "nearest",
        one_hot_label_interpolation: str | int = "linear",
        antialias: bool = False,
        default_pad_value: str | float = "minimum",
        default_pad_label: int | float = 0,
        **kwargs: Any,
    ) -> None:
        super().__init__(**kwargs)
        self.target = target
        _validate_isotropic(scales, isotropic)
        self.scales = _positive_range(scales)
        self.degrees = _parameter_range(degrees)
        self.translation = _parameter_range(translation)
        self.isotropic = isotropic
        self.center = _parse_center(center)
        self.control_points = None if control_points is None else _parse_control_points(control_points)
        self.num_control_points = _parse_num_control_points(num_control_points)
        self.max_displacement = _nonnegative_range(max_displacement)
        self.locked_borders = _parse_locked_borders(locked_borders)
        if self.locked_borders == 2 and 4 in self.num_control_points:
            raise ValueError("locked_borders=2 with 4 control points along any axis yields an identity elastic field")
        self.affine_first = affine_first
        parsed = _parse_interpolation(image_interpolation)
        if parsed == LABEL_INTERPOLATION:
            raise ValueError(
                f'image_interpolation cannot be "{LABEL_INTERPOLATION}"; that mode is only valid for label_interpolation'
            )
        self.image_interpolation = parsed
        self.label_interpolation = _parse_interpolation(label_interpolation)
        one_hot = _parse_interpolation(one_hot_label_interpolation)
        if one_hot == LABEL_INTERPOLATION:
            raise ValueError(f'one_hot_label_interpolation cannot be "{LABEL_INTERPOLATION}"')
        self.one_hot_label_interpolation = one_hot
        self.antialias = antialias
        self.default_pad_value = _parse_default_pad_value(default_pad_value)
        if not isinstance(default_pad_label, Number):
            raise TypeError(f"default_pad_label must be numeric, got {type(default_pad_label)}")
        self.default_pad_label = float(default_pad_label)

    @property
    def supports_per_instance_params(self) -> bool:
        return True

    @property
    def supports_per_instance_p(self) -> bool:
        # per-element gating needs a shape-preserving transform (spatial.py:375-380)
        return self.target is None

    @property
    def asks_minimum_fill(self) -> bool:
        """This transform's fill value is the minimum of the data it is handed (a producer that can fold it in does: `Compose`)."""
        return self.default_pad_value == "minimum"

    @property
    def draws_ahead(self) -> bool:
        # without a target the output grid is the input grid: sizes, affines and image names survive, and `make_params` reads
        # nothing else of the batch (transform.py: Transform.draws_ahead)
        return self.target is None and type(self).make_params is Spatial.make_params and type(self).apply_transform is Spatial.apply_transform

    # -- sampling: global-RNG order is scales, degrees, translation, max_displacement,
    #    control points (spatial.py:382-434) ----------------------------------
    def _scalar_plan(self):
        """All scalar draws of one element as one small ``uniform_`` call (same values, same RNG state)."""
        draw_displacement = self.control_points is None
        cached = self.__dict__.get("_scalar_plan_cache")
        # fast check: the very `_axes` tuples the plan was built from are still in place (the cache holds them, so a new
        # tuple can never reuse one of their ids); otherwise the ranges' VALUES decide
        axes = (self.scales._axes, self.degrees._axes, self.translation._axes, self.max_displacement._axes)
        if cached is not None and cached[2][0] is axes[0] and cached[2][1] is axes[1] and cached[2][2] is axes[2] and cached[2][3] is axes[3] \
                and cached[3] == (self.isotropic, draw_displacement):
            return cached[1]
        key = (_range_key(self.scales), _range_key(self.degrees), _range_key(self.translation), _range_key(self.max_displacement),
               self.isotropic, draw_displacement)
        if cached is None or cached[0] != key:
            ranges = [self.scales, self.degrees, self.translation] + ([self.max_displacement] if draw_displacement else [])
            counts = [1 if self.isotropic else 3, 3, 3] + ([3] if draw_displacement else [])
            cached = (key, ScalarDrawPlan.build(ranges, counts), axes, (self.isotropic, draw_displacement))
        else:
            cached = (cached[0], cached[1], axes, (self.isotropic, draw_displacement))
        self.__dict__["_scalar_plan_cache"] = cached
        return cached[1]

    _NO_PLAN = object()

    def _sample_one(self, shape, affine, *, build: bool = True, plan=_NO_PLAN):
        """Parameters of one element; with ``build=False`` the affine is returned as its three tuples.

        ``plan``: the scalar draw plan when the caller already holds it (one look-up per batch instead of one per
        element: the cache key is built from the four ranges' values)."""
        if plan is Spatial._NO_PLAN:
            plan = self._scalar_plan()
        displacement = None
        if plan is not None:
            values = plan.sample()
            n_scale = 1 if self.isotropic else 3
            scales = (values[0],) * 3 if self.isotropic else tuple(values[:3])
            degrees = tuple(values[n_scale : n_scale + 3])
            translation = tuple(values[n_scale + 3 : n_scale + 6])
            if self.control_points is None:
                displacement = tuple(values[n_scale + 6 : n_scale + 9])
        else:
            if self.isotropic:
                value = self.scales.sample_1d()
                scales = (value, value, value)
            else:
                scales = self.scales.sample()
            degrees = self.degrees.sample()
            translation = self.translation.sample()
        has_affine = not (
            _all_close(scales, 1.0) and _all_close(degrees, 0.0) and _all_close(translation, 0.0)
        )
        if self.control_points is not None:
            field = self.control_points.clone()
            displacement = _max_abs_displacement(field)
        else:
            if displacement is None:
                displacement = self.max_displacement.sample()
            if all(value == 0.0 for value in displacement):
                field, displacement = None, None
            else:
                field = _sample_control_points(self.num_control_points, displacement, self.locked_borders)
        forward = None
        if has_affine:
            forward = (scales, degrees, translation)
            if build:
                forward = _build_forward_affine(
                    scales=scales, degrees=degrees, translation=translation, center=self.center, shape=shape, affine=affine
                )
        return forward, field, displacement, (has_affine or field is not None)

    def _sample_block(self, count: int):
        """The draws of ``count`` elements as arrays, from ONE ``torch.rand`` call — or ``None`` when the per-element loop must run.

        The CPU generator hands out the same uniforms whether they are asked for element by element (a handful of
        scalars, then ``prod(grid) * 3`` control values, per element) or as one block, so the values AND the generator
        state afterwards are those of the loop (``tests/test_host_logic.py`` holds the two against each other); what
        the block saves is the per-element dispatcher round trips (8 x (uniform_ + rand + 3 tensor ops) per transform)
        and, with the arrays kept as arrays, the per-element Python bookkeeping behind them.
        ``None``: a draw decides whether later draws happen (a displacement that comes out as exactly zero draws no
        field), the scalar draws need the general path, or the control points are user-given.

        Returns ``(scales, degrees, translation, has_affine, fields, displacements)``: ``(count, 3)`` float64 arrays, a
        bool vector, the ``(count, ni, nj, nk, 3)`` float32 field block (or ``None``: no elastic part) and the ``(count, 3)``
        displacement array (or ``None``).
        """
        plan = self._scalar_plan()
        if plan is None or count < 2 or self.control_points is not None:
            return None
        n_scale = 1 if self.isotropic else 3
        displacement_entries = plan.entries[n_scale + 6 : n_scale + 9]
        never_field = all(constant == 0.0 for constant, _, _ in displacement_entries)
        grid = tuple(self.num_control_points)
        n_field = 0 if never_field else grid[0] * grid[1] * grid[2] * 3
        width = plan.n_random + n_field
        if width == 0:
            return None
        state = torch.get_rng_state() if not never_field else None
        block = torch.rand(count, width, dtype=torch.float32)
        values = plan.map_block_array(block[:, : plan.n_random].numpy())
        displacements = values[:, n_scale + 6 : n_scale + 9]
        if not never_field and bool((displacements == 0.0).all(axis=1).any()):
            torch.set_rng_state(state)  # measure zero: that element draws no field in the reference, the stream shifts
            return None
        fields = None
        if not never_field:
            fields = block[:, plan.n_random :].reshape(count, *grid, 3)
            fields -= 0.5
            fields *= torch.from_numpy((2.0 * displacements).astype(np.float32)).view(count, 1, 1, 1, 3)
            if self.locked_borders > 0:
                fields = torch.where(_interior_mask(grid, self.locked_borders), fields, torch.zeros((), dtype=torch.float32))
        scales = np.repeat(values[:, :1], 3, axis=1) if self.isotropic else values[:, :3]
        degrees = values[:, n_scale : n_scale + 3]
        translation = values[:, n_scale + 3 : n_scale + 6]
        # np.allclose(v, target) with its default tolerances, per element (the reference's no-op test, spatial.py:2237-2243)
        identity = (
            (np.abs(scales - 1.0) <= 1e-8 + 1e-5).all(axis=1) & (np.abs(degrees) <= 1e-8).all(axis=1) & (np.abs(translation) <= 1e-8).all(axis=1)
        )
        return scales, degrees, translation, ~identity, fields, (None if never_field else displacements)

    def _sample_many(self, shape, affine, count: int):
        """``[_sample_one(shape, affine, build=False) for _ in range(count)]`` with ONE draw for the whole batch (``_sample_block``)."""
        drawn = self._sample_block(count)
        if drawn is None:
            plan = self._scalar_plan()
            return [self._sample_one(shape, affine, build=False, plan=plan) for _ in range(count)]
        scales, degrees, translation, has_affine, fields, displacements = drawn
        rows_s, rows_d, rows_t = scales.tolist(), degrees.tolist(), translation.tolist()
        rows_m = None if displacements is None else displacements.tolist()
        out = []
        for index in range(count):
            forward = (tuple(rows_s[index]), tuple(rows_d[index]), tuple(rows_t[index])) if has_affine[index] else None
            field = None if fields is None else fields[index]
            displacement = None if field is None else tuple(rows_m[index])
            out.append((forward, field, displacement, bool(has_affine[index]) or field is not None))
        return out

    def make_params(self, batch: SubjectsBatch) -> dict[str, Any]:
        images = self._get_images(batch)
        if not images:
            return {"selected_images": []}
        first = next(iter(images.values()))
        shape, affine = _spatial_shape(first), first.affines[0]
        params: dict[str, Any] = LazyParams()
        params.update({
            "selected_images": list(images),
            "original": _serialize_space((shape, affine)),
            "affine_first": self.affine_first,
            "image_interpolation": self.image_interpolation,
            "label_interpolation": self.label_interpolation,
            "one_hot_label_interpolation": self.one_hot_label_interpolation,
            "antialias": self.antialias,
            "default_pad_value": self.default_pad_value,
            "default_pad_label": self.default_pad_label,
        })
        n = self._resolve_n(batch)
        if n is None:
            forward, field, displacement, has_geometry = self._sample_one(shape, affine)
            if has_geometry:
                _check_shared_space(images, shape, affine)
            # the (possibly random) target is resolved AFTER the geometry (spatial.py:474-481)
            params["target"] = _serialize_space(_resolve_target_space(self.target, batch, shape, affine))
            params["affine_matrix"] = None if forward is None else forward.tolist()
            params.set_lazy("control_points", None if field is None else field.detach().to(device="cpu", dtype=torch.float32))
            params["max_displacement"] = list(displacement) if displacement else None
            return params

        keep = self._keep_mask(batch, n)
        drawn = self._sample_block(n) if keep is None else None
        if drawn is not None:  # every element draws: the whole batch as arrays, lists only if somebody reads the history
            scales, degrees, translation, has_affine, field_block, displacement_block = drawn
            if bool(has_affine.any()) or field_block is not None:
                _check_shared_space(images, shape, affine)
            params["target"] = _serialize_space(_resolve_target_space(self.target, batch, shape, affine))
            everyone = bool(has_affine.all())
            built = _build_forward_affines_arrays(
                scales if everyone else scales[has_affine], degrees if everyone else degrees[has_affine],
                translation if everyone else translation[has_affine], center=self.center, shape=shape, affine=affine,
            )
            rows = iter(built)
            matrices = _MatrixList([next(rows) if flag else None for flag in has_affine.tolist()])
            matrices.stacked = built if everyone else None
            params.set_lazy("affine_matrix", matrices)
            params.set_lazy("control_points", [None] * n if field_block is None else _StackedFields(field_block))
            params["max_displacement"] = [None] * n if displacement_block is None else displacement_block.tolist()
            self._tag_batched(params, batch, n, keep, ["affine_matrix", "control_points", "max_displacement"])
            return params
        matrices, fields, displacements, any_geometry = [], [], [], False
        kept = [index for index in range(n) if keep is None or bool(keep[index])]
        drawn = iter(self._sample_many(shape, affine, len(kept)))  # gated-out elements draw nothing
        for index in range(n):
            if keep is not None and not bool(keep[index]):
                matrices.append(None), fields.append(None), displacements.append(None)
                continue
            forward, field, displacement, has_geometry = next(drawn)
            any_geometry = any_geometry or has_geometry
            matrices.append(forward)
            fields.append(None if field is None else field.detach().to(device="cpu", dtype=torch.float32))
            displacements.append(list(displacement) if displacement else None)
        if any_geometry:
            _check_shared_space(images, shape, affine)
        params["target"] = _serialize_space(_resolve_target_space(self.target, batch, shape, affine))
        # all elements' world affines in one vectorised pass; nested lists only if somebody reads the history
        built = _build_forward_affines([m for m in matrices if m is not None], center=self.center, shape=shape, affine=affine)
        rows = iter(built)
        params.set_lazy("affine_matrix", _MatrixList([None if m is None else next(rows) for m in matrices]))
        params.set_lazy("control_points", fields)  # nested lists only if somebody reads the history
        params["max_displacement"] = displacements
        self._tag_batched(params, batch, n, keep, ["affine_matrix", "control_points", "max_displacement"])
        return params

    def _prefetch(self, batch: SubjectsBatch, params: dict[str, Any]) -> None:
        """Draw-ahead road (Compose): mapping, control points, flags and the brick plan of the fused launch depend on the drawn
        parameters and the batch's geometry only — they go to the device on the side stream (`ops.ahead_stream`) now, while
        the children before this one are still being enqueued (or their kernels still run); `apply_transform` picks them up."""
        selected = params.get("selected_images", [])
        if not selected or not isinstance(params, LazyParams):
            return
        first = batch.images[selected[0]]
        tensor = first._data if hasattr(first, "_data") else first.data
        side = ops.ahead_stream(tensor.device) if tensor.is_cuda else None
        if side is None:
            return
        try:
            target_space = _deserialize_space(params["target"])
            matrix, field, displacement, per_sample = _resolve_spatial_params(params)
            if target_space is None and matrix is None and field is None and displacement is None and per_sample is None:
                return
            ahead = ops.Ahead()
            with torch.cuda.stream(side):
                geometry = _prepare_launch_geometry(
                    first, tensor.device, target_space=target_space, affine_matrix=matrix, control_points=field,
                    max_displacement=displacement, per_sample=per_sample,
                )
                images = [batch.images[name] for name in selected]
                if params["image_interpolation"] == "linear" and all(
                    (img._data if hasattr(img, "_data") else img.data).dtype == torch.float32 or _is_label_batch(img) for img in images
                ):  # (the launch of float32 trilinear images is the one that starts from a plan; label maps have their own kernel)
                    geometry.plan = ops.engine().resample_plan(
                        batch=first.batch_size, in_shape=geometry.in_shape, out_shape=geometry.out_shape, mapping=geometry.mapping_dev,
                        control_points=geometry.field_tensor, in_spacing=geometry.in_affine.spacing, out_spacing=geometry.out_affine.spacing,
                        affine_first=params["affine_first"], cp_skip=geometry.cp_skip, passthrough=geometry.passthrough_all,
                        large_boxes=geometry.large_boxes,
                    )
                ahead.event = side.record_event()
            ahead.tensors = [
                t for t in (geometry.mapping_dev, geometry.field_tensor, geometry.cp_skip, geometry.passthrough_all, geometry.plan) if t is not None
            ]
            ahead.payload = geometry
            params._ahead = ahead
        except Exception:  # noqa: BLE001 — whatever is wrong with these parameters is reported by `apply_transform`, in its turn
            params._ahead = None

    def apply_transform(self, batch: SubjectsBatch, params: dict[str, Any]) -> SubjectsBatch:
        selected = params.get("selected_images", [])
        if not selected:
            return batch
        ahead = getattr(params, "_ahead", None)
        prepared = None
        if ahead is not None:  # (Compose drew ahead: the launch geometry is on the device, or on its way on the side stream)
            params._ahead = None
            tensor = batch.images[selected[0]].data
            if ahead.payload is not None and tensor.is_cuda and ahead.payload.mapping_dev.device == tensor.device:
                ahead.join(tensor.device)
                prepared = ahead.payload
        target_space = _deserialize_space(params["target"])
        if prepared is not None:  # (the parameters were resolved by `_prefetch`, which only prepares launches that sample)
            matrix = field = displacement = per_sample = None
        else:
            matrix, field, displacement, per_sample = _resolve_spatial_params(params)
            if target_space is None and matrix is None and field is None and displacement is None and per_sample is None:
                return batch  # exact no-op: nothing sampled / every element gated out (spatial.py:579-590)
        _apply_spatial_to_batch(
            batch=batch,
            image_names=selected,
            target_space=target_space,
            affine_matrix=matrix,
            control_points=field,
            max_displacement=displacement,
            affine_first=params["affine_first"],
            image_interpolation=params["image_interpolation"],
            label_interpolation=params["label_interpolation"],
            one_hot_label_interpolation=params.get("one_hot_label_interpolation", "linear"),
            antialias=params.get("antialias", False),
            default_pad_value=params["default_pad_value"],
            default_pad_label=float(params["default_pad_label"]),
            per_sample=per_sample,
            prepared=prepared,
        )
        return batch

    @property
    def invertible(self) -> bool:
        return True

    def inverse(self, params: dict[str, Any]) -> "_SpatialInverse":
        """Exact inverse affine, negated elastic field, flipped order, back to the original grid."""
        original = _deserialize_space(params["original"])
        if original is None:
            raise RuntimeError("Spatial inverse needs the original output space")
        common: dict[str, Any] = {
            "target": original,
            "affine_first": not params["affine_first"],
            "image_interpolation": params["image_interpolation"],
            "label_interpolation": params["label_interpolation"],
            "one_hot_label_interpolation": params.get("one_hot_label_interpolation", "linear"),
            "default_pad_value": params["default_pad_value"],
            "default_pad_label": float(params["default_pad_label"]),
            "copy": False,
            "include": params["selected_images"],
        }
        if "affine_matrix" in (params.get("_batched_keys") or []):
            matrices, fields, displacements = [], [], []
            for matrix, field in zip(params["affine_matrix"], params["control_points"], strict=True):
                matrices.append(None if matrix is None else np.linalg.inv(np.asarray(matrix, dtype=np.float64)))
                negated = None if field is None else -torch.as_tensor(field, dtype=torch.float32)
                fields.append(negated)
                displacements.append(None if negated is None else _max_abs_displacement(negated))
            per_sample = _PerSampleGrids(matrices, fields, displacements)
            return _SpatialInverse(affine_matrix=None, control_points=None, per_sample=per_sample, **common)
        matrix = params["affine_matrix"]
        field = params["control_points"]
        return _SpatialInverse(
            affine_matrix=None if matrix is None else np.linalg.inv(np.asarray(matrix, dtype=np.float64)),
            control_points=None if field is None else -torch.as_tensor(field, dtype=torch.float32),
            **common,
        )


class _SpatialInverse(SpatialTransform):
    """Concrete inverse of ``Spatial`` used by history replay (spatial.py:679-756)."""

    def __init__(
        self,
        *,
        target,
        affine_matrix,
        control_points,
        affine_first: bool,
        image_interpolation: str,
        label_interpolation: str,
        one_hot_label_interpolation: str = "linear",
        default_pad_value,
        default_pad_label: float,
        per_sample: _PerSampleGrids | None = None,
        **kwargs: Any,
    ) -> None:
        super().__init__(**kwargs)
        self.target = target
        self.affine_matrix = None if affine_matrix is None else np.array(affine_matrix, dtype=np.float64)
        self.control_points = None if control_points is None else _parse_control_points(control_points)
        self.per_sample = per_sample
        self.affine_first = affine_first
        self.image_interpolation = _parse_interpolation(image_interpolation)
        self.label_interpolation = _parse_interpolation(label_interpolation)
        self.one_hot_label_interpolation = _parse_interpolation(one_hot_label_interpolation)
        self.default_pad_value = _parse_default_pad_value(default_pad_value)
        self.default_pad_label = float(default_pad_label)

    def apply_transform(self, batch: SubjectsBatch, params: dict[str, Any]) -> SubjectsBatch:
        displacement = None
        if self.per_sample is None and self.control_points is not None:
            displacement = _max_abs_displacement(self.control_points)
        _apply_spatial_to_batch(
            batch=batch,
            image_names=list(self._get_images(batch)),
            target_space=self.target,
            affine_matrix=self.affine_matrix,
            control_points=self.control_points,
            max_displacement=displacement,
            affine_first=self.affine_first,
            image_interpolation=self.image_interpolation,
            label_interpolation=self.label_interpolation,
            one_hot_label_interpolation=self.one_hot_label_interpolation,
            antialias=False,
            default_pad_value=self.default_pad_value,
            default_pad_label=self.default_pad_label,
            per_sample=self.per_sample,
        )
        return batch


class Resample(Spatial):
    """Resampling-only convenience wrapper (spatial.py:759-803); default target 1 mm isotropic."""

    def __init__(
        self,
        target=1,
        image_interpolation: str | int = "linear",
        label_interpolation: str | int = "nearest",
        one_hot_label_interpolation: str | int = "linear",
        antialias: bool = False,
        **kwargs: Any,
    ) -> None:
        super().__init__(
            target=target,
            image_interpolation=image_interpolation,
            label_interpolation=label_interpolation,
            one_hot_label_interpolation=one_hot_label_interpolation,
            antialias=antialias,
            **kwargs,
        )


class Affine(Spatial):
    """Affine-only convenience wrapper (spatial.py:806-869)."""

    def __init__(
        self,
        *,
        scales=1.0,
        degrees=0.0,
        translation=0.0,
        isotropic: bool = False,
        center: str = "image",
        default_pad_value: str | float = "minimum",
        default_pad_label: int | float = 0,
        image_interpolation: str | int = "linear",
        label_interpolation: str | int = "nearest",
        one_hot_label_interpolation: str | int = "linear",
        **kwargs: Any,
    ) -> None:
        super().__init__(
            scales=scales,
            degrees=degrees,
            translation=translation,
            isotropic=isotropic,
            center=center,
            default_pad_value=default_pad_value,
            default_pad_label=default_pad_label,
            image_interpolation=image_interpolation,
            label_interpolation=label_interpolation,
            one_hot_label_interpolation=one_hot_label_interpolation,
            **kwargs,
        )
        self._warn_if_noop(
            is_noop=self.scales.is_constant(1.0) and self.degrees.is_constant(0.0) and self.translation.is_constant(0.0),
            hint="degrees=(-15, 15)",
        )


class ElasticDeformation(Spatial):
    """Elastic-only convenience wrapper (spatial.py:872-922); default 7.5 mm on 7^3 control points."""

    def __init__(
        self,
        *,
        control_points=None,
        num_control_points: int | tuple[int, int, int] = 7,
        max_displacement=7.5,
        locked_borders: int = 2,
        image_interpolation: str | int = "linear",
        label_interpolation: str | int = "nearest",
        one_hot_label_interpolation: str | int = "linear",
        **kwargs: Any,
    ) -> None:
        super().__init__(
            control_points=control_points,
            num_control_points=num_control_points,
            max_displacement=max_displacement,
            locked_borders=locked_borders,
            image_interpolation=image_interpolation,
            label_interpolation=label_interpolation,
            one_hot_label_interpolation=one_hot_label_interpolation,
            **kwargs,
        )


class _LaunchGeometry:
    """What every launch of one `_apply_spatial_to_batch` call shares, on the device: grids, mapping, control points, flags
    (+ the brick plan of the fused launch when it was made ahead, `Spatial._prefetch`)."""

    __slots__ = ("in_shape", "in_affine", "out_shape", "out_affine", "mapping_dev", "field_tensor", "cp_skip", "passthrough_all", "flags", "plan",
                 "large_boxes")

    def __init__(self, in_shape, in_affine, out_shape, out_affine, mapping_dev, field_tensor, cp_skip, passthrough_all, flags, large_boxes=False) -> None:
        self.in_shape, self.in_affine, self.out_shape, self.out_affine = in_shape, in_affine, out_shape, out_affine
        self.mapping_dev, self.field_tensor, self.cp_skip, self.passthrough_all, self.flags = mapping_dev, field_tensor, cp_skip, passthrough_all, flags
        self.plan = None
        self.large_boxes = large_boxes  # 0 / 1 / 2: no / some / most bricks' input boxes exceed the planned roads' staging tile (`_expects_large_boxes`)


# floats of ONE staging tile of the planned roads (csrc/resample.hip: 160 KB of LDS per CU, three blocks, granules of 1 280 bytes)
_PLANNED_TILE_FLOATS = 13440
# how much of `d (15 / cell)` per output axis a displacement component typically varies over a brick (calibrated against the planner's
# own boxes: scripts/r5_box_estimate.py)
_FIELD_VARIATION = 0.2
# The ESTIMATE below is the largest box of an element (worst fractional position, worst alignment); calibrated against the planner's
# descriptors (scripts/r5_box_estimate.py): at 1.0 x the tile one brick in ten of the element really exceeds it, at 1.15 x more than half.
#   level 1 (TIO_GEOM_LARGE_BOXES): ANY element at 1.05 x (the bench's extreme — 10 degrees about all three axes — sits at 1.008) — the exact-coordinate lean road lists such bricks and stages them in passes
#     behind the launch (6 - 15 us: more than the odd brick on the per-voxel road costs, less than a tenth of an element's does);
#   level 2 (TIO_GEOM_MOSTLY_LARGE_BOXES): at least half of the elements at 1.15 x — the pass logic in every block of one launch.
_LARGE_BOX_MARGIN_SOME, _LARGE_BOX_MARGIN_MOST, _LARGE_BOX_FRACTION_MOST = 1.05, 1.15, 0.5


def _expects_large_boxes(mapping: np.ndarray | None, displacements, field_shape, out_shape, in_spacing) -> int:
    """Will the input box of a 16^3 output brick exceed the planned roads' staging tile — for some element (1), for most (2)?

    The box of a brick under the output -> input voxel mapping ``M`` spans ``15 sum_c |M_rc|`` voxels along input axis ``r``
    (plus the displacement field's variation over the brick, plus the taps), rows padded to 16-byte chunks — the planner's
    own arithmetic (csrc/resample_fast.hpp: plan_bricks_kernel) on the host copy of the mappings.  A HINT: it chooses between
    roads that compute the same values (``TIO_GEOM_LARGE_BOXES`` / ``TIO_GEOM_MOSTLY_LARGE_BOXES``), so an estimate is enough."""
    if mapping is None:
        return 0
    extent = 15.0 * np.abs(mapping[:, :, :3]).sum(axis=2)  # (n, 3): 15 (|M_r0| + |M_r1| + |M_r2|)
    if field_shape is not None and displacements is not None:
        # the field moves a point by up to d mm; between two control points it is linear, so over a brick edge (15 voxels) a
        # component varies by ~d (15 / cell) along each output axis for typical draws (the bound is twice that)
        share = 0.0
        for axis in range(3):
            share += 15.0 * max(int(field_shape[axis]) - 1, 1) / max(int(out_shape[axis]) - 1, 1)
        share = min(1.0, share / 3.0 * _FIELD_VARIATION)
        if isinstance(displacements, np.ndarray):  # (make_params' block draw: already (n, 3))
            limits = displacements
        else:  # a list with None for the elements without a field (one entry when the batch shares its parameters)
            limits = np.array([(0.0, 0.0, 0.0) if d is None else d for d in displacements])
        if limits.shape[0] not in (1, extent.shape[0]):
            limits = limits[:1]
        extent = extent + np.abs(limits) * np.array([share / float(in_spacing[0]), share / float(in_spacing[1]), share / float(in_spacing[2])])
    length = np.floor(extent) + 3.0  # first tap to last tap + 1, the fractional position
    chunks = np.ceil((length[:, 2] + 3.0) * 0.25)  # rows start on a 16-byte boundary: up to three floats in front
    floats = length[:, 0] * length[:, 1] * chunks * 4.0
    if int(np.count_nonzero(floats > _LARGE_BOX_MARGIN_MOST * _PLANNED_TILE_FLOATS)) >= _LARGE_BOX_FRACTION_MOST * floats.shape[0]:
        return 2
    return 1 if bool(np.any(floats > _LARGE_BOX_MARGIN_SOME * _PLANNED_TILE_FLOATS)) else 0


def _prepare_launch_geometry(
    first: ImagesBatch, device, *, target_space, affine_matrix: np.ndarray | None, control_points: Tensor | None, max_displacement,
    per_sample: _PerSampleGrids | None,
) -> _LaunchGeometry:
    """The parameter half of `_apply_spatial_to_batch`: reads the first image's geometry (never its voxels) and the drawn
    parameters, uploads on the CURRENT stream of *device*."""
    batch_size = first.batch_size
    in_shape, in_affine = _spatial_shape(first), first.affines[0]
    out_shape, out_affine = target_space if target_space is not None else (in_shape, in_affine)

    if per_sample is None:
        matrices = [affine_matrix]
        fields = [control_points]
        displacements = [max_displacement]
    else:
        if len(per_sample.affine_matrices) != batch_size:
            raise RuntimeError(
                f"Per-instance spatial parameters were recorded for {len(per_sample.affine_matrices)}"
                f" elements but the batch has {batch_size}"
            )
        matrices, fields, displacements = per_sample.affine_matrices, per_sample.control_points, per_sample.max_displacements

    # output voxel -> input voxel: inv(A_in) @ inv(T) @ A_out in float64, cast to float32 (spatial.py:1582-1601)
    # (stacked: numpy runs the same LAPACK / BLAS routine per 4x4 slice as the one-at-a-time form)
    in_inverse = in_affine.inverse_numpy() if hasattr(in_affine, "inverse_numpy") else np.linalg.inv(in_affine.numpy())
    out_matrix = out_affine.numpy()
    stacked_matrices = getattr(matrices, "stacked", None)
    present_matrices = [index for index, matrix in enumerate(matrices) if matrix is not None]
    mapping = None
    if stacked_matrices is not None:  # every element has a world affine, already one (n, 4, 4) array
        mapping = ((in_inverse @ np.linalg.inv(stacked_matrices)) @ out_matrix)[:, :3].astype(np.float32)
    elif not present_matrices:
        # pure elastic / gated-out batch: every element maps through inv(A_in) @ I @ A_out; when that IS the identity
        # in float32 (it need not be bit for bit: inv(A) @ A carries float64 rounding) one cached device tensor serves
        shared = ((in_inverse @ np.eye(4)) @ out_matrix)[:3].astype(np.float32)
        if not np.array_equal(shared, _EYE34):
            mapping = np.repeat(shared[None], len(matrices), axis=0)
    else:
        world = np.stack([np.eye(4) if matrix is None else np.asarray(matrix, dtype=np.float64) for matrix in matrices])
        world[present_matrices] = np.linalg.inv(world[present_matrices])
        mapping = ((in_inverse @ world) @ out_matrix)[:, :3].astype(np.float32)

    out_spacing = np.asarray(out_affine.spacing, dtype=np.float64)
    field_tensor = None
    cp_skip = None
    if isinstance(fields, _StackedFields):  # the whole batch's control points as one host block (make_params drew them so)
        block = fields.stacked
        limits = np.asarray(displacements, dtype=np.float64)
        grid_spacing = np.asarray(
            [_folding_grid_spacing(float(out_shape[axis]) * float(out_spacing[axis]), float(block.shape[1 + axis]) - _SPLINE_ORDER) for axis in range(3)]
        )
        if bool((limits > grid_spacing / 2).any()):  # (the warning, element by element, exactly as the loop below words it)
            for f, displacement in zip(block.unbind(0), displacements, strict=True):
                _check_folding(f.numpy(), displacement, out_shape, out_spacing)
        field_tensor = ops.h2d(block, device)
        present = [block]
    else:
        present = [f for f in fields if f is not None]
    if present and field_tensor is None:
        shapes = {tuple(f.shape) for f in present}
        if len(shapes) != 1:
            raise RuntimeError(f"All control-point fields of a batch must share one shape, got {sorted(shapes)}")
        stacked = []
        for f, displacement in zip(fields, displacements, strict=True):
            if f is None:
                stacked.append(torch.zeros(present[0].shape, dtype=torch.float32))
                continue
            f = f.detach().to(device="cpu", dtype=torch.float32)
            if displacement is None:
                displacement = _max_abs_displacement(f)
            _check_folding(f.numpy(), displacement, out_shape, out_spacing)
            stacked.append(f)
        field_tensor = ops.h2d(torch.stack(stacked), device)
        if len(present) != len(fields):
            cp_skip = ops.h2d(torch.tensor([f is None for f in fields], dtype=torch.uint8), device)

    passthrough_all = None
    if per_sample is not None and target_space is None:
        flags = [False] * batch_size if isinstance(fields, _StackedFields) else [m is None and f is None for m, f in zip(matrices, fields, strict=True)]
        if any(flags):
            passthrough_all = ops.h2d(torch.tensor(flags, dtype=torch.uint8), device)
    else:
        flags = [False] * batch_size

    mapping_dev = _identity_mapping(device) if mapping is None else ops.h2d(torch.from_numpy(mapping), device)
    large_boxes = _expects_large_boxes(
        mapping, displacements if field_tensor is not None else None, None if field_tensor is None else tuple(field_tensor.shape[1:4]), out_shape,
        in_affine.spacing,
    )
    return _LaunchGeometry(in_shape, in_affine, out_shape, out_affine, mapping_dev, field_tensor, cp_skip, passthrough_all, flags, large_boxes)



# =============================================================================
# functional seam S1: _apply_spatial_to_batch on the engine
# =============================================================================
def _apply_spatial_to_batch(
    *,
    batch: SubjectsBatch,
    image_names: list[str],
    target_space,
    affine_matrix: np.ndarray | None,
    control_points: Tensor | None,
    max_displacement,
    affine_first: bool,
    image_interpolation: str,
    label_interpolation: str,
    one_hot_label_interpolation: str = "linear",
    antialias: bool,
    default_pad_value,
    default_pad_label: float,
    per_sample: _PerSampleGrids | None = None,
    prepared: "_LaunchGeometry | None" = None,
) -> None:
    """Resample every selected image of *batch* with one fused launch (spatial.py:1110-1272).

    Same keyword signature as the reference function (plus *prepared*: the launch geometry when a `Compose` that draws
    ahead has already put it on the device, `Spatial._prefetch`).  The geometry comes from the
    first image (shape, affine); all selected images share it (checked in
    ``make_params``).  With *per_sample* every batch element gets its own 3x4
    mapping / control-point field; elements with no geometry and no target are
    passed through bit-exactly (spatial.py:1167-1174).
    """
    if not image_names:
        return
    first = batch.images[image_names[0]]
    if prepared is None:
        prepared = _prepare_launch_geometry(
            first, first.data.device, target_space=target_space, affine_matrix=affine_matrix, control_points=control_points,
            max_displacement=max_displacement, per_sample=per_sample,
        )
    in_shape, in_affine, out_shape, out_affine = prepared.in_shape, prepared.in_affine, prepared.out_shape, prepared.out_affine
    field_tensor, cp_skip, passthrough_all, flags, mapping_dev = (
        prepared.field_tensor, prepared.cp_skip, prepared.passthrough_all, prepared.flags, prepared.mapping_dev
    )
    engine = ops.engine()

    def resample(tensors, interps, fills, gated=True, **label_arguments):
        passthrough = passthrough_all if gated else None
        # images of another shape than the first (Resample onto a named image of a multi-resolution subject) are
        # sampled with the first image's grid like in the reference (spatial.py:1136-1191): one call per shape
        shapes = [tuple(t.shape[2:]) for t in tensors]
        if any(shape != shapes[0] for shape in shapes) or shapes[0] != tuple(in_shape):
            outputs: list = [None] * len(tensors)
            for shape in dict.fromkeys(shapes):
                members = [n for n, s in enumerate(shapes) if s == shape]
                extra = {key: [value[n] for n in members] for key, value in label_arguments.items()}
                results = engine.resample3d(
                    [tensors[n] for n in members], out_shape=out_shape, mapping=mapping_dev, control_points=field_tensor,
                    in_spacing=in_affine.spacing, out_spacing=out_affine.spacing, affine_first=affine_first,
                    interps=[interps[n] for n in members], fills=[fills[n] for n in members], cp_skip=cp_skip,
                    passthrough=passthrough, norm_shape=in_shape, large_boxes=prepared.large_boxes, **extra,
                )
                for n, result in zip(members, results, strict=True):
                    outputs[n] = result
            return outputs
        return engine.resample3d(
            tensors,
            out_shape=out_shape,
            mapping=mapping_dev,
            control_points=field_tensor,
            in_spacing=in_affine.spacing,
            out_spacing=out_affine.spacing,
            affine_first=affine_first,
            interps=interps,
            fills=fills,
            cp_skip=cp_skip,
            passthrough=passthrough,
            plan=prepared.plan if gated else None,  # (made for the gated geometry of the fused call; ignored by calls that take another road)
            large_boxes=prepared.large_boxes,
            **label_arguments,
        )

    # every image that the fused launch can take shares ONE tio_resample3d call; the
    # "label" partial-volume mode rides along (its own kernel inside the call) unless it needs
    # the materialised one-hot channels (antialias, nearest one-hot interpolation)
    tensors, interps, fills, tables, pads, fused_names = [], [], [], [], [], []
    finished: dict[str, Tensor] = {}
    for name in image_names:
        img_batch = batch.images[name]
        is_label = _is_label_batch(img_batch)
        interpolation = label_interpolation if is_label else image_interpolation
        data = img_batch.data
        table, pad = None, 0.0
        if interpolation == LABEL_INTERPOLATION:  # only reachable for label maps (validated by the constructors)
            if data.shape[1] > 1:
                # already one-hot / probabilistic: channels resampled as they are, zero outside,
                # floating-point result (spatial.py:1345-1358)
                work = data.float()
                if antialias:
                    work = _antialias(engine, work, in_affine, out_affine)
                if _ORDERS[one_hot_label_interpolation] >= 2:
                    coefficients = engine.bspline_prefilter(work, _ORDERS[one_hot_label_interpolation])
                    sampled = resample([coefficients], [one_hot_label_interpolation], [None], gated=False)[0]
                    sampled = sampled.to(data.dtype) if data.dtype.is_floating_point else sampled
                    if any(flags) and tuple(sampled.shape) == tuple(data.shape):
                        rows = ops.h2d(torch.tensor(flags, dtype=torch.bool), data.device)
                        sampled = torch.where(rows.view(-1, 1, 1, 1, 1), data.to(sampled.dtype), sampled)
                    finished[name] = sampled
                    continue
                if data.dtype.is_floating_point and data.dtype != torch.float32:
                    finished[name] = resample([work], [one_hot_label_interpolation], [None])[0].to(data.dtype)
                    continue
                data, interpolation, fill = work, one_hot_label_interpolation, None
            elif antialias or one_hot_label_interpolation != "linear":
                finished[name] = _label_partial_volume_composite(
                    engine, resample, data, in_affine, out_affine, antialias=antialias,
                    one_hot_label_interpolation=one_hot_label_interpolation, default_pad_label=default_pad_label,
                    passthrough_flags=flags,
                )
                continue
            else:
                fill = None
                table = engine.unique_labels(data)  # torch.unique(data), sorted; sizes the reference's one-hot (spatial.py:1360)
                pad = float(default_pad_label)
        elif _ORDERS[interpolation] >= 2:  # B-spline orders 2 - 7
            # interpol.grid_pull(data.float(), grid, interpolation=order, bound="dct2", extrapolate=False, prefilter=True)
            # .to(data.dtype) (spatial.py:1734-1761, 1860-1878): coefficients first, then the (order + 1)^3-tap sum at the
            # voxel coordinates; zero outside the field of view (the reference does not apply its fill value here).
            # Gated-out elements are restored on this side: the launch would copy their COEFFICIENTS.
            work = _antialias(engine, data, in_affine, out_affine) if antialias and not is_label else data
            coefficients = engine.bspline_prefilter(work, _ORDERS[interpolation])
            sampled = resample([coefficients], [interpolation], [None], gated=False)[0].to(data.dtype)
            if any(flags) and tuple(sampled.shape) == tuple(data.shape):
                rows = ops.h2d(torch.tensor(flags, dtype=torch.bool), data.device)
                sampled = torch.where(rows.view(-1, 1, 1, 1, 1), data, sampled)
            finished[name] = sampled
            continue
        else:
            fill = _fill_value(engine, img_batch, default_pad_value=default_pad_value, default_pad_label=default_pad_label)
            if antialias and not is_label:
                data = _antialias(engine, data, in_affine, out_affine)
        tensors.append(data)
        interps.append(interpolation)
        fills.append(fill)
        tables.append(table)
        pads.append(pad)
        fused_names.append(name)

    if tensors:
        label_arguments = {"label_tables": tables, "pad_labels": pads} if LABEL_INTERPOLATION in interps else {}
        finished.update(zip(fused_names, resample(tensors, interps, fills, **label_arguments), strict=True))
    for name in image_names:
        img_batch = batch.images[name]
        originals = list(img_batch.affines)
        img_batch.data = finished[name]
        keeps_values = target_space is None and hasattr(out_affine, "same_values")
        img_batch.affines[:] = [
            # every resampled element adopts the output grid (spatial.py:1100-1107); when the element's own matrix already holds
            # exactly those values (no target, and the shared-space check has passed) its object is kept instead of replaced
            # by a clone of the first element's: the same 4x4, one tensor clone per element less
            originals[index]
            if flags[index] or (target_space is None and originals[index] is out_affine) or (keeps_values and originals[index].same_values(out_affine))
            else out_affine.clone()
            for index in range(len(originals))
        ]


def _cascade_sum_channels(sampled: Tensor) -> Tensor:
    """``sampled.sum(dim=1)`` in the rounding order of ATen's CPU kernel, on any device.

    The reference thresholds this float32 sum at 0.5 (spatial.py:1378); ATen's
    ``cascade_sum`` adds the channels in order into one accumulator and moves it up a
    level every 16 channels (aten/src/ATen/native/cpu/SumKernel.cpp, ``multi_row_sum``).
    """
    size = sampled.shape[1]
    ceil_log2 = 1 if size <= 2 else (size - 1).bit_length()
    power = max(4, ceil_log2 // 4)
    step, mask = 1 << power, (1 << power) - 1
    zeros = torch.zeros_like(sampled[:, 0])
    levels = [zeros.clone() for _ in range(4)]
    index = 0
    while index + step <= size:
        for _ in range(step):
            levels[0] = levels[0] + sampled[:, index]
            index += 1
        for level in range(1, 4):
            levels[level] = levels[level] + levels[level - 1]
            levels[level - 1] = zeros.clone()
            if index & (mask << (level * power)):
                break
    for remaining in range(index, size):
        levels[0] = levels[0] + sampled[:, remaining]
    for level in range(1, 4):
        levels[0] = levels[0] + levels[level]
    return levels[0]


def _label_partial_volume_composite(
    engine, resample, data: Tensor, in_affine: AffineMatrix, out_affine: AffineMatrix, *, antialias: bool,
    one_hot_label_interpolation: str, default_pad_label: float, passthrough_flags: list[bool],
) -> Tensor:
    """The reference's four steps with the one-hot channels materialised (spatial.py:1360-1389).

    Needed when the channels are smoothed before sampling (``antialias=True``) or sampled
    with ``"nearest"``; the plain linear case is fused inside ``tio_resample3d`` instead.
    """
    labels = engine.unique_labels(data).to(data.dtype)  # torch.unique(data)
    one_hot = (data[:, :1] == labels.view(1, -1, 1, 1, 1)).float()
    if antialias:
        one_hot = _antialias(engine, one_hot, in_affine, out_affine)
    if _ORDERS[one_hot_label_interpolation] >= 2:  # B-spline channels, orders 2 - 7 (interpol.grid_pull in the reference; §4.9 of DESIGN.md)
        coefficients = engine.bspline_prefilter(one_hot, _ORDERS[one_hot_label_interpolation])
        sampled = resample([coefficients], [one_hot_label_interpolation], [None], gated=False)[0]
    else:
        sampled = resample([one_hot], [one_hot_label_interpolation], [None])[0]
    resampled = labels[sampled.argmax(dim=1)]
    in_bounds = _cascade_sum_channels(sampled) > 0.5
    resampled = torch.where(in_bounds, resampled, torch.full_like(resampled, default_pad_label))
    out = resampled.unsqueeze(1).to(data.dtype)
    for index, keep in enumerate(passthrough_flags):
        if keep:  # the kernel copied the one-hot rows; the labels of a gated-out element are the input's
            out[index] = data[index]
    return out


_IDENTITY_MAPPINGS: dict[str, Tensor] = {}
_EYE34 = np.eye(3, 4, dtype=np.float32)


def _identity_mapping(device) -> Tensor:
    """The shared ``(1, 3, 4)`` identity mapping, uploaded once per device (read-only)."""
    key = str(device)
    mapping = _IDENTITY_MAPPINGS.get(key)
    if mapping is None:
        mapping = _IDENTITY_MAPPINGS[key] = ops.h2d(torch.eye(3, 4, dtype=torch.float32)[None].contiguous(), device)
    return mapping


def _fill_value(engine, img_batch: ImagesBatch, *, default_pad_value, default_pad_label: float) -> Tensor | None:
    """Per-channel fill tensor, or ``None`` for the reference's "scalar 0 → no mask" branch.

    spatial.py:2034-2086: label maps use ``default_pad_label``; a numeric pad value
    is used as is; ``"minimum"`` / ``"mean"`` / ``"otsu"`` come from the FIRST batch
    element and always take the mask path, even when they evaluate to 0.
    """
    data = img_batch.data
    channels = data.shape[1]
    if _is_label_batch(img_batch):
        value: Any = float(default_pad_label)
    elif isinstance(default_pad_value, Number):
        value = float(default_pad_value)
    elif not isinstance(default_pad_value, str):
        raise TypeError(f"default_pad_value must be a string or number, got {type(default_pad_value)}")
    elif default_pad_value == "minimum":
        folded = ops.folded_channel_min(data)  # the resampler that wrote this tensor may have left it behind
        return folded if folded is not None else engine.channel_min(data)  # stays on the device: no .item() sync
    elif default_pad_value in ("mean", "otsu"):
        values = [_compute_channel_pad_value(channel, default_pad_value) for channel in data[0]]
        return ops.h2d(torch.tensor(values, dtype=torch.float32), data.device)
    else:
        raise ValueError(f'Unknown default_pad_value "{default_pad_value}"')
    if value == 0.0:
        return None
    return ops.h2d(torch.full((channels,), value, dtype=torch.float32), data.device)


def _compute_channel_pad_value(channel: Tensor, strategy: str) -> float:
    """One channel's pad value for ``"minimum"`` / ``"mean"`` / ``"otsu"`` (spatial.py:2089-2101); host-side form."""
    if strategy == "minimum":
        return float(channel.min().item())
    if strategy in ("mean", "otsu"):
        return _border_mean(channel, filter_otsu=strategy == "otsu")
    raise ValueError(f'Unknown default_pad_value "{strategy}"')


def _border_mean(channel: Tensor, *, filter_otsu: bool) -> float:
    """Mean of the six boundary faces, optionally of the voxels under their Otsu threshold.

    Rare, non-default pad modes (spatial.py:2104-2131): the faces (6 S^2 values)
    are brought to the host and reduced in float32/float64 numpy.
    """
    faces = [channel[0], channel[-1], channel[:, 0], channel[:, -1], channel[:, :, 0], channel[:, :, -1]]
    borders = torch.cat([f.reshape(-1) for f in faces]).float().cpu()
    if not filter_otsu:
        return float(borders.mean().item())
    below = borders[borders < _otsu_threshold(borders)]
    if below.numel():
        return float(below.mean().item())
    return float(borders.mean().item())


def _otsu_threshold(values: Tensor) -> float:
    """Otsu sweep over the sorted values: maximise ``w_b * w_f * (mean_b - mean_f)^2`` (spatial.py:2133-2168).

    Vectorised; the running sums are float64 like the reference's Python floats, the total is the float32
    ``Tensor.sum`` it starts from, and the first maximum wins like its strict ``>`` comparison.
    """
    ordered = np.sort(values.detach().reshape(-1).float().cpu().numpy())
    count = ordered.size
    if count == 0:
        return 0.0
    as_double = ordered.astype(np.float64)
    total = float(torch.from_numpy(ordered).sum().item())
    background_sum = np.cumsum(as_double[:-1])
    background_count = np.arange(1, count, dtype=np.float64)
    foreground_count = count - background_count
    mean_background = background_sum / background_count
    mean_foreground = (total - background_sum) / foreground_count
    variance = (background_count / count) * (foreground_count / count) * (mean_background - mean_foreground) ** 2
    if variance.size:
        best = int(np.argmax(variance))
        if variance[best] > 0.0:
            return float(as_double[best])
    return float(as_double[0])


def _antialias(engine, data: Tensor, in_affine: AffineMatrix, out_affine: AffineMatrix) -> Tensor:
    """Gaussian pre-filter along down-sampled axes (Cardoso et al. 2015; spatial.py:1921-2031)."""
    in_spacing = np.asarray(in_affine.spacing, dtype=np.float64)
    factors = np.asarray(out_affine.spacing, dtype=np.float64) / in_spacing
    sigmas = np.zeros(3, dtype=np.float64)
    for axis in range(3):
        k = factors[axis]
        if k > 1.0:
            variance = (k**2 - 1) * (2 * np.sqrt(2 * np.log(2))) ** (-2)
            sigmas[axis] = in_spacing[axis] * np.sqrt(variance) / in_spacing[axis]
    if np.all(sigmas == 0):
        return data
    taps, radius, _ = _stacked_gaussian_taps(sigmas[None])
    work = data if data.dtype in ops.FLOAT_DTYPES else data.float()
    return engine.separable_conv3d(work, ops.h2d(taps, data.device), radius).to(data.dtype)


# =============================================================================
# geometry helpers (float64 on the host, exactly like the reference)
# =============================================================================
def _spatial_shape(img_batch: ImagesBatch) -> tuple[int, int, int]:
    return tuple(int(s) for s in img_batch.data.shape[-3:])  # type: ignore[return-value]


def _euler_to_rotation_matrix(degrees: np.ndarray) -> np.ndarray:
    """``R = Rz @ Ry @ Rx`` from XYZ Euler angles in degrees (spatial.py:2328-2365)."""
    (cx, cy, cz), (sx, sy, sz) = np.cos(np.radians(degrees)), np.sin(np.radians(degrees))
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], dtype=np.float64)
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=np.float64)
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dtype=np.float64)
    return rz @ ry @ rx


def _build_forward_affine(*, scales, degrees, translation, center, shape, affine: AffineMatrix) -> np.ndarray:
    """World-space 4x4: ``A = R @ diag(s)``, pivot at the image centre, plus translation (spatial.py:2269-2325)."""
    scaling = np.asarray(scales, dtype=np.float64).copy()
    rotation = np.asarray(degrees, dtype=np.float64).copy()
    shift = np.asarray(translation, dtype=np.float64).copy()
    if shape[-1] == 1:  # 2-D slice: suppress out-of-plane components
        scaling[2] = 1.0
        rotation[0] = rotation[1] = 0.0
        shift[2] = 0.0
    rotation_scale = _euler_to_rotation_matrix(rotation) @ np.diag(scaling)
    transform = np.eye(4, dtype=np.float64)
    transform[:3, :3] = rotation_scale
    if center == "image":
        matrix = affine.numpy()
        center_world = matrix[:3, 3] + matrix[:3, :3] @ ((np.asarray(shape, dtype=np.float64) - 1) / 2)
        transform[:3, 3] = center_world - rotation_scale @ center_world
    transform[:3, 3] += shift
    return transform


def _build_forward_affines(parameters: list, *, center, shape, affine: AffineMatrix) -> np.ndarray:
    """``_build_forward_affine`` for a list of ``(scales, degrees, translation)`` at once -> ``(n, 4, 4)`` float64.

    Same float64 operations per element (numpy's stacked ``@`` runs the same 3x3 / 4x4
    products), so each matrix equals the one-at-a-time result bit for bit.
    """
    n = len(parameters)
    if n == 0:
        return np.zeros((0, 4, 4), dtype=np.float64)
    scaling = np.array([p[0] for p in parameters], dtype=np.float64)
    rotation = np.array([p[1] for p in parameters], dtype=np.float64)
    shift = np.array([p[2] for p in parameters], dtype=np.float64)
    return _build_forward_affines_arrays(scaling, rotation, shift, center=center, shape=shape, affine=affine)


def _build_forward_affines_arrays(scaling: np.ndarray, rotation: np.ndarray, shift: np.ndarray, *, center, shape, affine: AffineMatrix) -> np.ndarray:
    """``_build_forward_affines`` for ``(n, 3)`` float64 arrays of scales, degrees and translations (not modified)."""
    n = scaling.shape[0]
    if n == 0:
        return np.zeros((0, 4, 4), dtype=np.float64)
    if shape[-1] == 1:  # 2-D slice: suppress out-of-plane components
        scaling, rotation, shift = scaling.copy(), rotation.copy(), shift.copy()
        scaling[:, 2] = 1.0
        rotation[:, :2] = 0.0
        shift[:, 2] = 0.0
    radians = np.radians(rotation)
    cos, sin = np.cos(radians), np.sin(radians)
    zeros, ones = np.zeros(n), np.ones(n)
    rx = np.stack([ones, zeros, zeros, zeros, cos[:, 0], -sin[:, 0], zeros, sin[:, 0], cos[:, 0]], axis=1).reshape(n, 3, 3)
    ry = np.stack([cos[:, 1], zeros, sin[:, 1], zeros, ones, zeros, -sin[:, 1], zeros, cos[:, 1]], axis=1).reshape(n, 3, 3)
    rz = np.stack([cos[:, 2], -sin[:, 2], zeros, sin[:, 2], cos[:, 2], zeros, zeros, zeros, ones], axis=1).reshape(n, 3, 3)
    diagonal = np.zeros((n, 3, 3), dtype=np.float64)
    diagonal[:, [0, 1, 2], [0, 1, 2]] = scaling
    rotation_scale = ((rz @ ry) @ rx) @ diagonal
    transform = np.zeros((n, 4, 4), dtype=np.float64)
    transform[:, 3, 3] = 1.0
    transform[:, :3, :3] = rotation_scale
    if center == "image":
        matrix = affine.numpy()
        center_world = matrix[:3, 3] + matrix[:3, :3] @ ((np.asarray(shape, dtype=np.float64) - 1) / 2)
        transform[:, :3, 3] = center_world - (rotation_scale @ center_world[:, None])[:, :, 0]
    transform[:, :3, 3] += shift
    return transform


class _MatrixList(list):
    """Per-element 4x4 world affines (``np.ndarray`` or ``None``) that serialise as nested lists.

    ``stacked``: the same matrices as one ``(n, 4, 4)`` array when every element has one (saves re-stacking them)."""

    stacked: np.ndarray | None = None

    def tolist(self) -> list:
        return [None if m is None else m.tolist() for m in self]


class _StackedFields:
    """The control-point fields of a batch as ONE ``(n, ni, nj, nk, 3)`` float32 host tensor that reads like the list of
    per-element fields it stands for (``len``, indexing, iteration give ``(ni, nj, nk, 3)`` views)."""

    __slots__ = ("stacked",)

    def __init__(self, stacked: Tensor) -> None:
        self.stacked = stacked

    def __len__(self) -> int:
        return int(self.stacked.shape[0])

    def __getitem__(self, index):
        return self.stacked[index]

    def __iter__(self):
        return iter(self.stacked.unbind(0))


def _all_close(values, target: float) -> bool:
    """``np.allclose(values, target)`` (rtol 1e-5, atol 1e-8) for a short tuple of Python floats."""
    bound = 1e-8 + 1e-5 * abs(target)
    return all(abs(v - target) <= bound for v in values)


_BORDER_MASKS: dict[tuple, Tensor] = {}


def _interior_mask(grid_shape, locked_borders: int) -> Tensor:
    """``(ni, nj, nk, 1)`` bool: False on the *locked_borders* outer layers of the control grid."""
    key = (tuple(grid_shape), locked_borders)
    mask = _BORDER_MASKS.get(key)
    if mask is None:
        mask = torch.ones(*grid_shape, 1, dtype=torch.bool)
        for border in range(locked_borders):
            for dim in range(3):
                index = [slice(None)] * 3
                index[dim] = border
                mask[tuple(index)] = False
                index[dim] = -1 - border
                mask[tuple(index)] = False
        _BORDER_MASKS[key] = mask
    return mask


_SCALES: dict[tuple, Tensor] = {}


def _displacement_scale(max_displacement: tuple) -> Tensor:
    """``2 * max_displacement`` as a float32 tensor (read-only; constant ranges hit the cache every time)."""
    scale = _SCALES.get(max_displacement)
    if scale is None:
        if len(_SCALES) > 256:
            _SCALES.clear()
        scale = _SCALES[max_displacement] = torch.tensor([2.0 * m for m in max_displacement], dtype=torch.float32)
    return scale


def _sample_control_points(grid_shape, max_displacement, locked_borders: int) -> Tensor:
    """``U(-max, +max)`` per axis from ONE ``torch.rand(ni, nj, nk, 3)`` draw; outer layers zeroed.

    Same values as the reference's step-by-step version (``(u - 0.5) * 2`` is exact, so
    one multiply by ``2 * max`` rounds once, like ``* max`` after it), built with three
    tensor ops instead of ~25.
    """
    field = torch.rand(*grid_shape, 3, dtype=torch.float32)
    field -= 0.5
    field *= _displacement_scale(tuple(float(m) for m in max_displacement))
    if locked_borders > 0:
        field = torch.where(_interior_mask(grid_shape, locked_borders), field, torch.zeros((), dtype=torch.float32))
    return field


def _max_abs_displacement(control_points: Tensor) -> tuple[float, float, float]:
    absolute = control_points.abs()
    return tuple(float(absolute[..., axis].max().item()) for axis in range(3))  # type: ignore[return-value]


def _folding_grid_spacing(extent: float, mesh: float) -> float:
    """``extent / mesh`` as plain floats with numpy's division-by-zero results (a 3-point grid has mesh 0)."""
    return extent / mesh if mesh != 0 else (math.copysign(math.inf, extent) if extent != 0 else math.nan)


def _check_folding(control_points: np.ndarray, max_displacement, shape, spacing: np.ndarray) -> None:
    """Warn when the displacement exceeds half the coarse-grid spacing (spatial.py:2192-2216)."""
    where = []
    for axis in range(3):
        grid_spacing = _folding_grid_spacing(float(shape[axis]) * float(spacing[axis]), float(control_points.shape[axis]) - _SPLINE_ORDER)
        if float(max_displacement[axis]) > grid_spacing / 2:
            where.append(axis)
    if where:
        warnings.warn(
            "The maximum displacement is larger than half the coarse-grid"
            f" spacing for dimensions {where}, so folding may occur",
            RuntimeWarning,
            stacklevel=4,
        )


def _check_shared_space(images: dict[str, ImagesBatch], reference_shape, reference_affine: AffineMatrix) -> None:
    for name, img_batch in images.items():
        shape = _spatial_shape(img_batch)
        if shape != reference_shape:
            raise RuntimeError(f'Image "{name}" has shape {shape}, expected {reference_shape}')
        fast_compare = hasattr(reference_affine, "same_values")  # (the reference's own AffineMatrix under reference_binding has none)
        for affine in img_batch.affines:
            if affine is reference_affine or (fast_compare and affine.same_values(reference_affine)) or torch.equal(affine.data, reference_affine.data):
                continue
            if not torch.allclose(affine.data, reference_affine.data, rtol=1e-6, atol=1e-6):
                raise RuntimeError(
                    "Spatial transforms with affine or elastic components require"
                    " selected images to share the same affine"
                )


def _resolve_spatial_params(params: dict[str, Any]):
    """``(matrix, field, max_displacement, per_sample)`` from a params dict (spatial.py:962-1008)."""
    def matrix_of(value):
        return None if value is None else np.asarray(value, dtype=np.float64)

    def field_of(value):
        return None if value is None else torch.as_tensor(value, dtype=torch.float32)

    def displacement_of(value):
        return None if value is None else (float(value[0]), float(value[1]), float(value[2]))

    parked, raw_fields = params.raw("control_points") if isinstance(params, LazyParams) else (False, None)
    if "affine_matrix" not in (params.get("_batched_keys") or []):
        return (
            matrix_of(params["affine_matrix"]),
            raw_fields if parked else field_of(params["control_points"]),
            displacement_of(params["max_displacement"]),
            None,
        )
    parked_matrices, raw_matrices = params.raw("affine_matrix") if isinstance(params, LazyParams) else (False, None)
    matrices = raw_matrices if parked_matrices and isinstance(raw_matrices, _MatrixList) else (
        list(raw_matrices) if parked_matrices else [matrix_of(m) for m in params["affine_matrix"]])
    if parked and isinstance(raw_fields, _StackedFields):
        fields = raw_fields  # one block for the whole batch: handed to the launch as it is
    else:
        fields = list(raw_fields) if parked else [field_of(c) for c in params["control_points"]]
    displacements = [displacement_of(d) for d in params["max_displacement"]]
    if all(m is None for m in matrices) and not isinstance(fields, _StackedFields) and all(f is None for f in fields):
        return None, None, None, None
    return None, None, None, _PerSampleGrids(matrices, fields, displacements)


# -- target space -----------------------------------------------------------------
def _serialize_space(space):
    if space is None:
        return None
    shape, affine = space
    return {"shape": list(shape), "affine": affine.numpy().tolist()}


def _deserialize_space(data):
    if data is None:
        return None
    shape = tuple(int(s) for s in data["shape"][:3])
    return shape, AffineMatrix(np.asarray(data["affine"], dtype=np.float64))


def _resolve_target_space(target, batch: SubjectsBatch, first_shape, first_affine: AffineMatrix):
    """User-facing *target* → ``(shape, affine)`` or ``None`` (spatial.py:1392-1422)."""
    if target is None:
        return None
    if isinstance(target, Image):
        return target.spatial_shape, target.affine.clone()
    if isinstance(target, (str, Path)):
        if isinstance(target, str) and target in batch.images:
            reference = batch.images[target]
            return _spatial_shape(reference), reference.affines[0].clone()
        raise ValueError(
            f'Unknown target "{target}". Pass an image name in the subject, an Image, a (shape, affine) pair'
            " or a spacing specification (reading a target image from a file path is not supported here)"
        )
    if _is_target_space_tuple(target):
        return _parse_target_space_tuple(*target)
    if not isinstance(target, (int, float, tuple, list, np.ndarray, Choice, Distribution)):
        raise ValueError(f'Target not understood: "{target}"')
    return _new_shape_affine(first_shape, first_affine, _resolve_target_spacing(target))


def _is_target_space_tuple(target) -> bool:
    """``(shape, affine)``; a 2-tuple of plain numbers is a spacing range instead (spatial.py:1425-1443)."""
    return isinstance(target, tuple) and len(target) == 2 and not isinstance(target[0], Number)


def _parse_target_space_tuple(shape, affine):
    if len(shape) != 3:
        raise ValueError(f"Target shape must have length 3, got {len(shape)}")
    return tuple(int(s) for s in shape), AffineMatrix(affine)


def _is_spacing_tuple(value) -> bool:
    return isinstance(value, tuple) and len(value) == 3


def _is_spacing_list(value) -> bool:
    return isinstance(value, list) and len(value) == 3


def _parse_spacing(value) -> tuple[float, float, float]:
    """A fixed spacing: one number or three (tuple / list / array), strictly positive (spatial.py:1504-1524)."""
    if isinstance(value, np.ndarray):
        spacing = tuple(float(v) for v in value.flat)
    elif isinstance(value, Number):
        spacing = (float(value),) * 3
    else:
        spacing = tuple(float(v) for v in value)
    if len(spacing) != 3:
        raise ValueError(f"Spacing must have 3 values, got {len(spacing)}")
    if any(v <= 0 for v in spacing):
        raise ValueError(f"Spacing must be strictly positive, got {spacing}")
    return spacing  # type: ignore[return-value]


def _resolve_target_spacing(value) -> tuple[float, float, float]:
    """Deterministic or random spacing spec → positive 3-tuple (spatial.py:1446-1469)."""
    if isinstance(value, (np.ndarray, int, float)):
        return _parse_spacing(value)
    return _parse_spacing(_ParameterRange(tuple(value) if isinstance(value, list) else value).sample())


def _new_shape_affine(shape, affine: AffineMatrix, spacing):
    """Output grid for a new spacing, same physical centre (spatial.py:1472-1501)."""
    old_spacing = np.asarray(affine.spacing, dtype=np.float64)
    new_spacing = np.asarray(spacing, dtype=np.float64)
    old_shape = np.asarray(shape, dtype=np.float64)
    new_shape = np.floor(old_shape * old_spacing / new_spacing)
    new_shape[old_shape == 1] = 1
    rotation = affine.direction.cpu().numpy()
    old_center = np.asarray(affine.origin, dtype=np.float64) + rotation @ (((old_shape - 1) / 2) * old_spacing)
    new_affine = np.eye(4, dtype=np.float64)
    new_affine[:3, :3] = rotation * new_spacing
    new_affine[:3, 3] = old_center - rotation @ (((new_shape - 1) / 2) * new_spacing)
    return tuple(int(s) for s in new_shape), AffineMatrix(new_affine)


# -- argument parsing -------------------------------------------------------------
def _parse_interpolation(interpolation) -> str:
    if isinstance(interpolation, int) and not isinstance(interpolation, bool):
        names = {order: name for name, order in _ORDERS.items()}
        if interpolation not in names:
            raise ValueError(f"Interpolation order {interpolation} is not supported. Must be 0-7.")
        return names[interpolation]
    if not isinstance(interpolation, str):
        raise TypeError(f"Interpolation must be a string or int, got {type(interpolation)}")
    lowered = interpolation.lower()
    if lowered not in _SUPPORTED_INTERPOLATIONS:
        raise ValueError(
            f'Interpolation "{lowered}" is not supported. Supported values are {_SUPPORTED_INTERPOLATIONS}'
        )
    return lowered


def _parse_default_pad_value(value):
    if isinstance(value, Number):
        return float(value)
    if value in _SUPPORTED_PAD_VALUES:
        return value
    raise ValueError('default_pad_value must be "minimum", "mean", "otsu", or a numeric value')


def _parse_num_control_points(value) -> tuple[int, int, int]:
    parsed = (value, value, value) if isinstance(value, int) else tuple(value)
    for axis, number in enumerate(parsed):
        if not isinstance(number, int) or number < 4:
            raise ValueError(
                f"Each num_control_points value must be an integer greater than 3; axis {axis} got {number}"
            )
    return parsed  # type: ignore[return-value]


def _parse_control_points(control_points) -> Tensor:
    if isinstance(control_points, Tensor):
        tensor = control_points.clone().detach().to(torch.float32)
    else:
        tensor = torch.as_tensor(np.asarray(control_points), dtype=torch.float32)
    if tensor.ndim != 4 or tensor.shape[-1] != 3:
        raise ValueError(f"control_points must have shape (n_i, n_j, n_k, 3), got {tuple(tensor.shape)}")
    for axis, size in enumerate(tensor.shape[:-1]):
        if size < 4:
            raise ValueError(f"Each control-point axis must have at least 4 elements; axis {axis} got {size}")
    return tensor.contiguous()


def _parameter_range(value) -> _ParameterRange:
    """Ints become floats so ``_ParameterRange`` always receives floats (spatial.py:2732-2751)."""
    if isinstance(value, (int, float)):
        value = float(value)
    elif isinstance(value, tuple) and all(isinstance(v, (int, float)) for v in value):
        value = tuple(float(v) for v in value)
    return _ParameterRange(value)


def _validate_isotropic(value, isotropic: bool) -> None:
    """``isotropic=True`` takes one value or one range, not per-axis values (spatial.py:2674-2683)."""
    if isotropic and not isinstance(value, Distribution) and isinstance(value, tuple) and len(value) in (3, 6):
        raise ValueError("If isotropic=True, scales must be a single value or a 2-value range")


def _parse_center(center: str) -> str:
    if center not in ("image", "origin"):
        raise ValueError(f'center must be "image" or "origin", got "{center}"')
    return center


def _parse_locked_borders(value: int) -> int:
    if value not in (0, 1, 2):
        raise ValueError(f"locked_borders must be 0, 1, or 2, got {value}")
    return value


def _positive_range(value) -> _ParameterRange:
    parsed = _parameter_range(value)
    if parsed._distribution is None and any(lo <= 0 or hi <= 0 for lo, hi in parsed._ranges):
        raise ValueError(f"Scale factors must be strictly positive, got {value}")
    return parsed


def _nonnegative_range(value) -> _ParameterRange:
    parsed = _parameter_range(value)
    if parsed._distribution is None and any(lo < 0 or hi < 0 for lo, hi in parsed._ranges):
        raise ValueError(f"Value must be non-negative, got {value}")
    return parsed


# the reference's names for the two range validators (spatial.py:2661-2672, 2621-2632)
_to_positive_range = _positive_range
_to_nonnegative_parameter_range = _nonnegative_range
