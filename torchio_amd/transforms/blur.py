"""``Blur`` on the HIP engine (mirror of reference ``transforms/intensity/blur.py``).

The host builds the 1-D Gaussian taps with the same float32 torch expressions as
the reference (blur.py:179-183 shared, blur.py:292-328 per element: zero-extended
to the largest radius, delta for sigma = 0, normalised); the replicate-padded
separable cross-correlation itself (blur.py:185-203, 234-247) is one
``tio_separable_conv3d`` call instead of three ``F.pad`` + ``F.conv3d`` pairs.
"""
from __future__ import annotations

from typing import Any

import math

import numpy as np
import torch
from torch import Tensor

from .. import ops
from ..data import _pending
from ..data.batch import SubjectsBatch
from .parameter_range import to_nonneg_range
from .transform import IntensityTransform


class Blur(IntensityTransform):
    """Gaussian blur with per-axis standard deviations in mm (blur.py:19-90)."""

    def __init__(self, *, std=0.0, **kwargs: Any) -> None:
        super().__init__(**kwargs)
        self.std = to_nonneg_range(std)
        self._warn_if_noop(is_noop=self.std.is_constant(0.0), hint="std=(0, 2)")

    @property
    def supports_per_instance_params(self) -> bool:
        return True

    @property
    def supports_per_instance_p(self) -> bool:
        return True

    @property
    def draws_ahead(self) -> bool:
        # parameters from the batch size alone; intensities change, geometry does not (a subclass that overrides either half speaks for itself)
        return type(self).make_params is Blur.make_params and type(self).apply_transform is Blur.apply_transform

    def make_params(self, batch: SubjectsBatch) -> dict[str, Any]:
        n = self._resolve_n(batch)
        if n is None:
            return {"std": self.std.sample()}
        keep = self._keep_mask(batch, n)
        std = self.std.sample(n)
        if keep is not None:
            std[~keep] = 0.0
        params = {"std": self._serialize_param(std)}
        self._tag_batched(params, batch, n, keep, ["std"])
        return params

    def apply_transform(self, batch: SubjectsBatch, params: dict[str, Any]) -> SubjectsBatch:
        per_instance = self._is_per_instance_params(params)
        for img_batch in self._get_images(batch).values():
            if per_instance:
                sigmas_mm = np.asarray(params["std"], dtype=np.float64)
                spacings = np.asarray([affine.spacing for affine in img_batch.affines], dtype=np.float64)
                sigmas = np.divide(sigmas_mm, spacings, out=np.zeros_like(sigmas_mm), where=spacings > 0)
            else:
                spacing = np.asarray(img_batch.affines[0].spacing, dtype=np.float64)
                sigmas = [s / sp if sp > 0 else 0.0 for s, sp in zip(params["std"], spacing, strict=True)]
            if not _defer_blur(img_batch, sigmas):
                img_batch.data = _gaussian_smooth(img_batch.data, sigmas)
        return batch


def _defer_blur(img_batch, sigmas) -> bool:
    """Queue the stencil on the batch instead of launching it (data/_pending.py) when the data qualifies.

    A ``BiasField`` queued just before and a ``Noise`` arriving just after are then folded into
    the stencil's passes; any other reader of ``img_batch.data`` launches what is queued.
    """
    if not hasattr(img_batch, "_flush"):  # a foreign container (reference_binding): launch right away
        return False
    raw = img_batch._data
    if not _pending.eligible(raw):
        return False
    sigmas = np.asarray(sigmas, dtype=np.float64)
    if np.all(sigmas <= 0):
        return False  # identity: leave it to the plain path (returns the data untouched)
    if sigmas.ndim == 2 and np.all(sigmas == sigmas[0]):
        sigmas = sigmas[0]
    taps, radius, skip = _stacked_gaussian_taps(sigmas if sigmas.ndim == 2 else sigmas[None], per_element=sigmas.ndim == 2)
    if skip is not None:
        return False  # rows restored bit-exactly: plain path
    queue = img_batch._pending
    if queue is not None and queue.blur is not None:  # two stencils in a row: finish the first
        img_batch._flush()
        queue = None
    if queue is None:
        queue = _pending.Pending()
    queue.blur = (taps, [int(r) for r in radius])  # host tensor: uploaded by the flush, with its neighbours' blocks
    img_batch._pending = queue
    return True


def _gaussian_smooth(data: Tensor, sigmas) -> Tensor:
    """Separable Gaussian smoothing of a ``(B, C, I, J, K)`` tensor (functional seam S2, blur.py:129-154).

    *sigmas* is per-axis (length 3, voxels) or per-element ``(B, 3)``; zero skips
    an axis.  Computed in float32, returned in the input dtype.
    """
    sigmas = np.asarray(sigmas, dtype=np.float64)
    if np.all(sigmas <= 0):
        return data
    if sigmas.ndim == 2 and np.all(sigmas == sigmas[0]):
        sigmas = sigmas[0]  # identical rows collapse to the shared kernel set (blur.py:150-153)
    taps, radius, skip = _stacked_gaussian_taps(sigmas if sigmas.ndim == 2 else sigmas[None], per_element=sigmas.ndim == 2)
    work = data if data.dtype in ops.FLOAT_DTYPES else data.float()
    skip_flags = None if skip is None else ops.h2d(torch.from_numpy(skip), data.device)
    result = ops.engine().separable_conv3d(work, ops.h2d(taps, data.device), radius, skip=skip_flags)
    if result.dtype != data.dtype:
        result = result.to(data.dtype)
        if skip is not None:  # untouched rows keep their exact integer values
            rows = ops.h2d(torch.from_numpy(skip.astype(bool)), data.device)
            result[rows] = data[rows]
    return result


_OFFSETS: dict[int, Tensor] = {}


_DISTANCES: dict[int, Tensor] = {}
_ZERO = torch.zeros((), dtype=torch.float32)


def _tap_distance(r: int) -> Tensor:
    """``|arange(2r + 1) - r|`` as int64 (read-only, cached per radius)."""
    distance = _DISTANCES.get(r)
    if distance is None:
        distance = _DISTANCES[r] = (torch.arange(2 * r + 1, dtype=torch.int64) - r).abs()
    return distance


def _tap_offsets(r: int) -> Tensor:
    """``arange(2r + 1) - r`` as float32 (read-only, cached per radius)."""
    offsets = _OFFSETS.get(r)
    if offsets is None:
        offsets = _OFFSETS[r] = torch.arange(2 * r + 1, dtype=torch.float32) - r
    return offsets


def _stacked_gaussian_taps(sigmas: np.ndarray, per_element: bool = False):
    """Normalised 1-D kernels for every (element, axis): ``(n, 3, stride)`` float32 taps.

    Returns ``(taps, radius[3], skip)``; ``radius[a]`` is the largest radius on axis
    ``a`` (0 = axis inactive for every element) and ``skip`` flags elements whose
    three sigmas are all <= 0 (restored bit-exactly, blur.py:249-251).
    """
    n = sigmas.shape[0]
    rows = sigmas.tolist()  # a handful of values: plain Python beats numpy's per-call overhead here
    radii_rows = [[max(math.ceil(3 * value), 1) if value > 0 else 0 for value in row] for row in rows]
    radius = [max(row[axis] for row in radii_rows) for axis in range(3)]
    stride = 2 * max(radius) + 1
    taps = torch.zeros(n, 3, stride, dtype=torch.float32)
    radii = positive = None
    if per_element:
        radii = np.asarray(radii_rows, dtype=np.int64)
        positive = radii > 0
        if bool(positive.all()):
            # every element blurs every axis (the usual per-instance draw): the Gaussian of all three axes from ONE chain of
            # tensor ops on the (n, 3, stride) block — exp(-0.5 (o / sigma)^2) is elementwise, so each value is the one the
            # per-axis expression gives — zeroed beyond each element's own radius; only the normalisation stays per axis, on
            # the (n, 2 r_axis + 1) slice the per-axis code sums (ATen's float32 row sum depends on the row's LENGTH: a row
            # padded with zeros to the common stride rounds differently; tests/test_host_logic.py holds the two forms equal)
            reach = max(radius)
            offsets = _tap_offsets(reach)
            sigma_block = torch.as_tensor(sigmas, dtype=torch.float32)[:, :, None]
            kernels = torch.exp(-0.5 * (offsets[None, None, :] / sigma_block) ** 2)
            kernels = torch.where(_tap_distance(reach)[None, None, :] <= torch.from_numpy(radii)[:, :, None], kernels, _ZERO)
            for axis in range(3):
                r = radius[axis]
                window = kernels[:, axis, reach - r : reach + r + 1]
                taps[:, axis, : 2 * r + 1] = window / window.sum(dim=1, keepdim=True)
            return taps, radius, None
    for axis in range(3):
        r = radius[axis]
        if r == 0:
            continue
        offsets = _tap_offsets(r)
        if not per_element:
            sigma = float(sigmas[0, axis])
            kernel = torch.exp(-0.5 * (offsets / sigma) ** 2)
            taps[0, axis, : 2 * r + 1] = kernel / kernel.sum()
            continue
        sigma_column = torch.as_tensor(sigmas[:, axis], dtype=torch.float32)[:, None]
        all_active = bool(positive[:, axis].all())
        # (the selects below are identities when every element blurs this axis / shares the radius: skipped then)
        safe = sigma_column if all_active else torch.where(sigma_column > 0, sigma_column, torch.ones_like(sigma_column))
        kernels = torch.exp(-0.5 * (offsets[None, :] / safe) ** 2)
        if int(radii[:, axis].min()) != r:
            radius_column = torch.as_tensor(radii[:, axis])[:, None]
            kernels = torch.where(offsets[None, :].abs() <= radius_column, kernels, torch.zeros_like(kernels))
        if not all_active:
            delta = torch.zeros_like(kernels)
            delta[:, r] = 1.0
            kernels = torch.where(sigma_column > 0, kernels, delta)
        taps[:, axis, : 2 * r + 1] = kernels / kernels.sum(dim=1, keepdim=True)
    skip = None
    if per_element:
        no_blur = np.all(sigmas <= 0, axis=1)
        if no_blur.any():
            skip = no_blur.astype(np.uint8)
    return taps, radius, skip
