"""``Gamma`` on the HIP engine (mirror of reference ``transforms/intensity/gamma.py``).

``sign(x) * |x| ** exp(log_gamma)`` in one pass instead of four elementwise passes
(gamma.py:90).
"""
from __future__ import annotations

import math
from typing import Any

import torch
from torch import Tensor

from .. import ops
from ..data.batch import SubjectsBatch
from .parameter_range import to_range
from .transform import IntensityTransform


class Gamma(IntensityTransform):
    """Random gamma correction with ``gamma = exp(log_gamma)`` (gamma.py:17-100)."""

    def __init__(self, *, log_gamma=0.0, **kwargs: Any) -> None:
        super().__init__(**kwargs)
        self.log_gamma = to_range(log_gamma)
        self._warn_if_noop(is_noop=self.log_gamma.is_constant(0.0), hint="log_gamma=(-0.3, 0.3)")

    @property
    def supports_per_instance_params(self) -> bool:
        return True

    @property
    def supports_per_instance_p(self) -> bool:
        return True

    @property
    def draws_ahead(self) -> bool:
        # parameters from the batch size alone; intensities change, geometry does not (a subclass that overrides either half speaks for itself)
        return type(self).make_params is Gamma.make_params and type(self).apply_transform is Gamma.apply_transform

    def make_params(self, batch: SubjectsBatch) -> dict[str, Any]:
        n = self._resolve_n(batch)
        keep = self._keep_mask(batch, n)
        log_gamma = self._mask_identity(self.log_gamma.sample_1d(n), keep, identity=0.0)
        params = {"log_gamma": self._serialize_param(log_gamma)}
        self._tag_batched(params, batch, n, keep, ["log_gamma"])
        return params

    def apply_transform(self, batch: SubjectsBatch, params: dict[str, Any]) -> SubjectsBatch:
        for img_batch in self._get_images(batch).values():
            img_batch.data = _gamma_pow(img_batch.data, params["log_gamma"])
        return batch

    @property
    def invertible(self) -> bool:
        return True

    def inverse(self, params: dict[str, Any]) -> "_GammaInverse":
        return _GammaInverse(log_gamma=params["log_gamma"], copy=False)


class _GammaInverse(IntensityTransform):
    """Apply ``1 / gamma`` (gamma.py:123-142)."""

    def __init__(self, *, log_gamma, **kwargs: Any) -> None:
        super().__init__(**kwargs)
        self._log_gamma = log_gamma

    def apply_transform(self, batch: SubjectsBatch, params: dict[str, Any]) -> SubjectsBatch:
        negated = [-v for v in self._log_gamma] if isinstance(self._log_gamma, list) else -self._log_gamma
        for img_batch in self._get_images(batch).values():
            img_batch.data = _gamma_pow(img_batch.data, negated)
        return batch


def _gamma_pow(data: Tensor, log_gamma) -> Tensor:
    """``sign(x) |x|^gamma`` with the reference's dtype promotion (functional seam S5, gamma.py:88-120)."""
    if isinstance(log_gamma, list):
        gamma: Any = ops.h2d(torch.exp(torch.tensor(log_gamma, dtype=torch.float32)), data.device)
        # a (B,1,1,1,1) float32 exponent tensor promotes half / integer data to float32
        work = data if data.dtype in (torch.float32, torch.float64) else data.float()
    else:
        gamma = math.exp(log_gamma)
        work = data if data.dtype in ops.FLOAT_DTYPES else data.float()
    return ops.engine().gamma_pow(work, gamma)
