"""``BiasField`` on the HIP engine (mirror of reference ``transforms/intensity/bias_field.py``).

SynthSeg recipe: a coarse ``N(0, std)`` tensor drawn from a seeded CPU generator
(exactly the reference's draw, bias_field.py:321-330 / 283-293 — 216 values at
256^3), then ONE kernel does trilinear upsampling (align_corners) + ``exp`` +
multiply (or divide for the inverse) instead of four full-volume passes.
"""
from __future__ import annotations

from typing import Any

import torch
from torch import Tensor

from .. import ops
from ..data import _pending
from ..data.batch import SubjectsBatch
from .parameter_range import to_nonneg_range
from .transform import IntensityTransform


class BiasField(IntensityTransform):
    """Smooth multiplicative intensity inhomogeneity (bias_field.py:22-146)."""

    def __init__(self, *, std=0.5, scale: float = 0.025, **kwargs: Any) -> None:
        super().__init__(**kwargs)
        self.std = to_nonneg_range(std)
        if scale <= 0 or scale > 1:
            raise ValueError(f"scale must be in (0, 1], got {scale}")
        self.scale = scale

    @property
    def supports_per_instance_params(self) -> bool:
        return True

    @property
    def supports_per_instance_p(self) -> bool:
        return True

    @property
    def draws_ahead(self) -> bool:
        # parameters from the batch size alone; intensities change, geometry does not (a subclass that overrides either half speaks for itself)
        return type(self).make_params is BiasField.make_params and type(self).apply_transform is BiasField.apply_transform

    def make_params(self, batch: SubjectsBatch) -> dict[str, Any]:
        n = self._resolve_n(batch)
        if n is None:  # draw order: std, then the seed (bias_field.py:71-72)
            std = self.std.sample_1d()
            seed = int(torch.randint(0, 2**31, (1,)).item())
            return {"std": std, "seed": seed, "scale": self.scale}
        keep = self._keep_mask(batch, n)
        std = self._mask_identity(self.std.sample_1d(n), keep, identity=0.0)
        # (one call for the n seeds: the CPU generator hands out the same 32-bit draws, and stands where it would stand, as
        # with the reference's n calls of one — tests/test_host_logic.py holds the two forms against each other)
        seeds = torch.randint(0, 2**31, (n,)).tolist()
        params = {"std": self._serialize_param(std), "seed": seeds, "scale": self.scale}
        self._tag_batched(params, batch, n, keep, ["std", "seed"])
        return params

    def apply_transform(self, batch: SubjectsBatch, params: dict[str, Any]) -> SubjectsBatch:
        _apply_to_images(self, batch, params["std"], params["seed"], params["scale"], divide=False)
        return batch

    @property
    def invertible(self) -> bool:
        return True

    def inverse(self, params: dict[str, Any]) -> "_BiasFieldInverse":
        return _BiasFieldInverse(std=params["std"], seed=params["seed"], scale=params["scale"], copy=False)


class _BiasFieldInverse(IntensityTransform):
    """Divide by the regenerated field (bias_field.py:149-198)."""

    def __init__(self, *, std, seed, scale: float, **kwargs: Any) -> None:
        super().__init__(**kwargs)
        self._std, self._seed, self._scale = std, seed, scale

    def apply_transform(self, batch: SubjectsBatch, params: dict[str, Any]) -> SubjectsBatch:
        _apply_to_images(self, batch, self._std, self._seed, self._scale, divide=True)
        return batch


def _apply_to_images(transform, batch: SubjectsBatch, std, seed, scale: float, *, divide: bool) -> None:
    per_element = isinstance(std, list)
    if not per_element and std == 0:
        return
    for img_batch in transform._get_images(batch).values():
        if not divide and _defer_bias(img_batch, std, seed, scale, per_element):
            continue
        if per_element:
            img_batch.data = _apply_bias_per_element(img_batch.data, std, seed, scale, divide=divide)
        else:
            data = img_batch.data
            coarse = _sample_coarse_field(data.shape, std=std, scale=scale, seed=seed)
            # `data * field` promotes with the float32 field (bias_field.py:130,196)
            work = data if data.dtype in (torch.float32, torch.float64) else data.float()
            img_batch.data = ops.engine().bias_field_apply(work, ops.h2d(coarse, data.device), divide=divide)


def _defer_bias(img_batch, std, seed, scale: float, per_element: bool) -> bool:
    """Queue the multiply on the batch (data/_pending.py): a ``Blur`` that follows folds it into its loads."""
    if not hasattr(img_batch, "_flush"):  # a foreign container (reference_binding): launch right away
        return False
    data = img_batch.data  # finished values of whatever came before
    if not _pending.eligible(data):
        return False
    if per_element:
        if any(s == 0 for s in std):
            return False  # identity rows are restored bit-exactly: plain path
        small = _coarse_shape(data.shape[2:], scale)
        coarse = torch.empty((len(std), data.shape[1], *small), dtype=torch.float32)
        generator = torch.Generator(device="cpu")  # (re-seeded per element: the state a fresh generator would have)
        for index, (std_b, seed_b) in enumerate(zip(std, seed, strict=True)):
            generator.manual_seed(seed_b)
            torch.normal(mean=0.0, std=std_b, size=(1, data.shape[1], *small), generator=generator, out=coarse[index : index + 1])
    else:
        coarse = _sample_coarse_field(data.shape, std=std, scale=scale, seed=seed)
    img_batch._pending = _pending.Pending(bias_coarse=coarse)  # host tensor: uploaded by the flush, with its neighbours' blocks
    return True


def _coarse_shape(spatial, scale: float) -> list[int]:
    return [max(round(s * scale), 4) for s in spatial]  # Python banker's round: 256 -> 6 (bias_field.py:319)


def _sample_coarse_field(shape, *, std: float, scale: float, seed: int) -> Tensor:
    """``(B, C, si, sj, sk)`` coarse field from ONE seeded CPU generator (bias_field.py:316-330)."""
    generator = torch.Generator(device="cpu")
    generator.manual_seed(seed)
    return torch.normal(mean=0.0, std=std, size=(shape[0], shape[1], *_coarse_shape(shape[2:], scale)), generator=generator)


def _apply_bias_per_element(data: Tensor, std_per_element, seed_per_element, scale: float, *, divide: bool) -> Tensor:
    """Each element gets the field of its own ``(std, seed)`` (functional seam S3, bias_field.py:201-293)."""
    identity_rows = [std == 0 for std in std_per_element]
    if all(identity_rows):
        return data
    small = _coarse_shape(data.shape[2:], scale)
    fields = []
    for std, seed in zip(std_per_element, seed_per_element, strict=True):
        generator = torch.Generator(device="cpu")
        generator.manual_seed(seed)
        fields.append(torch.normal(mean=0.0, std=std, size=(1, data.shape[1], *small), generator=generator))
    coarse = ops.h2d(torch.cat(fields, dim=0), data.device)
    skip = ops.h2d(torch.tensor(identity_rows, dtype=torch.uint8), data.device) if any(identity_rows) else None
    work = data if data.dtype in ops.FLOAT_DTYPES else data.float()
    result = ops.engine().bias_field_apply(work, coarse, divide=divide, skip=skip)
    if result.dtype != data.dtype:  # `.to(data.dtype)` + exact restore of identity rows (bias_field.py:245-253)
        result = result.to(data.dtype)
        if skip is not None:
            rows = skip.bool()
            result[rows] = data[rows]
    return result
