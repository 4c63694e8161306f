"""``Motion`` on the HIP engine (mirror of reference ``transforms/intensity/motion.py``).

The reference corrupts k-space segment by segment: ``num_transforms`` rigidly moved copies of
the image (``affine_grid`` + ``grid_sample``), a 3-D FFT of each, slabs of planes along the first
spatial axis swapped into the still image's spectrum, one inverse FFT (motion.py:334-372) —
``num_transforms + 2`` complex 3-D FFTs per image.  Here the moved copies come from the fused
resampler (``tio_resample3d`` with the voxel-space matrix of the same rigid transform) and the
k-space surgery is ``tio_kspace_segment_mix``: because only the first axis is ever masked the
whole composite is one real float32 GEMM along that axis, which runs on the matrix cores with
no FFT and no complex volume in HBM.  Same constructor, sampling order, parameter dictionary
(``{"transforms": [{"degrees": ..., "translation": ...}, ...]}``), gating and errors.
"""
from __future__ import annotations

from typing import Any

import torch
from torch import Tensor

from .. import ops
from ..data.batch import SubjectsBatch
from .parameter_range import to_range
from .transform import IntensityTransform

_IDENTITY_TRANSFORM = {"degrees": (0.0, 0.0, 0.0), "translation": (0.0, 0.0, 0.0)}


class Motion(IntensityTransform):
    """Simulate MRI motion artifacts, Shaw et al. 2019 (motion.py:32-140)."""

    def __init__(self, *, degrees=10.0, translation=10.0, num_transforms: int = 2, **kwargs: Any) -> None:
        super().__init__(**kwargs)
        self.degrees = to_range(degrees)
        self.translation = to_range(translation)
        if not isinstance(num_transforms, int) or num_transforms < 1:
            raise ValueError(f"num_transforms must be a positive int, got {num_transforms}")
        self.num_transforms = num_transforms

    def _sample_transforms(self) -> list[dict[str, tuple[float, float, float]]]:
        return [{"degrees": self.degrees.sample(), "translation": self.translation.sample()} for _ in range(self.num_transforms)]

    @property
    def supports_per_instance_params(self) -> bool:
        return True

    @property
    def supports_per_instance_p(self) -> bool:
        return True

    def make_params(self, batch: SubjectsBatch) -> dict[str, Any]:
        n = self._resolve_n(batch)
        if n is None:
            return {"transforms": self._sample_transforms()}
        keep = self._keep_mask(batch, n)
        transforms_list: list[Any] = []
        for index in range(n):
            if keep is not None and not keep[index]:
                transforms_list.append([])  # gated out: nothing drawn
                continue
            transforms_list.append(self._sample_transforms())
        params = {"transforms": transforms_list}
        self._tag_batched(params, batch, n, keep, ["transforms"])
        return params

    def apply_transform(self, batch: SubjectsBatch, params: dict[str, Any]) -> SubjectsBatch:
        per_instance = self._is_per_instance_params(params)
        for img_batch in self._get_images(batch).values():
            if per_instance:
                img_batch.data = _apply_motion_per_instance(img_batch.data, params["transforms"])
            else:
                img_batch.data = _apply_motion(img_batch.data, params["transforms"])
        return batch


def _apply_motion(data: Tensor, motion_transforms) -> Tensor:
    """Shared parameters: every element moves the same way (motion.py:143-166)."""
    if not motion_transforms:
        return data
    batch_size = data.shape[0]
    segments = [
        (torch.tensor([t["degrees"]] * batch_size, dtype=torch.float32), torch.tensor([t["translation"]] * batch_size, dtype=torch.float32))
        for t in motion_transforms
    ]
    return _apply_motion_segments(data, segments, active=None)


def _apply_motion_per_instance(data: Tensor, motion_transforms) -> Tensor:
    """One transform list per element; empty lists mark gated-out elements (motion.py:169-196)."""
    if len(motion_transforms) != data.shape[0]:
        raise ValueError(f"Expected {data.shape[0]} motion parameter lists, got {len(motion_transforms)}")
    active = [bool(transforms) for transforms in motion_transforms]
    if not any(active):
        return data
    lengths = {len(transforms) for transforms in motion_transforms} - {0}
    if len(lengths) > 1:
        raise ValueError(f"Expected uniform motion transform counts, got {sorted(lengths)}")
    segments = []
    for index in range(max(lengths)):
        chosen = [transforms[index] if transforms else _IDENTITY_TRANSFORM for transforms in motion_transforms]
        segments.append((
            torch.tensor([t["degrees"] for t in chosen], dtype=torch.float32),
            torch.tensor([t["translation"] for t in chosen], dtype=torch.float32),
        ))
    transformed = _apply_motion_segments(data, segments, active=None if all(active) else torch.tensor(active))
    if all(active):
        return transformed
    inactive = ops.h2d(torch.tensor([i for i, a in enumerate(active) if not a]), data.device)
    transformed.index_copy_(0, inactive, data.index_select(0, inactive))  # rows restored exactly from the input
    return transformed


def _segment_bounds(num_segments: int, first_spatial_size: int) -> list[int]:
    """Plane ranges of the k-space segments: equal slabs, the last one takes the remainder (motion.py:375-390)."""
    segment_size = first_spatial_size // num_segments
    if segment_size == 0:
        raise ValueError(
            f"Cannot split {first_spatial_size} k-space slices into {num_segments} motion segments; reduce num_transforms or use a"
            " larger image along the first spatial axis."
        )
    return [index * segment_size for index in range(num_segments)] + [first_spatial_size]


def _apply_motion_segments(data: Tensor, segment_parameters, *, active: Tensor | None) -> Tensor:
    """``ifftn(fftn(still) with the segments' planes taken from fftn(moved_s)).real`` (motion.py:334-372)."""
    engine = ops.engine()
    shape = tuple(data.shape[-3:])
    bounds = _segment_bounds(len(segment_parameters) + 1, shape[0])
    still = data if data.dtype == torch.float32 else data.float()
    flags = None if active is None else ops.h2d(active.to(torch.uint8), data.device)
    skip = None if active is None else ops.h2d((~active.bool()).to(torch.uint8), data.device)  # inactive rows: plain copies
    images = [still]
    # every event's matrices in one host computation and one upload: (events * B, 3, 4)
    batch_size = data.shape[0]
    mappings = ops.h2d(
        _rigid_voxel_mappings(torch.cat([d for d, _ in segment_parameters]), torch.cat([t for _, t in segment_parameters]), shape),
        data.device,
    )
    for index in range(len(segment_parameters)):
        moved = engine.resample3d(
            [still], out_shape=shape, mapping=mappings[index * batch_size : (index + 1) * batch_size], control_points=None,
            in_spacing=(1.0, 1.0, 1.0), out_spacing=(1.0, 1.0, 1.0), affine_first=True, interps=["linear"], fills=[None], passthrough=skip,
        )[0]
        images.append(moved)
    return engine.kspace_segment_mix(images, bounds, data.dtype, active=flags)


def _rigid_voxel_mappings(degrees: Tensor, translation: Tensor, shape) -> Tensor:
    """``(B, 3, 4)`` float32 output-voxel → input-voxel matrices of ``_apply_rigid_transform`` (motion.py:393-480).

    The reference builds ``theta = [Rz Ry Rx | t / (shape / 2)]`` in float32, lets ``affine_grid``
    evaluate it on the normalised grid ``n = 2 p / (S - 1) - 1`` (axes in x, y, z = K, J, I order,
    one-voxel axes at n = 0) and ``grid_sample`` un-normalise with ``(g + 1) / 2 * (S - 1)``.  The
    same chain composed per element in float64: ``D (R (D^-1 p - 1) + t + 1)`` with
    ``D = diag((S - 1) / 2)``, then reordered to the engine's (i, j, k) convention.
    """
    theta_rotation = _rotation_matrices(degrees).double()  # the reference's float32 matrices, widened exactly
    sizes_xyz = torch.tensor([shape[2], shape[1], shape[0]], dtype=torch.float64)
    theta_translation = (translation / (torch.tensor([shape[0], shape[1], shape[2]], dtype=torch.float32) / 2)).double()
    half = (sizes_xyz - 1) / 2  # D; zero for one-voxel axes
    inverse_half = torch.where(half > 0, 1 / half.clamp_min(0.5), torch.zeros_like(half))
    ones = (half > 0).double()  # the "-1" of the normalisation only exists on axes longer than one voxel
    linear = half[None, :, None] * theta_rotation * inverse_half[None, None, :]
    offset = half[None, :] * (theta_translation + 1 - theta_rotation @ ones)
    xyz = torch.cat([linear, offset[:, :, None]], dim=2)  # (B, 3, 4) in (x, y, z) = (k, j, i) order
    ijk = xyz.flip(1)
    ijk = torch.cat([ijk[:, :, :3].flip(2), ijk[:, :, 3:]], dim=2)
    return ijk.float().contiguous()


def _rotation_matrices(degrees: Tensor) -> Tensor:
    """``Rz @ Ry @ Rx`` from Euler angles in degrees, float32 like the reference (motion.py:483-561)."""
    radians = torch.deg2rad(degrees)
    cos, sin = torch.cos(radians), torch.sin(radians)
    (cx, cy, cz), (sx, sy, sz) = cos.unbind(dim=-1), sin.unbind(dim=-1)
    zero, one = torch.zeros_like(cx), torch.ones_like(cx)
    r_x = torch.stack([one, zero, zero, zero, cx, -sx, zero, sx, cx], dim=1).reshape(-1, 3, 3)
    r_y = torch.stack([cy, zero, sy, zero, one, zero, -sy, zero, cy], dim=1).reshape(-1, 3, 3)
    r_z = torch.stack([cz, -sz, zero, sz, cz, zero, zero, zero, one], dim=1).reshape(-1, 3, 3)
    return r_z @ r_y @ r_x
