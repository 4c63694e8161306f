"""The reference's hot path restated with stock ``torch.nn.functional`` ops on a device tensor.

TEST / BENCH INFRASTRUCTURE.  The reference cannot travel to the GPU box, so this file
re-states — op for op — what its transforms dispatch to ATen when the subject lives on a
CUDA device: per resampling one ``(B, I, J, K, 3)`` grid (``cat`` + ``mm`` +
``interpolate`` + normalise), two ``grid_sample`` calls (data and the ones-mask) and a
``where``; BiasField = ``interpolate`` + ``exp`` + ``mul``; Blur = three ``pad`` + ``conv3d``;
Noise = ``randn_like`` + ``add``.  ``bench.py --aten-baseline`` times it as the honest
"before" of the HIP kernels (stock hipified ATen on the same MI355X); the parity tests use
``oracle/`` and the golden fixtures instead, this file makes no numerical claims.

Reference lines: ``transforms/spatial/spatial.py:1504-1648, 1695-1731, 2171-2189``;
``intensity/bias_field.py:296-341``; ``intensity/blur.py:157-204``; ``intensity/noise.py:166-178``.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import Tensor


def _grid(shape, mapping: Tensor, control_points: Tensor | None, device) -> Tensor:
    """Normalised sampling grid ``(I, J, K, 3)`` for one element (x ≡ i ≡ W, spatial.py:1627-1648)."""
    i, j, k = shape
    axes = [torch.arange(n, dtype=torch.float32, device=device) for n in (i, j, k)]
    coords = torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=-1)  # (I, J, K, 3)
    if control_points is not None:  # trilinear up-sampling of the coarse displacement field
        field = F.interpolate(control_points.permute(3, 0, 1, 2)[None], size=(i, j, k), mode="trilinear", align_corners=True)
        displacement = field[0].permute(1, 2, 3, 0)
    else:
        displacement = None
    homogeneous = torch.cat([coords, torch.ones_like(coords[..., :1])], dim=-1).reshape(-1, 4)
    voxels = (homogeneous @ mapping.T).reshape(i, j, k, 3)  # (3, 4) mapping
    if displacement is not None:
        voxels = voxels + displacement
    sizes = torch.tensor([max(i - 1, 1), max(j - 1, 1), max(k - 1, 1)], dtype=torch.float32, device=device)
    return 2 * voxels / sizes - 1


def resample(data: Tensor, mappings: Tensor, control_points: Tensor | None, fill: Tensor | None) -> Tensor:
    """Per-element grids (the reference's per-instance path stacks them, spatial.py:1881-1918)."""
    b, c, i, j, k = data.shape
    grids = torch.stack(
        [_grid((i, j, k), mappings[n], None if control_points is None else control_points[n], data.device) for n in range(b)]
    )
    # grid_sample wants (x, y, z) ≡ (W, H, D); TorchIO maps i ≡ x by permuting the volume instead
    volume = data.permute(0, 1, 4, 3, 2)
    out = F.grid_sample(volume, grids.permute(0, 3, 2, 1, 4), mode="bilinear", padding_mode="zeros", align_corners=True)
    if fill is not None:
        mask = F.grid_sample(torch.ones_like(volume), grids.permute(0, 3, 2, 1, 4), mode="bilinear", padding_mode="zeros", align_corners=True)
        out = torch.where(mask > 0.5, out, fill.view(1, -1, 1, 1, 1))
    return out.permute(0, 1, 4, 3, 2)


def bias_field(data: Tensor, coarse: Tensor) -> Tensor:
    field = F.interpolate(coarse, size=data.shape[2:], mode="trilinear", align_corners=True)
    return data * torch.exp(field)


def blur(data: Tensor, sigmas: Tensor) -> Tensor:
    """Per-element sigmas ``(B, 3)``: grouped conv per axis with replicate padding (blur.py:206-252)."""
    b, c = data.shape[:2]
    out = data.reshape(1, b * c, *data.shape[2:])
    for axis in range(3):
        radius = max(int(math.ceil(3 * float(sigmas[:, axis].max()))), 1)
        offsets = torch.arange(-radius, radius + 1, dtype=torch.float32, device=data.device)
        kernels = torch.exp(-0.5 * (offsets[None, :] / sigmas[:, axis : axis + 1]) ** 2)
        kernels = kernels / kernels.sum(dim=1, keepdim=True)
        weight = kernels.repeat_interleave(c, dim=0)  # (B*C, taps)
        shape = [b * c, 1, 1, 1, 1]
        shape[2 + axis] = 2 * radius + 1
        pad = [0, 0, 0, 0, 0, 0]
        pad[2 * (2 - axis)] = pad[2 * (2 - axis) + 1] = radius
        out = F.conv3d(F.pad(out, pad, mode="replicate"), weight.reshape(shape), groups=b * c)
    return out.reshape(data.shape)


def noise(data: Tensor, std: Tensor) -> Tensor:
    return data + std.view(-1, 1, 1, 1, 1) * torch.randn_like(data)


def compose_step(data: Tensor, rng: torch.Generator) -> Tensor:
    """One pass of Compose[Affine, ElasticDeformation, BiasField, Blur, Noise] with random per-element parameters."""
    b, c, i, j, k = data.shape
    device = data.device

    def rand(*shape, lo=0.0, hi=1.0):
        return torch.rand(*shape, generator=rng) * (hi - lo) + lo

    mappings = torch.eye(3, 4).repeat(b, 1, 1)
    mappings[:, :, :3] += 0.08 * (rand(b, 3, 3) - 0.5)
    mappings[:, :, 3] = rand(b, 3, lo=-5, hi=5)
    fill = data[0].amin(dim=(1, 2, 3))
    out = resample(data, mappings.to(device), None, fill)
    control = rand(b, 7, 7, 7, 3, lo=-7.5, hi=7.5)
    control[:, :2] = control[:, -2:] = 0
    control[:, :, :2] = control[:, :, -2:] = 0
    control[:, :, :, :2] = control[:, :, :, -2:] = 0
    fill = out[0].amin(dim=(1, 2, 3))
    out = resample(out, torch.eye(3, 4).repeat(b, 1, 1).to(device), control.to(device), fill)
    out = bias_field(out, (0.5 * torch.randn(b, c, 6, 6, 6, generator=rng)).to(device))
    out = blur(out, rand(b, 3, lo=0.5, hi=2.0).to(device))
    return noise(out, torch.full((b,), 0.25, device=device))
