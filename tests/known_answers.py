"""Known-answer vectors of the un-vendored ATen kernels, as pinned empirically in SURVEY.md §8(c).

``grid_sample(align_corners=True, padding_mode="zeros")`` on a length-5 axis holding ``[0, 1, 2, 3, 4]``,
sampled at the voxel coordinates below: nearest rounds half to even (``-0.5 -> -0 -> index 0``,
``S - 0.5 -> S - 1``), bilinear zero-pads, and the mask (sampling ones) is exactly 0.5 half a voxel
outside, which FAILS ``mask > 0.5`` and gets the fill value.
"""
from __future__ import annotations

import torch

COORDS = [0.5, 1.5, 2.5, 3.5, -0.5, 4.5, 4.4999, -0.49]
NEAREST = [0.0, 2.0, 2.0, 4.0, 0.0, 4.0, 4.0, 0.0]
BILINEAR = [0.5, 1.5, 2.5, 3.5, 0.0, 2.0, 2.0004, 0.0]
MASK = [1.0, 1.0, 1.0, 1.0, 0.5, 0.5, 0.5001, 0.51]
FILL = -7.0


def run(engine, device, interp: str, fill: bool):
    """One element per coordinate: a (1, 1, 5) axis sampled at a single output voxel shifted by the coordinate."""
    n = len(COORDS)
    data = torch.arange(5, dtype=torch.float32).view(1, 1, 1, 1, 5).repeat(n, 1, 1, 1, 1).to(device)
    mapping = torch.eye(3, 4).repeat(n, 1, 1)
    mapping[:, 2, 3] = torch.tensor(COORDS)
    out = engine.resample3d(
        [data], out_shape=(1, 1, 1), mapping=mapping.to(device), control_points=None, in_spacing=(1, 1, 1), out_spacing=(1, 1, 1),
        affine_first=True, interps=[interp], fills=[torch.tensor([FILL]).to(device) if fill else None],
    )[0]
    return out.reshape(n).cpu()


def check(engine, device) -> None:
    assert run(engine, device, "nearest", fill=False).tolist() == NEAREST
    torch.testing.assert_close(run(engine, device, "linear", fill=False), torch.tensor(BILINEAR), rtol=0, atol=2e-5)
    # with a fill value: kept where the in-bounds weight mask exceeds 0.5, i.e. NOT at exactly half a voxel outside
    expected = [value if mask > 0.5 else FILL for value, mask in zip(BILINEAR, MASK, strict=True)]
    torch.testing.assert_close(run(engine, device, "linear", fill=True), torch.tensor(expected), rtol=0, atol=2e-5)
    expected_nearest = [value if mask > 0.5 else FILL for value, mask in zip(NEAREST, MASK, strict=True)]
    assert run(engine, device, "nearest", fill=True).tolist() == expected_nearest
