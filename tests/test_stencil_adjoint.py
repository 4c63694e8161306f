"""Backward of the replicate-padded separable correlation (`tio_separable_conv3d_adjoint`, ABI 16; round 6: the last ATen compute
left in the package is now a kernel).  The checker is the reference's own formulation — `F.pad(mode="replicate")` + grouped `conv3d`
per axis (blur.py:157-252) differentiated by autograd — restated HERE, in the test, never in the product; the oracle's adjoint is the
literal scatter transpose, the HIP kernel a gather with the clamped taps folded onto the border voxels.
CPU tests: the oracle against autograd (dot-product identity, per-element taps, inactive axes, extents below the radius, skipped rows);
the `gpu` test: the HIP kernel against both."""
from __future__ import annotations

import pytest
import torch
import torch.nn.functional as F


def _aten_backward(data: torch.Tensor, taps: torch.Tensor, radius, skip, grad: torch.Tensor) -> torch.Tensor:
    batch, channels = data.shape[:2]
    with torch.enable_grad():
        leaf = data.float().clone().requires_grad_(True)
        work = leaf
        for axis in range(3):
            r = int(radius[axis])
            if r <= 0:
                continue
            pad = [0, 0, 0, 0, 0, 0]
            pad[2 * (2 - axis)] = pad[2 * (2 - axis) + 1] = r
            extent = work.shape[2 + axis]
            padded = work
            for _ in range(r):  # (F.pad's replicate mode refuses pads beyond the extent: one voxel at a time is the same clamp)
                step = [0, 0, 0, 0, 0, 0]
                step[2 * (2 - axis)] = step[2 * (2 - axis) + 1] = 1
                padded = F.pad(padded, step, mode="replicate")
            assert padded.shape[2 + axis] == extent + 2 * r
            kernel = taps[:, axis, : 2 * r + 1].to(leaf.device, torch.float32)
            shape = [1, 1, 1]
            shape[axis] = 2 * r + 1
            if kernel.shape[0] == 1:
                work = F.conv3d(padded, kernel.reshape(1, 1, *shape).expand(channels, 1, *shape), groups=channels)
            else:
                weight = kernel.reshape(batch, 1, 1, *shape).expand(batch, channels, 1, *shape).reshape(batch * channels, 1, *shape)
                work = F.conv3d(padded.reshape(1, batch * channels, *padded.shape[2:]), weight, groups=batch * channels).reshape(batch, channels, *work.shape[2:])
        if skip is not None:
            work = torch.where(skip.to(leaf.device).bool().reshape(-1, 1, 1, 1, 1), leaf, work)
        (result,) = torch.autograd.grad(work, leaf, grad.float())
    return result


CASES = [
    # shape (B, C, I, J, K), radius, per-element taps, skipped rows
    ((2, 2, 9, 8, 11), (2, 1, 3), False, None),
    ((3, 1, 7, 10, 6), (3, 3, 2), True, [0, 1, 0]),
    ((1, 2, 12, 5, 9), (0, 2, 0), False, None),  # one active axis: no scratch
    ((2, 1, 3, 2, 1), (4, 3, 2), True, None),  # extents below the radius: both borders collect clamped taps, K = 1 collects all
    ((1, 1, 6, 6, 6), (0, 0, 0), False, None),  # identity
    ((2, 1, 16, 12, 70), (8, 5, 8), True, [1, 0]),
]


def _problem(case, device):
    shape, radius, per_element, skip = case
    g = torch.Generator().manual_seed(sum(shape) + sum(radius))
    stride = 2 * max(max(radius), 1) + 1
    taps = torch.rand(shape[0] if per_element else 1, 3, stride, generator=g) + 0.05
    taps = taps / taps.sum(dim=2, keepdim=True)
    data = torch.rand(shape, generator=g)
    grad = torch.randn(shape, generator=g)
    flags = None if skip is None else torch.tensor(skip, dtype=torch.uint8)
    return data.to(device), taps.to(device), grad.to(device), None if flags is None else flags.to(device), radius


def _check(engine, device, case, other=None):
    data, taps, grad, skip, radius = _problem(case, device)
    ours = engine.separable_conv3d_adjoint(grad, taps, radius, skip=skip)
    expected = _aten_backward(data.cpu(), taps.cpu(), radius, None if skip is None else skip.cpu(), grad.cpu())
    scale = max(float(expected.abs().max()), 1e-6)
    assert float((ours.cpu() - expected).abs().max()) <= 2e-6 * scale * (2 * max(radius) + 1), case
    # <A x, g> == <x, A^T g> against the engine's own forward
    forward = engine.separable_conv3d(data, taps, list(radius), skip=skip)
    lhs = float((forward.double() * grad.double()).sum())
    rhs = float((data.double() * ours.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0), (case, lhs, rhs)
    # and autograd through the op uses it
    leaf = data.clone().requires_grad_(True)
    (engine.separable_conv3d(leaf, taps, list(radius), skip=skip) * grad).sum().backward()
    assert torch.equal(leaf.grad, ours)
    if other is not None:
        theirs = other.separable_conv3d_adjoint(grad.cpu(), taps.cpu(), radius, skip=None if skip is None else skip.cpu())
        assert float((ours.cpu() - theirs).abs().max()) <= 2e-6 * scale * (2 * max(radius) + 1), case


@pytest.mark.parametrize("case", CASES)
def test_oracle_adjoint_is_autograds_backward_of_the_padded_correlation(oracle, case):
    _check(oracle, "cpu", case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES + [((2, 2, 96, 80, 128), (6, 8, 7), True, None)])
def test_hip_adjoint_against_autograd_and_the_oracle(hip, oracle, case):
    _check(hip, "cuda", case, other=oracle)


def test_the_product_holds_no_aten_convolution():
    """`torchio_amd/ops.py` used to differentiate the stencil through `F.pad` + `F.conv3d` (missing #7)."""
    import inspect

    from torchio_amd import ops

    source = inspect.getsource(ops)
    assert "conv3d(" not in source.replace("separable_conv3d(", "") and "F.pad(" not in source.replace("``F.pad", "")
