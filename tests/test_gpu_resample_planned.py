"""The planned-brick roads of `tio_resample3d` (resample_fast.hpp): large launches only by default, forced here.

* FAST: `plan_bricks_kernel` + `resample_planned_kernel` against the exact kernel (1e-4 relative, the north-star
  tolerance) and against the brick kernel's FAST instantiation it replaces for large launches.
* exact, affine-only: the planned box only decides what is staged — results must stay bit-identical to the unplanned launch.
"""
from __future__ import annotations

import pytest
import torch

from test_gpu_ops_parity import _control_points
from test_gpu_ops_parity import _mapping

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4  # BASELINE.json: float intensities within 1e-4 relative


def _rel(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return (a.double() - b.double()).abs() / a.double().abs().clamp_min(1.0)


@pytest.mark.parametrize("elastic", [False, True])
@pytest.mark.parametrize("with_fill", [False, True])
@pytest.mark.parametrize("shape", [(64, 64, 64), (70, 52, 56)])
def test_planned_fast_bricks_stay_within_tolerance(hip, monkeypatch, elastic, with_fill, shape):
    batch = 3
    g = torch.Generator(device="cuda").manual_seed(5)
    t1 = torch.rand(batch, 1, *shape, generator=g, device="cuda") * 4 - 1
    t2 = torch.rand(batch, 2, *shape, generator=g, device="cuda")  # a second image with two channels: the tile is reused
    kwargs = dict(
        out_shape=shape, mapping=_mapping(batch, 11, scale=0.1, shift=4.0).cuda(),
        control_points=_control_points(batch, (5, 5, 5), 12, amplitude=5.0).cuda() if elastic else None,
        in_spacing=(1, 1, 1), out_spacing=(1, 1, 1), affine_first=True, interps=["linear", "linear"],
        fills=[torch.tensor([-1.0], device="cuda"), torch.tensor([0.25, 0.5], device="cuda")] if with_fill else [None, None],
    )
    exact = hip.resample3d([t1, t2], precision="exact", **kwargs)
    monkeypatch.setenv("TIO_FAST_KERNEL", "brick")
    brick = hip.resample3d([t1, t2], precision="fast", **kwargs)
    monkeypatch.setenv("TIO_FAST_KERNEL", "planned")
    planned = hip.resample3d([t1, t2], precision="fast", **kwargs)
    torch.cuda.synchronize()
    for e, b, p in zip(exact, brick, planned):
        assert not torch.equal(e, p)
        beyond = _rel(e, p) > REL_TOL
        # a voxel whose in-bounds weight sits within rounding of 0.5 may flip between sample and fill
        assert int(beyond.sum()) <= (8 if with_fill else 0), int(beyond.sum())
        assert int((_rel(b, p) > REL_TOL).sum()) <= (8 if with_fill else 0)


def test_planned_fast_bricks_handle_gated_and_far_away_elements(hip, monkeypatch):
    """Element 1 is gated out (bit-exact copy), element 2 looks far outside the volume (fill everywhere)."""
    batch, shape = 3, (48, 48, 48)
    g = torch.Generator(device="cuda").manual_seed(9)
    data = torch.rand(batch, 1, *shape, generator=g, device="cuda")
    mapping = _mapping(batch, 21, scale=0.05, shift=2.0)
    mapping[2, :, 3] += 500.0
    kwargs = dict(
        out_shape=shape, mapping=mapping.cuda(), control_points=None, in_spacing=(1, 1, 1), out_spacing=(1, 1, 1), affine_first=True,
        interps=["linear"], fills=[torch.tensor([-3.0], device="cuda")], passthrough=torch.tensor([0, 1, 0], dtype=torch.uint8).cuda(),
    )
    exact = hip.resample3d([data], precision="exact", **kwargs)[0]
    monkeypatch.setenv("TIO_FAST_KERNEL", "planned")
    planned = hip.resample3d([data], precision="fast", **kwargs)[0]
    torch.cuda.synchronize()
    assert torch.equal(planned[1], data[1])
    assert torch.all(planned[2] == -3.0) and torch.all(exact[2] == -3.0)
    assert int((_rel(exact[0], planned[0]) > REL_TOL).sum()) <= 8


def test_planned_fast_bricks_with_boxes_beyond_the_lds_budget(hip, monkeypatch):
    """Zooming out by 4: every brick's box exceeds the budget, the planner marks them for the per-voxel road."""
    batch, shape = 2, (64, 64, 64)
    g = torch.Generator(device="cuda").manual_seed(13)
    data = torch.rand(batch, 1, *shape, generator=g, device="cuda")
    mapping = torch.zeros(batch, 3, 4)
    for b in range(batch):
        mapping[b, :, :3] = torch.eye(3) * 4.0
        mapping[b, :, 3] = -96.0
    kwargs = dict(out_shape=shape, mapping=mapping.cuda(), control_points=None, in_spacing=(1, 1, 1), out_spacing=(1, 1, 1), affine_first=True,
                  interps=["linear"], fills=[None])
    exact = hip.resample3d([data], precision="exact", **kwargs)[0]
    monkeypatch.setenv("TIO_FAST_KERNEL", "planned")
    planned = hip.resample3d([data], precision="fast", **kwargs)[0]
    torch.cuda.synchronize()
    assert float(_rel(exact, planned).max()) <= REL_TOL


@pytest.mark.parametrize("shape", [(64, 64, 64), (70, 52, 56), (40, 40, 37)])
@pytest.mark.parametrize("strong", [False, True])
def test_planned_exact_affine_launches_are_bit_identical(hip, monkeypatch, shape, strong):
    batch = 3
    g = torch.Generator(device="cuda").manual_seed(17)
    t1 = torch.rand(batch, 1, *shape, generator=g, device="cuda")
    seg = (torch.rand(batch, 1, *shape, generator=g, device="cuda") * 5).to(torch.int16)
    mapping = _mapping(batch, 31, scale=0.4 if strong else 0.1, shift=20.0 if strong else 3.0)
    kwargs = dict(
        out_shape=shape, mapping=mapping.cuda(), control_points=None, in_spacing=(1, 1, 1), out_spacing=(1, 1, 1), affine_first=True,
        interps=["linear", "nearest"], fills=[torch.tensor([0.5], device="cuda"), None], precision="exact",
    )
    monkeypatch.setenv("TIO_EXACT_PLAN", "0")
    plain = hip.resample3d([t1, seg], **kwargs)
    monkeypatch.setenv("TIO_EXACT_PLAN", "2")
    planned = hip.resample3d([t1, seg], **kwargs)
    torch.cuda.synchronize()
    assert torch.equal(plain[0], planned[0]) and torch.equal(plain[1], planned[1])
