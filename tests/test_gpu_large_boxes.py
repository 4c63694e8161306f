"""GPU: boxes beyond the staging tile (TIO_GEOM_LARGE_BOXES / TIO_GEOM_MOSTLY_LARGE_BOXES, ABI 15; transforms/spatial.py:
`_expects_large_boxes`).  Round 6: the exact-coordinate lean kernels stage such a brick in two / four passes over its planes —
listed by the planner and walked by a second kernel (hint level 1: some elements), or in every block of one launch (level 2:
most) — instead of sampling it voxel by voxel from global memory (no hint).  A choice of road: the exact mode's values must not
change by a bit whatever the level, the tight mode stays inside the per-voxel bar against the exact one at every level (its per-voxel
road interpolates in ATen's order, its staged passes with fused lerps), label maps are untouched.
The reference allows any rotation (docs/examples/plot_3d_to_2d.py:27: 360 degrees).
"""
from __future__ import annotations

import copy

import pytest
import torch

import torchio_amd as tio
from parity_harness import nested_spheres
from torchio_amd.transforms import spatial as sp

pytestmark = pytest.mark.gpu


def _subjects(size: int, batch: int):
    g = torch.Generator().manual_seed(3)
    return [
        tio.Subject(t1=tio.ScalarImage(torch.rand(1, size, size, size, generator=g)), seg=tio.LabelMap(nested_spheres(size)))
        for _ in range(batch)
    ]


def _run_with_hint(monkeypatch, transform, subjects, forced):
    """The transform on the device with `_expects_large_boxes` answering *forced* (None: its own estimate); returns (batch, hints seen)."""
    hints = []
    original = sp._expects_large_boxes

    def spy(*args):
        value = original(*args) if forced is None else forced
        hints.append(value)
        return value

    monkeypatch.setattr(sp, "_expects_large_boxes", spy)
    torch.manual_seed(1)
    out = transform(tio.SubjectsBatch.from_subjects(copy.deepcopy(subjects)).to("cuda"))
    torch.cuda.synchronize()
    monkeypatch.setattr(sp, "_expects_large_boxes", original)
    return out, hints


def _per_voxel(a, b) -> float:
    value_range = float(b.max() - b.min())
    return float(((a - b).abs() / b.abs().clamp_min(1e-3 * value_range)).max())


@pytest.mark.parametrize("degrees", [25, 45, 180])
@pytest.mark.parametrize("precision", ["exact", "tight"])
def test_every_hint_level_computes_the_same_values(hip, monkeypatch, precision, degrees):
    """3 x 256^3 (12 288 bricks: the planned roads' size) rotated about every axis: at 25 / 45 degrees every brick's box is beyond the
    tile (two / four passes, at 45 some quarters voxel by voxel), at 180 none is.  No hint (per-voxel road) == level 1 (list +
    walkers) == level 2 (passes in every block), bit for bit in the exact mode; tight within the per-voxel bar of exact."""
    size, batch = 256, 3
    subjects = _subjects(size, batch)
    transform = tio.Affine(degrees=(degrees, degrees), scales=(1.0, 1.0), translation=(3, 3))
    previous = tio.get_resample_precision()
    try:
        tio.set_resample_precision("exact")
        reference, _ = _run_with_hint(monkeypatch, transform, subjects, 0)
        tio.set_resample_precision(precision)
        estimated, hints = _run_with_hint(monkeypatch, transform, subjects, None)
        assert hints == [0 if degrees == 180 else 2]
        for level in (0, 1, 2):
            out, _ = _run_with_hint(monkeypatch, transform, subjects, level)
            assert torch.equal(out.images["seg"].data, reference.images["seg"].data)
            if precision == "exact":  # a road, never a value
                assert torch.equal(out.images["t1"].data, reference.images["t1"].data), level
            else:  # (the per-voxel road interpolates in ATen's order, the staged passes with fused lerps: each inside the bar)
                assert _per_voxel(out.images["t1"].data, reference.images["t1"].data) <= 1e-4, level
        assert torch.equal(estimated.images["seg"].data, reference.images["seg"].data)
    finally:
        tio.set_resample_precision(previous)


def _rotation_mappings(degrees_per_element, size):
    """Output -> input voxel mappings (n, 3, 4): a rotation about every axis by the element's angle, about the volume's centre."""
    import numpy as np

    rows = []
    centre = (size - 1) / 2.0
    for degrees in degrees_per_element:
        rotation = sp._euler_to_rotation_matrix(np.asarray((degrees,) * 3, dtype=np.float64))
        shift = centre - rotation @ np.full(3, centre) + 1.5
        rows.append(np.concatenate([rotation, shift[:, None]], axis=1))
    return torch.from_numpy(np.stack(rows).astype(np.float32))


@pytest.mark.parametrize("elastic", [False, True])
@pytest.mark.parametrize("precision", ["exact", "tight"])
def test_one_large_element_in_a_batch_is_listed_not_the_launch(hip, precision, elastic):
    """Per ELEMENT: 7 x 5 degrees + 1 x 30 degrees — the estimate answers 1 (some), the large element's bricks
    are listed and staged in passes by the walkers, the others by their own blocks: the same values as without any hint and as
    with the pass logic in every block, and the mixed launch costs no more than 1.25 x the launch of eight small elements."""
    size, batch = 256, 8
    g = torch.Generator(device="cuda").manual_seed(2)
    data = torch.rand(batch, 1, size, size, size, generator=g, device="cuda")
    mixed = _rotation_mappings([5, 5, 5, 30, 5, 5, 5, 5], size)
    small = _rotation_mappings([5] * 8, size)
    assert sp._expects_large_boxes(mixed.numpy(), None, None, (size,) * 3, (1.0, 1.0, 1.0)) == 1
    assert sp._expects_large_boxes(small.numpy(), None, None, (size,) * 3, (1.0, 1.0, 1.0)) == 0
    field = ((torch.rand(batch, 7, 7, 7, 3, generator=torch.Generator().manual_seed(4)) - 0.5) * 15.0).cuda() if elastic else None
    kwargs = dict(out_shape=(size,) * 3, control_points=field, in_spacing=(1, 1, 1), out_spacing=(1, 1, 1), affine_first=True,
                  interps=["linear"], fills=[torch.tensor([0.5], device="cuda")], precision=precision)
    outs = [hip.resample3d([data], mapping=mixed.cuda(), large_boxes=level, **kwargs)[0] for level in (0, 1, 2)]
    torch.cuda.synchronize()
    if precision == "exact":
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    else:  # (tight: the per-voxel road interpolates in ATen's order, the staged passes with fused lerps — both inside the bar of exact)
        exact = hip.resample3d([data], mapping=mixed.cuda(), large_boxes=0, **{**kwargs, "precision": "exact"})[0]
        assert all(_per_voxel(out, exact) <= 1e-4 for out in outs)
        assert torch.equal(outs[1], outs[2])

    def timed(mapping, level):
        mapping = mapping.cuda()
        for _ in range(3):
            hip.resample3d([data], mapping=mapping, large_boxes=level, **kwargs)
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(10):
            hip.resample3d([data], mapping=mapping, large_boxes=level, **kwargs)
        end.record()
        torch.cuda.synchronize()
        return start.elapsed_time(end) / 10

    t_small, t_mixed, t_unhinted, t_all = timed(small, 0), timed(mixed, 1), timed(mixed, 0), timed(mixed, 2)
    print(f"8 x 5 deg {t_small:.3f} ms; 7 x 5 + 1 x 30 deg: listed {t_mixed:.3f} ms, passes in every block {t_all:.3f} ms, without the hint {t_unhinted:.3f} ms")
    assert t_mixed <= 1.4 * t_small, (t_small, t_mixed, t_unhinted)
