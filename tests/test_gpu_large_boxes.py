"""GPU: TIO_GEOM_LARGE_BOXES (ABI 14; transforms/spatial.py: `_expects_large_boxes`) — a launch whose bricks' input boxes exceed the
planned roads' staging tile takes the brick kernels, which split such a brick into passes over its planes, instead of sampling
it voxel by voxel from global memory.  A choice of road: the exact mode's values must not change by a bit, the tight and fast
modes stay inside their bars, label maps are untouched.
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


@pytest.mark.parametrize("precision", ["exact", "tight", "fast"])
def test_the_hint_changes_the_road_not_the_values(hip, monkeypatch, precision):
    """3 x 256^3 (12 288 bricks: the planned roads' size), 25 degrees about every axis: every brick's box is beyond the tile."""
    size, batch = 256, 3
    subjects = _subjects(size, batch)
    transform = tio.Affine(degrees=(25, 25), scales=(1.0, 1.0), translation=(3, 3))
    previous = tio.get_resample_precision()
    tio.set_resample_precision(precision, allow_out_of_tolerance=True)
    try:
        hints, results = [], []
        original = sp._expects_large_boxes
        for forced in (None, False):
            def spy(*args, _forced=forced):
                value = original(*args) if _forced is None else _forced
                hints.append(value)
                return value

            monkeypatch.setattr(sp, "_expects_large_boxes", spy)
            torch.manual_seed(1)
            results.append(transform(tio.SubjectsBatch.from_subjects(copy.deepcopy(subjects)).to("cuda")))
        torch.cuda.synchronize()
        assert hints == [True, False]
        with_hint, without = results
        assert torch.equal(with_hint.images["seg"].data, without.images["seg"].data)
        a, b = with_hint.images["t1"].data, without.images["t1"].data
        if precision == "exact":
            assert torch.equal(a, b)  # the brick kernel and the exact-coordinate kernel's per-voxel road: ATen's arithmetic both
        else:
            value_range = float(b.max() - b.min())
            worst = float(((a - b).abs() / b.abs().clamp_min(1e-3 * value_range)).max()) if precision == "tight" else float((a - b).abs().max()) / value_range
            assert worst <= 1e-4, worst
    finally:
        tio.set_resample_precision(previous)
