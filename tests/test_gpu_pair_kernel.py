"""GPU: two channels — and the call's label channel — per exact-coordinate launch (round 6: `resample_lean_exact_pair_kernel`, csrc/resample_lean_exact.hpp — a subject's
float32 images, or an image's channels, share one descriptor round trip, one set of control planes and ONE coordinate chain per
brick; the second channel's box is staged into the same tile behind the first channel's sampling).  A choice of launch shape,
never of values: with `TIO_LEAN_PAIR=0` (one launch per channel, the road until round 6) every output is the same bit pattern, in
the exact AND the tight mode; the exact mode also equals the oracle (golden cases, `tests/test_gpu_config5.py`).
Reference: one `grid_sample` per image over the same grid (spatial.py:1230-1262, 1695-1731).
"""
from __future__ import annotations

import copy

import pytest
import torch

import torchio_amd as tio
from parity_harness import nested_spheres
from torchio_amd import ops

pytestmark = pytest.mark.gpu


def _run(monkeypatch, transform, batch, pair: str, seed: int = 11, label: str | None = None):
    monkeypatch.setenv("TIO_LEAN_PAIR", pair)
    monkeypatch.setenv("TIO_LEAN_LABEL", pair if label is None else label)  # (the label channel riding along the last launch: with the pairs unless told otherwise)
    monkeypatch.setenv("TIO_EXACT_LEAN", "2")  # (the exact-coordinate kernels for launches below the planned roads' 12 288 bricks too)
    ops.reload_env()
    torch.manual_seed(seed)
    out = transform(copy.deepcopy(batch))
    torch.cuda.synchronize()
    return out


def _subjects(size, batch, channels=(1, 1), seed=7, with_labels=True):
    g = torch.Generator().manual_seed(seed)
    subjects = []
    for _ in range(batch):
        images = {f"im{n}": tio.ScalarImage(torch.rand(c, *size, generator=g) + n) for n, c in enumerate(channels)}
        if with_labels:
            images["seg"] = tio.LabelMap(nested_spheres(size[0])[:, : size[0], : size[1], : size[2]].contiguous())
        subjects.append(tio.Subject(**images))
    return tio.SubjectsBatch.from_subjects(subjects).to("cuda")


@pytest.mark.parametrize("precision", ["exact", "tight"])
@pytest.mark.parametrize(
    "size,batch,channels,transform",
    [
        ((256, 256, 256), 2, (1, 1), tio.Spatial(degrees=(-10, 10), scales=(0.9, 1.1), translation=(-5, 5), max_displacement=7.5)),  # config 5's call shape
        ((256, 256, 256), 1, (3,), tio.Affine(degrees=(-10, 10), scales=(0.9, 1.1), translation=(-5, 5))),  # three channels: a pair and an odd one out
        ((200, 184, 168), 2, (2, 1, 1), tio.Spatial(degrees=(-8, 8), scales=(0.95, 1.05), translation=(-3, 3), max_displacement=5.0)),  # partial bricks, two pairs
        ((256, 256, 256), 2, (1, 1), tio.Affine(degrees=(30, 30), scales=(1.0, 1.0), translation=(2, 2))),  # boxes beyond the tile (see below)
        ((256, 256, 256), 2, (1, 1), tio.ElasticDeformation(max_displacement=7.5)),
        ((256, 256, 256), 3, (1, 1), tio.Affine(degrees=(-10, 10), scales=(0.9, 1.1), translation=(-5, 5), p=0.5, per_instance=True)),  # gated-out elements: bit-exact copies
    ],
)
def test_pairs_of_channels_equal_one_launch_per_channel(hip, monkeypatch, precision, size, batch, channels, transform):
    data = _subjects(size, batch, channels)
    previous = tio.get_resample_precision()
    try:
        tio.set_resample_precision(precision)
        single = _run(monkeypatch, transform, data, "0")
        paired = _run(monkeypatch, transform, data, "1")
        alone_with_label = _run(monkeypatch, transform, data, "0", label="1")  # one launch per channel, the label map with the last
        for name in single.images:
            assert torch.equal(single.images[name].data, paired.images[name].data), name
            assert torch.equal(single.images[name].data, alone_with_label.images[name].data), name
        moved = any(not torch.equal(single.images[name].data, data.images[name].data) for name in single.images)
        assert moved  # (the transform did something)
    finally:
        tio.set_resample_precision(previous)
        monkeypatch.delenv("TIO_LEAN_PAIR", raising=False)
        monkeypatch.delenv("TIO_LEAN_LABEL", raising=False)
        monkeypatch.delenv("TIO_EXACT_LEAN", raising=False)
        ops.reload_env()


def test_large_rotation_without_the_hint_takes_the_per_voxel_road_for_both_channels(hip, monkeypatch):
    """No large-box hint (`_expects_large_boxes` answering 0): bricks whose box exceeds the tile sample voxel by voxel from global
    memory — in the pair kernel for BOTH channels (`lean_exact_slow_planes<.., SECOND>`)."""
    from torchio_amd.transforms import spatial as sp

    monkeypatch.setattr(sp, "_expects_large_boxes", lambda *args: 0)
    data = _subjects((256, 256, 256), 2, (1, 1), with_labels=True)  # (the label channel rides along: `lean_exact_slow_planes<.., LABEL>`)
    transform = tio.Affine(degrees=(25, 25), scales=(1.0, 1.0), translation=(3, 3))
    previous = tio.get_resample_precision()
    try:
        tio.set_resample_precision("exact")
        single = _run(monkeypatch, transform, data, "0")
        paired = _run(monkeypatch, transform, data, "1")
        for name in single.images:
            assert torch.equal(single.images[name].data, paired.images[name].data), name
    finally:
        tio.set_resample_precision(previous)
        monkeypatch.delenv("TIO_LEAN_PAIR", raising=False)
        monkeypatch.delenv("TIO_LEAN_LABEL", raising=False)
        monkeypatch.delenv("TIO_EXACT_LEAN", raising=False)
        ops.reload_env()


@pytest.mark.parametrize("dtype", [torch.uint8, torch.int16, torch.int32, torch.int64])
@pytest.mark.parametrize("precision", ["exact", "tight"])
def test_one_image_and_a_label_map_of_every_width(hip, monkeypatch, precision, dtype):
    """The label channel behind ONE float channel (`resample_lean_exact_label_kernel<.., PAIR = false>`), element bits of 1 / 2 / 4 bytes
    (8 bytes: not taken along — its own kernel), a volume that ends in partial bricks, per-element parameters with gated-out elements."""
    size = (232, 200, 176)
    g = torch.Generator().manual_seed(23)
    subjects = [
        tio.Subject(
            t1=tio.ScalarImage(torch.rand(1, *size, generator=g)),
            seg=tio.LabelMap(torch.randint(0, 7, (1, *size), generator=g).to(dtype)),
        )
        for _ in range(3)
    ]
    data = tio.SubjectsBatch.from_subjects(subjects).to("cuda")
    transform = tio.Spatial(degrees=(-10, 10), scales=(0.9, 1.1), translation=(-5, 5), max_displacement=6.0, p=0.7, per_instance=True)
    previous = tio.get_resample_precision()
    try:
        tio.set_resample_precision(precision)
        apart = _run(monkeypatch, transform, data, "0", seed=5)
        along = _run(monkeypatch, transform, data, "1", seed=5)
        for name in apart.images:
            assert torch.equal(apart.images[name].data, along.images[name].data), name
        assert not torch.equal(apart.images["seg"].data, data.images["seg"].data)
    finally:
        tio.set_resample_precision(previous)
        monkeypatch.delenv("TIO_LEAN_PAIR", raising=False)
        monkeypatch.delenv("TIO_LEAN_LABEL", raising=False)
        monkeypatch.delenv("TIO_EXACT_LEAN", raising=False)
        ops.reload_env()
