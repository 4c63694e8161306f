"""GPU: the fused "label" partial-volume mode of tio_resample3d (TIO_LABEL_PV) vs the CPU oracle.

The oracle restates the reference literally (one-hot channel vector per voxel, grid_sample
accumulation, argmax, ATen's cascade sum > 0.5; spatial.py:1275-1389) and is pinned against
golden vectors of the reference (tests/golden, cases ``*label_pv*``); the HIP kernel never
builds the channels.  Bar: bit-exact labels, every dtype, with and without the label table.
"""
from __future__ import annotations

import pytest
import torch

from test_gpu_ops_parity import _both
from test_gpu_ops_parity import _control_points
from test_gpu_ops_parity import _data
from test_gpu_ops_parity import _mapping

pytestmark = pytest.mark.gpu


def _labels(shape, dtype, seed, count):
    """Blocky label volumes (2-voxel blocks: plenty of mixed neighbourhoods) with `count` label values."""
    g = torch.Generator().manual_seed(seed)
    coarse = torch.randint(0, count, tuple((s + 1) // 2 for s in shape[-3:]), generator=g)
    full = coarse.repeat_interleave(2, 0).repeat_interleave(2, 1).repeat_interleave(2, 2)[: shape[-3], : shape[-2], : shape[-1]]
    values = full * 3 - (5 if dtype not in (torch.uint8,) else 0)  # non-contiguous, some negative
    noise = torch.randint(0, count, shape, generator=g) * 3 - (5 if dtype not in (torch.uint8,) else 0)
    pick = torch.rand(shape, generator=g) < 0.15
    return torch.where(pick, noise, values.expand(shape)).to(dtype)


@pytest.mark.parametrize("dtype", [torch.int16, torch.uint8, torch.int8, torch.int32, torch.int64, torch.float32, torch.float64, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("count", [4, 40])
def test_label_pv_matches_oracle_every_dtype(oracle, hip, dtype, count):
    batch, shape, out_shape = 2, (20, 18, 22), (19, 21, 20)
    data = _labels((batch, 1, *shape), dtype, 11, count)
    table = torch.unique(data).to(torch.float64)
    kwargs = dict(
        out_shape=out_shape, mapping=_mapping(batch, 12, scale=0.2, shift=3.0), control_points=None,
        in_spacing=(1, 1, 1), out_spacing=(1, 1, 1), affine_first=True, interps=["label"], fills=[None],
        label_tables=[table], pad_labels=[7.0],
    )
    cpu, gpu = _both(oracle, hip, "resample3d", ([data],), **kwargs)
    assert gpu[0].dtype == dtype
    assert torch.equal(cpu[0], gpu[0].cpu())
    # the oracle derives the table itself (torch.unique) when none is given: same result
    kwargs["label_tables"] = [None]
    cpu_own = oracle.resample3d([data], **kwargs)
    assert torch.equal(cpu_own[0], cpu[0])


@pytest.mark.parametrize("elastic", [False, True])
@pytest.mark.parametrize("count", [3, 17, 300])
def test_label_pv_rides_along_with_intensity_images(oracle, hip, elastic, count):
    """t1 (linear, brick kernel) + seg (label mode, own kernel) + seg2 (nearest) through ONE call."""
    batch, shape = 2, (32, 24, 40)
    t1 = _data((batch, 2, *shape), torch.float32, 21)
    seg = _labels((batch, 1, *shape), torch.int16, 22, count)
    seg2 = _labels((batch, 1, *shape), torch.uint8, 23, 5)
    kwargs = dict(
        out_shape=shape, mapping=_mapping(batch, 24, scale=0.12, shift=4.0),
        control_points=_control_points(batch, (7, 7, 7), 25, amplitude=4.0) if elastic else None,
        in_spacing=(1.0, 1.2, 0.9), out_spacing=(1.0, 1.2, 0.9), affine_first=True,
        interps=["linear", "label", "nearest"], fills=[torch.tensor([-1.0, 0.5]), None, torch.tensor([2.0])],
        label_tables=[None, torch.unique(seg).to(torch.float64), None], pad_labels=[0.0, -5.0, 0.0],
    )
    cpu, gpu = _both(oracle, hip, "resample3d", ([t1, seg, seg2],), **kwargs)
    for expected, actual in zip(cpu, gpu, strict=True):
        assert torch.equal(expected, actual.cpu())


def test_label_pv_half_voxel_ties_and_passthrough(oracle, hip):
    """Exact 0.5 / 0.25 weights: argmax ties go to the smaller label, the 0.5 in-bounds sum is NOT in bounds."""
    batch, shape = 3, (12, 10, 8)
    data = _labels((batch, 1, *shape), torch.int16, 31, 6)
    mapping = torch.eye(3, 4).repeat(batch, 1, 1)
    mapping[0, :, 3] = torch.tensor([0.5, -0.5, 1.5])
    mapping[1, :, 3] = torch.tensor([-0.5, 0.5, 0.5])
    kwargs = dict(
        out_shape=shape, mapping=mapping, control_points=None, in_spacing=(1, 1, 1), out_spacing=(1, 1, 1),
        affine_first=True, interps=["label"], fills=[None], label_tables=[torch.unique(data).to(torch.float64)],
        pad_labels=[99.0], passthrough=torch.tensor([0, 0, 1], dtype=torch.uint8),
    )
    cpu, gpu = _both(oracle, hip, "resample3d", ([data],), **kwargs)
    assert torch.equal(cpu[0], gpu[0].cpu())
    assert torch.equal(gpu[0][2].cpu(), data[2])  # gated-out element: bit-exact copy
    assert bool((gpu[0][0] == 99).any())


def test_label_pv_rejects_multichannel(hip):
    data = torch.zeros(1, 2, 8, 8, 8, dtype=torch.int16, device="cuda")
    with pytest.raises(RuntimeError, match="channels == 1"):
        hip.resample3d(
            [data], out_shape=(8, 8, 8), mapping=torch.eye(3, 4)[None].cuda(), control_points=None, in_spacing=(1, 1, 1),
            out_spacing=(1, 1, 1), affine_first=True, interps=["label"], fills=[None],
        )


def test_label_pv_transform_end_to_end_on_device(oracle):
    """tio.Spatial(label_interpolation="label") on a device-resident subject == the same call on the oracle."""
    import torchio_amd as tio
    from parity_harness import make_subjects
    from parity_harness import use_engine

    kwargs = dict(degrees=(-15, 15), scales=(0.85, 1.15), translation=(-4, 4), max_displacement=4.0, label_interpolation="label",
                  default_pad_label=3)
    subjects = make_subjects(24, 3, 41)
    torch.manual_seed(42)
    with use_engine(oracle):
        expected = tio.Spatial(**kwargs)(tio.SubjectsBatch.from_subjects(subjects))
    torch.manual_seed(42)
    actual = tio.Spatial(**kwargs)(tio.SubjectsBatch.from_subjects(subjects).to("cuda"))
    assert actual.images["seg"].data.is_cuda
    assert torch.equal(expected.images["seg"].data, actual.images["seg"].data.cpu())
    assert torch.equal(expected.images["t1"].data, actual.images["t1"].data.cpu())
