"""GPU: work AHEAD of the data (round 4) — brick plans made by `tio_resample3d_plan` on a side stream, parameter uploads of
a `Compose` that draws ahead, the folded minimum announced by the next child.  None of it may change a bit of any result:
every test compares against the same call / pipeline without the side stream.
"""
from __future__ import annotations

import copy
import warnings

import pytest
import torch

import torchio_amd as tio
from torchio_amd import ops

pytestmark = pytest.mark.gpu


def _mapping(batch: int, seed: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    m = torch.eye(3, 4).repeat(batch, 1, 1)
    m[:, :, :3] += (torch.rand(batch, 3, 3, generator=g) - 0.5) * 0.1
    m[:, :, 3] = (torch.rand(batch, 3, generator=g) - 0.5) * 6.0
    return m.contiguous()


def _geometry(batch, shape, elastic, seed):
    g = torch.Generator().manual_seed(seed + 1)
    return dict(
        out_shape=shape, mapping=_mapping(batch, seed).cuda(),
        control_points=((torch.rand(batch, 7, 7, 7, 3, generator=g) - 0.5) * 10.0).cuda() if elastic else None,
        in_spacing=(1, 1, 1), out_spacing=(1, 1, 1), affine_first=True,
    )


@pytest.mark.parametrize("precision,elastic", [("fast", False), ("fast", True), ("exact", False), ("exact", True), ("tight", False), ("tight", True)])
def test_plan_made_ahead_on_another_stream_is_the_plan_of_the_call(hip, precision, elastic):
    batch, shape = 3, (256, 256, 256)  # 12 288 bricks: the smallest launch that starts from a plan by itself
    data = torch.rand(batch, 1, *shape, generator=torch.Generator(device="cuda").manual_seed(5), device="cuda")
    geometry = _geometry(batch, shape, elastic, seed=7)
    fill = torch.tensor([-0.25], device="cuda")
    plain = hip.resample3d([data], interps=["linear"], fills=[fill], precision=precision, **geometry)[0]
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        plan = hip.resample_plan(batch=batch, in_shape=shape, precision=precision, **geometry)
        ready = side.record_event()
    assert plan is not None and plan.dtype == torch.int32 and plan.numel() * 4 >= (batch * 16 + batch * 4096 * 16) * 4
    torch.cuda.current_stream().wait_event(ready)
    plan.record_stream(torch.cuda.current_stream())
    calls = []
    original = hip._call
    hip._call = lambda name, *args: (calls.append(name), original(name, *args))[1]
    try:
        ahead = hip.resample3d([data], interps=["linear"], fills=[fill], precision=precision, plan=plan, **geometry)[0]
    finally:
        del hip._call
    torch.cuda.synchronize()
    assert calls == ["resample3d"]
    assert torch.equal(plain, ahead)
    # a plan too small for the launch is ignored, not trusted
    short = hip.resample3d([data], interps=["linear"], fills=[fill], precision=precision, plan=plan[:64].clone(), **geometry)[0]
    torch.cuda.synchronize()
    assert torch.equal(plain, short)


def test_no_plan_for_launches_that_take_another_road(hip):
    small = _geometry(2, (64, 64, 64), False, seed=9)  # 128 bricks: in-kernel boxes
    assert hip.resample_plan(batch=2, in_shape=(64, 64, 64), precision="fast", **small) is None
    # (round 5: the exact ELASTIC launch of unit-spacing volumes plans too — the lean exact-coordinate kernel; with another
    # spacing its displacements are divided per voxel and it stays on the brick kernel's in-kernel boxes)
    elastic_exact = _geometry(3, (256, 256, 256), True, seed=9)
    elastic_exact.update(in_spacing=(1, 1, 2), out_spacing=(1, 1, 2))
    assert hip.resample_plan(batch=3, in_shape=(256, 256, 256), precision="exact", **elastic_exact) is None
    odd = _geometry(3, (256, 256, 254), False, seed=9)  # K not a multiple of 4: no LDS-DMA rows
    assert hip.resample_plan(batch=3, in_shape=(256, 256, 254), precision="fast", **odd) is None


def test_a_plan_handed_to_a_call_with_other_images_is_ignored(hip):
    batch, shape = 3, (256, 256, 256)
    labels = torch.randint(0, 5, (batch, 1, *shape), generator=torch.Generator(device="cuda").manual_seed(2), device="cuda", dtype=torch.int16)
    geometry = _geometry(batch, shape, False, seed=11)
    plan = hip.resample_plan(batch=batch, in_shape=shape, precision="fast", **geometry)
    assert plan is not None
    plain = hip.resample3d([labels], interps=["nearest"], fills=[None], precision="fast", **geometry)[0]
    with_plan = hip.resample3d([labels], interps=["nearest"], fills=[None], precision="fast", plan=plan, **geometry)[0]
    torch.cuda.synchronize()
    assert torch.equal(plain, with_plan)


def _pipeline():
    return tio.Compose([
        tio.Affine(degrees=(-10, 10), scales=(0.9, 1.1), translation=(-5, 5)),
        tio.ElasticDeformation(),
        tio.BiasField(),
        tio.Blur(std=(0.5, 1.5)),
        tio.Noise(std=(0.02, 0.05)),
    ])


@pytest.fixture
def _modes():
    previous = (tio.get_noise_rng(), tio.get_resample_precision(), tio.get_stencil_precision())
    yield
    tio.set_noise_rng(previous[0]); tio.set_resample_precision(previous[1]); tio.set_stencil_precision(previous[2])
    ops.set_ahead_stream(True)


@pytest.mark.parametrize("mode", [("philox", "fast"), ("reference", "exact")])
def test_compose_on_the_side_stream_changes_nothing(hip, _modes, mode):
    """The headline pipeline (and the library default), 3 x 256^3 float32 + int16 labels: values bit for bit, history and the
    global generator's state with and without the side stream; steps issued back to back without a synchronisation in
    between (the uploads and plans of step k + 1 run while the kernels of step k are still in flight)."""
    tio.set_noise_rng(mode[0]); tio.set_resample_precision(mode[1], allow_out_of_tolerance=True); tio.set_stencil_precision(mode[1])
    size, batch, steps = 256, 3, 6
    g = torch.Generator().manual_seed(17)
    subjects = [
        tio.Subject(t1=tio.ScalarImage(torch.rand(1, size, size, size, generator=g)), seg=tio.LabelMap(torch.randint(0, 4, (1, size, size, size), generator=g, dtype=torch.int16)))
        for _ in range(batch)
    ]
    source = tio.SubjectsBatch.from_subjects(subjects).to("cuda")
    results = []
    for ahead in (True, False):
        ops.set_ahead_stream(ahead)
        pipeline = _pipeline()
        assert pipeline._may_draw_ahead()
        torch.manual_seed(23)
        outs = []
        seen = []
        original = hip._call
        hip._call = lambda name, *args: (seen.append(name), original(name, *args))[1]
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                for _ in range(steps):
                    out = pipeline(source)
                    outs.append((out.t1.data, out.seg.data, [(t.name, copy.deepcopy(dict(t.params))) for t in out.applied_transforms]))
        finally:
            del hip._call
        probe = torch.rand(3)
        torch.cuda.synchronize()
        if mode[1] == "fast":
            assert ("resample3d_plan" in seen) is ahead  # the planners really moved (FAST: both launches start from a plan)
        results.append((outs, probe))
    (with_ahead, probe_a), (without, probe_b) = results
    assert torch.equal(probe_a, probe_b)
    for a, b in zip(with_ahead, without, strict=True):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        assert a[2] == b[2]


def test_compose_announces_the_minimum_to_the_launch_before(hip, _modes):
    """`Compose[Affine, ElasticDeformation]` in the FAST mode: the affine launch folds the minimum of its first element into
    its stores because the elastic child will ask for it — one `channel_min` reduction per step instead of two, same values."""
    tio.set_resample_precision("fast", allow_out_of_tolerance=True)
    size, batch = 256, 3
    g = torch.Generator().manual_seed(29)
    subjects = [tio.Subject(t1=tio.ScalarImage(torch.rand(1, size, size, size, generator=g) - 0.3)) for _ in range(batch)]
    source = tio.SubjectsBatch.from_subjects(subjects).to("cuda")
    pipeline = tio.Compose([tio.Affine(degrees=(-10, 10), translation=(-5, 5)), tio.ElasticDeformation()])
    outs, reductions = [], []
    for announce in (True, False):
        seen = []
        original = hip._call
        hip._call = lambda name, *args: (seen.append(name), original(name, *args))[1]
        real = ops.expect_minimum_fill
        if not announce:
            ops.expect_minimum_fill = lambda flag: real(False)
        try:
            torch.manual_seed(31)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                outs.append(pipeline(source).t1.data)
        finally:
            del hip._call
            ops.expect_minimum_fill = real
        reductions.append(seen.count("channel_min"))
    torch.cuda.synchronize()
    assert reductions == [1, 2], reductions
    assert torch.equal(outs[0], outs[1])
