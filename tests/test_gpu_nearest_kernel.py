"""The nearest-neighbour kernel of `tio_resample3d` (csrc/resample_nearest.hpp): label maps without a fill rule.

The kernel decides a voxel's index from the FAST coordinate line unless a coordinate lies within a margin of a
half-integer, where it evaluates the reference's exact float32 chain.  Its results must be BIT-IDENTICAL to that chain
everywhere: compared here with the CPU oracle (small cases) and with the previous road of nearest images —
`TIO_NEAREST_KERNEL=0`: the gather / brick kernels, every voxel through the exact chain — at full size, on label volumes
of independent random values (a wrong index is a wrong value six times out of seven).
"""
from __future__ import annotations

import pytest
import torch

from test_gpu_ops_parity import _both
from test_gpu_ops_parity import _control_points
from test_gpu_ops_parity import _data
from test_gpu_ops_parity import _mapping

pytestmark = pytest.mark.gpu


def _labels(shape, dtype, seed, device="cuda"):
    g = torch.Generator(device=device).manual_seed(seed)
    if dtype.is_floating_point:
        return torch.rand(*shape, generator=g, device=device).to(dtype)
    return torch.randint(0, 7, shape, generator=g, device=device).to(dtype)


@pytest.mark.parametrize("dtype", [torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64, torch.float32, torch.float64])
@pytest.mark.parametrize("elastic", [False, True])
def test_nearest_kernel_matches_the_oracle(oracle, hip, dtype, elastic):
    batch, shape = 2, (37, 41, 70)  # K >= 48: rows of 64 voxels per wave, ragged on every axis
    data = _data((batch, 2, *shape), dtype, 3)
    kwargs = dict(
        out_shape=shape, mapping=_mapping(batch, 7, scale=0.12, shift=2.5),
        control_points=_control_points(batch, (6, 5, 7), 8, amplitude=5.0) if elastic else None,
        in_spacing=(1.0, 1.5, 0.8), out_spacing=(1.0, 1.5, 0.8), affine_first=not elastic, interps=["nearest"], fills=[None],
    )
    cpu, gpu = _both(oracle, hip, "resample3d", ([data],), **kwargs)
    assert torch.equal(cpu[0], gpu[0].cpu())


def test_every_voxel_a_tie(oracle, hip):
    """A shift of exactly half a voxel on every axis: every index is a rounding tie, every voxel takes the exact chain."""
    shape = (33, 40, 64)
    data = _data((1, 1, *shape), torch.int16, 5)
    mapping = torch.eye(3, 4).unsqueeze(0).clone()
    mapping[0, :, 3] = torch.tensor([0.5, -0.5, 1.5])
    kwargs = dict(out_shape=shape, mapping=mapping, control_points=None, in_spacing=(1, 1, 1), out_spacing=(1, 1, 1), affine_first=True,
                  interps=["nearest"], fills=[None])
    cpu, gpu = _both(oracle, hip, "resample3d", ([data],), **kwargs)
    assert torch.equal(cpu[0], gpu[0].cpu())


def test_non_finite_and_far_away_geometry(oracle, hip):
    """Element 0: a NaN in the mapping; element 1: the volume a thousand voxels away; element 2: gated out; element 3: plain."""
    batch, shape = 4, (32, 32, 64)
    data = _data((batch, 1, *shape), torch.uint8, 11)
    mapping = _mapping(batch, 13, scale=0.05, shift=1.0)
    mapping[0, 1, 2] = float("nan")
    mapping[1, :, 3] += 1000.0
    kwargs = dict(out_shape=shape, mapping=mapping, control_points=_control_points(batch, (4, 4, 4), 2, amplitude=3.0), in_spacing=(1, 1, 1),
                  out_spacing=(1, 1, 1), affine_first=True, interps=["nearest"], fills=[None],
                  passthrough=torch.tensor([0, 0, 1, 0], dtype=torch.uint8), cp_skip=torch.tensor([0, 0, 0, 1], dtype=torch.uint8))
    cpu, gpu = _both(oracle, hip, "resample3d", ([data],), **kwargs)
    assert torch.equal(cpu[0], gpu[0].cpu())
    assert torch.equal(gpu[0][2].cpu(), data[2])
    assert int(gpu[0][1].abs().sum()) == 0


@pytest.mark.parametrize("precision", ["exact", "fast"])
def test_label_maps_no_longer_hold_the_float_images_back(hip, monkeypatch, precision):
    """One call, two float32 images and two label maps: the labels are bit-identical to the all-exact road in both
    precision modes; the float images are bit-identical in exact mode and within the FAST tolerance in fast mode (before
    this kernel a label map in the call kept them on the exact kernels)."""
    batch, shape = 2, (96, 80, 128)
    g = torch.Generator(device="cuda").manual_seed(17)
    t1 = torch.rand(batch, 1, *shape, generator=g, device="cuda")
    t2 = torch.rand(batch, 1, *shape, generator=g, device="cuda") + 1
    seg = _labels((batch, 1, *shape), torch.int16, 19)
    mask = _labels((batch, 2, *shape), torch.uint8, 23)
    kwargs = dict(
        out_shape=shape, mapping=_mapping(batch, 29, scale=0.08, shift=3.0).cuda(), control_points=_control_points(batch, (5, 5, 5), 31, amplitude=4.0).cuda(),
        in_spacing=(1, 1, 1), out_spacing=(1, 1, 1), affine_first=True, interps=["linear", "linear", "nearest", "nearest"],
        fills=[torch.tensor([0.0], device="cuda"), torch.tensor([1.0], device="cuda"), None, None],
    )
    monkeypatch.setenv("TIO_NEAREST_KERNEL", "0")
    reference = hip.resample3d([t1, t2, seg, mask], precision="exact", **kwargs)
    monkeypatch.setenv("TIO_NEAREST_KERNEL", "1")
    got = hip.resample3d([t1, t2, seg, mask], precision=precision, **kwargs)
    torch.cuda.synchronize()
    assert torch.equal(reference[2], got[2]) and torch.equal(reference[3], got[3])
    for r, o in zip(reference[:2], got[:2]):
        if precision == "exact":
            assert torch.equal(r, o)
        else:
            assert not torch.equal(r, o)  # the FAST kernels did run
            beyond = (r.double() - o.double()).abs() > 1e-4
            assert int(beyond.sum()) == 0, int(beyond.sum())  # (fill decisions within rounding of 0.5 take the exact chain's answer)


@pytest.mark.parametrize("size,elastic", [(256, True), (512, False), (512, True)])
def test_full_size_label_maps_are_bit_identical_to_the_exact_chain(hip, monkeypatch, size, elastic):
    """Config 5's label map (512^3 int16) and the bench volume (256^3), every voxel an independent random label."""
    seg = _labels((1, 1, size, size, size), torch.int16, 41)
    kwargs = dict(
        out_shape=(size,) * 3, mapping=_mapping(1, 43, scale=0.06, shift=5.0).cuda(),
        control_points=_control_points(1, (7, 7, 7), 47, amplitude=7.5).cuda() if elastic else None,
        in_spacing=(1, 1, 1), out_spacing=(1, 1, 1), affine_first=True, interps=["nearest"], fills=[None],
    )
    monkeypatch.setenv("TIO_NEAREST_KERNEL", "0")
    reference = hip.resample3d([seg], **kwargs)[0]
    monkeypatch.setenv("TIO_NEAREST_KERNEL", "1")
    got = hip.resample3d([seg], **kwargs)[0]
    torch.cuda.synchronize()
    assert int((reference != got).sum()) == 0
    assert int((got != 0).sum()) > 0.5 * got.numel()  # (the volume is in sight)


def test_margin_zero_would_not_be_bit_identical(hip, monkeypatch):
    """The margin is what makes the kernel exact: with TIO_NEAREST_EPS=0 (the FAST line decides every voxel) a 256^3 launch
    differs from the exact chain in a few hundred voxels — the test above is sensitive to what it claims."""
    size = 256
    seg = _labels((2, 1, size, size, size), torch.int16, 53)
    kwargs = dict(out_shape=(size,) * 3, mapping=_mapping(2, 59, scale=0.06, shift=5.0).cuda(), control_points=None, in_spacing=(1, 1, 1),
                  out_spacing=(1, 1, 1), affine_first=True, interps=["nearest"], fills=[None])
    monkeypatch.setenv("TIO_NEAREST_KERNEL", "0")
    reference = hip.resample3d([seg], **kwargs)[0]
    monkeypatch.setenv("TIO_NEAREST_KERNEL", "1")
    monkeypatch.setenv("TIO_NEAREST_EPS", "0")
    monkeypatch.setenv("TIO_NEAREST_EXACT", "0")  # (round 6: images without a fill rule take the exact-plane kernel, which has no margin — this is about the FAST-line kernel)
    loose = hip.resample3d([seg], **kwargs)[0]
    torch.cuda.synchronize()
    wrong = int((reference != loose).sum())
    assert 0 < wrong < 1e-4 * seg.numel(), wrong


@pytest.mark.parametrize("seed", range(12))
def test_random_geometries_against_the_all_exact_road(hip, monkeypatch, seed):
    """Random shapes (ragged against the 16 x 4 x 64 and 16 x 16 x 16 bricks), dtypes, control grids (some denser than a
    brick), spacings, both composition orders, per-element or shared parameters, gate and skip flags, different input and
    output grids: the kernel == the previous road of nearest images, bit for bit."""
    g = torch.Generator().manual_seed(1000 + seed)

    def pick(*options):
        return options[int(torch.randint(0, len(options), (1,), generator=g))]

    batch = pick(1, 2, 3)
    k_lo, k_hi = pick((12, 47), (50, 140))  # narrower / wider than the 48 voxels from which a wave is one output row
    in_shape = tuple(int(torch.randint(lo, hi, (1,), generator=g)) for lo, hi in ((20, 70), (18, 60), (k_lo, k_hi)))
    out_shape = in_shape if pick(True, True, False) else tuple(max(8, int(s * pick(0.5, 1.5))) for s in in_shape)
    dtype = pick(torch.uint8, torch.int16, torch.int32, torch.int64, torch.float32)
    channels = pick(1, 1, 2)
    data = _labels((batch, channels, *in_shape), dtype, 2000 + seed)
    shared = pick(False, True)
    n_param = 1 if shared else batch
    mapping = _mapping(n_param, 3000 + seed, scale=pick(0.02, 0.1, 0.3), shift=pick(0.0, 3.0, 20.0))
    if out_shape != in_shape:
        for axis in range(3):
            mapping[:, :, axis] *= in_shape[axis] / out_shape[axis]
    elastic = pick(True, True, False)
    cp_shape = pick((4, 4, 4), (7, 7, 7), (5, 9, 6), (out_shape[0] // 3 + 2, 5, 5))
    control_points = _control_points(n_param, cp_shape, 4000 + seed, amplitude=pick(1.0, 6.0)) if elastic else None
    kwargs = dict(
        out_shape=out_shape, mapping=mapping.cuda(), control_points=control_points.cuda() if elastic else None,
        in_spacing=pick((1, 1, 1), (0.8, 1.0, 2.5)), out_spacing=pick((1, 1, 1), (1.5, 0.7, 1.0)), affine_first=pick(True, False),
        interps=["nearest"], fills=[None],
    )
    if out_shape == in_shape and pick(True, False):
        kwargs["passthrough"] = (torch.rand(batch, generator=g) < 0.4).to(torch.uint8).cuda()
    if elastic and pick(True, False):
        kwargs["cp_skip"] = (torch.rand(batch, generator=g) < 0.4).to(torch.uint8).cuda()
    monkeypatch.setenv("TIO_NEAREST_KERNEL", "0")
    reference = hip.resample3d([data], **kwargs)[0]
    monkeypatch.setenv("TIO_NEAREST_KERNEL", "1")
    got = hip.resample3d([data], **kwargs)[0]
    monkeypatch.setenv("TIO_NEAREST_EXACT", "0")  # (round 6: the FAST-line kernel of rounds 3 - 5 where the exact-plane kernel is the default)
    got_line = hip.resample3d([data], **kwargs)[0]
    torch.cuda.synchronize()
    assert torch.equal(reference, got), (seed, in_shape, out_shape, dtype, cp_shape if elastic else None)
    assert torch.equal(reference, got_line), (seed, in_shape, out_shape, dtype, cp_shape if elastic else None)


@pytest.mark.parametrize("norm_shape", [(48, 40, 64), (200, 150, 260), (97, 81, 130)])
def test_label_map_on_another_grid_than_the_normalising_one(hip, monkeypatch, norm_shape):
    """`Resample` onto a named image of a multi-resolution subject: the grid is normalised with the FIRST image's shape
    and un-normalised with the label map's own (spatial.py:1136-1191) — coarser, finer and odd ratios."""
    batch, shape = 2, (96, 80, 128)
    seg = _labels((batch, 1, *shape), torch.int16, 61)
    mapping = _mapping(batch, 67, scale=0.05, shift=2.0)
    for axis in range(3):  # the mapping lives on the normalising grid
        mapping[:, :, axis] *= norm_shape[axis] / shape[axis]
    kwargs = dict(out_shape=shape, mapping=mapping.cuda(), control_points=_control_points(batch, (5, 5, 5), 71, amplitude=3.0).cuda(),
                  in_spacing=(1, 1, 1), out_spacing=(1, 1, 1), affine_first=True, interps=["nearest"], fills=[None], norm_shape=norm_shape)
    monkeypatch.setenv("TIO_NEAREST_KERNEL", "0")
    reference = hip.resample3d([seg], **kwargs)[0]
    monkeypatch.setenv("TIO_NEAREST_KERNEL", "1")
    got = hip.resample3d([seg], **kwargs)[0]
    torch.cuda.synchronize()
    assert torch.equal(reference, got)
    assert int((got != 0).sum()) > 0.3 * got.numel()


def test_plane_larger_than_2_23(hip, monkeypatch):
    """the label kernel's offsets are 24-bit multiply-adds; the signed form (v_mad_i32_i24) sign-extended the
    inner product ix * J + iy from 2^23 on — a 2-D slice of 2900 x 2900 (I * J = 8.4e6 > 2^23) lost every "decided" voxel of
    its upper half to offset 0.  Unsigned multiplicands now; compared with the all-exact road."""
    shape = (2900, 2900, 2)
    seg = _labels((1, 1, *shape), torch.uint8, 51)
    mapping = torch.eye(3, 4)[None].clone()
    mapping[0, 0, 3], mapping[0, 1, 3] = 1.25, -0.75
    mapping[0, 0, 1], mapping[0, 1, 0] = 0.01, -0.01
    kwargs = dict(out_shape=shape, mapping=mapping.cuda(), control_points=None, in_spacing=(1, 1, 1), out_spacing=(1, 1, 1), affine_first=True,
                  interps=["nearest"], fills=[None])
    monkeypatch.setenv("TIO_NEAREST_KERNEL", "0")
    reference = hip.resample3d([seg], precision="exact", **kwargs)[0]
    monkeypatch.setenv("TIO_NEAREST_KERNEL", "1")
    got = hip.resample3d([seg], precision="exact", **kwargs)[0]
    torch.cuda.synchronize()
    assert torch.equal(reference, got)
    upper = got[0, 0, 2000:]  # rows whose ix * J lies beyond 2^23
    assert int((upper != 0).sum()) > upper.numel() // 2


# ---- nearest images WITH a fill rule (round 4): `mask > 0.5 ? nearest tap : fill`, the mask trilinear (spatial.py:1719-1728) -----

@pytest.mark.parametrize("dtype", [torch.uint8, torch.int16, torch.int32, torch.int64, torch.float32, torch.float16, torch.float64])
@pytest.mark.parametrize("elastic", [False, True])
def test_nearest_images_with_a_fill_value_match_the_oracle(oracle, hip, dtype, elastic):
    batch, shape = 2, (36, 30, 70)
    data = _labels((batch, 2, *shape), dtype, 61, device="cpu")
    kwargs = dict(
        out_shape=shape, mapping=_mapping(batch, 62, scale=0.1, shift=5.0),
        control_points=_control_points(batch, (4, 4, 4), 63, amplitude=4.0) if elastic else None, in_spacing=(1, 1, 1), out_spacing=(1, 1, 1),
        affine_first=True, interps=["nearest"], fills=[torch.tensor([9.0, 11.0])],
    )
    cpu, gpu = _both(oracle, hip, "resample3d", ([data],), **kwargs)
    assert torch.equal(cpu[0], gpu[0].cpu())
    assert int((gpu[0][:, 0] == 9).sum()) > 100 and int((gpu[0][:, 1] == 11).sum()) > 100  # the fill value does show


@pytest.mark.parametrize("size", [256])
@pytest.mark.parametrize("elastic", [False, True])
def test_full_size_label_maps_with_a_pad_label_are_bit_identical_to_the_exact_road(hip, monkeypatch, size, elastic):
    batch = 2
    seg = _labels((batch, 1, size, size, size), torch.int16, 71)
    kwargs = dict(
        out_shape=(size, size, size), mapping=_mapping(batch, 72, scale=0.1, shift=6.0).cuda(),
        control_points=_control_points(batch, (7, 7, 7), 73, amplitude=6.0).cuda() if elastic else None, in_spacing=(1, 1, 1),
        out_spacing=(1, 1, 1), affine_first=True, interps=["nearest"], fills=[torch.tensor([200.0], device="cuda")],
    )
    monkeypatch.setenv("TIO_NEAREST_KERNEL", "0")
    reference = hip.resample3d([seg], precision="exact", **kwargs)[0]
    monkeypatch.setenv("TIO_NEAREST_KERNEL", "1")
    got = hip.resample3d([seg], precision="exact", **kwargs)[0]
    torch.cuda.synchronize()
    assert int((reference == 200).sum()) > 10_000
    assert torch.equal(reference, got), int((reference != got).sum())


def test_label_fill_decisions_on_a_plane_that_sits_on_the_threshold(hip, monkeypatch):
    """The geometry of test_gpu_resample_planned.py: a whole output plane with in-bounds weights 0.50013 ... 0.49987."""
    batch, size = 2, 128
    seg = _labels((batch, 1, size, size, size), torch.uint8, 81) + 1  # (labels 1 ... 7: the zero padding is recognisable too)
    mapping = torch.eye(3, 4).repeat(batch, 1, 1)
    for b in range(batch):
        mapping[b, 0, 1] = 2e-6 * (b + 1)
        mapping[b, 0, 3] = 0.5 - 1.3e-4 * (b + 1)
        mapping[b, 1, 3], mapping[b, 2, 3] = 0.37, -0.21
    kwargs = dict(out_shape=(size, size, size), mapping=mapping.cuda(), control_points=None, in_spacing=(1, 1, 1), out_spacing=(1, 1, 1),
                  affine_first=True, interps=["nearest"], fills=[torch.tensor([99.0], device="cuda")])
    monkeypatch.setenv("TIO_NEAREST_KERNEL", "0")
    reference = hip.resample3d([seg], precision="exact", **kwargs)[0]
    monkeypatch.setenv("TIO_NEAREST_KERNEL", "1")
    got = hip.resample3d([seg], precision="exact", **kwargs)[0]
    torch.cuda.synchronize()
    last = reference[:, 0, -1]
    assert 0.05 < float((last == 99).float().mean()) < 0.95  # both answers occur on the plane
    assert torch.equal(reference, got), int((reference != got).sum())


@pytest.mark.parametrize("precision", ["exact", "fast"])
def test_a_label_map_with_a_pad_label_no_longer_holds_the_float_images_back(hip, monkeypatch, precision):
    """Config 5's shape with `default_pad_label != 0`: two float images and a label map WITH a fill value in one call.  The
    labels are bit-identical to the all-exact road in both precision modes; in fast mode the float images run the FAST
    kernels (not bit-identical to the exact ones, within their tolerance, fill decisions identical)."""
    batch, shape = 3, (128, 128, 128)
    g = torch.Generator(device="cuda").manual_seed(91)
    t1 = torch.rand(batch, 1, *shape, generator=g, device="cuda")
    t2 = torch.rand(batch, 1, *shape, generator=g, device="cuda") + 1
    seg = _labels((batch, 1, *shape), torch.int16, 93)
    kwargs = dict(
        out_shape=shape, mapping=_mapping(batch, 95, scale=0.08, shift=4.0).cuda(), control_points=_control_points(batch, (7, 7, 7), 97, amplitude=5.0).cuda(),
        in_spacing=(1, 1, 1), out_spacing=(1, 1, 1), affine_first=True, interps=["linear", "linear", "nearest"],
        fills=[torch.tensor([-50.0], device="cuda"), torch.tensor([-60.0], device="cuda"), torch.tensor([5.0], device="cuda")],
    )
    monkeypatch.setenv("TIO_NEAREST_KERNEL", "0")
    reference = hip.resample3d([t1, t2, seg], precision="exact", **kwargs)
    monkeypatch.setenv("TIO_NEAREST_KERNEL", "1")
    monkeypatch.setenv("TIO_FAST_KERNEL", "planned")
    got = hip.resample3d([t1, t2, seg], precision=precision, **kwargs)
    torch.cuda.synchronize()
    assert torch.equal(reference[2], got[2])
    for r, o, fill in zip(reference[:2], got[:2], (-50.0, -60.0)):
        if precision == "exact":
            assert torch.equal(r, o)
        else:
            assert not torch.equal(r, o)  # the FAST kernels did run
            assert int(((r == fill) != (o == fill)).sum()) == 0
            assert float((r.double() - o.double()).abs().max()) <= 1e-4


@pytest.mark.parametrize("seed", range(12))
def test_label_maps_with_a_pad_label_fuzz(hip, monkeypatch, seed):
    """Random geometries (zooms 0.5 - 2, rotations up to +-40 degrees, shifts, anisotropic spacings, both composition orders,
    several control grids, ragged shapes, three dtypes): label maps with a fill value are bit-identical to the all-exact road."""
    import math

    g = torch.Generator().manual_seed(3000 + seed)

    def rnd(lo, hi):
        return lo + (hi - lo) * float(torch.rand(1, generator=g))

    batch = 2
    shape = (int(rnd(30, 80)), int(rnd(30, 80)), int(rnd(30, 140)))
    dtype = [torch.uint8, torch.int16, torch.float32][seed % 3]
    seg = _labels((batch, 1, *shape), dtype, 3100 + seed)
    mapping = torch.zeros(batch, 3, 4)
    for b in range(batch):
        rot = torch.eye(3, dtype=torch.float64)
        for axis in range(3):
            a = math.radians(rnd(-40, 40))
            c, s = math.cos(a), math.sin(a)
            i, j = [(1, 2), (0, 2), (0, 1)][axis]
            r = torch.eye(3, dtype=torch.float64)
            r[i, i], r[i, j], r[j, i], r[j, j] = c, -s, s, c
            rot = rot @ r
        lin = rot @ torch.diag(torch.tensor([rnd(0.5, 2.0) for _ in range(3)], dtype=torch.float64))
        centre = torch.tensor([(s - 1) / 2 for s in shape], dtype=torch.float64)
        mapping[b, :, :3] = lin.float()
        mapping[b, :, 3] = (centre - lin @ centre + torch.tensor([rnd(-10, 10) for _ in range(3)], dtype=torch.float64)).float()
    elastic = seed % 4 != 0
    spacing = tuple(rnd(0.6, 1.8) for _ in range(3)) if seed % 3 == 1 else (1, 1, 1)
    kwargs = dict(
        out_shape=shape, mapping=mapping.cuda(),
        control_points=_control_points(batch, [(4, 4, 4), (6, 5, 7), (7, 7, 7)][seed % 3], 3200 + seed, amplitude=rnd(1.0, 7.0)).cuda() if elastic else None,
        in_spacing=spacing, out_spacing=spacing, affine_first=bool(seed % 2), interps=["nearest"], fills=[torch.tensor([42.0], device="cuda")],
    )
    monkeypatch.setenv("TIO_NEAREST_KERNEL", "0")
    reference = hip.resample3d([seg], precision="exact", **kwargs)[0]
    monkeypatch.setenv("TIO_NEAREST_KERNEL", "1")
    got = hip.resample3d([seg], precision="fast", **kwargs)[0]
    torch.cuda.synchronize()
    assert torch.equal(reference, got), (int((reference != got).sum()), shape, dtype, elastic)
