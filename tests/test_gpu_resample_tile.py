"""GPU: the LDS-staged brick path of tio_resample3d vs the CPU oracle and vs the gather path.

The default path for launches with a trilinear image is the brick ("tile") kernel;
``TIO_RESAMPLE_PATH=gather|tile`` pins one of them (read by the launcher per call).
Bars: bit-exact against the oracle for every dtype (the tile path keeps the oracle's
float32 operation order); tile == gather bit for bit at the bench size, where the
oracle would be slow.  The stand-alone C++ driver (tests/native) repeats the sweep
through the bare C ABI without Python in the loop.
"""
from __future__ import annotations

import os
import subprocess

import pytest
import torch

from test_gpu_ops_parity import _both
from test_gpu_ops_parity import _control_points
from test_gpu_ops_parity import _data
from test_gpu_ops_parity import _mapping

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def resample_path():
    """Pin the launcher's path for one test and restore the default afterwards."""
    previous = os.environ.get("TIO_RESAMPLE_PATH")

    def pin(value):
        if value is None:
            os.environ.pop("TIO_RESAMPLE_PATH", None)
        else:
            os.environ["TIO_RESAMPLE_PATH"] = value

    yield pin
    pin(previous)


SHAPES = [
    ((40, 37, 75), (40, 37, 75)),  # K % 4 != 0: scalar staging, partial bricks on every axis
    ((64, 48, 96), (64, 48, 96)),  # aligned rows: 16-byte staging
    ((33, 70, 52), (48, 40, 64)),  # different output grid
]


@pytest.mark.parametrize("in_shape,out_shape", SHAPES)
@pytest.mark.parametrize("elastic", [False, True])
@pytest.mark.parametrize("with_fill", [False, True])
@pytest.mark.parametrize("path", ["tile", "gather"])
def test_tile_path_matches_oracle_bit_exact(oracle, hip, resample_path, in_shape, out_shape, elastic, with_fill, path):
    resample_path(path)
    batch, channels = 2, 2
    data = _data((batch, channels, *in_shape), torch.float32, 21)
    mapping = _mapping(batch, 22, scale=0.12, shift=4.0)
    # rescale the mapping when the grids differ so that the output still looks at the volume
    for axis in range(3):
        mapping[:, :, axis] *= in_shape[axis] / out_shape[axis]
    kwargs = dict(
        out_shape=out_shape,
        mapping=mapping,
        control_points=_control_points(batch, (7, 7, 7), 23, amplitude=5.0) if elastic else None,
        in_spacing=(1.0, 1.0, 1.0),
        out_spacing=(1.0, 1.0, 1.0),
        affine_first=True,
        interps=["linear"],
        fills=[torch.tensor([-1.0, 0.5]) if with_fill else None],
    )
    cpu, gpu = _both(oracle, hip, "resample3d", ([data],), **kwargs)
    assert torch.equal(cpu[0], gpu[0].cpu())


def test_tile_far_out_of_view_and_anisotropic(oracle, hip, resample_path):
    """Bricks fully outside / straddling the volume, mm spacings != 1, affine after elastic."""
    resample_path("tile")
    data = _data((2, 1, 48, 56, 64), torch.float32, 31)
    mapping = _mapping(2, 32, scale=0.5, shift=25.0)
    kwargs = dict(
        out_shape=(48, 56, 64),
        mapping=mapping,
        control_points=_control_points(2, (7, 6, 5), 33, amplitude=6.0),
        in_spacing=(1.0, 1.25, 0.8),
        out_spacing=(0.9, 1.1, 0.75),
        affine_first=False,
        interps=["linear"],
        fills=[torch.tensor([7.0])],
    )
    cpu, gpu = _both(oracle, hip, "resample3d", ([data],), **kwargs)
    assert torch.equal(cpu[0], gpu[0].cpu())


def test_tile_oversize_boxes_split_or_fall_back(oracle, hip, resample_path):
    """Down-sampling by 3: a brick's input box exceeds the LDS budget → 2/4 passes or per-voxel gather."""
    resample_path("tile")
    data = _data((1, 1, 120, 120, 120), torch.float32, 41)
    mapping = _mapping(1, 42, scale=0.05, shift=2.0) * 1.0
    mapping[:, :, :3] *= 3.0
    kwargs = dict(
        out_shape=(40, 40, 40), mapping=mapping, control_points=None, in_spacing=(1, 1, 1), out_spacing=(3, 3, 3),
        affine_first=True, interps=["linear"], fills=[torch.tensor([0.25])],
    )
    cpu, gpu = _both(oracle, hip, "resample3d", ([data],), **kwargs)
    assert torch.equal(cpu[0], gpu[0].cpu())


def test_tile_subject_multimodal_labels_bit_exact(oracle, hip, resample_path):
    """t1, t2 float32 (trilinear, staged) + int16 labels (nearest, gathered) through one launch."""
    resample_path(None)  # default path selection
    batch, shape = 2, (64, 64, 64)
    t1 = _data((batch, 1, *shape), torch.float32, 51)
    t2 = _data((batch, 1, *shape), torch.float32, 52) + 1
    seg = _data((batch, 1, *shape), torch.int16, 53)
    kwargs = dict(
        out_shape=shape, mapping=_mapping(batch, 54, scale=0.1), control_points=_control_points(batch, (7, 7, 7), 55),
        in_spacing=(1, 1, 1), out_spacing=(1, 1, 1), affine_first=True, interps=["linear", "linear", "nearest"],
        fills=[torch.tensor([0.0]), torch.tensor([1.0]), None],
    )
    cpu, gpu = _both(oracle, hip, "resample3d", ([t1, t2, seg],), **kwargs)
    for c, g in zip(cpu, gpu, strict=True):
        assert torch.equal(c, g.cpu())


@pytest.mark.parametrize("elastic", [False, True])
def test_tile_equals_gather_at_bench_size(hip, resample_path, elastic):
    """256^3, per-element geometry: the two paths agree bit for bit (size-independent property)."""
    batch, shape = 2, (256, 256, 256)
    g = torch.Generator(device="cuda").manual_seed(61)
    data = torch.rand(batch, 1, *shape, generator=g, device="cuda") * 4 - 1
    kwargs = dict(
        out_shape=shape,
        mapping=_mapping(batch, 62, scale=0.08, shift=5.0).cuda(),
        control_points=_control_points(batch, (7, 7, 7), 63, amplitude=7.5).cuda() if elastic else None,
        in_spacing=(1, 1, 1), out_spacing=(1, 1, 1), affine_first=True, interps=["linear"],
        fills=[torch.tensor([-1.0], device="cuda")],
    )
    resample_path("gather")
    expected = hip.resample3d([data], **kwargs)[0]
    resample_path("tile")
    actual = hip.resample3d([data], **kwargs)[0]
    torch.cuda.synchronize()
    assert torch.equal(expected, actual)


def _same_bits_or_both_nan(a, b):
    return bool(((a == b) | (a.isnan() & b.isnan())).all())


@pytest.mark.parametrize("poison", ["nan", "inf", "huge"])
@pytest.mark.parametrize("with_fill", [False, True])
def test_tile_non_finite_control_points_match_gather(oracle, hip, resample_path, poison, with_fill):
    """Bricks whose displacement is NaN / Inf / absurd leave the brick path; the others stay on it."""
    batch, shape = 2, (64, 48, 96)
    data = _data((batch, 1, *shape), torch.float32, 71)
    control = _control_points(batch, (7, 7, 7), 72, amplitude=5.0)
    value = {"nan": float("nan"), "inf": float("inf"), "huge": 3.0e35}[poison]
    control[0, 3, 3, 3, 1] = value
    control[1, 2, 4, 1, 0] = -value
    kwargs = dict(
        out_shape=shape, mapping=_mapping(batch, 73, scale=0.1, shift=3.0), control_points=control,
        in_spacing=(1, 1, 1), out_spacing=(1, 1, 1), affine_first=True, interps=["linear"],
        fills=[torch.tensor([-2.0]) if with_fill else None],
    )
    resample_path("gather")
    cpu, expected = _both(oracle, hip, "resample3d", ([data],), **kwargs)
    resample_path("tile")
    _, actual = _both(oracle, hip, "resample3d", ([data],), **kwargs)
    assert _same_bits_or_both_nan(expected[0], actual[0])
    assert _same_bits_or_both_nan(cpu[0], actual[0].cpu())


def test_native_driver_parity_sweep():
    """tests/native/resample_bench --cases parity: every path vs the oracle through the bare C ABI."""
    binary = os.path.join(ROOT, "tests", "native", "_build", "resample_bench")
    if not os.path.isfile(binary):
        subprocess.run([os.path.join(ROOT, "tests", "native", "build.sh")], check=True)
    result = subprocess.run([binary, "--cases", "parity"], capture_output=True, text=True, timeout=600)
    assert result.returncode == 0, result.stdout[-4000:] + result.stderr[-2000:]
    assert "failures: 0" in result.stdout


# -- the opt-in fast intensity path (TIO_PRECISION_FAST) ----------------------------------------
NORTH_STAR_REL_TOL = 1e-4  # BASELINE.json: float intensities within 1e-4 relative


@pytest.mark.parametrize("elastic", [False, True])
@pytest.mark.parametrize("with_fill", [False, True])
def test_fast_precision_stays_within_the_north_star_tolerance(hip, elastic, with_fill):
    """Fast vs exact on white noise (the worst case: unit gradients everywhere), 2 x 128^3, per-element geometry."""
    batch, shape = 2, (128, 128, 128)
    g = torch.Generator(device="cuda").manual_seed(71)
    data = torch.rand(batch, 1, *shape, generator=g, device="cuda") * 4 - 1
    kwargs = dict(
        out_shape=shape, mapping=_mapping(batch, 72, scale=0.08, shift=5.0).cuda(),
        control_points=_control_points(batch, (7, 7, 7), 73, amplitude=7.5).cuda() if elastic else None,
        in_spacing=(1, 1, 1), out_spacing=(1, 1, 1), affine_first=True, interps=["linear"],
        fills=[torch.tensor([-1.0], device="cuda") if with_fill else None],
    )
    exact = hip.resample3d([data], precision="exact", **kwargs)[0]
    fast = hip.resample3d([data], precision="fast", **kwargs)[0]
    torch.cuda.synchronize()
    assert not torch.equal(exact, fast)  # it IS a different rounding sequence
    rel = ((exact - fast).abs() / exact.abs().clamp_min(1.0)).max().item()
    # a voxel whose in-bounds weight sum sits within rounding of 0.5 may flip between sample and fill: exclude
    # none here - the fill decision uses the same coordinates in both kernels up to a few ulps, count disagreements
    assert rel <= NORTH_STAR_REL_TOL or with_fill, f"relative error {rel:.3g}"
    if with_fill:
        disagree = ((exact == -1.0) != (fast == -1.0)).sum().item()
        inside = (exact != -1.0) & (fast != -1.0)
        rel_inside = ((exact - fast).abs() / exact.abs().clamp_min(1.0))[inside].max().item()
        assert rel_inside <= NORTH_STAR_REL_TOL and disagree <= 8, (rel_inside, disagree)


def test_fast_precision_never_touches_label_maps(oracle, hip, monkeypatch):
    """A nearest image is bit-identical to the reference in either precision mode.  With its own kernel
    (csrc/resample_nearest.hpp) it no longer pins the exact coordinates for the float image of the call: that one is FAST
    (within tolerance); without the kernel (TIO_NEAREST_KERNEL=0) the whole launch stays exact, as before round 3."""
    batch, shape = 2, (48, 40, 56)
    t1 = _data((batch, 1, *shape), torch.float32, 81)
    seg = _data((batch, 1, *shape), torch.int16, 82)
    kwargs = dict(
        out_shape=shape, mapping=_mapping(batch, 83, scale=0.1), control_points=_control_points(batch, (7, 7, 7), 84),
        in_spacing=(1, 1, 1), out_spacing=(1, 1, 1), affine_first=True, interps=["linear", "nearest"], fills=[torch.tensor([0.5]), None],
        precision="fast",
    )
    cpu, gpu = _both(oracle, hip, "resample3d", ([t1, seg],), **kwargs)
    assert torch.equal(cpu[1], gpu[1].cpu())
    assert int(((cpu[0].double() - gpu[0].cpu().double()).abs() > 1e-4).sum()) <= 8  # (fill decisions within rounding of 0.5)
    monkeypatch.setenv("TIO_NEAREST_KERNEL", "0")
    cpu, gpu = _both(oracle, hip, "resample3d", ([t1, seg],), **kwargs)
    assert torch.equal(cpu[0], gpu[0].cpu()) and torch.equal(cpu[1], gpu[1].cpu())
