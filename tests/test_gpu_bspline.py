"""GPU: the B-spline kernels (tio_bspline_prefilter, TIO_QUADRATIC / TIO_CUBIC images of tio_resample3d) against the CPU
oracle — same operations in the same order, so bit for bit.  What pins the oracle itself: tests/test_bspline.py."""
from __future__ import annotations

import copy

import pytest
import torch

import torchio_amd as tio
from parity_harness import use_engine
from test_gpu_ops_parity import _control_points
from test_gpu_ops_parity import _mapping

pytestmark = pytest.mark.gpu
ORDERS = {"quadratic": 2, "cubic": 3, "fourth": 4, "fifth": 5, "sixth": 6, "seventh": 7}
NAMES = list(ORDERS)


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.int16, torch.uint8])
@pytest.mark.parametrize("shape", [(24, 20, 28), (5, 3, 2), (40, 1, 17)])
def test_prefilter_is_bit_identical_to_the_oracle(oracle, hip, name, dtype, shape):
    g = torch.Generator().manual_seed(7)
    x = (torch.rand(2, 2, *shape, generator=g) * 40 - 10).to(dtype)
    want = oracle.bspline_prefilter(x, ORDERS[name])
    got = hip.bspline_prefilter(x.cuda(), ORDERS[name])
    torch.cuda.synchronize()
    assert got.dtype == torch.float32 and torch.equal(want, got.cpu())


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("elastic", [False, True])
@pytest.mark.parametrize("affine_first", [True, False])
def test_sampling_is_bit_identical_to_the_oracle(oracle, hip, name, elastic, affine_first):
    batch, shape, out_shape = 2, (30, 26, 34), (28, 30, 33)
    g = torch.Generator().manual_seed(8)
    coefficients = torch.rand(batch, 2, *shape, generator=g) * 2 - 0.5
    kwargs = dict(
        out_shape=out_shape, mapping=_mapping(batch, 9, scale=0.12, shift=3.0),
        control_points=_control_points(batch, (5, 6, 7), 10, amplitude=3.0) if elastic else None,
        in_spacing=(1.0, 1.5, 0.8), out_spacing=(1.1, 1.2, 0.9), affine_first=affine_first, interps=[name], fills=[None],
    )
    want = oracle.resample3d([coefficients], **kwargs)[0]
    moved = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in kwargs.items()}
    got = hip.resample3d([coefficients.cuda()], **moved)[0]
    torch.cuda.synchronize()
    assert (want == 0).any() and (want != 0).any()  # the mask is exercised
    assert torch.equal(want, got.cpu())


def test_spline_images_share_a_call_with_the_others(oracle, hip):
    """One tio_resample3d call: a trilinear image, a label map and a cubic one — three launches behind it."""
    batch, shape = 2, (20, 22, 24)
    g = torch.Generator().manual_seed(11)
    t1 = torch.rand(batch, 1, *shape, generator=g)
    seg = (torch.rand(batch, 1, *shape, generator=g) * 4).to(torch.int16)
    coefficients = oracle.bspline_prefilter(torch.rand(batch, 1, *shape, generator=g), 3)
    kwargs = dict(
        out_shape=shape, mapping=_mapping(batch, 12, scale=0.08), control_points=None, in_spacing=(1, 1, 1), out_spacing=(1, 1, 1),
        affine_first=True, interps=["linear", "nearest", "cubic"], fills=[torch.tensor([0.25]), None, None],
    )
    want = oracle.resample3d([t1, seg, coefficients], **kwargs)
    moved = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in kwargs.items()}
    moved["fills"] = [f.cuda() if f is not None else None for f in kwargs["fills"]]
    got = hip.resample3d([t1.cuda(), seg.cuda(), coefficients.cuda()], **moved)
    torch.cuda.synchronize()
    for w, h in zip(want, got):
        assert torch.equal(w, h.cpu())


@pytest.mark.parametrize("name", NAMES)
def test_transform_with_spline_interpolation_matches_the_oracle(oracle, hip, name):
    g = torch.Generator().manual_seed(13)
    subjects = [tio.Subject(t1=tio.ScalarImage(torch.rand(1, 32, 30, 36, generator=g))) for _ in range(3)]
    transform = tio.Compose([
        tio.Affine(degrees=(-10, 10), scales=(0.9, 1.1), translation=(-2, 2), image_interpolation=name, per_instance=True, p=0.7),
        tio.ElasticDeformation(image_interpolation=name, per_instance=True),
    ])
    torch.manual_seed(14)
    with use_engine(oracle):
        want = transform(tio.SubjectsBatch.from_subjects(copy.deepcopy(subjects)))
    torch.manual_seed(14)
    got = transform(tio.SubjectsBatch.from_subjects(subjects).to("cuda"))
    torch.cuda.synchronize()
    assert torch.equal(want.t1.data, got.t1.data.cpu())
