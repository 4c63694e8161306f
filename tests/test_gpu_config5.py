"""GPU: BASELINE.json config 5 — multi-modal Subject (2 x float32 + 1 x int16 label map) through
one fused ``tio.Spatial(affine + elastic)`` with trilinear / nearest interpolation.

256^3 is compared with the CPU oracle (labels bit-exact, intensities bit-exact as well since
the kernels keep the oracle's operation order); 512^3 — the configured size — is checked through
size-independent properties (tile path == gather path bit for bit, label values stay in the
input's label set, an identity Spatial reproduces the input) because the oracle would need
minutes there.
"""
from __future__ import annotations

import copy
import os

import pytest
import torch

import torchio_amd as tio
from parity_harness import nested_spheres
from parity_harness import use_engine

pytestmark = pytest.mark.gpu


def _subject(size: int, seed: int) -> tio.Subject:
    g = torch.Generator().manual_seed(seed)
    return tio.Subject(
        t1=tio.ScalarImage(torch.rand(1, size, size, size, generator=g)),
        t2=tio.ScalarImage(torch.rand(1, size, size, size, generator=g) + 1),
        seg=tio.LabelMap(nested_spheres(size)),
    )


def _spatial() -> tio.Spatial:
    return tio.Spatial(degrees=(-10, 10), scales=(0.9, 1.1), translation=(-5, 5), max_displacement=7.5)


def test_config5_256_matches_oracle(oracle, hip):
    subject = _subject(256, 0)
    transform = _spatial()
    cpu = tio.SubjectsBatch.from_subjects([copy.deepcopy(subject)])
    gpu = tio.SubjectsBatch.from_subjects([copy.deepcopy(subject)]).to("cuda")
    torch.manual_seed(5)
    with use_engine(oracle):
        expected = transform(cpu)
    torch.manual_seed(5)
    actual = transform(gpu)
    torch.cuda.synchronize()
    assert torch.equal(expected.seg.data, actual.seg.data.cpu()), "label map not bit-exact"
    for name in ("t1", "t2"):
        assert torch.equal(expected.images[name].data, actual.images[name].data.cpu()), name


def test_config5_512_properties(hip):
    subject = _subject(512, 1)
    batch = tio.SubjectsBatch.from_subjects([subject]).to("cuda")
    transform = _spatial()
    previous = os.environ.get("TIO_RESAMPLE_PATH")
    try:
        os.environ["TIO_RESAMPLE_PATH"] = "gather"
        torch.manual_seed(9)
        by_gather = transform(batch)
        os.environ["TIO_RESAMPLE_PATH"] = "tile"
        torch.manual_seed(9)
        by_tile = transform(batch)
    finally:
        if previous is None:
            os.environ.pop("TIO_RESAMPLE_PATH", None)
        else:
            os.environ["TIO_RESAMPLE_PATH"] = previous
    torch.cuda.synchronize()
    for name in ("t1", "t2", "seg"):
        assert torch.equal(by_gather.images[name].data, by_tile.images[name].data), name
    labels = set(torch.unique(by_tile.seg.data).tolist())
    assert labels <= {0, 1, 2, 3, 4} and by_tile.seg.data.dtype == torch.int16
    # identity geometry reproduces the input exactly (nearest and trilinear)
    identity = tio.Spatial(degrees=0, scales=1, translation=0)(batch)
    for name in ("t1", "t2", "seg"):
        assert torch.equal(identity.images[name].data, batch.images[name].data), name
