"""Test-only helpers: run the host-side transforms on a chosen engine and compare.

``use_engine`` swaps the engine that ``torchio_amd.ops.engine()`` returns.  It
exists ONLY here, in tests/: the product has a single engine (HIP) and no switch.
On a CPU box it lets the host logic (sampling order, params, history, fill /
passthrough handling) be exercised end to end against the CPU oracle.
"""
from __future__ import annotations

import contextlib
import copy

import torch

import torchio_amd as tio
from torchio_amd import ops


@contextlib.contextmanager
def use_engine(engine):
    previous = ops._ENGINE
    ops._ENGINE = engine
    try:
        yield engine
    finally:
        ops._ENGINE = previous


def nested_spheres(size: int, dtype=torch.int16) -> torch.Tensor:
    """Label map with values {0..4}: nested spheres."""
    axis = torch.arange(size, dtype=torch.float32) - (size - 1) / 2
    i, j, k = torch.meshgrid(axis, axis, axis, indexing="ij")
    dist = torch.sqrt(i * i + j * j + k * k)
    label = sum((dist <= r * size).to(torch.int32) for r in (0.45, 0.35, 0.25, 0.15))
    return label.to(dtype).unsqueeze(0)


def make_subjects(size: int, batch: int, seed: int, *, with_label: bool = True, second_modality: bool = False):
    g = torch.Generator().manual_seed(seed)
    subjects = []
    for _ in range(batch):
        entries = {"t1": tio.ScalarImage(torch.rand(1, size, size, size, generator=g))}
        if second_modality:
            entries["t2"] = tio.ScalarImage(torch.rand(1, size, size, size, generator=g) + 1)
        if with_label:
            entries["seg"] = tio.LabelMap(nested_spheres(size))
        subjects.append(tio.Subject(**entries))
    return subjects


def benchmark_compose(per_instance: bool = True) -> tio.Compose:
    """The metric's pipeline with the explicit ranges of SURVEY.md §8(d)."""
    return tio.Compose(
        [
            tio.Affine(degrees=(-10, 10), scales=(0.9, 1.1), translation=(-5, 5), per_instance=per_instance),
            tio.ElasticDeformation(per_instance=per_instance),
            tio.BiasField(per_instance=per_instance),
            tio.Blur(std=(0.5, 2), per_instance=per_instance),
            tio.Noise(per_instance=per_instance),
        ]
    )


def compare(reference: torch.Tensor, candidate: torch.Tensor) -> dict:
    reference, candidate = reference.cpu(), candidate.cpu()
    if not reference.dtype.is_floating_point:
        return {"mismatches": int((reference != candidate).sum())}
    diff = (reference.double() - candidate.double()).abs()
    # max_rel is relative to max(|ref|, 1): on the unit-range test volumes that is an ABSOLUTE bar in units of the intensity
    # range, and is reported as such (tests/test_gpu_full_size.py::test_headline_mode_256_matches_oracle checks scaled data)
    scale = reference.double().abs().clamp_min(1.0)
    return {"max_abs": float(diff.max()), "max_rel": float((diff / scale).max())}


def run_compose_parity(size: int, batch: int, seed: int, device: str) -> dict:
    """Same seeded Compose through the HIP engine (*device*) and the CPU oracle; report deviations."""
    from oracle.oracle import oracle_engine  # noqa: PLC0415

    subjects = make_subjects(size, batch, seed)
    transform = benchmark_compose()
    cpu_batch = tio.SubjectsBatch.from_subjects(copy.deepcopy(subjects))
    gpu_batch = tio.SubjectsBatch.from_subjects(copy.deepcopy(subjects)).to(device)
    torch.manual_seed(seed)
    with use_engine(oracle_engine()):
        expected = transform(cpu_batch)
    torch.manual_seed(seed)
    actual = transform(gpu_batch)
    torch.cuda.synchronize()
    assert [t.params for t in expected.applied_transforms] == [t.params for t in actual.applied_transforms]
    intensity = compare(expected.t1.data, actual.t1.data)
    labels = compare(expected.seg.data, actual.seg.data)
    return {
        "size": size,
        "batch": batch,
        "label_mismatches": labels["mismatches"],
        "max_rel_err": intensity["max_rel"],
        "max_abs_err": intensity["max_abs"],
    }
