"""GPU: parity with the CPU oracle AT THE SIZES BASELINE.json states.

* config 3: ``Compose[Affine, ElasticDeformation, BiasField, Blur, Noise]`` on a batch of 2 x 256^3
  (float32 intensity + int16 label map), noise in the reference-identical mode: label maps bit-exact,
  intensities within the north_star's 1e-4 relative (measured ~2e-6: only exp differs).
* config 5: the multi-modal subject (2 x float32 + 1 label map) at **512^3** through one fused
  ``tio.Spatial(affine + elastic)``, labels as int16 and as int32: label maps bit-exact, and — because the exact
  kernels keep the reference's operation order — the trilinear intensities bit-exact too.
* the opt-in fast intensity path at 256^3: within 1e-4 relative of the oracle, labels untouched.

The oracle (oracle/libtio_oracle.so, OpenMP) needs a few seconds per case on the GPU box's host cores.
"""
from __future__ import annotations

import copy

import pytest
import torch

import torchio_amd as tio
from parity_harness import nested_spheres
from parity_harness import run_compose_parity
from parity_harness import use_engine

pytestmark = pytest.mark.gpu


def test_config3_compose_256_batch2_matches_oracle(oracle, hip):
    assert tio.get_noise_rng() == "reference"  # the reference-identical stream (seeded CPU mt19937 draws)
    report = run_compose_parity(size=256, batch=2, seed=3, device="cuda")
    assert report["label_mismatches"] == 0, report
    assert report["max_rel_err"] <= 1e-4, report  # north_star tolerance
    assert report["max_rel_err"] <= 1e-5, report  # what the kernels actually deliver (exp / summation order only)


def test_config3_compose_256_batch2_scanner_range_true_relative_error(oracle, hip):
    """The same configuration — the LIBRARY DEFAULT mode (reference noise stream, exact resamplers, exact stencil) at the
    BASELINE size — on data of a 12-bit scanner range, with a true PER-VOXEL relative metric:
    |delta| / max(|reference|, 1e-3 range) <= 1e-4 for every voxel (`max(|ref|, 1)` is an absolute
    bar on unit-range data)."""
    from parity_harness import benchmark_compose, make_subjects  # noqa: PLC0415

    assert tio.get_noise_rng() == "reference" and tio.get_resample_precision() == "exact" and tio.get_stencil_precision() == "exact"
    size, batch, scale = 256, 2, 4095.0
    subjects = make_subjects(size, batch, seed=5)
    for subject in subjects:
        subject.t1.set_data(subject.t1.data * scale)
    transform = benchmark_compose()
    cpu = tio.SubjectsBatch.from_subjects(copy.deepcopy(subjects))
    gpu = tio.SubjectsBatch.from_subjects(subjects).to("cuda")
    torch.manual_seed(6)
    with use_engine(oracle):
        expected = transform(cpu)
    torch.manual_seed(6)
    actual = transform(gpu)
    torch.cuda.synchronize()
    assert int((expected.seg.data != actual.seg.data.cpu()).sum()) == 0
    want, got = expected.t1.data.double(), actual.t1.data.cpu().double()
    value_range = float(want.max() - want.min())
    assert value_range > 1000.0  # scanner scale indeed
    rel = (want - got).abs() / want.abs().clamp_min(1e-3 * value_range)
    assert float(rel.max()) <= 1e-4, {"max_rel": float(rel.max()), "range": value_range, "beyond": int((rel > 1e-4).sum())}
    assert float(rel.max()) <= 2e-5, float(rel.max())  # what the kernels deliver: exp and a summation order


@pytest.mark.parametrize("label_dtype", [torch.int16, torch.int32])
def test_config5_512_matches_oracle(oracle, hip, label_dtype):
    size = 512
    g = torch.Generator().manual_seed(11)
    subject = tio.Subject(
        t1=tio.ScalarImage(torch.rand(1, size, size, size, generator=g)),
        t2=tio.ScalarImage(torch.rand(1, size, size, size, generator=g) + 1),
        seg=tio.LabelMap(nested_spheres(size, dtype=label_dtype)),
    )
    transform = tio.Spatial(degrees=(-10, 10), scales=(0.9, 1.1), translation=(-5, 5), max_displacement=7.5)
    cpu = tio.SubjectsBatch.from_subjects([copy.deepcopy(subject)])
    gpu = tio.SubjectsBatch.from_subjects([subject]).to("cuda")
    torch.manual_seed(12)
    with use_engine(oracle):
        expected = transform(cpu)
    torch.manual_seed(12)
    actual = transform(gpu)
    torch.cuda.synchronize()
    assert actual.seg.data.dtype == label_dtype
    mismatches = int((expected.seg.data != actual.seg.data.cpu()).sum())
    assert mismatches == 0, f"{mismatches} label voxels differ from the oracle at 512^3 ({label_dtype})"
    assert set(torch.unique(actual.seg.data).tolist()) <= {0, 1, 2, 3, 4}
    for name in ("t1", "t2"):
        assert torch.equal(expected.images[name].data, actual.images[name].data.cpu()), f"{name} not bit-exact at 512^3"


def test_fast_precision_256_within_tolerance_of_the_oracle(oracle, hip):
    """`set_resample_precision("fast")` at the bench size: float intensities within 1e-4 relative of the ORACLE
    (not of the exact kernel), the label map of the same subject still bit-exact (its launch stays exact)."""
    size = 256
    g = torch.Generator().manual_seed(21)
    subjects = [
        tio.Subject(t1=tio.ScalarImage(torch.rand(1, size, size, size, generator=g)), seg=tio.LabelMap(nested_spheres(size)))
        for _ in range(2)
    ]
    transform = tio.Compose(
        [
            tio.Affine(degrees=(-10, 10), scales=(0.9, 1.1), translation=(-5, 5), per_instance=True),
            tio.ElasticDeformation(per_instance=True),
        ]
    )
    cpu = tio.SubjectsBatch.from_subjects(copy.deepcopy(subjects))
    gpu = tio.SubjectsBatch.from_subjects(subjects).to("cuda")
    torch.manual_seed(22)
    with use_engine(oracle):
        expected = transform(cpu)
    previous = tio.get_resample_precision()
    try:
        tio.set_resample_precision("fast", allow_out_of_tolerance=True)
        torch.manual_seed(22)
        actual = transform(gpu)
        torch.cuda.synchronize()
    finally:
        tio.set_resample_precision(previous)
    assert torch.equal(expected.seg.data, actual.seg.data.cpu()), "labels must never take the fast path"
    want, got = expected.t1.data.double(), actual.t1.data.cpu().double()
    # FAST's contract is 1e-4 OF THE INTENSITY RANGE (unit-range data: an absolute 1e-4), for EVERY voxel — the 64-voxel
    # exemption of rounds 1 - 4 is gone (the fill decisions are the exact chain's since round 4).  It does
    # NOT meet the per-voxel bar on white noise (one ulp of a coordinate is already more): that is `precision="tight"`
    # (tests/test_gpu_tight.py); the count is recorded here, not asserted.
    value_range = float(want.max() - want.min())
    err = (want - got).abs()
    assert int((err > 1e-4 * value_range).sum()) == 0, f"{int((err > 1e-4 * value_range).sum())} voxels beyond 1e-4 of the range"
    per_voxel_beyond = int((err / want.abs().clamp_min(1e-3 * value_range) > 1e-4).sum())
    import json, os  # noqa: E401, PLC0415
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/fast_precision_256.json", "w") as handle:
            json.dump({"max_over_range": float(err.max()) / value_range, "beyond_per_voxel_bar": per_voxel_beyond, "voxels": err.numel()}, handle)


class _Calls:
    """Records which engine entry points a run went through (name -> count)."""

    def __init__(self, engine):
        self.engine, self.seen = engine, {}
        self._call, self._fused = engine._call, engine.blur_fused

    def __enter__(self):
        def call(name, ref, *args):
            self.seen[name] = self.seen.get(name, 0) + 1
            return self._call(name, ref, *args)

        def fused(data, taps, radius, *, bias_coarse=None, noise=None):
            out = self._fused(data, taps, radius, bias_coarse=bias_coarse, noise=noise)
            if out is not None:
                key = "blur_fused" + ("+bias" if bias_coarse is not None else "") + ("+noise" if noise is not None else "")
                self.seen[key] = self.seen.get(key, 0) + 1
            return out

        self.engine._call, self.engine.blur_fused = call, fused
        return self

    def __exit__(self, *exc):
        del self.engine._call, self.engine.blur_fused


@pytest.mark.parametrize("precision", ["tight", "fast"])
@pytest.mark.parametrize("intensity_scale", [1.0, 4095.0])
def test_headline_mode_256_matches_oracle(oracle, hip, intensity_scale, precision):
    """The EXACT configuration bench.py's headline is quoted on: `noise=philox` + `resample=tight` (round 5; `fast` = the
    headline of rounds 2 - 4, still timed in the bench's mode matrix) + fused multiply-adds in the stencil + the lazily fused
    BiasField / Blur / Noise, per-instance parameters, 256^3 float32, a batch large enough (3 x 4096 bricks) for the planned
    kernels.  The oracle runs the same Philox stream, so the whole pipeline is comparable, on unit-range data and on data
    scaled to a 12-bit scanner range.  `tight`: EVERY voxel inside |d| <= 1e-4 max(|ref|, 1e-3 range) — the north-star
    bar per voxel (the same metric as the default mode's test).  `fast`: every voxel within 1e-4 OF THE
    INTENSITY RANGE (its contract); its per-voxel count is recorded, not asserted — it cannot be zero on white noise."""
    from parity_harness import benchmark_compose  # noqa: PLC0415

    size, batch = 256, 3
    g = torch.Generator().manual_seed(31)
    subjects = [tio.Subject(t1=tio.ScalarImage(torch.rand(1, size, size, size, generator=g) * intensity_scale)) for _ in range(batch)]
    transform = benchmark_compose()
    cpu = tio.SubjectsBatch.from_subjects(copy.deepcopy(subjects))
    gpu = tio.SubjectsBatch.from_subjects(subjects).to("cuda")
    previous = (tio.get_noise_rng(), tio.get_resample_precision(), tio.get_stencil_precision())
    try:
        tio.set_noise_rng("philox")
        tio.set_resample_precision(precision, allow_out_of_tolerance=True)
        tio.set_stencil_precision("fast")  # bench.py's headline: fused multiply-adds in the Blur's taps
        torch.manual_seed(32)
        with use_engine(oracle):
            expected = transform(cpu)
        torch.manual_seed(32)
        with _Calls(hip) as calls:
            actual = transform(gpu)
            got = actual.t1.data
        torch.cuda.synchronize()
    finally:
        tio.set_noise_rng(previous[0])
        tio.set_resample_precision(previous[1])
        tio.set_stencil_precision(previous[2])
    assert [t.params for t in expected.applied_transforms] == [t.params for t in actual.applied_transforms]
    # the headline's launches, and nothing else: two resampling launches, ONE fused stencil carrying bias field and noise
    assert calls.seen.get("resample3d") == 2, calls.seen
    assert calls.seen.get("blur_fused+bias+noise") == 1, calls.seen
    assert "add_noise" not in calls.seen and "bias_field_apply" not in calls.seen and "separable_conv3d" not in calls.seen, calls.seen
    want, got = expected.t1.data.double(), got.cpu().double()
    # "1e-4 relative" = 1e-4 OF THE INTENSITY RANGE of the image that is compared (max - min of the oracle's output: the bias
    # field stretches the input's range), an absolute bar in those units.
    value_range = float(want.max() - want.min())
    err = (want - got).abs() / value_range
    beyond = err > 1e-4
    rel = (want - got).abs() / want.abs().clamp_min(1e-3 * value_range)
    stats = {
        "precision": precision, "intensity_scale": intensity_scale, "value_range": value_range, "voxels": err.numel(), "beyond_1e-4": int(beyond.sum()),
        "beyond_2e-4": int((err > 2e-4).sum()), "beyond_5e-4": int((err > 5e-4).sum()), "max": float(err.max()),
        "mean": float(err.mean()), "p99.99": float(err.flatten()[:: 7].kthvalue(int(0.9999 * err.flatten()[:: 7].numel())).values),
        "per_voxel_max": float(rel.max()), "per_voxel_beyond_1e-4": int((rel > 1e-4).sum()),
    }
    import json, os  # noqa: E401, PLC0415
    if os.path.isdir("gpurun_out"):
        with open(f"gpurun_out/headline_parity_{precision}_{int(intensity_scale)}.json", "w") as handle:
            json.dump(stats, handle)
    # every voxel within 1e-4 of the intensity range, both modes, no exemption (round 4: the FAST kernels take the fill decision of
    # voxels within rounding of the threshold from the exact chain; tight: the reference's decisions by construction)
    assert stats["beyond_1e-4"] == 0, stats
    assert stats["max"] <= 1e-4, stats
    assert stats["p99.99"] <= 2e-5, stats
    if precision == "tight":  # the per-voxel bar, no exempt voxel
        assert stats["per_voxel_beyond_1e-4"] == 0 and stats["per_voxel_max"] <= 1e-4, stats
