"""GPU: PatchAggregator accumulating on the device (tio_patch_accumulate) vs the reference's golden vectors and the oracle.

Bar: bit-exact - the kernel applies the patches that cover a voxel in patch order, so the
float sums round like the reference's sequence of slice assignments.
"""
from __future__ import annotations

import pytest
import torch

import torchio_amd as tio
from feeding_cases import aggregator_ids
from feeding_cases import check_aggregator
from feeding_cases import load
from parity_harness import use_engine

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", aggregator_ids())
def test_aggregator_golden_on_hip(name, hip):
    case = next(c for c in load()["aggregator"] if c["name"] == name)
    check_aggregator(case, "cuda")


@pytest.mark.parametrize("mode", ["crop", "average", "hann"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_device_pipeline_equals_oracle_at_inference_size(oracle, hip, mode, dtype):
    """GridSampler on a device-resident subject -> "model" -> aggregator, 96x80x64 with 32^3 patches, vs the CPU oracle."""
    g = torch.Generator().manual_seed(5)
    volume = torch.rand(1, 96, 80, 64, generator=g)
    overlap = (16, 8, 16)

    def run(device):
        subject = tio.Subject(t1=tio.ScalarImage(volume.to(device)))
        sampler = tio.GridSampler(subject, 32, overlap)
        aggregator = tio.PatchAggregator(subject.spatial_shape, overlap_mode=mode, patch_overlap=overlap)
        for start in range(0, len(sampler), 6):
            patches = [sampler[i] for i in range(start, min(start + 6, len(sampler)))]
            inputs = torch.stack([p.t1.data for p in patches])
            assert inputs.device.type == torch.device(device).type
            outputs = torch.cat([inputs * 2 - 1, inputs.flip(-1)], dim=1).to(dtype)  # a 2-channel stand-in for a model
            aggregator.add_batch(outputs, [p.patch_location for p in patches])
        return aggregator.get_output()

    with use_engine(oracle):
        expected = run("cpu")
    actual = run("cuda")
    assert actual.is_cuda and torch.equal(expected, actual.cpu())


def test_more_patches_than_one_launch_holds(oracle, hip):
    """A batch above TIO_MAX_PATCHES is split into ordered launches."""
    g = torch.Generator().manual_seed(6)
    outputs = torch.randn(70, 1, 8, 8, 8, generator=g)
    locations = [tio.PatchLocation(index=(int(i) % 9, (int(i) * 3) % 9, (int(i) * 5) % 9), size=(8, 8, 8)) for i in range(70)]

    def run(device):
        aggregator = tio.PatchAggregator((16, 16, 16), overlap_mode="hann")
        aggregator.add_batch(outputs.to(device), locations)
        return aggregator.get_output()

    with use_engine(oracle):
        expected = run("cpu")
    assert torch.equal(expected, run("cuda").cpu())


def test_bad_placement_is_rejected(hip):
    aggregator = tio.PatchAggregator((8, 8, 8), overlap_mode="average")
    with pytest.raises(RuntimeError, match="leaves the patch or the volume"):
        aggregator.add_batch(torch.ones(1, 1, 4, 4, 4, device="cuda"), [tio.PatchLocation(index=(6, 0, 0), size=(4, 4, 4))])


@pytest.mark.parametrize("num_workers", [0, 2])
def test_queue_feeds_device_resident_patches(hip, num_workers):
    """Subjects on the GPU -> transform on the GPU (worker threads included) -> patches that never left the device."""
    import random

    g = torch.Generator().manual_seed(9)
    subjects = [
        tio.Subject(t1=tio.ScalarImage(torch.rand(1, 32, 32, 32, generator=g)), seg=tio.LabelMap(torch.randint(0, 4, (1, 32, 32, 32), generator=g).to(torch.int16)), index=i).to("cuda")
        for i in range(4)
    ]
    transform = tio.Compose([tio.Affine(degrees=(-10, 10), translation=(-2, 2)), tio.Blur(std=(0.5, 1.0)), tio.Noise(std=(0.01, 0.02))])
    queue = tio.Queue(subjects, tio.UniformSampler(subjects[0], 16), max_length=8, patches_per_volume=4, num_workers=num_workers, transform=transform)
    random.seed(0)
    torch.manual_seed(0)
    patches = list(queue)
    torch.cuda.synchronize()
    assert len(patches) == 16
    for patch in patches:
        assert patch.t1.data.is_cuda and patch.seg.data.is_cuda and patch.t1.shape == (1, 16, 16, 16)
        assert torch.isfinite(patch.t1.data).all()
        assert set(patch.seg.data.unique().tolist()) <= {0, 1, 2, 3}
    assert sorted(p.index for p in patches) == [0] * 4 + [1] * 4 + [2] * 4 + [3] * 4
    assert all(not s.t1.applied_transforms for s in subjects)  # the queue's subjects are left untouched (copy=True)
