"""CPU: samplers and PatchAggregator vs golden vectors of the reference (tests/golden/feeding_golden.pt).

The aggregation arithmetic runs on the CPU oracle (``tio_oracle_patch_accumulate``) through
the test-only engine hook; the samplers are pure host logic.
"""
from __future__ import annotations

import pytest
import torch

import torchio_amd as tio
from feeding_cases import aggregator_ids
from feeding_cases import check_aggregator
from feeding_cases import load
from parity_harness import use_engine


@pytest.fixture(autouse=True)
def _oracle_engine(oracle):
    with use_engine(oracle):
        yield


@pytest.mark.parametrize("name", aggregator_ids())
def test_aggregator_golden_on_oracle(name):
    case = next(c for c in load()["aggregator"] if c["name"] == name)
    check_aggregator(case, "cpu")


@pytest.mark.parametrize("index", range(len(load()["samplers"])))
def test_sampler_golden(index):
    case = load()["samplers"][index]
    config = dict(case["config"])
    config.pop("shape")
    subject = tio.Subject(t1=tio.ScalarImage(case["t1"]), prob=tio.ScalarImage(case["prob"]), seg=tio.LabelMap(case["seg"]), note="kept")
    if case["kind"] == "grid":
        sampler = tio.GridSampler(subject, **config)
        patches = [sampler[i] for i in range(len(sampler))]
    else:
        torch.manual_seed(case["seed"])
        if case["kind"] == "uniform":
            sampler = tio.UniformSampler(subject, **config)
        elif case["kind"] == "weighted":
            sampler = tio.WeightedSampler(subject, probability_map="prob", **config)
        else:
            sampler = tio.LabelSampler(subject, label_name="seg", **config)
        patches = list(sampler)
    if case["seed"] is not None:  # the grid sampler draws nothing
        assert float(torch.rand(1).item()) == case["rng_probe"], "global RNG consumed differently"
    locations = [(tuple(p.patch_location.index), tuple(p.patch_location.size)) for p in patches]
    assert locations == [(tuple(i), tuple(s)) for i, s in case["locations"]]
    assert torch.equal(torch.stack([p.t1.data for p in patches]), case["t1_patches"])
    assert torch.equal(torch.stack([p.t1.affine.data[:3, 3] for p in patches]), case["origins"])
    assert all(p.note == "kept" and isinstance(p.seg, tio.LabelMap) for p in patches)
    if config.get("padding_mode") is not None:
        return
    # patches are views of the subject's storage: nothing is copied on the way to the model
    assert patches[0].t1.data.untyped_storage().data_ptr() == subject.t1.data.untyped_storage().data_ptr()


def test_image_indexing_follows_the_reference_rules():
    image = tio.ScalarImage(torch.arange(3 * 6 * 5 * 4, dtype=torch.float32).reshape(3, 6, 5, 4), affine=torch.diag(torch.tensor([2.0, 3.0, 4.0, 1.0])))
    assert image[0].shape == (1, 6, 5, 4) and image[-1].shape == (1, 6, 5, 4)
    assert image[:, 2:5].shape == (3, 3, 5, 4)
    assert image[..., 1:3].shape == (3, 6, 5, 2)
    assert image[1:3, 1:2, ..., 2:].shape == (2, 1, 5, 2)
    cropped = image[:, 2:5, 1:, -2:]
    assert torch.equal(cropped.affine.data[:3, 3], torch.tensor([4.0, 3.0, 8.0], dtype=torch.float64))
    assert torch.equal(cropped.data, image.data[:, 2:5, 1:, -2:])
    with pytest.raises(IndexError):
        image[0, 0, 0, 0, 0]
    with pytest.raises(TypeError):
        image[1.5]
    with pytest.raises(KeyError):
        image["missing"]


def test_aggregator_errors_and_keys():
    with pytest.raises(ValueError, match="overlap_mode"):
        tio.PatchAggregator((8, 8, 8), overlap_mode="max")
    aggregator = tio.PatchAggregator((8, 8, 8), overlap_mode="average")
    with pytest.raises(KeyError, match="No output"):
        aggregator.get_output()
    location = tio.PatchLocation(index=(0, 0, 0), size=(4, 4, 4))
    aggregator.add_batch({"a": torch.ones(1, 2, 4, 4, 4)}, [location])
    with pytest.raises(KeyError, match=r"Available: \['a'\]"):
        aggregator.get_output("b")
    with pytest.raises(ValueError, match="locations"):
        aggregator.add_batch({"a": torch.ones(2, 2, 4, 4, 4)}, [location])
    with pytest.raises(ValueError, match="channels"):
        aggregator.add_batch({"a": torch.ones(1, 3, 4, 4, 4)}, [location])
    with pytest.raises(TypeError, match="floating"):
        tio.PatchAggregator((8, 8, 8), overlap_mode="hann").add_batch(torch.ones(1, 1, 4, 4, 4, dtype=torch.int32), [location])
    out = aggregator.get_output("a")
    assert out.shape == (2, 8, 8, 8) and float(out[:, :4, :4, :4].min()) == 1.0 and float(out[:, 4:].abs().max()) == 0.0


def test_grid_sampler_then_aggregator_round_trip():
    """Identity "model": aggregating the sampled patches gives the volume back, in every mode."""
    volume = torch.rand(2, 21, 18, 17)
    subject = tio.Subject(t1=tio.ScalarImage(volume))
    for mode in ("crop", "average", "hann"):
        sampler = tio.GridSampler(subject, patch_size=(8, 6, 8), patch_overlap=(4, 2, 4))
        aggregator = tio.PatchAggregator(subject.spatial_shape, overlap_mode=mode, patch_overlap=(4, 2, 4))
        for start in range(0, len(sampler), 5):
            patches = [sampler[i] for i in range(start, min(start + 5, len(sampler)))]
            aggregator.add_batch(torch.stack([p.t1.data for p in patches]), [p.patch_location for p in patches])
        restored = aggregator.get_output()
        if mode == "crop":
            assert torch.equal(restored, volume)
        elif mode == "average":
            torch.testing.assert_close(restored, volume, rtol=1e-5, atol=1e-6)
        else:  # the reference divides by clamp(weights, min=1): only voxels with a weight sum >= 1 come back
            covered = aggregator._counts["__default__"] >= 1
            assert bool(covered.any())
            torch.testing.assert_close(restored[covered], volume[covered], rtol=1e-5, atol=1e-6)


# -- Queue ---------------------------------------------------------------------------------
def _queue_subjects(n=5, size=12):
    g = torch.Generator().manual_seed(7)
    return [
        tio.Subject(t1=tio.ScalarImage(torch.rand(1, size, size, size, generator=g) + index), seg=tio.LabelMap(torch.full((1, size, size, size), index, dtype=torch.int16)), index=index)
        for index in range(n)
    ]


@pytest.mark.parametrize("num_workers", [0, 3])
def test_queue_yields_every_patch_once_and_respects_the_buffer(num_workers):
    import random

    subjects = _queue_subjects()
    sampler = tio.UniformSampler(subjects[0], patch_size=4)
    queue = tio.Queue(subjects, sampler, max_length=6, patches_per_volume=3, num_workers=num_workers)
    assert queue.num_subjects == 5 and queue.patches_per_epoch == 15
    assert queue.max_memory == 4 * 2 * 64 * 6 and queue.max_memory_pretty == "3.0 KiB"
    random.seed(3)
    torch.manual_seed(3)
    patches = list(queue)
    assert len(patches) == 15
    counts = {}
    for patch in patches:
        assert patch.t1.shape == (1, 4, 4, 4) and isinstance(patch.patch_location, tio.PatchLocation)
        source = subjects[patch.index]
        i, j, k = patch.patch_location.index
        assert torch.equal(patch.t1.data, source.t1.data[:, i : i + 4, j : j + 4, k : k + 4])  # a crop of its own subject
        assert bool((patch.seg.data == patch.index).all())
        counts[patch.index] = counts.get(patch.index, 0) + 1
    assert counts == {index: 3 for index in range(5)}


def test_queue_applies_the_transform_before_sampling_and_shuffles_reproducibly():
    import random

    subjects = _queue_subjects(4)
    transform = tio.Affine(degrees=(20, 20, 20), default_pad_value=0.0)
    sampler = tio.GridSampler(subjects[0], patch_size=6)

    class WholeGrid(tio.PatchSampler):
        def __call__(self, subject, num_patches=None):
            grid = tio.GridSampler(subject, self.patch_size)
            return (grid[i] for i in range(len(grid)))

    def epoch(seed):
        random.seed(seed)
        torch.manual_seed(seed)
        queue = tio.Queue(subjects, WholeGrid(6), max_length=100, patches_per_volume=8, transform=transform, shuffle_subjects=True)
        return list(queue)

    first, again, other = epoch(1), epoch(1), epoch(2)
    assert len(first) == 32 and len(sampler) == 8
    assert [p.index for p in first] == [p.index for p in again]
    assert all(torch.equal(a.t1.data, b.t1.data) for a, b in zip(first, again, strict=True))
    assert [p.index for p in first] != [p.index for p in other]
    # the patches come from the TRANSFORMED subjects (history recorded, values differ from the raw crop)
    patch = first[0]
    raw = subjects[patch.index]
    i, j, k = patch.patch_location.index
    assert not torch.equal(patch.t1.data, raw.t1.data[:, i : i + 6, j : j + 6, k : k + 6])


def test_queue_with_distributed_sampler_splits_the_subjects():
    from torch.utils.data import DistributedSampler

    subjects = _queue_subjects(6)
    seen = []
    for rank in range(2):
        split = DistributedSampler(subjects, num_replicas=2, rank=rank, shuffle=False)
        queue = tio.Queue(subjects, tio.UniformSampler(subjects[0], 4), patches_per_volume=2, shuffle_subjects=False, subject_sampler=split)
        assert queue.num_subjects == 3 and queue.patches_per_epoch == 6
        seen.append(sorted({patch.index for patch in queue}))
    assert seen == [[0, 2, 4], [1, 3, 5]]
    with pytest.raises(ValueError, match="shuffle_subjects must be False"):
        tio.Queue(subjects, tio.UniformSampler(subjects[0], 4), subject_sampler=DistributedSampler(subjects, num_replicas=2, rank=0))


# -- SubjectsLoader ----------------------------------------------------------------------
def test_subjects_loader_runs_the_documented_dense_inference_loop():
    """sampler.py:84-96: GridSampler -> SubjectsLoader -> model -> aggregator.add_batch(outputs, locations)."""
    from torchio_amd.loader import ImagesLoader
    from torchio_amd.loader import SubjectsLoader

    volume = torch.rand(1, 20, 18, 16)
    subject = tio.Subject(t1=tio.ScalarImage(volume), note="kept")
    sampler = tio.GridSampler(subject, patch_size=8, patch_overlap=4)
    aggregator = tio.PatchAggregator(subject.spatial_shape, overlap_mode="crop", patch_overlap=4)
    seen = 0
    for batch in SubjectsLoader(sampler, batch_size=5):
        assert isinstance(batch, tio.SubjectsBatch) and batch.t1.data.shape[1:] == (1, 8, 8, 8)
        locations = batch.metadata["patch_location"]  # per-sample metadata lists (data/batch.py:124-160)
        assert len(locations) == batch.batch_size and batch.metadata["note"] == ["kept"] * batch.batch_size
        aggregator.add_batch(batch.t1.data * 2, locations)  # "model": doubles the intensities
        seen += batch.batch_size
    assert seen == len(sampler)
    assert torch.equal(aggregator.get_output(), volume * 2)
    with pytest.raises(ValueError, match="sets collate_fn automatically"):
        SubjectsLoader(sampler, collate_fn=lambda b: b)
    images = [tio.ScalarImage(torch.rand(1, 4, 4, 4)) for _ in range(3)]
    (image_batch,) = list(ImagesLoader(images, batch_size=3))
    assert isinstance(image_batch, tio.ImagesBatch) and image_batch.data.shape == (3, 1, 4, 4, 4)
