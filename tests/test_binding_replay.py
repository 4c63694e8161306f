"""The reference-side binding on device memory.

``tests/golden/make_binding_calls.py`` recorded, in the build container, every call the UNMODIFIED reference made to the
five seams ``torchio_amd.reference_binding.bind`` replaces (inputs as plain data + what the reference's own function left
behind).  Here those calls are replayed through THIS package's seam functions — the ones ``bind()`` installs — on
reference-SHAPED containers (the reference package does not travel to the GPU box; the containers below expose exactly
the attributes its ``ImagesBatch`` / ``SubjectsBatch`` / ``AffineMatrix`` do and nothing of this package's own classes):
on the CPU oracle here, on CUDA tensors through ``libtio_hip.so`` on the GPU box (``-m gpu``).  Results are held to the
reference's recorded outputs: label maps bit for bit, intensities to float rounding.
"""
from __future__ import annotations

import os

import numpy as np
import pytest
import torch

import torchio_amd as tio
from parity_harness import use_engine
from torchio_amd.transforms import bias_field as our_bias
from torchio_amd.transforms import blur as our_blur
from torchio_amd.transforms import gamma as our_gamma
from torchio_amd.transforms import noise as our_noise
from torchio_amd.transforms import spatial as our_spatial

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "binding_calls.pt")


class RefLikeAffine:
    """The surface of the reference's ``AffineMatrix`` that the seams touch (data/affine.py:20-248)."""

    def __init__(self, matrix) -> None:
        self._matrix = torch.as_tensor(np.asarray(matrix, dtype=np.float64)).clone()

    @property
    def data(self):
        return self._matrix

    def numpy(self):
        return self._matrix.cpu().numpy()

    @property
    def spacing(self):
        norms = torch.sqrt(torch.sum(self._matrix[:3, :3] ** 2, dim=0)).tolist()
        return (float(norms[0]), float(norms[1]), float(norms[2]))

    def clone(self):
        return RefLikeAffine(self._matrix)


class ScalarImage:  # class NAMES are what `_is_label_batch` / `IntensityTransform._get_images` look at on foreign containers
    pass


class LabelMap:
    pass


class RefLikeImagesBatch:
    """``data`` / ``affines`` / ``batch_size`` / ``_image_class`` like reference data/batch.py:21-120; no ``_flush`` / ``_pending``."""

    def __init__(self, data, affines, image_class) -> None:
        self.data = data
        self.affines = affines
        self._image_class = image_class
        self.applied_transforms = []

    @property
    def batch_size(self) -> int:
        return int(self.data.shape[0])


class RefLikeSubjectsBatch:
    def __init__(self, images) -> None:
        self.images = images
        self.applied_transforms = []

    @property
    def batch_size(self) -> int:
        return next(iter(self.images.values())).batch_size


class RefLikeTransform:
    """``self`` of a bound ``apply_transform``: under ``bind()`` it is the REFERENCE's transform instance, so the seam may only
    use what that class offers — ``_get_images`` (intensity transforms: ScalarImage batches, then include / exclude;
    transform.py:684-693) and ``_is_per_instance_params`` (transform.py:358-360)."""

    def __init__(self, include=None, exclude=None) -> None:
        self.include, self.exclude = include, exclude

    def _get_images(self, batch):
        images = {k: v for k, v in batch.images.items() if v._image_class is ScalarImage}
        if self.include is not None:
            images = {k: v for k, v in images.items() if k in self.include}
        if self.exclude is not None:
            images = {k: v for k, v in images.items() if k not in self.exclude}
        return images

    @staticmethod
    def _is_per_instance_params(params) -> bool:
        return "_batched_keys" in params


def _build(snapshot, device):
    classes = {"ScalarImage": ScalarImage, "LabelMap": LabelMap}
    return RefLikeSubjectsBatch({
        name: RefLikeImagesBatch(entry["data"].to(device), [RefLikeAffine(a) for a in entry["affines"]], classes[entry["image_class"]])
        for name, entry in snapshot.items()
    })


def _revive(value):
    if isinstance(value, dict) and "__affine__" in value:
        return RefLikeAffine(value["__affine__"])
    if isinstance(value, dict) and value.get("__per_sample__"):
        return our_spatial._PerSampleGrids(list(value["affine_matrices"]), list(value["control_points"]), list(value["max_displacements"]))
    if isinstance(value, tuple) and len(value) == 2 and isinstance(value[1], dict) and "__affine__" in value[1]:
        return (value[0], RefLikeAffine(value[1]["__affine__"]))
    return value


def _compare(entry, batch, what):
    for name, expected in entry["after"].items():
        got = batch.images[name].data.cpu()
        want = expected["data"]
        assert got.shape == want.shape and got.dtype == want.dtype, (what, name)
        if expected["image_class"] == "LabelMap" or not want.dtype.is_floating_point:
            assert torch.equal(got, want), f"{what}: label map {name} differs from the reference's result"
        else:
            scale = float(want.abs().max().clamp_min(1.0))
            assert float((got.double() - want.double()).abs().max()) <= 1e-5 * scale, f"{what}: image {name}"
        for ours, theirs in zip(batch.images[name].affines, expected["affines"], strict=True):
            assert np.allclose(ours.numpy(), theirs, atol=1e-12), f"{what}: affine of {name}"


def _replay(device):
    calls = torch.load(FIXTURE, weights_only=False)
    assert {c["seam"] for c in calls} == {
        "_apply_spatial_to_batch", "_gaussian_smooth", "BiasField.apply_transform", "Noise.apply_transform", "Gamma.apply_transform"}
    owners = {"BiasField": our_bias.BiasField, "Noise": our_noise.Noise, "Gamma": our_gamma.Gamma}
    for index, entry in enumerate(calls):
        what = f"call {index} ({entry['pipeline']}: {entry['seam']})"
        if entry["seam"] == "_gaussian_smooth":
            result = our_blur._gaussian_smooth(entry["data"].to(device), entry["sigmas"])
            scale = float(entry["result"].abs().max().clamp_min(1.0))
            assert float((result.cpu().double() - entry["result"].double()).abs().max()) <= 1e-5 * scale, what
            continue
        batch = _build(entry["before"], device)
        if entry["seam"] == "_apply_spatial_to_batch":
            kwargs = {key: _revive(value) for key, value in entry["kwargs"].items()}
            our_spatial._apply_spatial_to_batch(batch=batch, **kwargs)
        else:
            owner = owners[entry["seam"].split(".")[0]]
            instance = RefLikeTransform(include=entry["init"].get("include"), exclude=entry["init"].get("exclude"))
            owner.apply_transform(instance, batch, entry["params"])
        _compare(entry, batch, what)
    return len(calls)


def test_recorded_reference_seam_calls_replay_on_the_oracle(oracle):
    with use_engine(oracle):
        assert _replay("cpu") >= 10


@pytest.mark.gpu
def test_recorded_reference_seam_calls_replay_on_the_gpu(hip):
    """The seams of the binding with CUDA tensors in reference-shaped containers: every recorded call of the reference is
    answered by ``libtio_hip.so`` with the reference's own result."""
    assert tio.get_noise_rng() == "reference"
    assert _replay("cuda") >= 10
    torch.cuda.synchronize()
