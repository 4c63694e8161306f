"""Record the five seams of the REAL reference at work, for replay on the GPU box.

Run in the build container (where /root/reference exists):

    python tests/golden/make_binding_calls.py          # writes tests/golden/binding_calls.pt

Every seam that ``torchio_amd.reference_binding.bind`` replaces is wrapped with a recorder while the UNMODIFIED reference
runs its own transforms on small subjects: the recorder stores the call's inputs as plain data (image names, tensors,
4x4 affines, image class names, keyword arguments / parameter dictionaries), lets the reference's ORIGINAL function do the
work on the CPU, and stores what it left behind.  ``tests/test_gpu_binding_replay.py`` rebuilds reference-shaped
containers from that data on the GPU box (the reference itself does not travel), calls THIS package's seam functions on
CUDA tensors — the functions ``bind()`` installs — and compares with the reference's recorded results.
Test infrastructure: nothing under ``torchio_amd/`` imports this file.
"""
from __future__ import annotations

import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_import  # noqa: E402

tio = ref_import.import_reference()
import importlib  # noqa: E402

ref_spatial = importlib.import_module("torchio.transforms.spatial.spatial")
ref_blur = importlib.import_module("torchio.transforms.intensity.blur")
ref_bias = importlib.import_module("torchio.transforms.intensity.bias_field")
ref_noise = importlib.import_module("torchio.transforms.intensity.noise")
ref_gamma = importlib.import_module("torchio.transforms.intensity.gamma")

CALLS: list[dict] = []


def plain(value):
    """Parameters as plain data (tensors / arrays / numbers / containers of those)."""
    if isinstance(value, torch.Tensor):
        return value.detach().clone()
    if isinstance(value, np.ndarray):
        return value.copy()
    if isinstance(value, dict):
        return {k: plain(v) for k, v in value.items()}
    if isinstance(value, (list, tuple)):
        return type(value)(plain(v) for v in value)
    if hasattr(value, "affine_matrices") and hasattr(value, "control_points"):  # _PerSampleGrids
        return {"__per_sample__": True, "affine_matrices": plain(list(value.affine_matrices)),
                "control_points": plain(list(value.control_points)), "max_displacements": plain(list(value.max_displacements))}
    if type(value).__name__ == "AffineMatrix":
        return {"__affine__": value.numpy().copy()}
    return value


def snapshot(batch, names=None):
    images = {}
    for name, img in batch.images.items():
        if names is not None and name not in names:
            continue
        images[name] = {
            "data": img.data.detach().clone(), "affines": [a.numpy().copy() for a in img.affines],
            "image_class": img._image_class.__name__,
        }
    return images


def record_spatial(original):
    def seam(**kwargs):
        entry = {"seam": "_apply_spatial_to_batch", "before": snapshot(kwargs["batch"]),
                 "kwargs": {k: plain(v) for k, v in kwargs.items() if k != "batch"}}
        if kwargs.get("target_space") is not None:
            shape, affine = kwargs["target_space"]
            entry["kwargs"]["target_space"] = (tuple(shape), {"__affine__": affine.numpy().copy()})
        original(**kwargs)
        entry["after"] = snapshot(kwargs["batch"], kwargs["image_names"])
        CALLS.append(entry)

    return seam


def record_apply(owner_name, original):
    def seam(self, batch, params):
        names = list(self._get_images(batch))
        entry = {"seam": f"{owner_name}.apply_transform", "before": snapshot(batch), "params": plain(dict(params)), "image_names": names,
                 "init": {k: plain(getattr(self, k)) for k in ("include", "exclude") if hasattr(self, k)}}
        out = original(self, batch, params)
        entry["after"] = snapshot(batch, names)
        CALLS.append(entry)
        return out

    return seam


def record_smooth(original):
    def seam(data, sigmas):
        entry = {"seam": "_gaussian_smooth", "data": data.detach().clone(), "sigmas": plain(np.asarray(sigmas, dtype=np.float64))}
        out = original(data, sigmas)
        entry["result"] = out.detach().clone()
        CALLS.append(entry)
        return out

    return seam


def nested_spheres(size, dtype=torch.int16):
    axis = torch.arange(size, dtype=torch.float32) - (size - 1) / 2
    i, j, k = torch.meshgrid(axis, axis, axis, indexing="ij")
    dist = torch.sqrt(i * i + j * j + k * k)
    label = sum((dist <= r * size).to(torch.int32) for r in (0.45, 0.35, 0.25, 0.15))
    return label.to(dtype).unsqueeze(0)


def subjects(size, n, seed):
    g = torch.Generator().manual_seed(seed)
    return [
        tio.Subject(t1=tio.ScalarImage(torch.rand(1, size, size, size, generator=g)), t2=tio.ScalarImage(torch.rand(2, size, size, size, generator=g) + 1),
                    seg=tio.LabelMap(nested_spheres(size)))
        for _ in range(n)
    ]


def main() -> None:
    ref_spatial._apply_spatial_to_batch = record_spatial(ref_spatial._apply_spatial_to_batch)
    ref_blur._gaussian_smooth = record_smooth(ref_blur._gaussian_smooth)
    for module, names in ((ref_bias, ("BiasField",)), (ref_noise, ("Noise",)), (ref_gamma, ("Gamma",))):
        for name in names:
            cls = getattr(module, name)
            cls.apply_transform = record_apply(name, cls.apply_transform)

    size = 16
    pipelines = [
        ("affine_per_instance", lambda: tio.Affine(degrees=(-10, 10), scales=(0.9, 1.1), translation=(-3, 3)), 3),
        ("elastic_shared", lambda: tio.ElasticDeformation(per_instance=False), 2),
        ("spatial_fused", lambda: tio.Spatial(degrees=(-8, 8), scales=(0.95, 1.05), max_displacement=2.5, num_control_points=5), 2),
        ("resample_target", lambda: tio.Resample(target=1.5), 2),
        ("blur", lambda: tio.Blur(std=(0.5, 1.5)), 3),
        ("bias", lambda: tio.BiasField(), 3),
        ("noise", lambda: tio.Noise(std=(0.05, 0.2)), 2),
        ("gamma", lambda: tio.Gamma(log_gamma=(-0.3, 0.3)), 2),
        ("compose", lambda: tio.Compose([tio.Affine(degrees=(-5, 5)), tio.BiasField(), tio.Blur(std=(0.5, 1.0)), tio.Noise(std=0.05)]), 2),
    ]
    import warnings

    for index, (label, make, n) in enumerate(pipelines):
        first = len(CALLS)
        batch = tio.SubjectsBatch.from_subjects(copy.deepcopy(subjects(size, n, 100 + index)))
        torch.manual_seed(200 + index)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            make()(batch)
        for entry in CALLS[first:]:
            entry["pipeline"] = label
    path = os.path.join(HERE, "binding_calls.pt")
    torch.save(CALLS, path)
    sizes = {}
    for entry in CALLS:
        sizes[entry["seam"]] = sizes.get(entry["seam"], 0) + 1
    print(f"wrote {path}: {len(CALLS)} seam calls {sizes}, {os.path.getsize(path) / 1e6:.1f} MB")


if __name__ == "__main__":
    main()
