"""The other configurations of BASELINE.json / SURVEY.md §8(d), one JSON line each (1 GPU).

  2a  tio.Spatial(affine + elastic)          — one fused resampling, 8 x 1x256^3 f32
  2b  Compose[Affine, ElasticDeformation]    — two resamplings,      8 x 1x256^3 f32
  5   Subject{t1, t2: f32 512^3; seg: int16 512^3} through tio.Spatial(affine + elastic),
      trilinear for the intensities, nearest for the labels (one launch, shared coordinates)

bench.py stays the contract's single line (config 3); this script only documents the rest.
"""
from __future__ import annotations

import json
import os
import sys
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torchio_amd as tio  # noqa: E402
from parity_harness import nested_spheres  # noqa: E402

AFFINE = dict(degrees=(-10, 10), scales=(0.9, 1.1), translation=(-5, 5))


def timed(transform, batch, steps=20, warmup=3):
    for _ in range(warmup):
        transform(batch)
    torch.cuda.synchronize()
    start = time.perf_counter()
    for _ in range(steps):
        out = transform(batch)
    torch.cuda.synchronize()
    del out
    return (time.perf_counter() - start) / steps


def main() -> None:
    warnings.simplefilter("ignore")
    precision = os.environ.get("TIO_CONFIGS_PRECISION", "fast")  # like bench.py: the throughput mode; "exact" = the library default
    tio.set_resample_precision(precision)
    print(json.dumps({"resample_precision": precision, "note": "launches with a label map are always exact (config 5)"}))
    device = torch.device("cuda", 0)
    torch.manual_seed(0)
    g = torch.Generator(device=device).manual_seed(1)
    data = torch.rand(8, 1, 256, 256, 256, generator=g, device=device)
    batch = tio.SubjectsBatch({"t1": tio.ImagesBatch(data, [tio.AffineMatrix() for _ in range(8)], image_class=tio.ScalarImage)})
    volume = 256**3 * 4
    fused = tio.Spatial(**AFFINE, max_displacement=7.5)
    s = timed(fused, batch)
    print(json.dumps({"config": "2a tio.Spatial(affine+elastic), 8 x 1x256^3 f32", "volumes_per_s": 8 / s, "ms_per_step": 1e3 * s,
                      "algorithmic_GBps": 8 * 2 * volume / s / 1e9, "frac_of_8TBps": 8 * 2 * volume / s / 8e12}))
    two = tio.Compose([tio.Affine(**AFFINE), tio.ElasticDeformation()])
    s = timed(two, batch)
    print(json.dumps({"config": "2b Compose[Affine, ElasticDeformation], 8 x 1x256^3 f32", "volumes_per_s": 8 / s, "ms_per_step": 1e3 * s,
                      "algorithmic_GBps": 8 * 4 * volume / s / 1e9, "frac_of_8TBps": 8 * 4 * volume / s / 8e12}))
    del batch, data
    torch.cuda.empty_cache()
    size = 512
    subject = tio.Subject(
        t1=tio.ScalarImage(torch.rand(1, size, size, size)),
        t2=tio.ScalarImage(torch.rand(1, size, size, size) + 1),
        seg=tio.LabelMap(nested_spheres(size)),
    )
    big = tio.SubjectsBatch.from_subjects([subject]).to(device)
    s = timed(fused, big, steps=10)
    nbytes = 2 * (2 * 4 + 2) * size**3
    print(json.dumps({"config": "5 Subject{t1,t2 f32; seg int16} 512^3, tio.Spatial(affine+elastic), linear + nearest",
                      "subjects_per_s": 1 / s, "ms_per_step": 1e3 * s, "algorithmic_GBps": nbytes / s / 1e9, "frac_of_8TBps": nbytes / s / 8e12}))
    # the same subject with the label map resampled in the partial-volume ("label") mode: fused TIO_LABEL_PV launch,
    # the (1, 5, 512, 512, 512) float one-hot tensor of the reference (2.5 GiB) is never built
    label_mode = tio.Spatial(**AFFINE, max_displacement=7.5, label_interpolation="label")
    s = timed(label_mode, big, steps=10)
    print(json.dumps({"config": "5b same subject, label_interpolation=\"label\" (fused partial-volume mode; includes tio_unique_labels of the label map)",
                      "subjects_per_s": 1 / s, "ms_per_step": 1e3 * s, "algorithmic_GBps": nbytes / s / 1e9, "frac_of_8TBps": nbytes / s / 8e12}))


if __name__ == "__main__":
    main()
