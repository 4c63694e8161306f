"""Golden fixtures for the feeding side (samplers + PatchAggregator) from the UNMODIFIED reference.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_feeding.py

Build container only (imports /root/reference through ref_import.py).  Writes
``tests/golden/feeding_golden.pt``: plain tensors / lists / dicts.
"""
from __future__ import annotations

import os
import sys
import warnings

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from ref_import import import_reference  # noqa: E402

tio = import_reference()
warnings.simplefilter("ignore")

# name, volume shape, patch size, patch overlap, mode, dtype, channels, batch size, output_shape (or None), dict keys
AGGREGATOR_CASES = [
    ("crop_f32", (20, 18, 16), 8, 4, "crop", "float32", 2, 4, None, None),
    ("crop_aniso_overlap_i64", (21, 17, 19), (8, 6, 10), (2, 4, 6), "crop", "int64", 1, 3, None, None),
    ("crop_odd_overlap_u8", (16, 16, 16), 8, (3, 2, 0), "crop", "uint8", 3, 5, None, None),
    ("average_f32", (20, 18, 16), 8, 4, "average", "float32", 2, 4, None, None),
    ("average_f64_dict", (18, 14, 12), (8, 6, 4), (4, 2, 2), "average", "float64", 1, 3, None, ("logits", "embedding")),
    ("average_f16", (16, 16, 12), 8, 4, "average", "float16", 2, 4, None, None),
    ("average_bf16", (16, 16, 12), 8, 4, "average", "bfloat16", 2, 4, None, None),
    ("hann_f32", (20, 18, 16), 8, 4, "hann", "float32", 2, 4, None, None),
    ("hann_f32_big_batch", (24, 24, 24), 8, 6, "hann", "float32", 1, 40, None, None),
    ("hann_f64", (18, 14, 12), (8, 6, 4), (4, 2, 2), "hann", "float64", 2, 3, None, None),
    ("hann_f16", (16, 16, 12), 8, 4, "hann", "float16", 2, 4, None, None),
    ("hann_bf16", (16, 16, 12), 8, 4, "hann", "bfloat16", 1, 4, None, None),
    ("average_downsampled_output", (32, 24, 16), 16, 8, "average", "float32", 2, 2, (16, 12, 8), None),
    ("crop_downsampled_output", (32, 24, 16), 16, 8, "crop", "float32", 1, 3, (16, 12, 8), None),
    ("hann_no_overlap", (16, 16, 16), 8, 0, "hann", "float32", 1, 4, None, None),
]

SAMPLER_CASES = [
    ("grid", dict(shape=(20, 18, 16), patch_size=8, patch_overlap=4)),
    ("grid", dict(shape=(21, 17, 19), patch_size=(8, 6, 10), patch_overlap=(2, 4, 6))),
    ("grid", dict(shape=(7, 30, 9), patch_size=(8, 8, 8), patch_overlap=0)),  # patch larger than the volume on two axes
    ("uniform", dict(shape=(20, 18, 16), patch_size=(8, 6, 4), num_patches=6, seed=11)),
    ("weighted", dict(shape=(14, 12, 10), patch_size=(4, 6, 4), num_patches=5, seed=12)),
    ("label", dict(shape=(14, 12, 10), patch_size=4, num_patches=5, seed=13, label_probabilities=None)),
    ("label", dict(shape=(14, 12, 10), patch_size=(4, 4, 6), num_patches=5, seed=14, label_probabilities={1: 1.0, 2: 3.0})),
    # padded before sampling (sampler.py:127-147): overlap // 2 voxels per side through Pad
    ("grid", dict(shape=(20, 18, 16), patch_size=8, patch_overlap=4, padding_mode="constant", fill=-2.0)),
    ("grid", dict(shape=(13, 11, 12), patch_size=(8, 6, 10), patch_overlap=(2, 4, 6), padding_mode="reflect")),
    ("grid", dict(shape=(10, 12, 9), patch_size=6, patch_overlap=(2, 0, 4), padding_mode="mean")),
]


def subject_for(shape, seed):
    g = torch.Generator().manual_seed(seed)
    t1 = torch.rand(1, *shape, generator=g)
    prob = torch.rand(1, *shape, generator=g) ** 4
    seg = (torch.rand(1, *shape, generator=g) * 3).to(torch.int16)
    return t1, prob, seg


def main():
    aggregator_cases = []
    for index, (name, shape, patch, overlap, mode, dtype, channels, batch, output_shape, keys) in enumerate(AGGREGATOR_CASES):
        t1, _, _ = subject_for(shape, 100 + index)
        subject = tio.Subject(t1=tio.ScalarImage(t1))
        sampler = tio.GridSampler(subject, patch, overlap)
        locations = [(loc.index, loc.size) for loc in sampler.locations]
        aggregator = tio.PatchAggregator(shape, overlap_mode=mode, patch_overlap=overlap, output_shape=output_shape)
        g = torch.Generator().manual_seed(200 + index)
        scale = (1, 1, 1) if output_shape is None else tuple(o / s for o, s in zip(output_shape, shape))
        batches = []
        for start in range(0, len(sampler), batch):
            chunk = sampler.locations[start : start + batch]
            size = tuple(round(s * f) for s, f in zip(chunk[0].size, scale))

            def draw():
                values = torch.randn(len(chunk), channels, *size, generator=g, dtype=torch.float64)
                if dtype in ("int64", "uint8"):
                    return (values * 20).abs().to(getattr(torch, dtype))
                return values.to(getattr(torch, dtype))

            outputs = draw() if keys is None else {key: draw() for key in keys}
            aggregator.add_batch(outputs, chunk)
            batches.append({"outputs": outputs, "first": start, "count": len(chunk)})
        expected = {"__default__": aggregator.get_output()} if keys is None else {key: aggregator.get_output(key) for key in keys}
        aggregator_cases.append({
            "name": name, "shape": shape, "patch_size": patch, "patch_overlap": overlap, "mode": mode, "output_shape": output_shape,
            "locations": locations, "batches": batches, "expected": expected,
        })
        print(f"{name:32s} patches={len(sampler)} out={tuple(next(iter(expected.values())).shape)}")

    sampler_cases = []
    for index, (kind, cfg) in enumerate(SAMPLER_CASES):
        cfg = dict(cfg)
        shape = cfg.pop("shape")
        t1, prob, seg = subject_for(shape, 300 + index)
        subject = tio.Subject(t1=tio.ScalarImage(t1), prob=tio.ScalarImage(prob), seg=tio.LabelMap(seg), note="kept")
        seed = cfg.pop("seed", None)
        if kind == "grid":
            sampler = tio.GridSampler(subject, **cfg)
            patches = [sampler[i] for i in range(len(sampler))]
        else:
            torch.manual_seed(seed)
            if kind == "uniform":
                sampler = tio.UniformSampler(subject, **cfg)
            elif kind == "weighted":
                sampler = tio.WeightedSampler(subject, probability_map="prob", **cfg)
            else:
                sampler = tio.LabelSampler(subject, label_name="seg", **cfg)
            patches = list(sampler)
        rng_probe = float(torch.rand(1).item())
        sampler_cases.append({
            "kind": kind, "config": {**cfg, "shape": shape}, "seed": seed, "t1": t1, "prob": prob, "seg": seg,
            "locations": [(p.patch_location.index, p.patch_location.size) for p in patches],
            "t1_patches": torch.stack([p.t1.data for p in patches]),
            "origins": torch.stack([p.t1.affine.data[:3, 3].clone() for p in patches]),
            "rng_probe": rng_probe,
        })
        print(f"{kind:10s} {cfg} -> {len(patches)} patches")

    path = os.path.join(HERE, "feeding_golden.pt")
    torch.save({"torch": str(torch.__version__), "torchio": str(tio.__version__), "aggregator": aggregator_cases,
                "samplers": sampler_cases}, path)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
