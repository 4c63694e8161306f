"""Generate the golden fixtures from the UNMODIFIED reference (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Imports TorchIO 2.0.0a2 from /root/reference (stubbed non-hot-path deps, see
ref_import.py), runs each hot-path transform on small seeded inputs and stores
inputs, the recorded history (sampled params) and outputs in
``tests/golden/transforms_golden.pt``.  The fixtures travel to the GPU box; the
reference does not.  Everything is plain tensors / lists / dicts so the file
loads with ``torch.load(weights_only=True)``.
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


def spheres(shape, dtype=torch.int16):
    axes = [torch.arange(s, dtype=torch.float32) - (s - 1) / 2 for s in shape]
    i, j, k = torch.meshgrid(*axes, indexing="ij")
    dist = torch.sqrt(i * i + j * j + k * k)
    size = max(shape)
    return sum((dist <= r * size).to(torch.int32) for r in (0.45, 0.35, 0.25, 0.15)).to(dtype).unsqueeze(0)


def affine_matrix(kind):
    if kind == "identity":
        return torch.eye(4, dtype=torch.float64)
    if kind == "aniso":  # anisotropic spacing, offset origin
        m = torch.diag(torch.tensor([0.8, 1.25, 1.5, 1.0], dtype=torch.float64))
        m[:3, 3] = torch.tensor([-10.0, 4.0, 2.5], dtype=torch.float64)
        return m
    if kind == "oblique":  # rotated direction cosines + spacing
        a = 0.3
        r = torch.tensor(
            [[1, 0, 0], [0, torch.cos(torch.tensor(a)), -torch.sin(torch.tensor(a))],
             [0, torch.sin(torch.tensor(a)), torch.cos(torch.tensor(a))]], dtype=torch.float64)
        m = torch.eye(4, dtype=torch.float64)
        m[:3, :3] = r * torch.tensor([1.0, 0.9, 1.2], dtype=torch.float64)
        m[:3, 3] = torch.tensor([3.0, -2.0, 1.0], dtype=torch.float64)
        return m
    raise ValueError(kind)


# name, class, kwargs, shape, batch, affine kind, t1 dtype, seg dtype
CASES = [
    ("affine", "Affine", dict(degrees=(-10, 10), scales=(0.9, 1.1), translation=(-5, 5)), (16, 14, 12), 1, "identity", "float32", "int16"),
    ("affine_batch", "Affine", dict(degrees=(-10, 10), scales=(0.9, 1.1), translation=(-5, 5)), (12, 14, 16), 3, "aniso", "float32", "int32"),
    ("affine_pad0", "Affine", dict(degrees=(-25, 25), default_pad_value=0.0), (12, 12, 12), 1, "identity", "float32", "uint8"),
    ("affine_pad_number_label5", "Affine", dict(degrees=(-25, 25), default_pad_value=-3.5, default_pad_label=5), (12, 12, 12), 1, "identity", "float32", "int16"),
    ("affine_pad_mean", "Affine", dict(degrees=(-15, 15), default_pad_value="mean"), (12, 12, 12), 2, "identity", "float32", "int16"),
    ("affine_pad_otsu", "Affine", dict(degrees=(-15, 15), default_pad_value="otsu"), (12, 12, 12), 1, "identity", "float32", "int16"),
    ("affine_rot90_ties", "Affine", dict(degrees=(0, 0, 90)), (9, 9, 6), 1, "identity", "float32", "int16"),
    ("affine_half_voxel_ties", "Affine", dict(translation=(0.5, -0.5, 1.5), center="origin"), (8, 7, 6), 1, "identity", "float32", "int16"),
    ("affine_out_of_view", "Affine", dict(translation=(100.0, 0.0, 0.0)), (8, 8, 8), 1, "identity", "float32", "int16"),
    ("affine_2d", "Affine", dict(degrees=(-20, 20), scales=(0.8, 1.2), translation=(-2, 2)), (20, 18, 1), 1, "identity", "float32", "int16"),
    ("affine_isotropic_origin", "Affine", dict(scales=(0.8, 1.2), isotropic=True, degrees=(0, 0, -30, 30, 0, 0), center="origin"), (12, 12, 12), 1, "oblique", "float32", "int16"),
    ("affine_nearest_image", "Affine", dict(degrees=(-10, 10), image_interpolation="nearest"), (12, 12, 12), 1, "identity", "float32", "int16"),
    ("affine_f64", "Affine", dict(degrees=(-10, 10)), (10, 10, 10), 1, "identity", "float64", "int64"),
    ("affine_f16", "Affine", dict(degrees=(-10, 10)), (10, 10, 10), 1, "identity", "float16", "int8"),
    ("affine_p_gate", "Affine", dict(degrees=(-10, 10), p=0.5), (10, 10, 10), 4, "identity", "float32", "int16"),
    ("elastic", "ElasticDeformation", dict(), (16, 16, 16), 1, "identity", "float32", "int16"),
    ("elastic_batch", "ElasticDeformation", dict(max_displacement=(2.0, 6.0), num_control_points=(5, 6, 7), locked_borders=1), (14, 12, 16), 3, "aniso", "float32", "int16"),
    ("spatial_fused", "Spatial", dict(degrees=(-10, 10), scales=(0.9, 1.1), translation=(-5, 5), max_displacement=7.5), (16, 16, 16), 1, "identity", "float32", "int16"),
    ("spatial_elastic_first_batch_p", "Spatial", dict(degrees=(-10, 10), max_displacement=5.0, affine_first=False, p=0.6), (12, 12, 14), 4, "oblique", "float32", "int16"),
    ("spatial_target_and_affine", "Spatial", dict(target=1.5, degrees=(-10, 10), max_displacement=4.0), (16, 16, 16), 2, "aniso", "float32", "int16"),
    ("resample_2mm", "Resample", dict(target=2), (16, 14, 12), 1, "identity", "float32", "int16"),
    ("resample_aniso_antialias", "Resample", dict(target=(1.5, 0.8, 1.2), antialias=True), (16, 14, 12), 2, "identity", "float32", "int16"),
    ("resample_random_spacing", "Resample", dict(target=(0.8, 1.6)), (12, 12, 12), 1, "aniso", "float32", "int16"),
    ("bias", "BiasField", dict(), (16, 14, 12), 1, "identity", "float32", "int16"),
    ("bias_batch_p", "BiasField", dict(std=(0.2, 0.6), scale=0.3, p=0.6), (12, 12, 12), 4, "identity", "float32", "int16"),
    ("bias_f64", "BiasField", dict(), (10, 10, 10), 1, "identity", "float64", "int16"),
    ("blur", "Blur", dict(std=(0.5, 2)), (16, 14, 12), 1, "identity", "float32", "int16"),
    ("blur_batch_aniso_p", "Blur", dict(std=(0.0, 1.0, 0.5, 2.0, 0.3, 0.9), p=0.6), (12, 14, 16), 4, "aniso", "float32", "int16"),
    ("blur_single_axis", "Blur", dict(std=(0.0, 0.0, 1.3)), (10, 10, 12), 1, "identity", "float32", "int16"),
    ("noise", "Noise", dict(), (16, 14, 12), 1, "identity", "float32", "int16"),
    ("noise_batch_rician_p", "Noise", dict(mean=(-0.1, 0.1), std=(0.1, 0.3), rician=True, p=0.6), (10, 10, 10), 4, "identity", "float32", "int16"),
    ("noise_f64", "Noise", dict(std=0.1), (8, 8, 8), 2, "identity", "float64", "int16"),
    ("gamma", "Gamma", dict(log_gamma=(-0.3, 0.3)), (16, 14, 12), 1, "identity", "float32", "int16"),
    ("gamma_batch_p", "Gamma", dict(log_gamma=(-0.5, 0.5), p=0.6), (10, 10, 10), 4, "identity", "float32", "int16"),
    # label_interpolation="label": partial-volume resampling of label maps (spatial.py:1275-1389).
    # New cases are appended so that the seeds (1000 + index) of the earlier ones never move.
    ("affine_label_pv", "Affine", dict(degrees=(-10, 10), scales=(0.9, 1.1), translation=(-5, 5), label_interpolation="label"), (16, 14, 12), 1, "identity", "float32", "int16"),
    ("affine_label_pv_batch_pad7", "Affine", dict(degrees=(-25, 25), label_interpolation="label", default_pad_label=7), (12, 14, 16), 3, "aniso", "float32", "int32"),
    ("resample_label_pv_half_voxel_ties", "Resample", dict(target=2, label_interpolation="label"), (16, 14, 12), 1, "identity", "float32", "int16"),
    ("resample_label_pv_upsample", "Resample", dict(target=0.6, label_interpolation="label"), (10, 9, 8), 1, "identity", "float32", "uint8"),
    ("spatial_label_pv_elastic_p", "Spatial", dict(degrees=(-10, 10), max_displacement=5.0, label_interpolation="label", p=0.6), (12, 12, 14), 4, "oblique", "float32", "int16"),
    ("affine_label_pv_out_of_view", "Affine", dict(translation=(100.0, 0.0, 0.0), label_interpolation="label", default_pad_label=3), (8, 8, 8), 1, "identity", "float32", "int16"),
    ("affine_label_pv_40_labels", "Affine", dict(degrees=(-20, 20), scales=(0.8, 1.2), label_interpolation="label"), (16, 16, 16), 2, "identity", "float32", "int16", "many"),
    ("affine_label_pv_float_labels", "Affine", dict(degrees=(-15, 15), label_interpolation="label"), (12, 12, 12), 1, "identity", "float32", "float32", "many"),
    ("resample_label_pv_antialias", "Resample", dict(target=(1.5, 0.8, 2.2), antialias=True, label_interpolation="label"), (16, 14, 12), 2, "identity", "float32", "int16"),
    ("affine_label_pv_nearest_one_hot", "Affine", dict(degrees=(-10, 10), label_interpolation="label", one_hot_label_interpolation="nearest"), (12, 12, 12), 1, "identity", "float32", "int16"),
    ("affine_label_pv_multichannel_int", "Affine", dict(degrees=(-10, 10), translation=(-3, 3), label_interpolation="label"), (12, 12, 12), 2, "identity", "float32", "uint8", "onehot"),
    ("affine_label_pv_multichannel_f64", "Affine", dict(degrees=(-10, 10), label_interpolation="label", default_pad_label=2), (10, 10, 10), 1, "identity", "float32", "float64", "onehot"),
    # F.interpolate users: Resize, Anisotropy
    ("resize_mixed", "Resize", dict(target_shape=(20, 9, 17)), (12, 14, 10), 2, "aniso", "float32", "int16"),
    ("resize_down_cube_nearest_image", "Resize", dict(target_shape=7, image_interpolation="nearest"), (12, 14, 10), 1, "identity", "float32", "uint8"),
    ("resize_f16_to_one_voxel_axis", "Resize", dict(target_shape=(9, 1, 12)), (6, 5, 8), 1, "oblique", "float16", "int32"),
    ("resize_f64_many_labels", "Resize", dict(target_shape=(24, 20, 11)), (16, 16, 16), 1, "identity", "float64", "int64", "many"),
    ("anisotropy", "Anisotropy", dict(downsampling=(1.5, 5)), (16, 14, 12), 1, "identity", "float32", "int16"),
    ("anisotropy_batch_p", "Anisotropy", dict(axes=(0, 2), downsampling=(1.5, 4), p=0.6), (12, 14, 16), 4, "identity", "float32", "int16"),
    ("anisotropy_batch_nearest_image", "Anisotropy", dict(downsampling=(2, 6), image_interpolation="nearest"), (12, 10, 14), 3, "aniso", "float32", "uint8"),
    ("anisotropy_batch_shared", "Anisotropy", dict(axes=(1,), downsampling=(2, 3), per_instance=False), (10, 16, 12), 3, "identity", "float32", "int16"),
    ("anisotropy_f16_extreme_factor", "Anisotropy", dict(downsampling=40), (12, 12, 12), 2, "identity", "float16", "int16"),
    # images of different shapes resampled onto a named image: the reference samples all of them with the first image's grid
    ("resample_named_target_multires", "Resample", dict(target="t1"), (8, 10, 12), 1, "aniso", "float32", "int16", "multires"),
    ("resample_spacing_multires_batch", "Resample", dict(target=(1.5, 2.0, 1.0)), (8, 10, 12), 2, "identity", "float32", "uint8", "multires"),
    # Pad / Crop (pad.py:36-122, crop.py:33-112, _padding.py): element moves + an origin shift
    ("pad_six_constant_fill", "Pad", dict(padding=(2, 3, 1, 0, 4, 2), fill=-1.5), (8, 10, 12), 2, "aniso", "float32", "int16"),
    ("pad_one_value_oblique", "Pad", dict(padding=3), (6, 5, 4), 1, "oblique", "float64", "uint8"),
    ("pad_three_reflect", "Pad", dict(padding=(3, 2, 1), padding_mode="reflect"), (8, 10, 12), 2, "identity", "float32", "float32", "many"),
    ("pad_replicate_f16", "Pad", dict(padding=(0, 4, 2, 2, 5, 0), padding_mode="replicate"), (6, 7, 8), 1, "aniso", "float16", "float32", "many"),
    ("pad_circular_full_wrap", "Pad", dict(padding=(6, 1, 0, 7, 3, 3), padding_mode="circular"), (6, 7, 8), 2, "identity", "float32", "float32", "many"),
    ("pad_mean_batch", "Pad", dict(padding=(1, 2, 3), padding_mode="mean"), (10, 9, 8), 3, "identity", "float32", "int16"),
    ("pad_median_batch", "Pad", dict(padding=(2, 1, 2, 1, 0, 3), padding_mode="median"), (10, 9, 8), 3, "identity", "float32", "int16"),
    ("pad_minimum_multires", "Pad", dict(padding=2, padding_mode="minimum"), (8, 10, 12), 2, "identity", "float32", "int16", "multires"),
    ("crop_six", "Crop", dict(cropping=(2, 3, 1, 0, 4, 2)), (12, 10, 14), 2, "oblique", "float32", "int16"),
    ("crop_three_to_one_voxel_axis", "Crop", dict(cropping=(0, 2, 5)), (6, 7, 11), 1, "aniso", "float16", "int32"),
    # Flip (flip.py:76-236): per-axis coins, per-element axes, anatomical labels
    ("flip_axis0", "Flip", dict(axes=0), (8, 10, 12), 1, "identity", "float32", "int16"),
    ("flip_all_axes_coin_batch", "Flip", dict(axes=(0, 1, 2), flip_probability=0.5), (8, 10, 12), 4, "aniso", "float32", "int16"),
    ("flip_anatomical_oblique_f16", "Flip", dict(axes=("Left", "A")), (7, 9, 6), 2, "oblique", "float16", "uint8"),
    ("flip_batch_p_shared", "Flip", dict(axes=(1, 2), flip_probability=0.7, p=0.8, per_instance=False), (6, 6, 6), 3, "identity", "float64", "int64"),
    ("flip_multires", "Flip", dict(axes=(0, 2)), (8, 10, 12), 2, "identity", "float32", "int16", "multires"),
    # Motion (motion.py:32-561): rigid copies + k-space slabs; float rounding only (FFT vs GEMM evaluation)
    ("motion", "Motion", dict(), (16, 14, 12), 1, "identity", "float32", "int16"),
    ("motion_batch_p_three_events", "Motion", dict(degrees=(5, 25), translation=(-3, 3), num_transforms=3, p=0.6), (13, 10, 12), 4, "aniso", "float32", "int16"),
    ("motion_2d_shared_one_event", "Motion", dict(num_transforms=1, per_instance=False), (9, 12, 1), 3, "identity", "float32", "uint8"),
    ("motion_f64_many_events", "Motion", dict(degrees=3.0, translation=1.5, num_transforms=7), (8, 6, 7), 2, "oblique", "float64", "int16"),
    ("motion_f16", "Motion", dict(), (12, 12, 12), 1, "identity", "float16", "int16"),
]

COMPOSE = [
    ("Affine", dict(degrees=(-10, 10), scales=(0.9, 1.1), translation=(-5, 5))),
    ("ElasticDeformation", dict()),
    ("BiasField", dict()),
    ("Blur", dict(std=(0.5, 2))),
    ("Noise", dict()),
    ("Gamma", dict(log_gamma=(-0.3, 0.3))),
]


def label_map(shape, dtype, seg_kind, element):
    if seg_kind == "spheres":
        return spheres(shape, dtype)
    if seg_kind == "many":  # 40 non-contiguous label values (> 16: ATen's channel sum cascades), 4-voxel blocks
        axes = [torch.arange(s) // 4 for s in shape]
        i, j, k = torch.meshgrid(*axes, indexing="ij")
        index = (i * 7 + j * 3 + k + element) % 40
        return (index * 3 - 17).to(dtype).unsqueeze(0)
    if seg_kind == "onehot":  # an already one-hot map: three channels
        base = spheres(shape, torch.int16)[0]
        return torch.stack([(base == 0), (base == 1) | (base == 2), (base >= 3)]).to(dtype)
    if seg_kind == "multires":  # the label map at twice the resolution of the intensity image (same field of view)
        return spheres(tuple(2 * n for n in shape), dtype)
    raise ValueError(seg_kind)


CONTAINER_STEPS = [
    ("Affine", dict(degrees=(-10, 10), translation=(-2, 2))),
    ("Gamma", dict(log_gamma=(-0.3, 0.3))),
    ("Noise", dict(std=(0.05, 0.1))),
    ("Blur", dict(std=(0.5, 1.5))),
]
# name, class, container kwargs, batch size
CONTAINERS = [
    ("oneof_single", "OneOf", dict(), 1),
    ("oneof_weights_p_batch", "OneOf", dict(weights=[0.1, 0.4, 0.3, 0.2], p=0.7), 4),
    ("oneof_batch_shared", "OneOf", dict(per_instance=False), 3),
    ("someof_single_range", "SomeOf", dict(num_transforms=(1, 3)), 1),
    ("someof_batch_replace_p", "SomeOf", dict(num_transforms=(2, 4), replace=True, p=0.8), 3),
    ("someof_batch_shared", "SomeOf", dict(num_transforms=2, per_instance=False), 2),
]


def make_inputs(shape, batch, kind, t1_dtype, seg_dtype, seed, seg_kind="spheres"):
    g = torch.Generator().manual_seed(seed)
    items = []
    for element in range(batch):
        t1 = (torch.rand(2, *shape, generator=g) * 2 - 0.5).to(getattr(torch, t1_dtype))
        seg = label_map(shape, getattr(torch, seg_dtype), seg_kind, element)
        item = {"t1": t1, "seg": seg, "affine": affine_matrix(kind)}
        if seg_kind == "multires":
            finer = affine_matrix(kind).clone()
            finer[:3, :3] *= 0.5
            item["seg_affine"] = finer
        items.append(item)
    return items


def to_subjects(lib, items):
    return [
        lib.Subject(
            t1=lib.ScalarImage(it["t1"].clone(), affine=lib.AffineMatrix(it["affine"])),
            seg=lib.LabelMap(it["seg"].clone(), affine=lib.AffineMatrix(it.get("seg_affine", it["affine"]))),
        )
        for it in items
    ]


def run(lib, transform, items, seed):
    subjects = to_subjects(lib, items)
    data = subjects[0] if len(subjects) == 1 else lib.SubjectsBatch.from_subjects(subjects)
    torch.manual_seed(seed)
    out = transform(data)
    rng_probe = float(torch.rand(1).item())  # position of the global RNG after the call
    outs = [out] if len(subjects) == 1 else out.unbatch()
    history = [{"name": t.name, "params": t.params} for t in out.applied_transforms]
    result = {
        "history": history,
        "rng_probe": rng_probe,
        "t1": torch.stack([o.t1.data.contiguous() for o in outs]),
        "seg": torch.stack([o.seg.data.contiguous() for o in outs]),
        "affines": torch.stack([o.t1.affine.data.clone() for o in outs]),
        # what each unbatched element carries (per-element branches of OneOf / SomeOf live only here)
        "element_history": [[{"name": t.name, "params": t.params} for t in o.applied_transforms] for o in outs],
    }
    return out, result


def main():
    cases = []
    for index, (name, cls, kwargs, shape, batch, kind, t1_dtype, seg_dtype, *rest) in enumerate(CASES):
        items = make_inputs(shape, batch, kind, t1_dtype, seg_dtype, seed=1000 + index, seg_kind=rest[0] if rest else "spheres")
        transform = getattr(tio, cls)(**kwargs)
        out, result = run(tio, transform, items, seed=2000 + index)
        entry = {"name": name, "cls": cls, "kwargs": kwargs, "seed": 2000 + index, "inputs": items, "expected": result}
        if cls in ("Affine", "ElasticDeformation", "Spatial", "Resample", "BiasField", "Gamma", "Pad", "Crop", "Flip") and batch <= 2:
            restored = out.apply_inverse_transform()
            outs = [restored] if batch == 1 else restored.unbatch()
            entry["inverse"] = {
                "t1": torch.stack([o.t1.data.contiguous() for o in outs]),
                "seg": torch.stack([o.seg.data.contiguous() for o in outs]),
            }
        cases.append(entry)
        print(f"{name:32s} history={[h['name'] for h in result['history']]} out={tuple(result['t1'].shape)}")
    for batch in (1, 3):
        items = make_inputs((16, 16, 16), batch, "identity", "float32", "int16", seed=5000 + batch)
        transform = tio.Compose([getattr(tio, cls)(**kw) for cls, kw in COMPOSE])
        _, result = run(tio, transform, items, seed=6000 + batch)
        cases.append({"name": f"compose6_b{batch}", "cls": "Compose", "kwargs": {"steps": COMPOSE}, "seed": 6000 + batch,
                      "inputs": items, "expected": result})
        print(f"compose6_b{batch} history={[h['name'] for h in result['history']]}")
    # OneOf / SomeOf (compose.py:101-280): the containers only draw gates / choices and delegate
    for index, (name, cls, extra, batch) in enumerate(CONTAINERS):
        items = make_inputs((12, 12, 12), batch, "identity", "float32", "int16", seed=7000 + index)
        children = [getattr(tio, c)(**kw) for c, kw in CONTAINER_STEPS]
        if cls == "OneOf" and "weights" in extra:
            transform = tio.OneOf(dict(zip(children, extra["weights"])), **{k: v for k, v in extra.items() if k != "weights"})
        else:
            transform = getattr(tio, cls)(children, **extra)
        _, result = run(tio, transform, items, seed=8000 + index)
        cases.append({"name": name, "cls": cls, "kwargs": {"steps": CONTAINER_STEPS, "extra": extra}, "seed": 8000 + index,
                      "inputs": items, "expected": result})
        print(f"{name:32s} history={[h['name'] for h in result['history']]}")
    path = os.path.join(HERE, "transforms_golden.pt")
    torch.save({"torch": str(torch.__version__), "torchio": str(tio.__version__), "cases": cases}, path)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
