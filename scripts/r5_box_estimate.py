"""Calibration of transforms/spatial.py `_expects_large_boxes` against the planner: for random draws of the bench's parameter
ranges (and wider rotations) the fraction of a volume's bricks the planner could NOT stage (descriptor kind "slow") next to
the host-side estimate of the brick box in floats.  256^3, per-instance batches of 8."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torchio_amd as tio  # noqa: E402
from torchio_amd import ops  # noqa: E402
from torchio_amd.transforms import spatial as sp  # noqa: E402

device = torch.device("cuda:0")
batch = bench.make_batch(256, 8, 0, device)
first = batch.images["t1"]
tio.set_resample_precision("tight")
rows = []
captured = {}
original = sp._expects_large_boxes


def spy(mapping, displacements, field_shape, out_shape, in_spacing):
    captured["args"] = (None if mapping is None else mapping.copy(), displacements, field_shape, out_shape, in_spacing)
    return False


sp._expects_large_boxes = spy
for label, kwargs in (
    ("affine10", dict(degrees=(-10, 10), scales=(0.9, 1.1), translation=(-5, 5))),
    ("spatial10", dict(degrees=(-10, 10), scales=(0.9, 1.1), translation=(-5, 5), max_displacement=7.5)),
    ("elastic", dict(max_displacement=7.5)),
    ("affine15", dict(degrees=(-15, 15), scales=(0.9, 1.1), translation=(-5, 5))),
    ("affine20", dict(degrees=(-20, 20), scales=(0.9, 1.1), translation=(-5, 5))),
    ("spatial15", dict(degrees=(-15, 15), scales=(0.9, 1.1), translation=(-5, 5), max_displacement=7.5)),
):
    transform = tio.Spatial(**kwargs, per_instance=True)
    for seed in range(10):
        torch.manual_seed(seed)
        params = transform.make_params(batch)
        _, _, _, per_sample = sp._resolve_spatial_params(params)
        matrix, field, displacement, per_sample = sp._resolve_spatial_params(params)
        geometry = sp._prepare_launch_geometry(first, device, target_space=None, affine_matrix=matrix, control_points=field,
                                               max_displacement=displacement, per_sample=per_sample)
        plan = ops.engine().resample_plan(batch=8, in_shape=geometry.in_shape, out_shape=geometry.out_shape, mapping=geometry.mapping_dev,
                                          control_points=geometry.field_tensor, in_spacing=geometry.in_affine.spacing,
                                          out_spacing=geometry.out_affine.spacing, affine_first=params["affine_first"], cp_skip=geometry.cp_skip,
                                          passthrough=geometry.passthrough_all)
        torch.cuda.synchronize()
        desc = plan.cpu().numpy()[8 * 16:].reshape(-1, 16)
        kind = desc[:, 0] & 0xFF
        box = desc[:, 4].astype(np.int64) * desc[:, 5] * desc[:, 6] * 4
        per_element = kind.reshape(8, -1)
        boxes = box.reshape(8, -1)
        mapping, displacements, field_shape, out_shape, in_spacing = captured["args"]
        for n in range(8):
            one = None if mapping is None else mapping[n:n + 1]
            d = None if displacements is None else [displacements[n]] if len(displacements) == 8 else displacements
            # the estimate as floats: re-run the arithmetic of _expects_large_boxes for ONE element with a threshold sweep
            est = None
            for floats in range(4000, 60000, 200):
                sp._PLANNED_TILE_FLOATS = floats
                if not original(one, d, field_shape, out_shape, in_spacing):
                    est = floats
                    break
            sp._PLANNED_TILE_FLOATS = 13440
            staged = boxes[n][per_element[n] == 0]
            rows.append({"mapping": None if one is None else one.tolist(), "displacement": None if d is None else [None if x is None else list(map(float, x)) for x in d],
                         "field_shape": None if field_shape is None else list(field_shape), "case": label, "slow_fraction": float((per_element[n] == 2).mean()), "kinds": np.bincount(per_element[n], minlength=4).tolist(),
                         "planner_box_p50": float(np.median(staged)) if staged.size else None, "planner_box_max": float(staged.max()) if staged.size else None,
                         "estimate_floats": est})
for label in sorted({r["case"] for r in rows}):
    sel = [r for r in rows if r["case"] == label]
    slow = np.array([r["slow_fraction"] for r in sel])
    est = np.array([r["estimate_floats"] or 0 for r in sel])
    big = slow > 0.5
    print(label, "elements", len(sel), "mostly slow:", int(big.sum()), "| estimate > 13440:", int((est > 13440).sum()), "| both:", int((big & (est > 13440)).sum()),
          "| median planner max box", np.median([r["planner_box_max"] or 0 for r in sel]), "median estimate", np.median(est))
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "r5_box_estimate.json"), "w"))
