"""How the resampling roads behave beyond the bench's +-10 degrees: a batch of 8 x 256^3 float32 volumes under tio.Affine with
rotations fixed at R degrees about all three axes (the input box of a 16^3 output brick grows with the rotation; beyond the LDS
tile a brick takes its road's fallback).  Prints ms per call for every precision mode (and the exact mode on the brick kernel)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torchio_amd as tio  # noqa: E402

from torchio_amd.transforms import spatial as _sp  # noqa: E402

if os.environ.get("TIO_NO_LARGE_BOX_HINT", "") not in ("", "0"):  # A/B: the planned roads whatever the boxes
    _sp._expects_large_boxes = lambda *args: False
device = torch.device("cuda:0")
batch = bench.make_batch(256, 8, 0, device)
out = {}
for degrees in (5, 10, 15, 20, 30, 45):
    transform = tio.Affine(degrees=(degrees, degrees), scales=(1.0, 1.0), translation=(0, 0))
    for precision in ("tight", "exact", "fast"):
        tio.set_resample_precision(precision)
        for _ in range(5):
            transform(batch)
        torch.cuda.synchronize()
        start = time.perf_counter()
        for _ in range(20):
            transform(batch)
        torch.cuda.synchronize()
        out.setdefault(f"{degrees} deg", {})[precision] = round(1e3 * (time.perf_counter() - start) / 20, 3)
print(json.dumps({"what": "ms per Affine call, 8 x 256^3 f32, rotation about all three axes", "TIO_EXACT_LEAN": os.environ.get("TIO_EXACT_LEAN", "default"),
                  "large_box_hint": os.environ.get("TIO_NO_LARGE_BOX_HINT", "") in ("", "0"), **out}, indent=1))
