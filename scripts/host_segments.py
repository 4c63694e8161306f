"""Where the host time of one bench step goes, per transform and phase, WITHOUT a GPU (the null engine of
`host_null_profile.py`, perf_counter around every `make_params` / `apply_transform`; test infrastructure only).
    python scripts/host_segments.py [philox|reference]
"""
import collections
import os
import sys
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import torchio_amd as tio  # noqa: E402
from torchio_amd import ops  # noqa: E402
from torchio_amd.data import _pending  # noqa: E402

warnings.simplefilter("ignore")


class _Null(dict):
    def __missing__(self, key):
        return lambda *a: 0

    def __contains__(self, key):
        return key != "last_error"


ops._ENGINE = ops.Engine(_Null(), "cpu", "null")
_pending.eligible = lambda data: _pending.enabled() and data.dtype == torch.float32 and data.ndim == 5 and not data.requires_grad
ops.h2d = lambda tensor, device: tensor
from torchio_amd.transforms import spatial as _sp  # noqa: E402
_sp._folding_grid_spacing = lambda extent, mesh: float("inf")
tio.set_noise_rng("philox")
tio.set_resample_precision("fast")
transform = bench.build_transform()
batch = bench.make_batch(16, 8, 0, "cpu")
spent = collections.OrderedDict()


def timed(obj, name, label):
    inner = getattr(obj, name)

    def wrapper(*a, **k):
        t = time.perf_counter()
        try:
            return inner(*a, **k)
        finally:
            spent[label] = spent.get(label, 0.0) + time.perf_counter() - t
    setattr(obj, name, wrapper)


for child in transform.transforms:
    timed(child, "make_params", type(child).__name__ + ".make_params")
    timed(child, "apply_transform", type(child).__name__ + ".apply_transform")
for _ in range(300):
    transform(batch)
spent.clear()
N = 500
t0 = time.perf_counter()
for _ in range(N):
    transform(batch)
total = (time.perf_counter() - t0) / N * 1e3
for label, seconds in spent.items():
    print(f"{label:40s} {seconds / N * 1e3:7.3f} ms")
print(f"{'envelope (copy, history, Compose)':40s} {total - sum(spent.values()) / N * 1e3:7.3f} ms")
print(f"{'step':40s} {total:7.3f} ms")
