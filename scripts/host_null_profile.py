"""Host-side cost of one bench step WITHOUT a GPU: the HIP entry points replaced by no-ops (test infrastructure only).

Runs in the build container: every `tio_*` call returns TIO_OK at once, tensors live on the host, so the time per step is
the Python / marshalling time that `bench.py` reports as `host_enqueue_ms_per_step` (minus the driver's launch cost).
    python scripts/host_null_profile.py [tottime|cumtime] [--profile]
"""
import cProfile
import os
import pstats
import sys
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import torchio_amd as tio  # noqa: E402
from torchio_amd import _abi, ops  # noqa: E402
from torchio_amd.data import _pending  # noqa: E402

warnings.simplefilter("ignore")


class _Null(dict):
    def __missing__(self, key):
        return lambda *a: 0

    def __contains__(self, key):
        return key != "last_error"


engine = ops.Engine(_Null(), "cpu", "null")
ops._ENGINE = engine
_pending.eligible = lambda data: _pending.enabled() and data.dtype == torch.float32 and data.ndim == 5 and not data.requires_grad
ops.h2d = lambda tensor, device: tensor
from torchio_amd.transforms import spatial as _sp
_sp._folding_grid_spacing = lambda extent, mesh: float("inf")  # (16^3 stand-in volumes: the 256^3 bench never trips the folding warning)
tio.set_noise_rng("philox")
tio.set_resample_precision("tight")
transform = bench.build_transform()
batch = bench.make_batch(16, 8, 0, "cpu")
for _ in range(200):
    transform(batch)
for rep in range(3):
    t = time.perf_counter()
    for _ in range(200):
        out = transform(batch)
    print("host ms/step (null engine)", (time.perf_counter() - t) / 200 * 1e3)
if "--profile" in sys.argv:
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        transform(batch)
    pr.disable()
    key = [a for a in sys.argv[1:] if not a.startswith("--")]
    pstats.Stats(pr).sort_stats(key[0] if key else "tottime").print_stats(60)
