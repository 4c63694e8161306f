"""GPU box: where the HOST time of one bench step goes, per transform phase and per engine entry (perf_counter around
`make_params` / `_prefetch` / `apply_transform` of every child and around the host-side pieces of the noise stream).
    python scripts/host_segments_gpu.py [philox,fast|reference,fast|reference,exact]
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

warnings.simplefilter("ignore")
mode = (sys.argv[1] if len(sys.argv) > 1 else "philox,fast").split(",")
tio.set_noise_rng(mode[0]); tio.set_resample_precision(mode[1]); tio.set_stencil_precision(mode[1])
transform = bench.build_transform()
batch = bench.make_batch(256, 8, 0, "cuda")
spent = collections.OrderedDict()
calls = collections.Counter()


def timed(obj, name, label):
    inner = getattr(obj, name, None)
    if inner is None:
        return

    def wrapper(*a, **k):
        t = time.perf_counter()
        try:
            return inner(*a, **k)
        finally:
            spent[label] = spent.get(label, 0.0) + time.perf_counter() - t
            calls[label] += 1
    setattr(obj, name, wrapper)


for child in transform.transforms:
    for phase in ("make_params", "_prefetch", "apply_transform"):
        timed(child, phase, type(child).__name__ + "." + phase)
engine = ops.engine()
for entry in ("resample3d", "blur_fused", "channel_min", "add_noise", "philox_normal"):
    timed(engine, entry, "  engine." + entry)
for entry in ("prefetch_plan", "_device_plan", "randn_ahead", "_randn_on_device", "add_noise", "__init__"):
    timed(ops.HostNormalStream, entry, "  HostNormalStream." + entry)
timed(ops, "h2d_packed", "  ops.h2d_packed")
timed(ops, "on_draw_stream", "  ops.on_draw_stream")
for _ in range(30):
    transform(batch)
torch.cuda.synchronize()
spent.clear(); calls.clear()
N = 60
t0 = time.perf_counter()
for _ in range(N):
    transform(batch)
host = (time.perf_counter() - t0) / N * 1e3
torch.cuda.synchronize()
total = (time.perf_counter() - t0) / N * 1e3
print(f"mode {mode}: host {host:.3f} ms/step, total {total:.3f} ms/step")
top = sum(v for k, v in spent.items() if not k.startswith("  "))
for label, seconds in spent.items():
    print(f"{label:44s} {seconds / N * 1e3:7.3f} ms  ({calls[label] / N:.1f} calls/step)")
print(f"{'envelope (copy, history, Compose)':44s} {host - top / N * 1e3:7.3f} ms")
