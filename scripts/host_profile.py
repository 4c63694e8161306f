"""cProfile of the host side of one bench step on the GPU box (where does the enqueue time go?)."""
import cProfile
import os
import pstats
import sys
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torchio_amd as tio  # noqa: E402

warnings.simplefilter("ignore")
mode = os.environ.get("TIO_PROFILE_MODE", "philox,tight").split(",")  # noise rng, resample precision (the stencil's follows as in bench.py)
tio.set_noise_rng(mode[0]); tio.set_resample_precision(mode[1]); tio.set_stencil_precision(bench.stencil_mode(mode[1]))
transform = bench.build_transform()
batch = bench.make_batch(256, 8, 0, "cuda")
for _ in range(10):
    transform(batch)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(50):
    transform(batch)
enqueue = time.perf_counter() - t
torch.cuda.synchronize()
print("host enqueue ms/step", enqueue / 50 * 1e3, "total ms/step", (time.perf_counter() - t) / 50 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    transform(batch)
pr.disable()
torch.cuda.synchronize()
for key in (sys.argv[1:] or ["tottime", "cumulative"]):
    pstats.Stats(pr).sort_stats(key).print_stats(110)
