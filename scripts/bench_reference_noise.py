"""GPU: the two reference-noise modes of the bench step with the draws made AHEAD on the draw stream (tio_blur_fused with
explicit draws: the sum rides on the stencil's stores) against the previous road (draws + sum in one kernel behind the
stencil: TIO_NO_DRAW_STREAM=1), same process, alternating.

    python scripts/bench_reference_noise.py [--steps 40] [--rounds 2]
"""
import argparse
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


def timed(transform, batch, steps):
    for _ in range(25):
        out = transform(batch)
    torch.cuda.synchronize()
    start = time.perf_counter()
    for _ in range(steps):
        out = transform(batch)
    host = time.perf_counter() - start
    del out
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - start
    return elapsed / steps * 1e3, host / steps * 1e3


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--steps", type=int, default=40)
    parser.add_argument("--rounds", type=int, default=2)
    parser.add_argument("--modes", default="reference,fast;reference,exact;philox,fast")
    parser.add_argument("--toggle", default="TIO_NO_DRAW_STREAM", help="the Python-level switch that is alternated (unset / 1) inside the process")
    args = parser.parse_args()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    transform = bench.build_transform()
    batch = bench.make_batch(256, 8, 1234, device)
    tio.set_noise_rng("philox")
    for _ in range(60):
        transform(batch)
    torch.cuda.synchronize()
    for rep in range(args.rounds):
        for mode in args.modes.split(";"):
            rng_mode, precision = mode.split(",")
            tio.set_noise_rng(rng_mode)
            tio.set_resample_precision(precision)
            tio.set_stencil_precision(precision)
            # (one road per PROCESS for the stream's priority and the draw kernel's grid — TIO_DRAW_STREAM_PRIORITY, TIO_DRAW_BLOCKS_PER_CU:
            # a new stream has its own pool in the caching allocator, its first steps are device allocations)
            for road, flag in ((f"{args.toggle} unset", ""), (f"{args.toggle}=1", "1")):
                if rng_mode != "reference" and flag == "1" and args.toggle == "TIO_NO_DRAW_STREAM":
                    continue
                if flag == "1":
                    os.environ[args.toggle] = flag
                else:
                    os.environ.pop(args.toggle, None)
                torch.manual_seed(7)
                ms, host = timed(transform, batch, args.steps)
                print(f"round {rep} noise={rng_mode},resample={precision} [{road}]: {8e3 / ms:8.1f} volumes/s  {ms:.3f} ms/step  host {host:.3f} ms/step", flush=True)
    os.environ.pop(args.toggle, None)


if __name__ == "__main__":
    main()
