"""GPU: the bench step issued from ONE host thread on S alternating HIP streams (step n on stream n mod S): consecutive steps
are independent, so their kernels may overlap on the device — no second host thread, no GIL contention.

    python scripts/bench_two_streams.py [--steps 60] [--streams 1,2,3]
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

warnings.simplefilter("ignore")


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--steps", type=int, default=60)
    parser.add_argument("--streams", default="1,2,3")
    parser.add_argument("--mode", default="philox,fast")
    args = parser.parse_args()
    rng_mode, precision = args.mode.split(",")
    tio.set_noise_rng(rng_mode)
    tio.set_resample_precision(precision)
    tio.set_stencil_precision(precision)
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    transform = bench.build_transform()
    batch = bench.make_batch(256, 8, 1234, device)
    for _ in range(100):
        out = transform(batch)
    torch.cuda.synchronize()
    for rep in range(2):
        for n_streams in [int(s) for s in args.streams.split(",")]:
            streams = [torch.cuda.Stream() for _ in range(n_streams)]
            outs = [None] * n_streams
            for step in range(20):  # warm the allocator's per-stream pools
                with torch.cuda.stream(streams[step % n_streams]):
                    outs[step % n_streams] = transform(batch)
            torch.cuda.synchronize()
            start = time.perf_counter()
            for step in range(args.steps):
                with torch.cuda.stream(streams[step % n_streams]):
                    outs[step % n_streams] = transform(batch)
            host = time.perf_counter() - start
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - start
            print(f"streams {n_streams}: {args.steps * 8 / elapsed:8.1f} volumes/s  {1e3 * elapsed / args.steps:.3f} ms/step  host {1e3 * host / args.steps:.3f} ms/step", flush=True)
            del outs


if __name__ == "__main__":
    main()
