"""GPU: the bench step issued from T host threads, each on its own HIP stream with its own batch (what `Queue`'s worker threads do
when every worker takes a stream): do the kernels of different steps overlap usefully?  Prints volumes/s for T = 1, 2, 3, 4.

    python scripts/bench_concurrent.py [--steps 40]
"""
import argparse
import json
import os
import sys
import threading
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import torchio_amd as tio  # noqa: E402

warnings.simplefilter("ignore")


def worker(index, transform, batch, steps, barrier, out):
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        torch.manual_seed(100 + index)
        for _ in range(10):
            result = transform(batch)
        stream.synchronize()
        barrier.wait()
        start = time.perf_counter()
        for _ in range(steps):
            result = transform(batch)
        stream.synchronize()
        out[index] = (start, time.perf_counter())
        del result


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--steps", type=int, default=40)
    parser.add_argument("--batch", type=int, default=8)
    parser.add_argument("--mode", default="philox,fast")
    parser.add_argument("--switch", type=float, default=0.0, help="sys.setswitchinterval (0 = Python's default, 5 ms)")
    parser.add_argument("--threads", default="1,2,3,4")
    args = parser.parse_args()
    if args.switch > 0:
        sys.setswitchinterval(args.switch)
    rng_mode, precision = args.mode.split(",")
    tio.set_noise_rng(rng_mode)
    tio.set_resample_precision(precision)
    tio.set_stencil_precision(precision)
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    results = {}
    for threads in [int(t) for t in args.threads.split(',')]:
        transforms = [bench.build_transform() for _ in range(threads)]
        batches = [bench.make_batch(256, args.batch, 1234 + t, device) for t in range(threads)]
        for t in range(threads):  # allocator / process warm-up on the default stream
            for _ in range(20):
                transforms[t](batches[t])
        torch.cuda.synchronize()
        barrier = threading.Barrier(threads)
        out = [None] * threads
        pool = [threading.Thread(target=worker, args=(t, transforms[t], batches[t], args.steps, barrier, out)) for t in range(threads)]
        for th in pool:
            th.start()
        for th in pool:
            th.join()
        torch.cuda.synchronize()
        elapsed = max(e for _, e in out) - min(s for s, _ in out)
        total = threads * args.steps * args.batch
        results[threads] = {"volumes_per_s": total / elapsed, "ms_per_step": 1e3 * elapsed / (threads * args.steps)}
        print(threads, "threads:", results[threads], flush=True)
    print(json.dumps(results))


if __name__ == "__main__":
    main()
