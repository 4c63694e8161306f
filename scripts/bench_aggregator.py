"""Dense-inference aggregation: device-side tio_patch_accumulate vs the reference's host loop.

    python scripts/bench_aggregator.py            # on the GPU box

Workload: a 256 x 256 x 176 volume, 128^3 patches with 64-voxel overlap (27 patches), 4-channel
float32 "model outputs" already on the GPU, batches of 4.  The reference's PatchAggregator
moves every batch to the host and adds it with Python slice assignments
(src/torchio/data/aggregator.py:94-99, 206-232); that loop is restated here with plain torch
ops on CPU tensors as the "before" (the reference itself cannot travel to the GPU box).
"""
from __future__ import annotations

import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torchio_amd as tio  # noqa: E402


def host_loop(outputs, locations, shape, mode):
    """aggregator.py as written: tensor.cpu() then one slice update per patch."""
    channels = outputs[0].shape[1]
    out = torch.zeros(channels, *shape)
    counts = torch.zeros(channels, *shape)
    window = None
    for batch, locs in zip(outputs, locations, strict=True):
        batch = batch.cpu()
        for patch, loc in zip(batch, locs, strict=True):
            si, sj, sk = loc.to_slices()
            if mode == "average":
                out[:, si, sj, sk] += patch
                counts[:, si, sj, sk] += 1
            else:
                if window is None:
                    window = torch.ones(1)
                    for dim, size in enumerate(patch.shape[-3:]):
                        view = [1, 1, 1]
                        view[dim] = size
                        window = window * torch.hann_window(size + 2, periodic=False)[1:-1].reshape(view)
                out[:, si, sj, sk] += patch * window
                counts[:, si, sj, sk] += window
    return out / counts.clamp(min=1)


def main():
    shape, patch, overlap, channels, batch_size = (256, 256, 176), 128, 64, 4, 4
    device = torch.device("cuda")
    subject = tio.Subject(t1=tio.ScalarImage(torch.zeros(1, *shape)))
    sampler = tio.GridSampler(subject, patch, overlap)
    g = torch.Generator(device=device).manual_seed(0)
    outputs, locations = [], []
    for start in range(0, len(sampler), batch_size):
        locs = sampler.locations[start : start + batch_size]
        outputs.append(torch.randn(len(locs), channels, patch, patch, patch, generator=g, device=device))
        locations.append(locs)
    results = {}
    for mode in ("average", "hann"):
        def device_run():
            aggregator = tio.PatchAggregator(shape, overlap_mode=mode, patch_overlap=overlap)
            for batch, locs in zip(outputs, locations, strict=True):
                aggregator.add_batch(batch, locs)
            return aggregator.get_output()

        got = device_run()
        torch.cuda.synchronize()
        start = time.perf_counter()
        reps = 10
        for _ in range(reps):
            got = device_run()
        torch.cuda.synchronize()
        device_s = (time.perf_counter() - start) / reps
        start = time.perf_counter()
        expected = host_loop(outputs, locations, shape, mode)
        host_s = time.perf_counter() - start
        patch_bytes = sum(o.numel() for o in outputs) * 4
        results[mode] = {
            "patches": len(sampler), "device_ms": 1e3 * device_s, "host_loop_ms": 1e3 * host_s,
            "speedup": host_s / device_s, "bit_exact_vs_host_loop": bool(torch.equal(got.cpu(), expected)),
            # per run: every patch read once, the touched accumulator region read + written once per batch
            "patch_GBps": patch_bytes / device_s / 1e9,
        }
    print(json.dumps({"workload": f"{shape} volume, {patch}^3 patches, overlap {overlap}, {channels} ch f32, batch {batch_size}", **results}))


if __name__ == "__main__":
    main()
