"""GPU: Motion on the engine (2 rigid resamples + one MFMA GEMM) vs the reference's op sequence on stock ATen.

    python scripts/bench_motion.py [--size 256] [--batch 4] [--events 2]

Prints one JSON line: kernel time of `tio_kspace_segment_mix` (HIP events, TFLOP/s of the real GEMM), the
whole transform, and `affine_grid` + `grid_sample` + `fftn` / `ifftn` with torch on the same device.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchio_amd as tio  # noqa: E402
from torchio_amd import ops  # noqa: E402
from torchio_amd.transforms.motion import _rigid_voxel_mappings  # noqa: E402
from torchio_amd.transforms.motion import _rotation_matrices  # noqa: E402
from torchio_amd.transforms.motion import _segment_bounds  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(reps):
        fn()
    end.record()
    torch.cuda.synchronize()
    return start.elapsed_time(end) / reps


def aten_motion(data, degrees, translations, bounds):
    import torch.nn.functional as F

    batch, channels, *shape = data.shape
    spectrum = torch.fft.fftn(data, dim=(-3, -2, -1))
    for index, (deg, tr) in enumerate(zip(degrees, translations), start=1):
        theta = torch.zeros(batch, 3, 4, device=data.device)
        theta[:, :3, :3] = _rotation_matrices(deg).to(data.device)
        theta[:, :3, 3] = (tr / (torch.tensor(shape, dtype=torch.float32) / 2)).to(data.device)
        grid = F.affine_grid(theta, [batch, 1, *shape], align_corners=True)
        moved = F.grid_sample(data, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
        spectrum[:, :, bounds[index] : bounds[index + 1]] = torch.fft.fftn(moved, dim=(-3, -2, -1))[:, :, bounds[index] : bounds[index + 1]]
    return torch.fft.ifftn(spectrum, dim=(-3, -2, -1)).real


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--size", type=int, default=256)
    parser.add_argument("--batch", type=int, default=4)
    parser.add_argument("--events", type=int, default=2)
    parser.add_argument("--reps", type=int, default=10)
    parser.add_argument("--profile", action="store_true", help="cProfile of the public Motion call (stderr)")
    args = parser.parse_args()
    engine = ops.engine()
    shape = (args.size,) * 3
    torch.manual_seed(0)
    data = torch.rand(args.batch, 1, *shape, device="cuda")
    degrees = [torch.rand(args.batch, 3) * 20 - 10 for _ in range(args.events)]
    translations = [torch.rand(args.batch, 3) * 20 - 10 for _ in range(args.events)]
    bounds = _segment_bounds(args.events + 1, shape[0])

    def moved_images():
        images = [data]
        for deg, tr in zip(degrees, translations):
            mapping = ops.h2d(_rigid_voxel_mappings(deg, tr, shape), data.device)
            images.append(engine.resample3d([data], out_shape=shape, mapping=mapping, control_points=None, in_spacing=(1, 1, 1),
                                            out_spacing=(1, 1, 1), affine_first=True, interps=["linear"], fills=[None])[0])
        return images

    images = moved_images()
    mix_ms = timed(lambda: engine.kspace_segment_mix(images, bounds, torch.float32), args.reps)
    whole_ms = timed(lambda: engine.kspace_segment_mix(moved_images(), bounds, torch.float32), args.reps)
    # executed: the still image's block is not multiplied (out = x_0 + sum_{s>=1} W_s (x_s - x_0))
    flops = 2.0 * args.batch * shape[0] * args.events * shape[0] * shape[1] * shape[2]
    ours = engine.kspace_segment_mix(images, bounds, torch.float32)
    reference = aten_motion(data, degrees, translations, bounds)
    error = float((ours - reference).abs().max())
    del reference
    aten_ms = timed(lambda: aten_motion(data, degrees, translations, bounds), max(2, args.reps // 3))
    subject_batch = tio.SubjectsBatch.from_subjects([tio.Subject(t1=tio.ScalarImage(data[i])) for i in range(args.batch)])
    transform = tio.Motion(num_transforms=args.events, copy=False)
    # public call, synchronised per call: back-to-back enqueueing of a short loop mostly measures the pinned
    # staging pool growing (one hipHostMalloc per call until enough blocks exist), not the transform
    import time

    for _ in range(6):
        transform(subject_batch).t1.data
    torch.cuda.synchronize()
    begin = time.perf_counter()
    for _ in range(args.reps):
        transform(subject_batch).t1.data
        torch.cuda.synchronize()
    api_ms = (time.perf_counter() - begin) * 1e3 / args.reps
    if args.profile:
        import time

        walls = []
        for _ in range(12):
            t0 = time.perf_counter()
            transform(subject_batch).t1.data
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            walls.append((round((t1 - t0) * 1e3, 2), round((time.perf_counter() - t0) * 1e3, 2)))
        print("per call (host ms, host+sync ms):", walls, "history length", len(subject_batch.applied_transforms), file=sys.stderr)
        import cProfile
        import pstats

        profiler = cProfile.Profile()
        profiler.enable()
        for _ in range(10):
            transform(subject_batch).t1.data
        torch.cuda.synchronize()
        profiler.disable()
        pstats.Stats(profiler, stream=sys.stderr).sort_stats("tottime").print_stats(14)
    print(json.dumps({
        "workload": f"Motion(num_transforms={args.events}) on {args.batch} x 1 x {args.size}^3 float32",
        "kspace_segment_mix_ms": round(mix_ms, 3), "gemm_tflops_executed": round(flops / mix_ms / 1e9, 1), "f32_mfma_peak_tflops": 157.3,
        "resamples_plus_mix_ms": round(whole_ms, 3), "tio_motion_call_synchronised_ms": round(api_ms, 3),
        "aten_grid_sample_fft_ms": round(aten_ms, 3), "speedup_vs_aten": round(aten_ms / whole_ms, 2),
        "max_abs_diff_vs_aten": error,
    }))


if __name__ == "__main__":
    main()
