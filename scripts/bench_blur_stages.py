"""Stage timing of the Blur path (I pass, fused J+K pass, with / without BiasField and Noise) on 8 x 256^3 f32.

    python scripts/bench_blur_stages.py [radius] [exact|fast]     # on the GPU box; TIO_CONV_RING=1 selects the LDS-ring kernels
"""
from __future__ import annotations

import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from torchio_amd import ops  # noqa: E402
from torchio_amd.transforms.blur import _stacked_gaussian_taps  # noqa: E402


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(reps):
        fn()
    end.record()
    torch.cuda.synchronize()
    return start.elapsed_time(end) / reps * 1e3  # us


def main():
    radius = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    precision = sys.argv[2] if len(sys.argv) > 2 else "exact"
    ops.set_stencil_precision(precision)
    engine = ops.engine()
    data = torch.rand(8, 1, 256, 256, 256, device="cuda")
    sigma = radius / 3.0
    import numpy as np

    taps, radii, _ = _stacked_gaussian_taps(np.full((8, 3), sigma))
    taps = taps.cuda()
    coarse = (0.5 * torch.randn(8, 1, 6, 6, 6)).cuda()
    std = torch.full((8,), 0.25, device="cuda")
    mean = torch.zeros(8, device="cuda")
    rows = {
        "blur (I pass + fused J+K)": lambda: engine.blur_fused(data, taps, radii),
        "bias + blur": lambda: engine.blur_fused(data, taps, radii, bias_coarse=coarse),
        "blur + noise": lambda: engine.blur_fused(data, taps, radii, noise=(mean, std, 1234)),
        "bias + blur + noise": lambda: engine.blur_fused(data, taps, radii, bias_coarse=coarse, noise=(mean, std, 1234)),
        "separable_conv3d (same kernels, no pointwise)": lambda: engine.separable_conv3d(data, taps, radii),
    }
    draws = torch.randn(data.shape, device="cuda")
    rows["bias + blur + explicit draws (the reference's stream)"] = lambda: engine.blur_fused(data, taps, radii, bias_coarse=coarse, noise=(mean, std, draws))
    print(f"radius {list(radii)}  ring={'1' if os.environ.get('TIO_CONV_RING') else '0'}  stencil precision {precision}")
    for name, fn in rows.items():
        print(f"{name:48s} {timed(fn):8.1f} us")


if __name__ == "__main__":
    main()
