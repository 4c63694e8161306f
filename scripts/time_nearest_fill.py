"""GPU: the label kernel with and without a fill value (8 x 256^3 int16, affine + elastic), and config 5 with a pad label."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchio_amd as tio  # noqa: E402
from parity_harness import nested_spheres  # noqa: E402
from test_gpu_ops_parity import _control_points, _mapping  # noqa: E402
from torchio_amd import ops  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(reps):
        fn()
    end.record()
    torch.cuda.synchronize()
    return start.elapsed_time(end) / reps


def main():
    engine = ops.engine()
    batch, size = 8, 256
    seg = torch.randint(0, 5, (batch, 1, size, size, size), device="cuda").to(torch.int16)
    kwargs = dict(out_shape=(size,) * 3, mapping=_mapping(batch, 1, scale=0.08, shift=4.0).cuda(), control_points=_control_points(batch, (7, 7, 7), 2, amplitude=6.0).cuda(),
                  in_spacing=(1, 1, 1), out_spacing=(1, 1, 1), affine_first=True, interps=["nearest"])
    fill = torch.tensor([7.0], device="cuda")
    print(f"label kernel, int16 8 x 256^3, affine + elastic: no fill {timed(lambda: engine.resample3d([seg], fills=[None], **kwargs)):.3f} ms, "
          f"with a fill value {timed(lambda: engine.resample3d([seg], fills=[fill], **kwargs)):.3f} ms")
    os.environ["TIO_NEAREST_KERNEL"] = "0"
    ops.reload_env()
    print(f"  the same with a fill value on the exact gather road (round 3): {timed(lambda: engine.resample3d([seg], fills=[fill], **kwargs), 5):.3f} ms")
    os.environ.pop("TIO_NEAREST_KERNEL")
    ops.reload_env()
    big = 512
    g = torch.Generator(device="cuda").manual_seed(5)
    subject = tio.SubjectsBatch({
        "t1": tio.ImagesBatch(torch.rand(1, 1, big, big, big, generator=g, device="cuda"), [tio.AffineMatrix()], image_class=tio.ScalarImage),
        "t2": tio.ImagesBatch(torch.rand(1, 1, big, big, big, generator=g, device="cuda") + 1, [tio.AffineMatrix()], image_class=tio.ScalarImage),
        "seg": tio.ImagesBatch(nested_spheres(big).unsqueeze(0).cuda(), [tio.AffineMatrix()], image_class=tio.LabelMap),
    })
    for pad in (0, 9):
        fused = tio.Spatial(degrees=(-10, 10), scales=(0.9, 1.1), translation=(-5, 5), max_displacement=7.5, default_pad_label=pad)
        for precision in ("fast", "exact"):
            tio.set_resample_precision(precision)
            for _ in range(5):
                out = fused(subject)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(5):
                out = fused(subject)
            torch.cuda.synchronize()
            print(f"config 5 (512^3, 2 x f32 + int16 labels), default_pad_label={pad}, resample={precision}: {(time.perf_counter() - t) / 5 * 1e3:.3f} ms per subject")
            del out
    tio.set_resample_precision("exact")


if __name__ == "__main__":
    main()
