"""Exact vs fast (TIO_PRECISION_FAST) resampling on the bench batch: launch time and maximum deviation."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_ops_parity import _control_points, _mapping
from torchio_amd import ops
e = ops.engine()
data = torch.rand(8, 1, 256, 256, 256, device="cuda")
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for name, cp in (("affine", None), ("elastic", _control_points(8, (7, 7, 7), 3, amplitude=7.5).cuda())):
    mapping = (_mapping(8, 2, scale=0.08, shift=5.0) if cp is None else torch.eye(3, 4)[None].repeat(8, 1, 1)).cuda()
    kw = dict(out_shape=(256,) * 3, mapping=mapping, control_points=cp, in_spacing=(1, 1, 1), out_spacing=(1, 1, 1), affine_first=True,
              interps=["linear"], fills=[torch.tensor([0.0], device="cuda") + 1e-9])
    exact = e.resample3d([data], precision="exact", **kw)[0]
    fast = e.resample3d([data], precision="fast", **kw)[0]
    err = (exact - fast).abs().max().item()
    t_exact = timed(lambda: e.resample3d([data], precision="exact", **kw))
    t_fast = timed(lambda: e.resample3d([data], precision="fast", **kw))
    gb = 2 * data.numel() * 4 / 1e9
    print(f"{name:8s} exact {t_exact:.3f} ms ({gb / t_exact / 8:.1%} of 8 TB/s)   fast {t_fast:.3f} ms ({gb / t_fast / 8:.1%})   max |exact - fast| = {err:.2e}")
