"""Where the time of the fused "label" mode goes (512^3 int16 label map): torch.unique vs the TIO_LABEL_PV launch."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from parity_harness import nested_spheres
from torchio_amd import ops
e = ops.engine()
seg = nested_spheres(512).unsqueeze(0).cuda()
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
print("torch.unique            %.2f ms" % timed(lambda: torch.unique(seg)))
print("tio_unique_labels       %.2f ms" % timed(lambda: e.unique_labels(seg)))
table = torch.unique(seg).double()
m = torch.eye(3, 4)[None].clone(); m[0, :, :3] += 0.05 * torch.randn(3, 3); m = m.cuda()
kw = dict(out_shape=(512,)*3, mapping=m, control_points=None, in_spacing=(1,1,1), out_spacing=(1,1,1), affine_first=True, fills=[None])
print("label launch            %.2f ms" % timed(lambda: e.resample3d([seg], interps=["label"], label_tables=[table], pad_labels=[0.0], **kw)))
print("nearest launch          %.2f ms" % timed(lambda: e.resample3d([seg], interps=["nearest"], **kw)))
flat = torch.full_like(seg, 3)
print("label launch, uniform map %.2f ms" % timed(lambda: e.resample3d([flat], interps=["label"], label_tables=[torch.tensor([3.0], dtype=torch.float64)], pad_labels=[0.0], **kw)))
noisy = torch.randint(0, 5, seg.shape, device="cuda", dtype=torch.int16)
print("label launch, random map  %.2f ms" % timed(lambda: e.resample3d([noisy], interps=["label"], label_tables=[torch.arange(5, dtype=torch.float64)], pad_labels=[0.0], **kw)))
