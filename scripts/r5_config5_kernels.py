"""Config 5 (512^3 subject: two float32 images + an int16 label map, fused tio.Spatial, tight mode), 30 calls — run under
`rocprofv3 --kernel-trace --stats` for the per-kernel breakdown of a call (scripts/r5_rocprof_config5.sh)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchio_amd as tio  # noqa: E402
from parity_harness import nested_spheres  # noqa: E402

device = torch.device("cuda:0")
big = 512
g = torch.Generator(device=device).manual_seed(5)
subject = tio.SubjectsBatch({
    "t1": tio.ImagesBatch(torch.rand(1, 1, big, big, big, generator=g, device=device), [tio.AffineMatrix()], image_class=tio.ScalarImage),
    "t2": tio.ImagesBatch(torch.rand(1, 1, big, big, big, generator=g, device=device) + 1, [tio.AffineMatrix()], image_class=tio.ScalarImage),
    "seg": tio.ImagesBatch(nested_spheres(big).unsqueeze(0).to(device), [tio.AffineMatrix()], image_class=tio.LabelMap),
})
tio.set_resample_precision("tight")
fused = tio.Spatial(degrees=(-10, 10), scales=(0.9, 1.1), translation=(-5, 5), max_displacement=7.5)
for _ in range(30):
    fused(subject)
torch.cuda.synchronize()
