"""Round 6: config 5 (one 512^3 subject: 2 x float32 + int16 labels through tio.Spatial) per step, by switch:
the label channel riding along the images' last launch (TIO_LEAN_LABEL), two channels per exact-coordinate launch (TIO_LEAN_PAIR)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchio_amd as tio  # noqa: E402
from torchio_amd import ops  # noqa: E402
from parity_harness import nested_spheres  # noqa: E402

device = torch.device("cuda:0")
big = int(os.environ.get("SIZE", "512"))
g = torch.Generator(device=device).manual_seed(5)
subject = tio.SubjectsBatch({
    "t1": tio.ImagesBatch(torch.rand(1, 1, big, big, big, generator=g, device=device), [tio.AffineMatrix()], image_class=tio.ScalarImage),
    "t2": tio.ImagesBatch(torch.rand(1, 1, big, big, big, generator=g, device=device) + 1, [tio.AffineMatrix()], image_class=tio.ScalarImage),
    "seg": tio.ImagesBatch(nested_spheres(big).unsqueeze(0).to(device), [tio.AffineMatrix()], image_class=tio.LabelMap),
})
fused = tio.Spatial(degrees=(-10, 10), scales=(0.9, 1.1), translation=(-5, 5), max_displacement=7.5)


def timed(steps=30):
    torch.manual_seed(4242)
    for _ in range(10):
        result = fused(subject)
    torch.cuda.synchronize()
    torch.manual_seed(4243)
    start = time.perf_counter()
    for _ in range(steps):
        result = fused(subject)
    torch.cuda.synchronize()
    return result, (time.perf_counter() - start) / steps * 1e3


out = {}
reference = None
for precision in ("tight", "exact"):
    tio.set_resample_precision(precision)
    for rep in range(2):
        for label in ("1", "0"):
            for pair in ("1", "0"):
                os.environ["TIO_LEAN_PAIR"] = pair
                os.environ["TIO_LEAN_LABEL"] = label
                ops.reload_env()
                result, ms = timed()
                key = f"{precision},label_rides={label},pair={pair}"
                out.setdefault(key, []).append(round(ms, 4))
                images = {name: result.images[name].data for name in ("t1", "t2", "seg")}
                if reference is None or reference[0] != precision:
                    reference = (precision, {k: v.clone() for k, v in images.items()})
                else:
                    for name, data in images.items():
                        assert torch.equal(reference[1][name], data), (key, name)
print(json.dumps(out, indent=1))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r6_config5_ab.json"), "w"), indent=1)
