"""tio_channel_min on one 512^3 float32 channel (config 5's fill-value minima) and on a bench batch's element, per block cap."""
import os
import sys

import torch

sys.path.insert(0, ".")
from torchio_amd import ops  # noqa: E402

engine = ops.engine()
for shape in ((1, 1, 512, 512, 512), (8, 1, 256, 256, 256)):
    data = torch.rand(*shape, device="cuda")
    expected = data[0].amin().item()
    for cap in ("512", "1024", "2048", "4096"):
        os.environ["TIO_MIN_BLOCKS"] = cap
        for _ in range(3):
            out = engine.channel_min(data)
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(20):
            out = engine.channel_min(data)
        stop.record()
        torch.cuda.synchronize()
        ms = start.elapsed_time(stop) / 20
        n = data[0].numel() * 4
        print(f"{shape} cap {cap:>4}: {ms * 1e3:7.1f} us, {n / ms / 1e9:6.2f} TB/s, value ok: {out.item() == expected}")
