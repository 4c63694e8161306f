"""Round 6: who runs the mt19937 state chain of the reference's noise stream — the host's worker threads or the device
(`ops.set_noise_plan`).  Host wall time and device time of the PLAN of a bench batch's 134 M draws, and of plan + draws."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from torchio_amd import ops  # noqa: E402

n = 8 * 256**3
out = {}
for where in ("host", "device"):
    ops.set_noise_plan(where)
    for _ in range(3):
        ops.HostNormalStream(3)._device_plan(n, torch.device("cuda"))
    torch.cuda.synchronize()
    host_ms, dev_ms = [], []
    for rep in range(10):
        stream = ops.HostNormalStream(rep)
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        start.record()
        plan = stream._device_plan(n, torch.device("cuda"))
        end.record()
        host_ms.append(1e3 * (time.perf_counter() - t0))
        torch.cuda.synchronize()
        dev_ms.append(start.elapsed_time(end))
    total = []
    for rep in range(10):
        stream = ops.HostNormalStream(rep)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        stream.randn((n,), "cuda")
        torch.cuda.synchronize()
        total.append(1e3 * (time.perf_counter() - t0))
    out[where] = {"plan_host_ms": sorted(host_ms)[len(host_ms) // 2], "plan_device_ms": sorted(dev_ms)[len(dev_ms) // 2],
                  "plan_plus_draws_wall_ms": sorted(total)[len(total) // 2], "threads": ops.HostNormalStream(0).threads}
print(json.dumps(out))
