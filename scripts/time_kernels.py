"""Quick per-kernel timing at 256^3 (development aid; bench.py is the contract)."""
from __future__ import annotations

import sys
import time

import torch

sys.path.insert(0, ".")
from torchio_amd import ops  # noqa: E402

E = ops.engine()
dev = "cuda"
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
torch.manual_seed(0)
x = torch.rand(B, 1, S, S, S, device=dev)
V = x.numel() * 4


def timeit(name, fn, bytes_moved, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(iters):
        fn()
    end.record()
    torch.cuda.synchronize()
    ms = start.elapsed_time(end) / iters
    print(f"{name:42s} {ms*1e3:9.1f} us  {bytes_moved/ms/1e9:8.2f} TB/s(alg)  {B/ms*1e3:9.1f} vol/s", flush=True)


g = torch.Generator().manual_seed(0)
M = torch.eye(3, 4).repeat(B, 1, 1)
M[:, :, :3] += 0.05 * torch.randn(B, 3, 3, generator=g)
M[:, :, 3] = 3 * torch.randn(B, 3, generator=g)
c = (S - 1) / 2
M[:, :, 3] += c - (M[:, :, :3] @ torch.full((3,), c))
M = M.to(dev)
cp = ((torch.rand(B, 7, 7, 7, 3, generator=g) - 0.5) * 15).to(dev)
fill = torch.zeros(1, device=dev)
common = dict(out_shape=(S, S, S), in_spacing=(1, 1, 1), out_spacing=(1, 1, 1), affine_first=True)
timeit("copy (torch clone)", lambda: x.clone(), 2 * V)
timeit("resample affine linear nofill", lambda: E.resample3d([x], mapping=M, control_points=None, interps=["linear"], fills=[None], **common), 2 * V)
timeit("resample affine linear fill", lambda: E.resample3d([x], mapping=M, control_points=None, interps=["linear"], fills=[fill], **common), 2 * V)
timeit("resample affine+elastic linear fill", lambda: E.resample3d([x], mapping=M, control_points=cp, interps=["linear"], fills=[fill], **common), 2 * V)
timeit("resample affine+elastic nearest", lambda: E.resample3d([x], mapping=M, control_points=cp, interps=["nearest"], fills=[None], **common), 2 * V)
timeit("channel_min", lambda: E.channel_min(x), V / B)
r = 6
taps = torch.zeros(1, 3, 16)
for a, s in enumerate((2.0, 1.2, 0.7)):
    rr = max(int(-(-3 * s // 1)), 1)
    k = torch.exp(-0.5 * ((torch.arange(2 * rr + 1) - rr) / s) ** 2)
    taps[0, a, : 2 * rr + 1] = k / k.sum()
radius = [6, 4, 3]
taps = taps.to(dev)
timeit("separable_conv3d r=(6,4,3)", lambda: E.separable_conv3d(x, taps, radius), 2 * V)
coarse = (0.5 * torch.randn(B, 1, 6, 6, 6)).to(dev)
timeit("bias_field_apply", lambda: E.bias_field_apply(x, coarse), 2 * V)
timeit("add_noise philox", lambda: E.add_noise(x, 0.0, 0.25, philox_seed=1), 2 * V)
base = torch.randn_like(x)
timeit("add_noise base tensor", lambda: E.add_noise(x, 0.0, 0.25, base1=base), 2 * V)
timeit("gamma_pow", lambda: E.gamma_pow(x, 1.2), 2 * V)
t0 = time.perf_counter()
z = torch.randn(x.shape)
t1 = time.perf_counter()
zz = z.to(dev)
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host torch.randn {t1-t0:.3f}s  H2D {t2-t1:.3f}s for {B} volumes; host threads={torch.get_num_threads()}")
