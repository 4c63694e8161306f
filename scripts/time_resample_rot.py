"""A/B: resample time vs rotation angle (cache-line divergence of the gathers)."""
import math, sys, torch
sys.path.insert(0, ".")
from torchio_amd import ops
E = ops.engine(); dev = "cuda"; S = 256
x = torch.rand(1, 1, S, S, S, device=dev)
fill = torch.zeros(1, device=dev)
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for deg in (0.0, 1.0, 3.0, 10.0, 30.0):
    for axis in ("z", "x"):
        c, s = math.cos(math.radians(deg)), math.sin(math.radians(deg))
        R = torch.tensor([[c, -s, 0], [s, c, 0], [0, 0, 1.0]]) if axis == "z" else torch.tensor([[1.0, 0, 0], [0, c, -s], [0, s, c]])
        ctr = torch.full((3,), (S - 1) / 2)
        M = torch.zeros(1, 3, 4); M[0, :, :3] = R; M[0, :, 3] = ctr - R @ ctr + torch.tensor([0.3, 0.4, 0.2])
        M = M.to(dev)
        t = timeit(lambda: E.resample3d([x], out_shape=(S, S, S), mapping=M, control_points=None, in_spacing=(1, 1, 1),
                                        out_spacing=(1, 1, 1), affine_first=True, interps=["linear"], fills=[fill]))
        print(f"rot {deg:5.1f} deg about {axis}: {t:8.1f} us")
