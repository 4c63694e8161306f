"""Launch each resample variant a few times (for rocprofv3 kernel-trace / PMC passes)."""
import sys
import torch
sys.path.insert(0, ".")
from torchio_amd import ops
E = ops.engine()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
which = sys.argv[3] if len(sys.argv) > 3 else "all"
dev = "cuda"
g = torch.Generator().manual_seed(0)
x = torch.rand(B, 1, S, S, S, device=dev)
import math
def rot(deg):
    a, b_, c = [math.radians(d) for d in deg]
    rx = torch.tensor([[1, 0, 0], [0, math.cos(a), -math.sin(a)], [0, math.sin(a), math.cos(a)]])
    ry = torch.tensor([[math.cos(b_), 0, math.sin(b_)], [0, 1, 0], [-math.sin(b_), 0, math.cos(b_)]])
    rz = torch.tensor([[math.cos(c), -math.sin(c), 0], [math.sin(c), math.cos(c), 0], [0, 0, 1]])
    return rz @ ry @ rx
M = torch.zeros(B, 3, 4)
for b in range(B):
    R = rot((torch.rand(3, generator=g) * 20 - 10).tolist()) * (0.9 + 0.2 * torch.rand(3, generator=g))
    c = torch.full((3,), (S - 1) / 2)
    M[b, :, :3] = R
    M[b, :, 3] = c - R @ c + (torch.rand(3, generator=g) * 10 - 5)
M = M.to(dev)
cp = ((torch.rand(B, 7, 7, 7, 3, generator=g) - 0.5) * 15)
cp[:, :2] = 0; cp[:, -2:] = 0; cp[:, :, :2] = 0; cp[:, :, -2:] = 0; cp[:, :, :, :2] = 0; cp[:, :, :, -2:] = 0
cp = cp.to(dev)
fill = torch.zeros(1, device=dev)
common = dict(out_shape=(S, S, S), in_spacing=(1, 1, 1), out_spacing=(1, 1, 1), affine_first=True)
for _ in range(3):
    if which in ("all", "affine"):
        E.resample3d([x], mapping=M, control_points=None, interps=["linear"], fills=[fill], **common)
    if which in ("all", "elastic"):
        E.resample3d([x], mapping=M, control_points=cp, interps=["linear"], fills=[fill], **common)
torch.cuda.synchronize()
print("done")
