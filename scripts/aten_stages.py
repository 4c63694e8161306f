import sys, time, torch
sys.path.insert(0, 'tests')
import aten_pipeline as ap
dev = 'cuda'
b, s = 2, 256
data = torch.rand(b, 1, s, s, s, device=dev)
rng = torch.Generator().manual_seed(0)
def t(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
m = torch.eye(3, 4).repeat(b, 1, 1); m[:, :, :3] += 0.05 * torch.rand(b, 3, 3); m = m.to(dev)
cp = (torch.rand(b, 7, 7, 7, 3) * 15 - 7.5).to(dev)
fill = torch.zeros(1, device=dev)
print("resample affine  ms/2vol", t(lambda: ap.resample(data, m, None, fill)))
print("resample elastic ms/2vol", t(lambda: ap.resample(data, torch.eye(3,4).repeat(b,1,1).to(dev), cp, fill)))
print("bias             ms/2vol", t(lambda: ap.bias_field(data, torch.randn(b, 1, 6, 6, 6, device=dev))))
sig = torch.tensor([[1.0, 1.5, 2.0], [0.7, 1.2, 1.9]], device=dev)
print("blur             ms/2vol", t(lambda: ap.blur(data, sig)))
print("noise            ms/2vol", t(lambda: ap.noise(data, torch.full((b,), 0.25, device=dev))))
