"""Host-side enqueue time per bench step on the GPU box: first vs warm calls, GC on / off, and the host-only floor (tiny volumes)."""
import gc, os, sys, time, warnings
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torchio_amd as tio
warnings.simplefilter("ignore")
tio.set_noise_rng("philox")
transform = bench.build_transform()
batch = bench.make_batch(256, 8, 0, "cuda")
def run(n, label):
    for _ in range(5): transform(batch)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): transform(batch)
    e = time.perf_counter() - t
    torch.cuda.synchronize()
    print(f"{label:24s} enqueue {e/n*1e3:.3f} ms/step   total {(time.perf_counter()-t)/n*1e3:.3f} ms/step")
run(100, "baseline")
run(100, "baseline again")
gc.disable(); run(100, "gc disabled"); gc.enable()
gc.collect(); gc.freeze(); run(100, "gc frozen")
small = bench.make_batch(32, 8, 0, "cuda")
batch = small; run(100, "tiny volumes (host only)")
