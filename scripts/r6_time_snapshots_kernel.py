import ctypes as C, sys, torch
sys.path.insert(0, ".")
from torchio_amd import ops, _abi
st = ops.HostNormalStream(1)
fn = st._fn
S = 211
polys = torch.empty((S) * 624, dtype=torch.int32)
assert fn["host_mt19937_segment_polynomials"](1024, S, C.c_void_p(polys.data_ptr())) == 0
polys = polys.cuda()
plan = torch.zeros(656 + 1700 * 624, dtype=torch.int32, device="cuda")
plan[656:656+624] = torch.randint(0, 2**31-1, (624,), dtype=torch.int32, device="cuda")
raw = torch._C._cuda_getCurrentRawStream(0)
def run(total, seg):
    for _ in range(2): fn["mt19937_device_snapshots"](C.c_void_p(plan.data_ptr()), total, seg, C.c_void_p(polys.data_ptr()), C.c_void_p(raw))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): fn["mt19937_device_snapshots"](C.c_void_p(plan.data_ptr()), total, seg, C.c_void_p(polys.data_ptr()), C.c_void_p(raw))
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 5
print("211 segments x 1024 twists", run(211 * 1024, 1024))
print("211 segments x  128 twists", run(211 * 128, 128))
print("  1 segment  x 1024 twists (no jump)", run(1024, 1024))
print("  2 segments x 128", run(256, 128))
