"""Time tio_host_mt19937_plan (the host half of the device-side noise stream) per thread count on this host."""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, ".")
from torchio_amd import _abi, _lib  # noqa: E402

_, fn = _lib.load()


def state(seed):
    st = (C.c_uint64 * (_abi.HOST_MT_STATE_BYTES // 8))()
    assert fn["host_mt19937_seed"](C.addressof(st), seed) == 0
    return st


n = 8 * 256**3
words = fn["host_mt19937_plan_words"](n)
plan = torch.zeros(words, dtype=torch.int32).pin_memory() if torch.cuda.is_available() else torch.zeros(words, dtype=torch.int32)
used = C.c_int64()
reference = None
for threads in (1, 2, 4, 8, 12, 16, 24, 32, 64):
    best = 1e9
    for rep in range(5):
        st = state(7)
        t = time.perf_counter()
        assert fn["host_mt19937_plan"](C.addressof(st), n, C.c_void_p(plan.data_ptr()), words, C.byref(used), threads) == 0
        best = min(best, time.perf_counter() - t)
    if reference is None:
        reference = bytes(st)
    print(f"threads {threads:2d}: plan of {n} draws in {best * 1e3:6.2f} ms (best of 5), state as the serial one: {bytes(st) == reference}")
