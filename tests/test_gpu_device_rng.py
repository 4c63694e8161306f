"""`torch.randn(n, generator=cpu_gen)` drawn ON THE DEVICE (csrc/mt19937.hip) from the host's plan of the mt19937 state
chain (csrc/host_rng.cpp: tio_host_mt19937_plan).  Bit-identical to torch's CPU stream — the reference's Noise draws from
exactly that (transforms/intensity/noise.py:108-116) — for every count, continuation and tail the transform can produce.
"""
from __future__ import annotations

import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["host", "device"])
def plan_where(request):
    """Who runs the mt19937 state chain (round 6, `ops.set_noise_plan`): this rank's host threads, or the device — one workgroup per
    segment jumps to its first state (a correlation of the jump polynomial's bits with the generator's word sequence) and chains."""
    from torchio_amd import ops

    previous = ops.get_noise_plan()
    ops.set_noise_plan(request.param)
    yield request.param
    ops.set_noise_plan(previous)


def _torch_stream(seed, counts):
    generator = torch.Generator().manual_seed(seed)
    return [torch.randn(count, generator=generator) for count in counts]


@pytest.mark.parametrize("seed", [0, 1234567, 2**31 - 1])
@pytest.mark.parametrize(
    "counts",
    [
        [1 << 20],                        # whole groups, whole units
        [(1 << 20) + 16 * 39 * 3 + 32],   # ends inside a state block, on a group boundary
        [1_500_003],                      # not a multiple of 16: torch's tail rule (16 fresh draws)
        [1 << 20, 1 << 21, 1 << 20],      # one generator, three images (the second call starts inside a block)
        [1_200_000 + 7, 1 << 20],         # after a tail the stream stands inside a group: the second draw takes the host road
        [64 * 64 * 64, 1 << 22],          # a small draw first (host road), then a large one
        [6_000_000 + 16, 12_000_000 + 5], # long chains: the host jumps ahead to the segments of its plan (host_rng_jump.cpp)
    ],
)
def test_device_draws_equal_torch_randn(hip, plan_where, seed, counts):
    from torchio_amd import ops

    stream = ops.HostNormalStream(seed)
    got = [stream.randn((count,), "cuda") for count in counts]
    torch.cuda.synchronize()
    for expected, result in zip(_torch_stream(seed, counts), got):
        assert torch.equal(expected.view(torch.int32), result.cpu().view(torch.int32))  # bit for bit (signed zeros included)


def test_bench_batch_of_draws_equals_torch_randn(hip, plan_where):
    """8 x 256^3: 134 M draws, 1 681 snapshots."""
    from torchio_amd import ops

    shape = (8, 1, 256, 256, 256)
    result = ops.HostNormalStream(99).randn(shape, "cuda")
    expected = torch.randn(shape, generator=torch.Generator().manual_seed(99))
    assert torch.equal(expected.view(torch.int32), result.cpu().view(torch.int32))


def test_device_and_host_roads_agree_and_leave_the_same_state(hip, plan_where, monkeypatch):
    from torchio_amd import ops

    counts = [3_000_000, 1_000_000 + 9, 2_000_000]
    device_stream = ops.HostNormalStream(5)
    on_device = [device_stream.randn((count,), "cuda").cpu() for count in counts]
    monkeypatch.setenv("TIO_DEVICE_RNG", "0")
    host_stream = ops.HostNormalStream(5)
    on_host = [host_stream.randn((count,), "cuda").cpu() for count in counts]
    for a, b in zip(on_device, on_host):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    # (a device-made plan leaves the host state OWING its twists: one more draw settles it — and both streams go on alike)
    assert torch.equal(device_stream.randn((4096,), "cpu").view(torch.int32), host_stream.randn((4096,), "cpu").view(torch.int32))
    assert bytes(device_stream._state) == bytes(host_stream._state)


def test_plan_refuses_what_it_cannot_express_and_leaves_the_state_alone(hip):
    from torchio_amd import _abi
    from torchio_amd import ops

    stream = ops.HostNormalStream(3)
    stream.randn((1_000_000 + 5,), "cpu")  # a tail: the stream now stands inside a group of 16
    before = bytes(stream._state)
    words = int(stream._fn["host_mt19937_plan_words"](1 << 20))
    plan = torch.empty(words, dtype=torch.int32)
    used = C.c_int64(0)
    status = stream._fn["host_mt19937_plan"](C.addressof(stream._state), 1 << 20, C.c_void_p(plan.data_ptr()), words, C.byref(used), 1)
    assert status == _abi.UNSUPPORTED_CONFIG and bytes(stream._state) == before
    status = stream._fn["host_mt19937_plan"](C.addressof(stream._state), 8, C.c_void_p(plan.data_ptr()), words, C.byref(used), 1)
    assert status == _abi.UNSUPPORTED_CONFIG and bytes(stream._state) == before


@pytest.mark.parametrize("batched", [False, True])
@pytest.mark.parametrize("shape", [(2, 1, 96, 96, 96), (3, 2, 64, 80, 70 + 1)])
def test_draw_and_sum_in_one_kernel_equals_the_two_steps(hip, plan_where, monkeypatch, batched, shape):
    """tio_mt19937_add_noise_device == torch.randn on the host + tio_add_noise, bit for bit, and the same stream afterwards."""
    from torchio_amd import ops

    g = torch.Generator(device="cuda").manual_seed(1)
    data = torch.rand(*shape, generator=g, device="cuda") * 100 - 20
    mean = torch.tensor([0.5, -1.25, 3.0][: shape[0]], device="cuda") if batched else 0.75
    std = torch.tensor([0.1, 2.0, 0.5][: shape[0]], device="cuda") if batched else 1.5
    stream = ops.HostNormalStream(2024)
    fused = stream.add_noise(data, mean, std)
    assert fused is not None
    follow_up = stream.randn((1 << 20,), "cuda").cpu()
    generator = torch.Generator().manual_seed(2024)
    base = torch.randn(shape, generator=generator)
    expected = hip.add_noise(data, mean, std, rician=False, base1=base.cuda())
    torch.cuda.synchronize()
    assert torch.equal(expected.view(torch.int32), fused.view(torch.int32))
    assert torch.equal(torch.randn(1 << 20, generator=generator).view(torch.int32), follow_up.view(torch.int32))


def test_fused_noise_keeps_the_gradient_and_refuses_what_it_cannot_read(hip):
    """The reference is differentiable through Noise (tests/test_noise.py:75-80): the fused form passes the gradient on
    (dy/dx = 1); parameter vectors of another length / dtype / device are left to the general road (`None`)."""
    from torchio_amd import ops

    g = torch.Generator(device="cuda").manual_seed(5)
    data = torch.rand(2, 1, 96, 96, 128, generator=g, device="cuda")
    leaf = data.clone().requires_grad_(True)
    noisy = ops.HostNormalStream(77).add_noise(leaf, 0.5, 2.0)
    assert noisy is not None and noisy.requires_grad
    weight = torch.rand(data.shape, generator=g, device="cuda")
    (noisy * weight).sum().backward()
    assert torch.equal(leaf.grad, weight)
    assert torch.equal(noisy.detach(), ops.HostNormalStream(77).add_noise(data, 0.5, 2.0))
    with torch.no_grad():  # (a tensor that requires grad is just data here)
        assert not ops.HostNormalStream(77).add_noise(leaf, 0.5, 2.0).requires_grad
    stream = ops.HostNormalStream(77)
    state = bytes(stream._state)
    assert stream.add_noise(data, torch.ones(3, device="cuda"), 1.0) is None          # not one value per element
    assert stream.add_noise(data, torch.ones(2), 1.0) is None                         # on the host
    assert stream.add_noise(data, torch.ones(2, device="cuda", dtype=torch.float64), 1.0) is None
    assert bytes(stream._state) == state, "a refused call draws nothing"


def test_in_place_sum_only_where_the_tail_rule_does_not_reread(hip):
    """out == x through the C ABI: fine for whole groups of 16; with torch's tail rule (the last 16 values drawn again, from x)
    the call is refused instead of adding noise twice."""
    from torchio_amd import _abi, ops

    for count, expected in (((1 << 20) + 32, _abi.OK), ((1 << 20) + 5, _abi.UNSUPPORTED_CONFIG)):
        stream = ops.HostNormalStream(9)
        data = torch.zeros(count, device="cuda")
        with torch.cuda.device(data.device):
            plan_host, plan_dev = stream._device_plan(count, data.device)
        raw_stream = torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())
        status = stream._fn["mt19937_add_noise_device"](
            C.c_void_p(plan_host.data_ptr()), C.c_void_p(plan_dev.data_ptr()), C.c_void_p(data.data_ptr()), C.c_void_p(data.data_ptr()),
            count, 0.0, 1.0, None, None, C.c_void_p(raw_stream),
        )
        assert status == expected
        if status == _abi.OK:
            torch.cuda.synchronize()
            reference = torch.randn(count, generator=torch.Generator().manual_seed(9))
            assert torch.equal(data.cpu(), 0.0 + (0.0 + 1.0 * reference))


def test_noise_transform_in_reference_mode_uses_the_fused_kernel_and_matches_torch(hip, monkeypatch):
    """`tio.Noise` on device-resident float32 images, reference RNG mode: identical to the reference's arithmetic
    `data + (mean + std * torch.randn(shape, generator=cpu(seed)))`, image after image from one generator."""
    import torchio_amd as tio
    from torchio_amd import ops

    calls = []
    original = ops.HostNormalStream.add_noise
    monkeypatch.setattr(ops.HostNormalStream, "add_noise", lambda self, *a: calls.append(1) or original(self, *a))
    previous = tio.get_noise_rng()
    tio.set_noise_rng("reference")
    try:
        g = torch.Generator(device="cuda").manual_seed(3)
        t1 = torch.rand(2, 1, 96, 96, 128, generator=g, device="cuda")
        t2 = torch.rand(2, 1, 96, 96, 128, generator=g, device="cuda") + 5
        batch = tio.SubjectsBatch({
            "t1": tio.ImagesBatch(t1, [tio.AffineMatrix(), tio.AffineMatrix()], image_class=tio.ScalarImage),
            "t2": tio.ImagesBatch(t2, [tio.AffineMatrix(), tio.AffineMatrix()], image_class=tio.ScalarImage),
        })
        transform = tio.Noise(mean=0.25, std=(0.5, 0.5))
        torch.manual_seed(11)
        out = transform(batch)
        params = out.applied_transforms[-1].params if hasattr(out, "applied_transforms") else None
    finally:
        tio.set_noise_rng(previous)
    assert len(calls) == 2
    torch.manual_seed(11)
    torch.rand(1)  # the p-gate draw of the envelope
    seed = int(torch.randint(0, 2**31, (1,)).item())
    generator = torch.Generator().manual_seed(seed)
    for name, source in (("t1", t1), ("t2", t2)):
        base = torch.randn(source.shape, generator=generator).cuda()
        expected = source + (0.25 + 0.5 * base)
        assert torch.equal(expected, out.images[name].data), name


def test_concurrent_streams_do_not_share_staging_buffers(hip):
    """`Queue`'s worker threads each run a Noise: the pinned buffers a draw is staged in are per thread (a shared ring once
    handed two threads the same buffer — one's raw state words went up as the other's draws)."""
    import threading

    from torchio_amd import ops

    counts = [40_000, 300_000, 40_000, 2_000_000]  # host road and device road
    results: dict = {}

    def worker(seed):
        drawn = []
        for repeat in range(6):
            stream = ops.HostNormalStream(seed + repeat)
            drawn.append([stream.randn((count,), "cuda").cpu() for count in counts])
        results[seed] = drawn

    threads = [threading.Thread(target=worker, args=(seed,)) for seed in (100, 200, 300, 400)]
    for thread in threads:
        thread.start()
    for thread in threads:
        thread.join()
    for seed, drawn in results.items():
        for repeat, tensors in enumerate(drawn):
            for expected, got in zip(_torch_stream(seed + repeat, counts), tensors):
                assert torch.equal(expected.view(torch.int32), got.view(torch.int32)), (seed, repeat)


def _untemper(y: "np.ndarray") -> "np.ndarray":
    """Inverse of mt19937's tempering on uint32 arrays (a bijection: every 24-bit uniform can be asked for)."""
    import numpy as np

    y = y.astype(np.uint64)
    y ^= y >> 18
    y ^= (y << 15) & 0xEFC60000
    x = y.copy()
    for _ in range(5):  # y ^= (y << 7) & mask, undone seven bits at a time
        x = y ^ ((x << 7) & 0x9D2C5680)
    y = x & 0xFFFFFFFF
    x = y.copy()
    for _ in range(3):
        x = y ^ (x >> 11)
    return (x & 0xFFFFFFFF).astype(np.uint32)


def test_every_24_bit_uniform_through_both_transforms(hip):
    """The Box-Muller step depends on a draw only through two 24-bit uniforms: u1 (radius = sqrt(-2 log(1 - u1))) and u2
    (cos / sin of 2 pi u2).  EVERY value of each — 2^24 radii, 2^24 angles — goes through the host restatement
    (pinned against torch.randn, tests/test_host_rng.py) and through the device kernel: identical bits.  The raw words
    are placed in the rest-of-block slot of a generator state (which is emitted without a twist), 624 per call."""
    import ctypes as C

    import numpy as np

    from torchio_amd import _abi
    from torchio_amd import _lib

    _, fn = _lib.load()
    values = np.arange(1 << 24, dtype=np.uint32)
    high = (np.arange(1 << 24, dtype=np.uint64) * 2654435761 % 256).astype(np.uint32) << 24  # the 8 bits the uniform drops: anything
    raw = _untemper(values | high)
    check = raw.astype(np.uint64)
    check ^= check >> 11
    check ^= (check << 7) & 0x9D2C5680
    check ^= (check << 15) & 0xEFC60000
    check ^= check >> 18
    assert np.array_equal((check & 0xFFFFFF).astype(np.uint32), values)  # the tempered words carry the uniforms we asked for
    # groups of 16 words: lanes 0..7 carry u1 = k, lanes 8..15 u2 = a permutation of k (every value of both occurs once)
    angle = np.roll(raw, 12345)
    words = np.empty(2 << 24, dtype=np.uint32).reshape(-1, 2, 8)
    words[:, 0, :] = raw.reshape(-1, 8)
    words[:, 1, :] = angle.reshape(-1, 8)
    words = words.reshape(-1)
    words = np.concatenate([words, words[: (-words.size) % 624]]).reshape(-1, 624)  # whole state blocks (the last one padded with repeats)
    n_blocks = words.shape[0]
    state = (C.c_uint64 * (_abi.HOST_MT_STATE_BYTES // 8))()
    view = np.frombuffer(state, dtype=np.uint32)  # MtState: s[624 + 16], pos, seeded (csrc/host_rng.cpp)
    host = np.empty((n_blocks, 624), dtype=np.float32)
    device = torch.empty((n_blocks, 624), dtype=torch.float32, device="cuda")
    plan_words = int(fn["host_mt19937_plan_words"](624))
    plan_host = torch.empty((n_blocks, plan_words), dtype=torch.int32).pin_memory()
    used = C.c_int64(0)
    for block in range(n_blocks):
        for target in ("host", "plan"):
            fn["host_mt19937_seed"](C.addressof(state), 1)
            view[:624] = words[block]
            view[640] = 0  # pos: the whole block is still to be read
            if target == "host":
                assert fn["host_mt19937_randn"](C.addressof(state), C.c_void_p(host[block].ctypes.data), 624, 1) == _abi.OK
            else:
                assert fn["host_mt19937_plan"](C.addressof(state), 624, C.c_void_p(plan_host[block].data_ptr()), plan_words, C.byref(used), 1) == _abi.OK
    plan_dev = plan_host.cuda()
    stream = torch.cuda.current_stream().cuda_stream
    for block in range(n_blocks):
        status = fn["mt19937_randn_device"](C.c_void_p(plan_host[block].data_ptr()), C.c_void_p(plan_dev[block].data_ptr()),
                                            C.c_void_p(device[block].data_ptr()), C.c_void_p(stream))
        assert status == _abi.OK
    torch.cuda.synchronize()
    got = device.cpu().numpy()
    assert np.array_equal(host.view(np.uint32), got.view(np.uint32))
    assert n_blocks * 312 >= 1 << 24


def test_the_device_made_plan_is_the_hosts_plan(hip):
    """Snapshot for snapshot: `tio_mt19937_device_snapshots` against `tio_host_mt19937_plan` on ONE thread (the plain chain) —
    every word but the 31 low bits of a JUMPED snapshot's first word (bits that are not part of the generator's state: the
    twist never reads them), 12 M draws = 151 snapshots on 19 segments of 1 024 blocks... and the states the two leave behind."""
    from torchio_amd import _abi, ops

    count = 12_000_000 + 16 * 5
    host_stream, device_stream = ops.HostNormalStream(77), ops.HostNormalStream(77)
    words = int(host_stream._fn["host_mt19937_plan_words"](count))
    plan = torch.empty(words, dtype=torch.int32)
    used = C.c_int64(0)
    assert host_stream._fn["host_mt19937_plan"](C.addressof(host_stream._state), count, C.c_void_p(plan.data_ptr()), words, C.byref(used), 1) == _abi.OK
    previous = ops.get_noise_plan()
    ops.set_noise_plan("device")
    try:
        made = device_stream._device_made_plan(count, torch.device("cuda"))
    finally:
        ops.set_noise_plan(previous)
    assert made is not None
    torch.cuda.synchronize()
    theirs, ours = plan[: used.value], made[1].cpu()
    assert ours.numel() == used.value and torch.equal(ours[:8], theirs[:8])
    snapshots_at = 16 + 624 + 16
    a = theirs[snapshots_at:].view(-1, 624).clone()
    b = ours[snapshots_at:].view(-1, 624).clone()
    a[:, 0] &= -(2**31)
    b[:, 0] &= -(2**31)
    assert torch.equal(a, b)
    assert torch.equal(device_stream.randn((1 << 16,), "cpu").view(torch.int32), host_stream.randn((1 << 16,), "cpu").view(torch.int32))
