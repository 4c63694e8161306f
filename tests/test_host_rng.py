"""``tio_host_mt19937_*`` (csrc/host_rng.cpp): torch's CPU ``randn`` stream on all host cores, bit for bit.

The reference's Noise draws ``torch.randn(data.shape, generator=Generator().manual_seed(seed))``
(transforms/intensity/noise.py:108-116, 166-178); this is the pin of the restatement against torch itself — sizes around the
16-value groups and the 624-word state blocks, continuation across calls (one generator serves every image of a batch),
thread counts, and a 32 M-draw run.  No GPU involved: the helper is host code of ``libtio_hip.so``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import pytest
import torch

from torchio_amd import _abi
from torchio_amd import _lib


@pytest.fixture(scope="module")
def functions():
    return _lib.load()[1]


def _draw(functions, state, n, threads):
    out = torch.empty(n, dtype=torch.float32)
    status = functions["host_mt19937_randn"](C.addressof(state), C.c_void_p(out.data_ptr()), n, threads)
    return status, out


@pytest.mark.parametrize("threads", [1, 2, 5])
@pytest.mark.parametrize("n", [16, 32, 608, 624, 640, 1248, 16 * 1000, 16 * 1000 + 7, 624 * 16 * 3 + 5, 100003, 624 * 300, 624 * 300 + 5, 624 * 517 + 16 * 11])
def test_randn_equals_torch_bit_for_bit(functions, n, threads):
    seed = 12345 + n
    state = (C.c_uint64 * (_abi.HOST_MT_STATE_BYTES // 8))()
    assert functions["host_mt19937_seed"](C.addressof(state), seed) == _abi.OK
    status, ours = _draw(functions, state, n, threads)
    assert status == _abi.OK
    expected = torch.randn(n, generator=torch.Generator().manual_seed(seed))
    assert torch.equal(ours.view(torch.int32), expected.view(torch.int32))


def test_stream_continues_across_calls_like_one_generator(functions):
    """Several images / a Rician second draw share one generator: call after call, including sizes that leave the state
    in the middle of a block and sizes that are not multiples of 16 (torch spends 16 extra draws there)."""
    seed = 2**31 - 7
    state = (C.c_uint64 * (_abi.HOST_MT_STATE_BYTES // 8))()
    functions["host_mt19937_seed"](C.addressof(state), seed)
    generator = torch.Generator().manual_seed(seed)
    for n, threads in [(1000 * 16, 3), (624 * 400 + 32, 6), (16 * 77 + 3, 1), (624 * 5, 4), (624 * 333, 5), (48, 2), (20 * 20 * 20, 8), (16, 1)]:
        status, ours = _draw(functions, state, n, threads)
        assert status == _abi.OK
        expected = torch.randn(n, generator=generator)
        assert torch.equal(ours.view(torch.int32), expected.view(torch.int32)), n


def test_small_draws_are_left_to_torch(functions):
    state = (C.c_uint64 * (_abi.HOST_MT_STATE_BYTES // 8))()
    functions["host_mt19937_seed"](C.addressof(state), 1)
    status, _ = _draw(functions, state, 15, 1)
    assert status == _abi.UNSUPPORTED_CONFIG


def test_a_volume_sized_draw(functions):
    n = 2 * 256 * 256 * 256  # 32 M values: two bench volumes
    seed = 424242
    state = (C.c_uint64 * (_abi.HOST_MT_STATE_BYTES // 8))()
    functions["host_mt19937_seed"](C.addressof(state), seed)
    status, ours = _draw(functions, state, n, 8)
    assert status == _abi.OK
    expected = torch.randn(n, generator=torch.Generator().manual_seed(seed))
    assert torch.equal(ours.view(torch.int32), expected.view(torch.int32))


def _plan(functions, state, n, threads=1):
    words = int(functions["host_mt19937_plan_words"](n))
    plan = torch.empty(max(words, 1), dtype=torch.int32)
    used = C.c_int64(0)
    status = functions["host_mt19937_plan"](C.addressof(state), n, C.c_void_p(plan.data_ptr()), words, C.byref(used), threads)
    return status, plan, used.value, words


@pytest.mark.parametrize("n", [16, 624, 640, 16 * 39 * 128, 100_000, 1_000_003, 5_000_000])
def test_plan_of_a_device_draw_advances_the_state_like_the_host_draw(functions, n):
    """tio_host_mt19937_plan (the host half of the device-side stream, csrc/mt19937.hip): same state afterwards as
    tio_host_mt19937_randn, a snapshot per 128 blocks, the tail draws final in the plan."""
    drawn = (C.c_uint64 * (_abi.HOST_MT_STATE_BYTES // 8))()
    planned = (C.c_uint64 * (_abi.HOST_MT_STATE_BYTES // 8))()
    for state in (drawn, planned):
        functions["host_mt19937_seed"](C.addressof(state), 77)
    status, values = _draw(functions, drawn, n, 4)
    assert status == _abi.OK
    status, plan, used, words = _plan(functions, planned, n)
    assert status == _abi.OK and used <= words
    assert bytes(drawn) == bytes(planned)
    blocks = -(-n // 624)  # (a fresh generator: no words left in the current block)
    assert plan[0].item() == 0x4D54504C and plan[1].item() == 0 and plan[4].item() == -(-blocks // 128)
    assert used == 656 + plan[4].item() * 624
    if n % 16:
        assert torch.equal(plan[640:656], values[-16:].view(torch.int32))
    # the stream continues identically after either call
    _, after_draw = _draw(functions, drawn, 4096, 1)
    _, after_plan = _draw(functions, planned, 4096, 1)
    assert torch.equal(after_draw.view(torch.int32), after_plan.view(torch.int32))


def test_plan_refuses_a_stream_inside_a_group(functions):
    state = (C.c_uint64 * (_abi.HOST_MT_STATE_BYTES // 8))()
    functions["host_mt19937_seed"](C.addressof(state), 5)
    _draw(functions, state, 1000 * 16 + 5, 1)  # a tail: 16 extra draws, the stream now stands inside a group of 16
    before = bytes(state)
    status, _, _, _ = _plan(functions, state, 1 << 20)
    assert status == _abi.UNSUPPORTED_CONFIG and bytes(state) == before
    status, _, _, _ = _plan(functions, state, 8)
    assert status == _abi.UNSUPPORTED_CONFIG and bytes(state) == before


@pytest.mark.parametrize("n,threads", [(2 * 4096 * 624, 2), (6_000_000 + 7, 3), (20_000_000, 8), (20_000_000 + 624 * 3 + 48, 16)])
def test_plan_with_jump_ahead_equals_the_serial_plan(functions, n, threads):
    """Long chains are cut into segments whose starts are reached by jumping (csrc/host_rng_jump.cpp: g(f) applied to the
    state, g = x^J mod the computed characteristic polynomial).  Same generator state afterwards, same plan — except the
    31 low bits of the first word of a jumped segment's first snapshot, which are not part of mt19937's state — and the
    stream continues identically."""
    serial = (C.c_uint64 * (_abi.HOST_MT_STATE_BYTES // 8))()
    jumped = (C.c_uint64 * (_abi.HOST_MT_STATE_BYTES // 8))()
    for state in (serial, jumped):
        functions["host_mt19937_seed"](C.addressof(state), 31337)
        _draw(functions, state, 624 * 7 + 160, 1)  # (start inside a block, on a group boundary)
    status, plan_serial, used_serial, _ = _plan(functions, serial, n, 1)
    assert status == _abi.OK
    status, plan_jumped, used_jumped, _ = _plan(functions, jumped, n, threads)
    assert status == _abi.OK and used_serial == used_jumped
    assert bytes(serial) == bytes(jumped)
    different = (plan_serial[:used_serial] != plan_jumped[:used_jumped]).nonzero().flatten()
    assert different.numel() < threads  # at most one word per jumped segment ...
    assert all((int(i) - 656) % 624 == 0 and (int(i) - 656) // 624 % 32 == 0 for i in different)  # ... the first of a snapshot at a segment start
    xor = plan_serial[different] ^ plan_jumped[different]
    assert bool((xor >= 0).all())  # ... and never its top bit (int32: a set top bit would be negative)
    _, after_serial = _draw(functions, serial, 10_000, 1)
    _, after_jumped = _draw(functions, jumped, 10_000, 1)
    assert torch.equal(after_serial.view(torch.int32), after_jumped.view(torch.int32))


def test_concurrent_plans_share_the_polynomial_cache_safely(functions):
    """Several threads planning at once (Queue workers): same segment length, different segment counts — the cache of
    jump polynomials grows behind the plans that are using it."""
    import threading

    counts = [2 * 4096 * 624 + 16 * k for k in (0, 5)] + [5 * 4096 * 624, 9 * 4096 * 624 + 160]
    expected = {}
    for n in counts:
        state = (C.c_uint64 * (_abi.HOST_MT_STATE_BYTES // 8))()
        functions["host_mt19937_seed"](C.addressof(state), n & 0xFFFF)
        assert _plan(functions, state, n, 1)[0] == _abi.OK
        expected[n] = bytes(state)
    failures = []

    def worker(n):
        for _ in range(3):
            state = (C.c_uint64 * (_abi.HOST_MT_STATE_BYTES // 8))()
            functions["host_mt19937_seed"](C.addressof(state), n & 0xFFFF)
            status = _plan(functions, state, n, 2 + (n // (4096 * 624)))[0]
            if status != _abi.OK or bytes(state) != expected[n]:
                failures.append(n)

    threads = [threading.Thread(target=worker, args=(n,)) for n in counts]
    for thread in threads:
        thread.start()
    for thread in threads:
        thread.join()
    assert not failures


def test_worker_pool_survives_a_fork():
    """DataLoader workers fork: the child must not inherit the parent's worker threads' synchronisation objects
    (csrc/host_rng.cpp: one pool per process, re-created behind pthread_atfork).  Run in a subprocess: the child of a
    fork is not a place for pytest, and torch itself must not be touched in it."""
    import subprocess
    import sys
    import textwrap

    script = textwrap.dedent(
        """
        import ctypes as C, os, sys, torch
        sys.path.insert(0, %r)
        from torchio_amd import _abi, _lib
        _, fn = _lib.load()
        def state(seed):
            st = (C.c_uint64 * (_abi.HOST_MT_STATE_BYTES // 8))(); fn["host_mt19937_seed"](C.addressof(st), seed); return st
        n = 12_000_000
        words = fn["host_mt19937_plan_words"](n); plan = torch.zeros(words, dtype=torch.int32); used = C.c_int64()
        def run(st, threads): return fn["host_mt19937_plan"](C.addressof(st), n, C.c_void_p(plan.data_ptr()), words, C.byref(used), threads)
        serial = state(1); assert run(serial, 1) == 0
        pooled = state(1); assert run(pooled, 6) == 0 and bytes(pooled) == bytes(serial)   # the pool exists in the parent
        pid = os.fork()
        if pid == 0:
            child = state(1)
            os._exit(0 if (run(child, 6) == 0 and bytes(child) == bytes(serial)) else 3)
        _, status = os.waitpid(pid, 0)
        again = state(1)
        sys.exit(0 if (os.WEXITSTATUS(status) == 0 and run(again, 6) == 0 and bytes(again) == bytes(serial)) else 4)
        """
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    result = subprocess.run([sys.executable, "-c", script], timeout=120, capture_output=True, text=True)
    assert result.returncode == 0, result.stderr[-2000:]


def _untemper(y):
    """Inverse of mt19937's tempering on uint32 arrays (a bijection: every 24-bit uniform can be asked for)."""
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


def test_every_24_bit_uniform_against_torch_itself(functions):
    """The Box-Muller step sees a draw only through two 24-bit uniforms.  EVERY value of each (2^24 radii through
    log256_ps and the square root, 2^24 angles through sincos256_ps) goes through torch's own kernel — a CPU generator
    whose state is set to crafted words, read without a twist — and through the restatement: identical bits.  With
    tests/test_gpu_device_rng.py::test_every_24_bit_uniform_through_both_transforms (host == device on the same words)
    the device stream is pinned to torch for every input the arithmetic can see, not for a sample of them."""
    values = np.arange(1 << 24, dtype=np.uint32)
    high = (np.arange(1 << 24, dtype=np.uint64) * 2654435761 % 256).astype(np.uint32) << 24  # the 8 bits the uniform drops
    raw = _untemper(values | high)
    words = np.empty(2 << 24, dtype=np.uint32).reshape(-1, 2, 8)
    words[:, 0, :] = raw.reshape(-1, 8)                    # lanes 0 .. 7 of a group of 16: u1
    words[:, 1, :] = np.roll(raw, 12345).reshape(-1, 8)    # lanes 8 .. 15: u2 (a permutation: every value once)
    words = words.reshape(-1)
    words = np.concatenate([words, words[: (-words.size) % 608]]).reshape(-1, 608)  # 38 groups of 16 per generator state
    generator = torch.Generator()
    template = generator.manual_seed(1).get_state().clone()
    raw_state = template.numpy().view(np.uint8)
    # THGeneratorState: uint64 seed; int left; int seeded; uint64 next; uint64 state[624]; ... (left = 624 — the largest
    # valid value — and next = 0: the next 623 draws read state[0 .. 623) as it stands; 608 of them are used)
    raw_state[8:12] = np.frombuffer(np.int32(624).tobytes(), dtype=np.uint8)
    raw_state[16:24] = 0
    slots = raw_state[24 : 24 + 624 * 8].view(np.uint64)
    state = (C.c_uint64 * (_abi.HOST_MT_STATE_BYTES // 8))()
    view = np.frombuffer(state, dtype=np.uint32)  # MtState: s[624 + 16], pos, seeded (csrc/host_rng.cpp)
    ours = np.empty(608, dtype=np.float32)
    step = 1 if os.environ.get("TIO_TEST_EXHAUSTIVE", "1") != "0" else 97
    for block in range(0, words.shape[0], step):
        slots[:608] = words[block]
        generator.set_state(template)
        theirs = torch.randn(608, generator=generator).numpy()
        functions["host_mt19937_seed"](C.addressof(state), 1)
        view[:608] = words[block]
        view[640] = 0
        assert functions["host_mt19937_randn"](C.addressof(state), C.c_void_p(ours.ctypes.data), 608, 1) == _abi.OK
        assert np.array_equal(theirs.view(np.uint32), ours.view(np.uint32)), block


def test_the_stream_checks_itself_against_this_torch_build(monkeypatch):
    """the restatement is tied to one torch build's arithmetic.  The first use compares 4 096 + 48 draws with
    `torch.randn`; a stream that is not verified is not used (`takes` -> False: the callers keep `torch.randn`)."""
    from torchio_amd import ops

    monkeypatch.setattr(ops.HostNormalStream, "_self_check", None)
    assert ops.HostNormalStream.verified() is True          # this image: torch 2.10, AVX2
    assert ops.HostNormalStream.takes((4, 4)) is True
    monkeypatch.setattr(ops.HostNormalStream, "_self_check", False)
    assert ops.HostNormalStream.takes((1 << 20,)) is False  # switched off for the process: nobody draws from it
