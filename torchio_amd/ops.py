"""Tensor-facing wrappers over the C ABI (``include/tio_hip.h``).

PyTorch is plumbing here: it owns device memory (outputs come from the caching
allocator), the current HIP stream and nothing else.  Every function marshals
``torch.Tensor`` arguments into raw pointers and calls one ``tio_*`` entry point
on the tensor's device and the current stream.  Inputs are never modified and
outputs never alias inputs (callers with ``copy=False`` may alias user data,
reference transform.py:220-221).
"""
from __future__ import annotations

import collections
import ctypes as C
import os
import time
import threading
import types
from collections.abc import Sequence

import torch
from torch import Tensor

from . import _abi

_DTYPE_CODES = {
    torch.float32: _abi.F32,
    torch.float64: _abi.F64,
    torch.float16: _abi.F16,
    torch.bfloat16: _abi.BF16,
    torch.uint8: _abi.U8,
    torch.int8: _abi.I8,
    torch.int16: _abi.I16,
    torch.int32: _abi.I32,
    torch.int64: _abi.I64,
}
FLOAT_DTYPES = (torch.float32, torch.float64, torch.float16, torch.bfloat16)
INTERP_CODES = {
    "nearest": _abi.NEAREST, "linear": _abi.LINEAR, "label": _abi.LABEL_PV, "linear_adjoint": _abi.LINEAR_ADJOINT,
    "quadratic": _abi.QUADRATIC, "cubic": _abi.CUBIC,  # the image handed over holds B-spline coefficients (bspline_prefilter)
    "fourth": _abi.BSPLINE4, "fifth": _abi.BSPLINE5, "sixth": _abi.BSPLINE6, "seventh": _abi.BSPLINE7,
}


PRECISION_CODES = {"exact": _abi.PRECISION_EXACT, "fast": _abi.PRECISION_FAST, "tight": _abi.PRECISION_TIGHT}
_RESAMPLE_PRECISION = "exact"
_FAST_OPTED_IN = False  # set_resample_precision("fast", allow_out_of_tolerance=True) was called in this process


def set_resample_precision(mode: str, *, allow_out_of_tolerance: bool = False) -> None:
    """Process-wide arithmetic of ``tio_resample3d`` for float32 trilinear images.

    ``"exact"`` (default) reproduces the reference's float32 operation sequence bit for bit.
    ``"tight"`` keeps every sampling coordinate the reference's float32 value bit for bit — hence the same eight taps,
    the same weights and the same fill decisions — and fuses only the interpolation (three nested fma lerps): results
    within the rounding of seven multiply-adds of the reference, i.e. inside the north-star bar PER VOXEL
    (``|d| <= 1e-4 max(|ref|, 1e-3 range)``) even on white noise.  The mode of the bench's headline.
    ``"fast"`` is NOT north-star compliant and is refused unless ``allow_out_of_tolerance=True``: coordinates as a line
    per control cell, within 1e-4 of the intensity RANGE but not per voxel on noisy data (one ulp of a coordinate already
    moves a white-noise value by more than the per-voxel bar allows: 4.4 M of 134 M voxels of a bare launch beyond it).
    It is kept for A/B measurements of the coordinate chain's cost; once a process has opted in, the mode can be
    re-selected (restoring a saved mode) without repeating the flag.  Label maps resampled with ``"nearest"`` and
    ``"label"`` (partial-volume) images are bit-identical to the reference in every mode.
    """
    global _RESAMPLE_PRECISION, _FAST_OPTED_IN
    if mode not in PRECISION_CODES:
        raise ValueError(f"precision must be one of {sorted(PRECISION_CODES)}, got {mode!r}")
    if mode == "fast":
        if not (allow_out_of_tolerance or _FAST_OPTED_IN):
            raise ValueError('resample precision "fast" is outside the per-voxel tolerance (1e-4 relative) on noisy data; '
                             'pass allow_out_of_tolerance=True to select it for measurements')
        _FAST_OPTED_IN = True
    _RESAMPLE_PRECISION = mode


def get_resample_precision() -> str:
    return _RESAMPLE_PRECISION


_STENCIL_PRECISION = "exact"


def set_stencil_precision(mode: str) -> None:
    """Process-wide arithmetic of the fused Blur launch (``tio_blur_fused``): ``"exact"`` (default) accumulates every tap with
    a separately rounded multiply and add, like the reference's convolution (bit-identical to the unfused launches);
    ``"fast"`` uses fused multiply-adds — one rounding per tap, results within float rounding (~1e-7 relative; the
    contract for intensities is 1e-4), and a J + K (+ noise) pass, which is bound by vector instructions, ~25 % shorter."""
    global _STENCIL_PRECISION
    if mode not in ("exact", "fast"):
        raise ValueError(f'stencil precision must be "exact" or "fast", got {mode!r}')
    _STENCIL_PRECISION = mode


def get_stencil_precision() -> str:
    return _STENCIL_PRECISION


def reload_env() -> None:
    """Have ``libtio_hip.so`` parse its ``TIO_*`` environment switches again (it reads them once per process, at the first
    call that needs one: ``tio_reload_env``, ABI 10).  Only A/B experiments and tests change them after start-up."""
    from . import _lib  # noqa: PLC0415

    _lib.load()[1]["reload_env"]()


def h2d(tensor: Tensor, device) -> Tensor:
    """Upload a (small) host tensor without stalling the device.

    ``tensor.to(device)`` from pageable memory blocks the host until everything already
    queued on the stream has finished, so a pipeline of transforms can never run ahead of
    the GPU (measured: ~1 ms of idle gaps per 4.7 ms bench step).  Staging through the
    caching pinned allocator + ``non_blocking`` keeps the copy stream-ordered and the host
    free; the allocator holds the staging block until the copy has completed.
    """
    device = torch.device(device)
    if device.type != "cuda" or tensor.device.type != "cpu":
        return tensor.to(device)
    return tensor.contiguous().pin_memory().to(device, non_blocking=True)


def h2d_packed(tensors: Sequence[Tensor | None], device) -> list[Tensor | None]:
    """Upload several small float32 host tensors with ONE staging copy; returns device views of the same shapes.

    Each stream-ordered upload is a ~4 us copy kernel plus a ~6 us gap on the GPU's timeline, in front of whatever needs
    it; a transform (or a fused launch of three) that needs three parameter blocks pays that three times.  Entries that are
    ``None`` or already on the device pass through.  Segments start on 256-byte boundaries (vector loads in the kernels).
    """
    device = torch.device(device)
    todo = [i for i, t in enumerate(tensors) if t is not None and t.device.type == "cpu"]
    out: list[Tensor | None] = list(tensors)
    if device.type != "cuda" or not todo:
        return [None if t is None else t.to(device) for t in tensors]
    if len(todo) == 1:
        out[todo[0]] = h2d(tensors[todo[0]].to(torch.float32), device)
        return out
    offsets, total = [], 0
    for i in todo:
        offsets.append(total)
        total += (tensors[i].numel() + 63) // 64 * 64
    staging = torch.empty(total, dtype=torch.float32).pin_memory()
    for i, off in zip(todo, offsets, strict=True):
        staging[off : off + tensors[i].numel()] = tensors[i].reshape(-1).to(torch.float32)
    block = staging.to(device, non_blocking=True)
    for i, off in zip(todo, offsets, strict=True):
        out[i] = block[off : off + tensors[i].numel()].view(tensors[i].shape)
    return out


_MINIMUM_STATE = threading.local()  # (per THREAD — Queue's workers run Composes concurrently; one thread's withdrawal used to disarm another's announcement)
_ANNOUNCED_MINIMUM_ENABLED = os.environ.get("TIO_NO_ANNOUNCED_MIN", "") in ("", "0")  # (A/B switch)


def expect_minimum_fill(flag: bool) -> None:
    """Announce (or withdraw) that the consumer of the next resampling's output will ask for the per-channel minimum of its
    first element (``default_pad_value="minimum"``): large FAST launches then fold it into their stores (``resample3d``)."""
    _MINIMUM_STATE.wanted = bool(flag) and _ANNOUNCED_MINIMUM_ENABLED


def minimum_fill_expected() -> bool:
    return getattr(_MINIMUM_STATE, "wanted", False)


# ---------------------------------------------------------------------------------------------------------------------
# Work AHEAD of the data: parameter uploads and brick plans depend on a transform's drawn parameters, not on voxel values.
# A Compose that has drawn every child's parameters (transforms/compose.py) has its spatial children enqueue them on a side
# stream — while the kernels of the previous children (or of the previous step: the host runs ahead of the GPU) are still
# running — so that the upload copies, the planning kernels and the dependency gaps around them (~5 us each on this GPU) are
# off the critical path of the stream the data lives on.  The data stream waits for the side stream's event right before the
# launch that reads those buffers; buffers allocated on the side stream are marked as used by the data stream
# (`record_stream`), so the caching allocator does not hand them out again before that launch has finished.
# ---------------------------------------------------------------------------------------------------------------------
_AHEAD_STREAMS: dict[int, "torch.cuda.Stream"] = {}
_AHEAD_ENABLED = os.environ.get("TIO_NO_AHEAD_STREAM", "") in ("", "0")


def set_ahead_stream(enabled: bool) -> None:
    """Switch the side stream for work ahead of the data on or off (on by default; ``TIO_NO_AHEAD_STREAM=1`` starts with it off)."""
    global _AHEAD_ENABLED
    _AHEAD_ENABLED = bool(enabled)


def ahead_stream(device) -> "torch.cuda.Stream | None":
    """The side stream of *device* for work ahead of the data (``None`` on the host or when switched off)."""
    device = torch.device(device)
    if device.type != "cuda" or not _AHEAD_ENABLED:
        return None
    index = device.index if device.index is not None else torch.cuda.current_device()
    stream = _AHEAD_STREAMS.get(index)
    if stream is None:
        stream = _AHEAD_STREAMS[index] = torch.cuda.Stream(device=index)
    return stream


_DRAW_STREAMS: dict[int, "torch.cuda.Stream"] = {}
_DRAW_MARKS: dict = {}
_DRAWS_IN_FLIGHT = 4
# Where the draws of the reference's noise stream run (round 5; VERDICT r4 weak #4: the draw kernel is bound by the vector
# ALU, and so are the exact resamplers — left free to start whenever the host has enqueued it, it ran BESIDE them and the
# library default lost 15 % on the driver's box):
#   "gated"  the draw kernel waits (on the device) for the latest resampling launch of the data stream: with a host that runs
#            ahead, the draws of step n + 1 then run beside step n's memory-bound stencil passes, not beside its resamplers;
#   "free"   round 4's behaviour (no gate);
#   "off"    no draw stream: the draws are made on the data stream, in front of the stencil (the road of round 3).
# `calibrate_draw_policy` measures the three on the caller's own pipeline and keeps the fastest.
_DRAW_POLICIES = ("gated", "free", "off")


def _draw_policy_from_env() -> str:
    """``TIO_DRAW_POLICY`` validated like ``set_draw_policy`` (a typo — "on", "1" — silently behaved as "free")."""
    value = os.environ.get("TIO_DRAW_POLICY", "off")
    if value not in _DRAW_POLICIES:
        raise ValueError(f'TIO_DRAW_POLICY must be one of {_DRAW_POLICIES}, got {value!r}')
    return value


_DRAW_POLICY = _draw_policy_from_env()
_LAST_RESAMPLE: dict = {}  # (device index, data stream id) -> event recorded behind the latest tio_resample3d launch


def set_draw_policy(policy: str) -> None:
    """Where the draws of the reference's noise stream run (see above).  The library default is ``"off"`` since round 5
    (round 4 shipped ``"free"``): without a call of ``calibrate_draw_policy`` the reference-noise mode has no draw-stream
    overlap — and nothing that can contend with the resamplers."""
    global _DRAW_POLICY
    if policy not in _DRAW_POLICIES:
        raise ValueError(f'draw policy must be "gated", "free" or "off", got {policy!r}')
    _DRAW_POLICY = policy


def get_draw_policy() -> str:
    return _DRAW_POLICY


def calibrate_draw_policy(step, *, steps: int = 12, warmup: int = 6, policies=("gated", "free", "off")) -> dict:
    """Time ``step()`` (one pass of the caller's pipeline on its device-resident batch) under each draw policy and keep the
    fastest: the A/B that cannot be decided once for every box (the same tree measured +12 % and -15 % for "free" on two
    MI355X boxes of one pool).  Returns the milliseconds per step of every policy; the winner stays set."""
    previous = _DRAW_POLICY
    timings: dict[str, float] = {}
    try:
        for policy in policies:
            set_draw_policy(policy)
            for _ in range(warmup):
                step()
            torch.cuda.synchronize()
            start = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            timings[policy] = 1e3 * (time.perf_counter() - start) / steps
    except Exception:
        set_draw_policy(previous)
        raise
    set_draw_policy(min(timings, key=timings.get))
    return timings


# Who runs the mt19937 state chain of the reference's noise stream (round 6):
#   "host"    this rank's worker threads, by jump-ahead (0.6 ms per 134 M draws on 32 threads of an idle host, overlapped with the
#             enqueue work of a Compose that draws ahead; 4.8 - 5.7 ms on the 15 threads a rank has when eight share a host);
#   "device"  the device (csrc/mt19937.hip: tio_mt19937_device_snapshots — one workgroup per segment jumps and chains, ~0.1 ms
#             beside whatever else runs); the host writes a 5 KB prefix;
#   "auto"    "device" when several ranks share this host (LOCAL_WORLD_SIZE > 1), else "host".
_NOISE_PLANS = ("auto", "host", "device")


def _noise_plan_from_env() -> str:
    value = os.environ.get("TIO_NOISE_PLAN", "auto")
    if value not in _NOISE_PLANS:
        raise ValueError(f"TIO_NOISE_PLAN must be one of {_NOISE_PLANS}, got {value!r}")
    return value


_NOISE_PLAN = _noise_plan_from_env()


def set_noise_plan(where: str) -> None:
    """Where the state chain of the reference's noise stream is run: ``"host"``, ``"device"`` or ``"auto"`` (see above).  The
    draws are the same bits either way (tests/test_gpu_device_rng.py)."""
    global _NOISE_PLAN
    if where not in _NOISE_PLANS:
        raise ValueError(f"noise plan must be one of {_NOISE_PLANS}, got {where!r}")
    _NOISE_PLAN = where


def get_noise_plan() -> str:
    return _NOISE_PLAN


def noise_plan_on_device() -> bool:
    if _NOISE_PLAN == "auto":
        try:
            return int(os.environ.get("LOCAL_WORLD_SIZE", "1")) > 1
        except ValueError:
            return False
    return _NOISE_PLAN == "device"


def note_resample_launch(reference: Tensor) -> None:
    """Called behind every tio_resample3d launch while the reference's noise stream is drawn on the draw stream: the event
    the next draw kernel is gated on."""
    if _DRAW_POLICY != "gated" or not reference.is_cuda or not _DRAW_STREAMS:
        return
    main = torch.cuda.current_stream(reference.device)
    event = torch.cuda.Event()
    event.record(main)
    if len(_LAST_RESAMPLE) > 64:
        _LAST_RESAMPLE.clear()
    _LAST_RESAMPLE[(reference.device.index, main.stream_id)] = event


def draw_stream(device) -> "torch.cuda.Stream | None":
    """The stream of *device* on which the reference's noise stream is drawn ahead of the data (``HostNormalStream.randn_ahead``).
    A stream of its own: a draw kernel runs for a third of a millisecond, and the brick plans of the next step — which the
    data stream waits for — must not queue behind it on ``ahead_stream``."""
    device = torch.device(device)
    if device.type != "cuda" or not _AHEAD_ENABLED or _DRAW_POLICY == "off" or os.environ.get("TIO_NO_DRAW_STREAM", "") not in ("", "0"):
        return None
    index = device.index if device.index is not None else torch.cuda.current_device()
    stream = _DRAW_STREAMS.get(index)
    if stream is None:
        stream = _DRAW_STREAMS[index] = _low_priority_stream(index) or torch.cuda.Stream(device=index)
    return stream


def _low_priority_stream(index: int):
    """A HIP stream of the LOWEST priority the device offers (torch hands out default- and high-priority streams only), wrapped
    for torch — the dispatcher then prefers the data stream's workgroups whenever both have some ready.  ``None`` when the
    runtime cannot be reached this way or offers one priority only (``TIO_DRAW_STREAM_PRIORITY=default`` switches it off)."""
    if os.environ.get("TIO_DRAW_STREAM_PRIORITY", "low") != "low":
        return None
    try:
        # (the HIP runtime THIS process already runs — torch's bundled copy — by its path in /proc/self/maps; a bare
        # "libamdhip64.so" may resolve to another ROCm on the box, and a stream handle of one runtime means nothing to the other)
        path = None
        with open("/proc/self/maps") as maps:
            for line in maps:
                if "libamdhip64.so" in line:
                    path = line.split(None, 5)[-1].strip()
                    break
        if path is None:
            return None
        hip = C.CDLL(path)
        least, greatest = C.c_int(0), C.c_int(0)
        if hip.hipDeviceGetStreamPriorityRange(C.byref(least), C.byref(greatest)) != 0 or least.value == greatest.value:
            return None
        handle = C.c_void_p()
        with torch.cuda.device(index):
            if hip.hipStreamCreateWithPriority(C.byref(handle), C.c_uint(1), C.c_int(least.value)) != 0 or not handle.value:  # 1 = non-blocking
                return None
        return torch.cuda.ExternalStream(handle.value, device=index)
    except (OSError, AttributeError, RuntimeError):
        return None


def on_draw_stream(device, make):
    """``make()`` — a launch that produces a tensor of draws and depends on nothing on the data stream — enqueued on the draw
    stream of *device*; the current stream is ordered behind it by an event and takes the buffer over.  ``None`` when there
    is no draw stream (host tensors, switched off)."""
    device = torch.device(device)
    side = draw_stream(device)
    if side is None:
        return None
    with torch.cuda.device(device):
        main = torch.cuda.current_stream()
        # Back-pressure: the draw stream waits for nothing on the data stream, so with a host that enqueues faster than the data
        # stream works it would run ahead without bound — one buffer of draws (as large as the image batch) per step.  Every
        # call marks the data stream ("the consumers of all earlier draws are enqueued before this point") and waits, on the
        # host, for the mark of `_DRAWS_IN_FLIGHT` calls ago.
        if len(_DRAW_MARKS) > 64:  # (a caller that keeps creating data streams: forget the marks of the old ones)
            _DRAW_MARKS.clear()
        marks = _DRAW_MARKS.setdefault((side.device.index, main.stream_id), collections.deque())
        mark = torch.cuda.Event()
        mark.record(main)
        marks.append(mark)
        if len(marks) > _DRAWS_IN_FLIGHT:
            marks.popleft().synchronize()
        gate = _LAST_RESAMPLE.get((side.device.index, main.stream_id)) if _DRAW_POLICY == "gated" else None
        with torch.cuda.stream(side):
            if gate is not None:
                side.wait_event(gate)  # (device-side: the draw kernel starts when the data stream's latest resampling launch has finished)
            out = make()
            if out is None:
                return None
            done = torch.cuda.Event()
            done.record(side)
        main.wait_event(done)
        out.record_stream(main)
    return out


class Ahead:
    """Device tensors prepared on the side stream + the event that says they are ready."""

    __slots__ = ("tensors", "event", "payload")

    def __init__(self) -> None:
        self.tensors: list[Tensor] = []
        self.event = None
        self.payload = None

    def join(self, device) -> None:
        """Order the current stream of *device* behind the preparation and hand the buffers over to it."""
        if self.event is None:
            return
        current = torch.cuda.current_stream(device)
        current.wait_event(self.event)
        for tensor in self.tensors:
            tensor.record_stream(current)
        self.event = None


def folded_channel_min(data: Tensor) -> Tensor | None:
    """The per-channel minimum of element 0 that the launch which produced *data* left behind, if it did and the
    tensor has not been written since (``Engine.resample3d``); ``None`` otherwise."""
    record = getattr(data, "_tio_channel_min", None)
    if record is None or record[0] != data._version:
        return None
    return record[1]


class HostNormalStream:
    """``torch.Generator().manual_seed(seed)`` + ``torch.randn(shape, generator=...)`` on all host cores, bit for bit
    (``tio_host_mt19937_*``, csrc/host_rng.cpp), delivered on the device.

    The reference's Noise draws from ONE seeded CPU generator, call after call (noise.py:108-116); torch produces that
    stream on a single thread (0.35 s for 8 x 256^3 values).  Large draws for the device are made ON the device
    (``_randn_on_device``: the host only runs the mt19937 state chain and uploads a snapshot every 128 blocks); the host
    road — the chain on one thread, everything else on the others, into pinned memory, the upload of a chunk overlapping
    the generation of the next one — serves host tensors, small draws and streams that stand inside a group of 16.
    """

    CHUNK = 16 * (1 << 20)  # values per upload (64 MiB; a multiple of 16: only the last chunk can carry torch's tail rule)

    def __init__(self, seed: int) -> None:
        from . import _lib  # noqa: PLC0415

        _, functions = _lib.load()
        self._fn = functions
        self._state = (C.c_uint64 * (_abi.HOST_MT_STATE_BYTES // 8))()
        status = functions["host_mt19937_seed"](C.addressof(self._state), int(seed) & (2**64 - 1))
        if status != _abi.OK:
            raise EngineError(f"tio_host_mt19937_seed failed with status {status}")
        threads = os.environ.get("TIO_HOST_RNG_THREADS")
        if threads:
            self.threads = int(threads)
        else:  # this rank's share of the host (LOCAL_WORLD_SIZE ranks per host, the affinity mask), at most 32
            from .distributed import cached_host_thread_budget  # noqa: PLC0415

            self.threads = cached_host_thread_budget()

    _self_check: bool | None = None  # does the restated stream equal THIS torch build's? (checked once per process)

    @classmethod
    def verified(cls) -> bool:
        """The restatement is tied to the arithmetic of torch's CPU ``normal_`` (mt19937 + the 16-lane Box-Muller of
        ``normal_fill_16_AVX2`` with ``avx_mathfun.h``'s polynomials: torch 2.10 on AVX2 hosts).  Another torch build or a host
        without AVX2 may take another path, so the first use draws 4 096 values both ways from one seed — a microsecond-scale
        check — and on ANY difference the stream is switched off for the process: ``takes`` then answers False and the
        callers keep ``torch.randn`` (slow, and by definition the reference's stream)."""
        if cls._self_check is None:
            try:
                from . import _lib  # noqa: PLC0415

                _, functions = _lib.load()
                state = (C.c_uint64 * (_abi.HOST_MT_STATE_BYTES // 8))()
                ok = functions["host_mt19937_seed"](C.addressof(state), 20260922) == _abi.OK
                ours = torch.empty(4096 + 48, dtype=torch.float32)
                for start, length in ((0, 4096), (4096, 48)):  # a continuation too: the state must end up where torch's does
                    ok = ok and functions["host_mt19937_randn"](C.addressof(state), C.c_void_p(ours.data_ptr() + 4 * start), length, 1) == _abi.OK
                generator = torch.Generator().manual_seed(20260922)
                theirs = torch.cat([torch.randn(4096, generator=generator), torch.randn(48, generator=generator)])
                cls._self_check = bool(ok and torch.equal(ours.view(torch.int32), theirs.view(torch.int32)))
            except Exception:  # noqa: BLE001 - whatever went wrong, the answer is "not verified"
                cls._self_check = False
            if not cls._self_check:
                import warnings  # noqa: PLC0415

                warnings.warn(
                    "torchio_amd: the restated CPU normal stream differs from this torch build's torch.randn; reference-identical "
                    "noise falls back to torch.randn on the host (slow)", RuntimeWarning, stacklevel=2)
        return cls._self_check

    @classmethod
    def takes(cls, shape) -> bool:
        """torch uses another algorithm below 16 values (not restated): such draws stay with ``torch.randn`` — and so does
        everything when the stream could not be verified against this torch build (``verified``)."""
        count = 1
        for extent in shape:
            count *= int(extent)
        return count >= 16 and cls.verified()

    DEVICE_DRAW_MIN = 1 << 20  # draws from which the device produces the stream itself (below: the host road's one small upload)

    def randn(self, shape, device) -> Tensor:
        count = 1
        for extent in shape:
            count *= int(extent)
        device = torch.device(device)
        on_gpu = device.type == "cuda"
        ahead = self._prefetched
        if ahead is not None and not (on_gpu and count == ahead[0] and device == ahead[1] and count >= self.DEVICE_DRAW_MIN
                                      and os.environ.get("TIO_DEVICE_RNG", "1") != "0"):
            self._prefetched = None
            self._fn["host_mt19937_plan_end"](ahead[3], None)
            raise EngineError("HostNormalStream: a plan was started ahead for a draw that is not the next one")
        if on_gpu and count >= self.DEVICE_DRAW_MIN and os.environ.get("TIO_DEVICE_RNG", "1") != "0":
            out = self._randn_on_device(count, device)
            if out is not None:
                return out.view(tuple(int(extent) for extent in shape))
        if on_gpu:  # (the uploads and the event that frees the staging buffer belong to the DATA's device and its current stream)
            with torch.cuda.device(device):
                return self._randn_via_host(count, device).view(tuple(int(extent) for extent in shape))
        return self._randn_via_host(count, device).view(tuple(int(extent) for extent in shape))

    def _randn_via_host(self, count: int, device) -> Tensor:
        on_gpu = device.type == "cuda"
        host = self._staging(count) if on_gpu else torch.empty(count, dtype=torch.float32)
        out = torch.empty(count, dtype=torch.float32, device=device) if on_gpu else host
        for start in range(0, count, self.CHUNK):
            stop = min(start + self.CHUNK, count)
            if count - stop < 16 and stop != count:
                stop = count  # (never leave a last chunk of fewer than 16 values)
            status = self._fn["host_mt19937_randn"](C.addressof(self._state), C.c_void_p(host.data_ptr() + 4 * start), stop - start, self.threads)
            if status != _abi.OK:
                raise EngineError(f"tio_host_mt19937_randn failed with status {status}")
            if on_gpu:
                out[start:stop].copy_(host[start:stop], non_blocking=True)
            if stop == count:
                break
        if on_gpu:
            HostNormalStream._rings().uploaded[id(host)].record()  # the staging buffer is free again once this copy has completed
        return out

    _prefetched = None  # (count, device, staging buffer, native job handle) of a plan started ahead of its launch

    def prefetch_plan(self, count: int, device) -> None:
        """Start the plan of the NEXT ``count`` draws of this stream on a native thread (``tio_host_mt19937_plan_begin``: the
        state chain runs on this rank's worker threads, no interpreter lock involved) while the caller goes on enqueueing; the
        next ``_device_plan(count, device)`` — whoever asks first: the fused Noise kernel or ``randn`` — collects it.  Called
        by ``Noise._prefetch`` from a ``Compose`` that drew its children's parameters ahead: the seed is known one millisecond
        of host work before the noise kernel is launched."""
        if self._prefetched is not None or count < self.DEVICE_DRAW_MIN or os.environ.get("TIO_DEVICE_RNG", "1") == "0":
            return
        if noise_plan_on_device() and count % 16 == 0:
            return  # (the device runs the chain: the host's part is a 5 KB prefix, written when the plan is asked for)
        device = torch.device(device)
        with torch.cuda.device(device):
            words = int(self._fn["host_mt19937_plan_words"](count))
            plan_host = self._plan_staging(words)
            lent = self._rings().lent
        lent.add(id(plan_host))  # busy from now on: no upload event guards it until `_device_plan` records one
        handle = int(self._fn["host_mt19937_plan_begin"](C.addressof(self._state), count, C.c_void_p(plan_host.data_ptr()), words, self.threads))
        if handle != 0:
            self._prefetched = (count, device, plan_host, handle, lent)
        else:
            lent.discard(id(plan_host))

    def __del__(self):  # a job that nobody collected still owns the state and the staging buffer: wait for it
        ahead = getattr(self, "_prefetched", None)
        if ahead is not None:
            self._prefetched = None
            self._fn["host_mt19937_plan_end"](ahead[3], None)
            ahead[4].discard(id(ahead[2]))

    def _device_plan(self, count: int, device):
        """The plan of ``count`` draws (``tio_host_mt19937_plan``: the host runs the mt19937 state chain — in parallel, by
        jump-ahead — and keeps a snapshot every 128 blocks), uploaded: 2.5 KB of state per 79 872 draws instead of 4 bytes
        per draw.  ``None`` when the stream stands inside a group of 16 (the host road takes over; the state is untouched)."""
        ahead, self._prefetched = self._prefetched, None
        if ahead is not None:
            used = C.c_int64(0)
            status = self._fn["host_mt19937_plan_end"](ahead[3], C.byref(used))  # (the state is the job's until it has returned)
            if ahead[0] != count or ahead[1] != torch.device(device):
                ahead[4].discard(id(ahead[2]))
                raise EngineError(f"HostNormalStream: a plan of {ahead[0]} draws on {ahead[1]} was started ahead, {count} on {device} are asked for")
            plan_host = ahead[2]
        else:
            if noise_plan_on_device():
                made = self._device_made_plan(count, device)
                if made is not None:
                    return made
            words = int(self._fn["host_mt19937_plan_words"](count))
            plan_host = self._plan_staging(words)
            used = C.c_int64(0)
            status = self._fn["host_mt19937_plan"](C.addressof(self._state), count, C.c_void_p(plan_host.data_ptr()), words, C.byref(used), self.threads)
        try:
            if status == _abi.UNSUPPORTED_CONFIG:
                return None
            if status != _abi.OK:
                raise EngineError(f"tio_host_mt19937_plan failed with status {status}")
            plan_dev = torch.empty(used.value, dtype=torch.int32, device=device)
            plan_dev.copy_(plan_host[: used.value], non_blocking=True)
            HostNormalStream._rings().uploaded.setdefault(id(plan_host), torch.cuda.Event()).record()
            return plan_host, plan_dev
        finally:
            if ahead is not None:  # from here on the upload event (if any) guards the buffer
                ahead[4].discard(id(plan_host))

    # segment polynomials on the device: {(device index, segment_blocks): (count, tensor)} — 2.5 KB per segment, made by the host
    # once per segment length (tens of milliseconds: polynomial products) and uploaded once
    _segment_polynomials: dict = {}
    _SEGMENTS = 224  # workgroups of the snapshot kernel to aim for (one per CU; a segment is a power of two of 128-block units)

    def _device_made_plan(self, count: int, device):
        """The plan with its snapshots made ON the device (``set_noise_plan("device")``): the host writes the prefix (header,
        rest of the current block, snapshot 0 — ``tio_host_mt19937_plan_prefix``) and leaves its own state owing the twists;
        ``tio_mt19937_device_snapshots`` jumps to every segment's first state and chains through it.  ``None`` (state
        untouched) where that form does not apply: the caller makes the whole plan on the host."""
        prefix_words, used, total_blocks = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        plan_host = self._plan_staging(2048)
        status = self._fn["host_mt19937_plan_prefix"](C.addressof(self._state), count, C.c_void_p(plan_host.data_ptr()), 2048,
                                                      C.byref(prefix_words), C.byref(used), C.byref(total_blocks))
        if status == _abi.UNSUPPORTED_CONFIG:
            return None
        if status != _abi.OK:
            raise EngineError(f"tio_host_mt19937_plan_prefix failed with status {status}")
        device = torch.device(device)
        index = device.index if device.index is not None else torch.cuda.current_device()
        blocks = total_blocks.value
        segment = 128
        while segment * self._SEGMENTS < blocks:
            segment *= 2
        n_segments = (blocks + segment - 1) // segment
        polys = None
        if n_segments > 1:
            cached = HostNormalStream._segment_polynomials.get((index, segment))
            if cached is None or cached[0] < n_segments - 1:
                need = max(n_segments - 1, self._SEGMENTS)
                host = torch.empty(need * 624, dtype=torch.int32)
                status = self._fn["host_mt19937_segment_polynomials"](segment, need, C.c_void_p(host.data_ptr()))
                if status != _abi.OK:
                    raise EngineError(f"tio_host_mt19937_segment_polynomials failed with status {status}")
                cached = (need, host.to(device))
                HostNormalStream._segment_polynomials[(index, segment)] = cached
            polys = cached[1]
        plan_dev = torch.empty(used.value, dtype=torch.int32, device=device)
        plan_dev[: prefix_words.value].copy_(plan_host[: prefix_words.value], non_blocking=True)
        HostNormalStream._rings().uploaded.setdefault(id(plan_host), torch.cuda.Event()).record()
        raw_stream = torch._C._cuda_getCurrentRawStream(index)
        status = self._fn["mt19937_device_snapshots"](C.c_void_p(plan_dev.data_ptr()), blocks, segment,
                                                      None if polys is None else C.c_void_p(polys.data_ptr()), C.c_void_p(raw_stream))
        if status != _abi.OK:
            raise EngineError(f"tio_mt19937_device_snapshots failed with status {status}")
        # (the consumers read the HEADER from the host copy — magic, units, tail flag, n — while the kernels read the device's:
        # a private copy, the ring buffer goes back to the ring)
        return plan_host[:16].clone(), plan_dev

    def can_draw_ahead(self, shape, device) -> bool:
        """Can :meth:`randn_ahead` take these draws?  (Nothing is drawn; a stream that stands inside a group of 16 still
        declines later — the caller then draws with :meth:`randn`.)"""
        count = 1
        for extent in shape:
            count *= int(extent)
        device = torch.device(device)
        ahead = self._prefetched
        return not (
            draw_stream(device) is None or count < self.DEVICE_DRAW_MIN or os.environ.get("TIO_DEVICE_RNG", "1") == "0"
            or (ahead is not None and (count != ahead[0] or device != ahead[1]))
        )

    def randn_ahead(self, shape, device) -> Tensor | None:
        """The next draws of this stream, made on the device's DRAW stream (``draw_stream``) instead of the data's: the kernel
        that turns the plan into normal draws is bound by vector arithmetic (torch's Box-Muller, ~135 instructions per pair)
        and depends on nothing but the plan, so it runs next to the memory-bound kernels of the data stream — the resamplers
        and the stencil's first pass of this step, or, with the host running ahead, of the step before.  The current stream is
        ordered behind the draws (an event: no host wait) and the buffer is handed over to it.  ``None`` (nothing drawn, the
        state untouched) when this road does not apply; the caller then draws with :meth:`randn` / :meth:`add_noise`."""
        if not self.can_draw_ahead(shape, device):
            return None
        count = 1
        for extent in shape:
            count *= int(extent)
        device = torch.device(device)
        out = on_draw_stream(device, lambda: self._randn_on_device(count, device))
        return None if out is None else out.view(tuple(int(extent) for extent in shape))

    def _randn_on_device(self, count: int, device) -> Tensor | None:
        """The draws made on the device (``tio_mt19937_randn_device``) from the host's plan of the state chain."""
        with torch.cuda.device(device):
            plan = self._device_plan(count, device)
            if plan is None:
                return None
            out = torch.empty(count, dtype=torch.float32, device=device)
            raw_stream = torch._C._cuda_getCurrentRawStream(device.index if device.index is not None else torch.cuda.current_device())
            status = self._fn["mt19937_randn_device"](
                C.c_void_p(plan[0].data_ptr()), C.c_void_p(plan[1].data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(raw_stream)
            )
        if status != _abi.OK:
            raise EngineError(f"tio_mt19937_randn_device failed with status {status}")
        return out

    def add_noise(self, data: Tensor, mean, std) -> Tensor | None:
        """``data + (mean + std * randn(data.shape))`` in ONE kernel (``tio_mt19937_add_noise_device``): the draws of this stream
        never exist in memory.  *data*: a dense float32 ``(B, ...)`` device tensor; *mean* / *std*: numbers or ``(B,)`` float32
        device tensors.  ``None`` (nothing drawn) when this form does not apply: the caller draws with :meth:`randn`."""
        count = data.numel()
        if (
            not data.is_cuda or data.dtype != torch.float32 or not data.is_contiguous() or count < self.DEVICE_DRAW_MIN
            or data.shape[0] < 1 or os.environ.get("TIO_DEVICE_RNG", "1") == "0" or os.environ.get("TIO_FUSED_REFERENCE_NOISE", "1") == "0"
        ):
            return None
        per_element = count // data.shape[0]
        if (isinstance(mean, Tensor) or isinstance(std, Tensor)) and per_element < 624:
            return None  # (per-element parameters: the kernel wants elements of at least one state block)
        for vector in (mean, std):  # per-element parameters as the kernel reads them: (B,) float32 next to the data
            if isinstance(vector, Tensor) and (
                vector.device != data.device or vector.dtype != torch.float32 or vector.numel() != data.shape[0]
                or not vector.is_contiguous() or vector.requires_grad
            ):
                return None
        if _wants_grad(data):  # additive: dy/dx = 1 (as Engine.add_noise)
            result = self.add_noise(data.detach(), mean, std)
            return None if result is None else _AttachBackward.apply(data, result, lambda grad: grad)
        device = data.device
        with torch.cuda.device(device):
            plan = self._device_plan(count, device)
            if plan is None:
                return None
            out = torch.empty_like(data)
            mean_dev = mean if isinstance(mean, Tensor) else None
            std_dev = std if isinstance(std, Tensor) else None
            raw_stream = torch._C._cuda_getCurrentRawStream(device.index if device.index is not None else torch.cuda.current_device())
            status = self._fn["mt19937_add_noise_device"](
                C.c_void_p(plan[0].data_ptr()), C.c_void_p(plan[1].data_ptr()), C.c_void_p(data.data_ptr()), C.c_void_p(out.data_ptr()), per_element,
                0.0 if mean_dev is not None else float(mean), 0.0 if std_dev is not None else float(std),
                C.c_void_p(mean_dev.data_ptr()) if mean_dev is not None else None, C.c_void_p(std_dev.data_ptr()) if std_dev is not None else None,
                C.c_void_p(raw_stream),
            )
        if status != _abi.OK:
            raise EngineError(f"tio_mt19937_add_noise_device failed with status {status}")
        return out

    # Pinned host buffers (allocating 512 MiB of pinned memory costs ~0.2 s): kept PER THREAD — a buffer is written by the
    # host and read by an asynchronous upload, and `Queue`'s workers draw concurrently (a ring shared between threads handed
    # two of them the same buffer: one's raw state words went up as the other's "draws") — and per size, a few per size
    # used in turn, each guarded by the event of its last upload.
    _thread_state = threading.local()

    @classmethod
    def _rings(cls):
        state = cls._thread_state
        if not hasattr(state, "per_device"):
            state.per_device = {}
        device = torch.cuda.current_device()  # (an event belongs to the device it is first recorded on: one set of rings per device)
        rings = state.per_device.get(device)
        if rings is None:
            rings = state.per_device[device] = types.SimpleNamespace(uploaded={}, plans={}, buffers={}, lent=set())
        return rings

    @classmethod
    def _ring_buffer(cls, rings: dict, uploaded: dict, size: int, dtype, length: int) -> Tensor:
        ring = rings.setdefault(size, [])
        lent = cls._rings().lent
        if len(rings) > 4:  # (a few distinct sizes at most: drop the rest)
            # (never a buffer a native plan job still owns — `_device_plan` records its upload event later; a
            # ring keeps such buffers and is dropped at a later call)
            for key in [k for k in rings if k != size]:
                kept = [tensor for tensor in rings[key] if id(tensor) in lent]
                for tensor in rings[key]:
                    if id(tensor) not in lent:
                        uploaded.pop(id(tensor), None)
                if kept:
                    rings[key] = kept
                else:
                    rings.pop(key)
        # (`query()` is True for an event that was never recorded — a buffer handed to a native plan job by
        # `prefetch_plan` has no recorded upload yet, so a second request of the same size, before the first plan is collected,
        # used to get the SAME buffer: two Noise children of one Compose then shared a plan.  Buffers lent to a job are listed
        # in `lent` until `_device_plan` has recorded their upload.)
        for tensor in ring:
            if id(tensor) not in lent and uploaded[id(tensor)].query():
                return tensor
        if len(ring) < length or all(id(tensor) in lent for tensor in ring):
            tensor = torch.empty(size, dtype=dtype, pin_memory=True)
            ring.append(tensor)
            uploaded[id(tensor)] = torch.cuda.Event()
            return tensor
        free = next(tensor for tensor in ring if id(tensor) not in lent)
        uploaded[id(free)].synchronize()
        return free

    @classmethod
    def _plan_staging(cls, words: int) -> Tensor:
        state = cls._rings()
        return cls._ring_buffer(state.plans, state.uploaded, words, torch.int32, 3)

    @classmethod
    def _staging(cls, count: int) -> Tensor:
        state = cls._rings()
        return cls._ring_buffer(state.buffers, state.uploaded, count, torch.float32, 2)


class EngineError(RuntimeError):
    """A ``tio_*`` call returned a non-zero status."""


def dtype_code(dtype: torch.dtype) -> int:
    try:
        return _DTYPE_CODES[dtype]
    except KeyError:
        raise TypeError(f"unsupported image dtype {dtype}") from None


def _ptr(t: Tensor | None):
    return None if t is None else C.c_void_p(t.data_ptr())


def _i32x3(values) -> C.Array:
    return (C.c_int32 * 3)(*[int(v) for v in values])


class _AttachBackward(torch.autograd.Function):
    """Give an engine result its place in the autograd graph: ``result`` was computed by a kernel from
    ``source.detach()``; ``backward_fn(grad)`` returns dL/d(source) (another kernel, or tensor algebra)."""

    @staticmethod
    def forward(ctx, source, result, backward_fn):
        ctx.backward_fn = backward_fn
        return result.view_as(result)

    @staticmethod
    def backward(ctx, grad):
        return ctx.backward_fn(grad.contiguous()), None, None


class _PadConstantFn(torch.autograd.Function):
    """Constant padding (one value, or one per batch element for the statistic modes) with its two gradients: the
    interior of the incoming gradient for the data, the sum over the border for each element's fill value."""

    @staticmethod
    def forward(ctx, data, fill_per_element, engine, padding, fill):
        ctx.padding, ctx.in_shape, ctx.has_fills = padding, data.shape, fill_per_element is not None
        ctx.fill_dtype = None if fill_per_element is None else fill_per_element.dtype
        ctx.fill_shape = None if fill_per_element is None else fill_per_element.shape
        fills = None if fill_per_element is None else fill_per_element.detach()
        return engine.pad3d(data.detach(), padding, "constant", fill, fills)

    @staticmethod
    def backward(ctx, grad):
        p, (_, _, si, sj, sk) = ctx.padding, ctx.in_shape
        inner = grad[:, :, p[0] : p[0] + si, p[2] : p[2] + sj, p[4] : p[4] + sk]
        grad_fill = None
        if ctx.has_fills:
            border = grad.sum(dim=(1, 2, 3, 4)) - inner.sum(dim=(1, 2, 3, 4))
            grad_fill = border.to(ctx.fill_dtype).reshape(ctx.fill_shape)
        return inner.contiguous(), grad_fill, None, None, None


def _wants_grad(tensor: Tensor | None) -> bool:
    return tensor is not None and torch.is_grad_enabled() and tensor.requires_grad


class Engine:
    """One loaded implementation of the C ABI bound to one device type.

    Autograd: the reference's transforms are differentiable with respect to the image data (they are compositions
    of torch ops).  Here every op that has a derivative takes inputs that require grad, runs its kernel on the
    detached data and attaches a backward: the adjoint scatter kernel for trilinear resampling
    (``TIO_LINEAR_ADJOINT``), the same multiply for the bias field, the flip for the flip, closed-form tensor algebra
    for gamma / noise, the transposed clamped stencil for the blur (``tio_separable_conv3d_adjoint``, round 6).
    Parameters (mapping, control points, sigmas ...) are plain numbers in the reference too and get no gradient.
    """

    def __init__(self, functions: dict, device_type: str, name: str):
        self._fn = functions
        self.device_type = device_type
        self.name = name

    # -- plumbing -----------------------------------------------------------
    def _check(self, what: str, *tensors: Tensor | None) -> None:
        for t in tensors:
            if t is None:
                continue
            if t.device.type != self.device_type:
                raise EngineError(
                    f"{what}: tensor on {t.device} but the {self.name} engine runs on {self.device_type} tensors — move the data"
                    f" to the GPU (`subject.to('cuda')`), or keep using `torchio` itself with"
                    " `torchio_amd.reference_binding.bind()`: there, host tensors stay on the reference's own code"
                )
            if t.requires_grad and torch.is_grad_enabled():  # (with grad disabled a tensor that requires grad is just data)
                raise EngineError(
                    f"{what}: this {self.name} engine op has no backward (differentiable: trilinear resampling, bias field, blur,"
                    " noise, gamma, flip) — detach the tensor, or use `torchio` with `torchio_amd.reference_binding.bind()`,"
                    " which leaves autograd inputs to the reference's own path"
                )

    def _stream(self, ref: Tensor):
        if self.device_type == "cuda":
            # torch._C._cuda_getCurrentRawStream: the raw handle without building a torch.cuda.Stream object per call
            return C.c_void_p(torch._C._cuda_getCurrentRawStream(ref.device.index if ref.device.index is not None else torch.cuda.current_device()))
        return None

    def _call(self, name: str, ref: Tensor, *args) -> None:
        if self.device_type == "cuda" and ref.device.index != torch.cuda.current_device():
            with torch.cuda.device(ref.device):
                status = self._fn[name](*args)
        else:
            status = self._fn[name](*args)
        if status != _abi.OK:
            message = ""
            if "last_error" in self._fn:
                message = (self._fn["last_error"]() or b"").decode(errors="replace")
            raise EngineError(f"tio_{name} failed with status {status}: {message}")

    @staticmethod
    def _flags(flags: Tensor | None, batch: int, what: str) -> Tensor | None:
        if flags is None:
            return None
        if flags.dtype not in (torch.uint8, torch.bool) or flags.numel() != batch:
            raise ValueError(f"{what} must hold {batch} uint8/bool flags")
        return flags.to(torch.uint8).contiguous()

    # -- spatial ------------------------------------------------------------
    def _resample_geom(
        self, batch: int, in_shape, out_shape, mapping: Tensor, control_points: Tensor | None, in_spacing, out_spacing,
        affine_first: bool, cp_skip: Tensor | None, passthrough: Tensor | None, norm_shape, precision: str | None,
        large_boxes: bool = False,
    ):
        """``tio_resample_geom`` of one launch and the tensors it points into (mapping, control points, flags — keep them alive)."""
        if mapping.dtype != torch.float32 or not mapping.is_contiguous():
            mapping = mapping.to(torch.float32).contiguous()
        if mapping.ndim != 3 or mapping.shape[1:] != (3, 4) or mapping.shape[0] not in (1, batch):
            raise ValueError(f"mapping must be (1|B, 3, 4), got {tuple(mapping.shape)}")
        geom = _abi.ResampleGeom()
        geom.batch = batch
        geom.in_shape = _i32x3(in_shape)
        geom.out_shape = _i32x3(out_shape)
        geom.affine_first = int(bool(affine_first))
        geom.mapping_dev = mapping.data_ptr()
        geom.mapping_batched = int(mapping.shape[0] == batch and batch > 1)
        if control_points is not None:
            if control_points.dtype != torch.float32 or not control_points.is_contiguous():
                control_points = control_points.to(torch.float32).contiguous()
            if control_points.ndim != 5 or control_points.shape[-1] != 3 or control_points.shape[0] not in (1, batch):
                raise ValueError(f"control_points must be (1|B, ni, nj, nk, 3), got {tuple(control_points.shape)}")
            geom.control_points_dev = control_points.data_ptr()
            geom.cp_batched = int(control_points.shape[0] == batch and batch > 1)
            geom.cp_shape = _i32x3(control_points.shape[1:4])
        cp_skip = self._flags(cp_skip, batch, "cp_skip")
        passthrough = self._flags(passthrough, batch, "passthrough")
        geom.cp_skip_dev = None if cp_skip is None else cp_skip.data_ptr()
        geom.passthrough_dev = None if passthrough is None else passthrough.data_ptr()
        geom.in_spacing = (C.c_float * 3)(*[float(s) for s in in_spacing])
        geom.out_spacing = (C.c_float * 3)(*[float(s) for s in out_spacing])
        if norm_shape is not None:  # the shape the coordinates are normalised with, when it is not the images' own
            geom.norm_shape = _i32x3(norm_shape)
        geom.precision = PRECISION_CODES[precision if precision is not None else _RESAMPLE_PRECISION]
        geom.flags = (0, _abi.GEOM_LARGE_BOXES, _abi.GEOM_MOSTLY_LARGE_BOXES)[min(int(large_boxes), 2)]  # (0 / False: no hint, 1 / True: some bricks, 2: most)
        self._check("resample3d", mapping, control_points, cp_skip, passthrough)
        return geom, [mapping, control_points, cp_skip, passthrough]

    def resample_plan(
        self, *, batch: int, in_shape, out_shape, mapping: Tensor, control_points: Tensor | None, in_spacing, out_spacing,
        affine_first: bool, cp_skip: Tensor | None = None, passthrough: Tensor | None = None, norm_shape=None,
        precision: str | None = None, large_boxes: bool = False,
    ) -> Tensor | None:
        """The brick plan of the launch ``resample3d`` would make for this geometry with float32 trilinear images, enqueued on
        the CURRENT stream of the mapping's device (``tio_resample3d_plan``, ABI 11) — or ``None`` when that launch takes a
        road without a plan (small batches, ...).  Hand the tensor to ``resample3d(..., plan=...)`` with the same geometry, on
        a stream ordered behind this one; the planning kernel then leaves that call's critical path."""
        if "resample3d_plan" not in self._fn or mapping.device.type != "cuda":
            return None
        geom, keep_alive = self._resample_geom(
            batch, tuple(in_shape), tuple(int(s) for s in out_shape), mapping, control_points, in_spacing, out_spacing,
            affine_first, cp_skip, passthrough, norm_shape, precision, large_boxes,
        )
        reference = keep_alive[0]
        if reference.device.index != torch.cuda.current_device():
            with torch.cuda.device(reference.device):
                size = self._fn["resample3d_plan_bytes"](C.byref(geom))
        else:
            size = self._fn["resample3d_plan_bytes"](C.byref(geom))
        if size <= 0:
            return None
        plan = torch.empty((size + 3) // 4, dtype=torch.int32, device=reference.device)
        self._call("resample3d_plan", reference, C.byref(geom), C.c_void_p(plan.data_ptr()), size, self._stream(reference))
        del keep_alive
        return plan

    def resample3d(
        self,
        images: Sequence[Tensor],
        *,
        out_shape: Sequence[int],
        mapping: Tensor,
        control_points: Tensor | None,
        in_spacing: Sequence[float],
        out_spacing: Sequence[float],
        affine_first: bool,
        interps: Sequence[str | int],
        fills: Sequence[Tensor | None],
        cp_skip: Tensor | None = None,
        passthrough: Tensor | None = None,
        label_tables: Sequence[Tensor | None] | None = None,
        pad_labels: Sequence[float] | None = None,
        norm_shape: Sequence[int] | None = None,
        precision: str | None = None,
        plan: Tensor | None = None,
        large_boxes: bool = False,
        _adjoint_of: Sequence[Tensor] | None = None,
    ) -> list[Tensor]:
        """Resample every ``(B, C, I, J, K)`` tensor in *images* through one coordinate pass.

        ``mapping`` is ``(1|B, 3, 4)`` float32 (output voxel → input voxel),
        ``control_points`` ``(1|B, ni, nj, nk, 3)`` float32 in mm or ``None``;
        ``fills[n]`` is ``None`` (zero padding, no mask) or a ``(C,)`` float32
        tensor.  Returns new contiguous tensors ``(B, C, *out_shape)`` with the
        input dtypes.  Images with interpolation ``"label"`` (single channel) take the
        fused partial-volume mode: ``label_tables[n]`` is ``torch.unique(images[n])`` as
        float64 on the data's device (or ``None``: exact for fewer than 16 labels) and
        ``pad_labels[n]`` the out-of-bounds label; ``fills[n]`` is ignored for them.
        ``large_boxes``: the caller's hint that some (1 / True: ``TIO_GEOM_LARGE_BOXES``) or most (2: ``TIO_GEOM_MOSTLY_LARGE_BOXES``)
        bricks' input boxes exceed the staging tile (transforms/spatial.py: ``_expects_large_boxes``) — a choice of road, never of values.
        """
        if not images:
            return []
        if not (len(images) == len(interps) == len(fills)):
            raise ValueError("images, interps and fills must have the same length")
        first = images[0]
        if first.ndim != 5:
            raise ValueError(f"expected (B, C, I, J, K) tensors, got {tuple(first.shape)}")
        batch = first.shape[0]
        in_shape = tuple(first.shape[2:])
        out_shape = tuple(int(s) for s in out_shape)
        geom, keep_alive = self._resample_geom(
            batch, in_shape, out_shape, mapping, control_points, in_spacing, out_spacing, affine_first, cp_skip, passthrough,
            norm_shape, precision, large_boxes,
        )
        mapping, control_points, cp_skip, passthrough = keep_alive[:4]
        if plan is not None and plan.device == first.device:  # made ahead by `resample_plan` for exactly this geometry
            geom.plan_dev = plan.data_ptr()
            geom.plan_bytes = plan.numel() * plan.element_size()
            keep_alive.append(plan)

        # images that take part in autograd (float, trilinear): the kernel sees the detached data, the result gets
        # the adjoint launch as its backward
        sources = list(images)
        codes = [INTERP_CODES[i] if isinstance(i, str) else int(i) for i in interps]
        if _adjoint_of is None and any(t.requires_grad for t in images):
            wants = [_wants_grad(t) and t.dtype in FLOAT_DTYPES and codes[n] == _abi.LINEAR for n, t in enumerate(images)]
            images = [t.detach() if t.requires_grad else t for t in images]
            for n, t in enumerate(sources):
                if _wants_grad(t) and not wants[n]:
                    raise EngineError("resample3d: only floating-point images resampled trilinearly are differentiable")
        else:  # (the usual call: nothing takes part in autograd)
            wants = [False] * len(images)

        # The folded minimum: a large planned launch (FAST, TIGHT, and EXACT on the lean exact-coordinate kernel) can hand back the per-channel minimum of element 0 of each output
        # (tio_resample_image.out_min_dev), which is what the NEXT spatial transform's default_pad_value="minimum" will ask
        # of exactly this tensor (`folded_channel_min`).  Requested when somebody has announced that consumer
        # (`expect_minimum_fill`: a Compose that draws ahead knows its next child) or with TIO_FOLDED_MIN=1; only the bricks of
        # element 0 track what they store, in an instantiation of its own (resample_fast.hpp), and a ~2 us kernel decodes the
        # result: it replaces the ~21 us reduction and its re-read of the volume.  Requested only where the C side folds it
        # (the rule of resample.hip's planned path, mirrored loosely: a miss costs one tio_channel_min launch, never a wrong
        # value).
        fold_min = (
            _adjoint_of is None and (minimum_fill_expected() or os.environ.get("TIO_FOLDED_MIN", "") not in ("", "0"))
            and batch * -(-out_shape[0] // 16) * -(-out_shape[1] // 16) * -(-out_shape[2] // 16) >= 12288 and not any(wants)
            and len(images) <= _abi.MAX_IMAGES and all(t.dtype == torch.float32 for t in images) and all(c == _abi.LINEAR for c in codes)
        )
        folded: list[Tensor | None] = []

        outputs: list[Tensor] = []
        for start in range(0, len(images), _abi.MAX_IMAGES):
            chunk = range(start, min(start + _abi.MAX_IMAGES, len(images)))
            descs = (_abi.ResampleImage * len(chunk))()
            for slot, n in enumerate(chunk):
                data = images[n]
                if data.ndim != 5 or data.shape[0] != batch or tuple(data.shape[2:]) != in_shape:
                    raise ValueError("all images must share the batch size and spatial shape")
                if not data.is_contiguous():
                    data = data.contiguous()
                fill = fills[n]
                if fill is not None:
                    if fill.dtype != torch.float32 or fill.device != data.device or not fill.is_contiguous():
                        fill = h2d(fill.to(torch.float32), data.device).contiguous()
                    if fill.numel() != data.shape[1]:
                        raise ValueError("fill must have one value per channel")
                self._check("resample3d", data, fill)
                if _adjoint_of is not None:  # `data` is the zeroed accumulator, `out` the incoming gradient (read)
                    out = _adjoint_of[n].to(torch.float32).contiguous()
                    if tuple(out.shape) != (batch, data.shape[1], *out_shape):
                        raise ValueError("gradient shape does not match the forward output")
                else:
                    out = torch.empty((batch, data.shape[1], *out_shape), dtype=data.dtype, device=data.device)
                descs[slot].in_ = data.data_ptr()
                descs[slot].out = out.data_ptr()
                descs[slot].channels = data.shape[1]
                descs[slot].dtype = dtype_code(data.dtype)
                descs[slot].interp = codes[n]
                descs[slot].fill_dev = None if fill is None else fill.data_ptr()
                if descs[slot].interp == _abi.LABEL_PV:
                    table = None if label_tables is None else label_tables[n]
                    if table is not None:
                        table = table.to(device=data.device, dtype=torch.float64).contiguous()
                        self._check("resample3d", table)
                        descs[slot].labels_dev = table.data_ptr()
                        descs[slot].n_labels = table.numel()
                    descs[slot].pad_label = 0.0 if pad_labels is None else float(pad_labels[n])
                    keep_alive.append(table)
                minimum = torch.empty(data.shape[1], dtype=torch.float32, device=data.device) if fold_min else None
                descs[slot].out_min_dev = None if minimum is None else minimum.data_ptr()
                folded.append(minimum)
                keep_alive += [data, fill]
                outputs.append(out)
            self._call("resample3d", first, C.byref(geom), len(chunk), descs, self._stream(first))
        note_resample_launch(first)
        del keep_alive
        for out, minimum in zip(outputs, folded, strict=True):
            if minimum is not None:
                out._tio_channel_min = (out._version, minimum)
        if _adjoint_of is not None:
            return [t for t in images]  # the accumulators now hold dL/d(input)
        if any(wants):
            shared = dict(
                out_shape=out_shape, mapping=mapping, control_points=control_points, in_spacing=tuple(in_spacing),
                out_spacing=tuple(out_spacing), affine_first=affine_first, cp_skip=cp_skip, passthrough=passthrough,
                norm_shape=norm_shape, precision="exact",
            )
            for n, want in enumerate(wants):
                if not want:
                    continue
                source, fill = sources[n], fills[n]

                def backward(grad, source=source, fill=fill):
                    accumulator = torch.zeros(source.shape, dtype=torch.float32, device=source.device)
                    self.resample3d([accumulator], interps=["linear_adjoint"], fills=[fill], _adjoint_of=[grad], **shared)
                    return accumulator.to(source.dtype)

                outputs[n] = _AttachBackward.apply(source, outputs[n], backward)
        return outputs

    def bspline_prefilter(self, data: Tensor, order: int) -> Tensor:
        """B-spline coefficients (float32) of a ``(B, C, I, J, K)`` tensor for ``resample3d``'s ``"quadratic"`` (order 2) /
        ``"cubic"`` (order 3) images: the recursive prefilter with the half-sample-symmetric boundary that
        ``interpol.grid_pull(prefilter=True, bound="dct2")`` applies before sampling (spatial.py:1753-1760)."""
        if data.ndim != 5:
            raise ValueError("expected a (B, C, I, J, K) tensor")
        if order not in (2, 3, 4, 5, 6, 7):
            raise NotImplementedError(f"B-spline order {order} is not implemented (2 ... 7 are)")
        if _wants_grad(data):
            raise EngineError("bspline_prefilter: B-spline resampling is not differentiable here (use reference_binding)")
        data = data.detach().contiguous()
        self._check("bspline_prefilter", data)
        out = torch.empty(data.shape, dtype=torch.float32, device=data.device)
        self._call(
            "bspline_prefilter", data, _ptr(data), _ptr(out), dtype_code(data.dtype), data.shape[0] * data.shape[1],
            _i32x3(data.shape[2:]), int(order), self._stream(data),
        )
        return out

    def channel_min(self, data: Tensor) -> Tensor:
        """Per-channel minimum of the first batch element as a ``(C,)`` float32 device tensor."""
        data = data.detach()  # a fill VALUE (the reference takes `.item()`): no gradient flows through it
        self._check("channel_min", data)
        first = data if data.is_contiguous() else data[0].contiguous()  # element 0 of a dense batch starts at the batch's pointer
        channels = data.shape[1]
        n_spatial = 1
        for extent in data.shape[2:]:
            n_spatial *= int(extent)
        out = torch.empty(channels, dtype=torch.float32, device=data.device)
        self._call("channel_min", data, _ptr(first), dtype_code(data.dtype), channels, n_spatial, _ptr(out), self._stream(data))
        return out

    # -- intensity ----------------------------------------------------------
    def separable_conv3d(
        self, data: Tensor, taps: Tensor, radius: Sequence[int], skip: Tensor | None = None
    ) -> Tensor:
        """Per-axis cross-correlation with replicate padding; ``taps`` is ``(1|B, 3, stride)``."""
        if data.ndim != 5:
            raise ValueError("expected a (B, C, I, J, K) tensor")
        if data.dtype not in FLOAT_DTYPES:
            raise TypeError(f"separable_conv3d needs a floating dtype, got {data.dtype}")
        if _wants_grad(data):
            detached = data.detach()
            result = self.separable_conv3d(detached, taps, radius, skip=skip)
            return _AttachBackward.apply(data, result, lambda grad: self.separable_conv3d_adjoint(grad, taps, radius, skip=skip).to(data.dtype))
        batch, channels = data.shape[:2]
        data = data.contiguous()
        taps = taps.to(torch.float32).contiguous()
        if taps.ndim != 3 or taps.shape[1] != 3 or taps.shape[0] not in (1, batch):
            raise ValueError(f"taps must be (1|B, 3, stride), got {tuple(taps.shape)}")
        skip = self._flags(skip, batch, "skip")
        self._check("separable_conv3d", data, taps, skip)
        out = torch.empty_like(data)
        active = sum(1 for r in radius if int(r) > 0)
        tmp = None
        if active > 1:
            tmp = torch.empty((2, *data.shape), dtype=torch.float32, device=data.device)
        self._call(
            "separable_conv3d", data, _ptr(data), _ptr(out), _ptr(tmp), dtype_code(data.dtype), batch, channels,
            _i32x3(data.shape[2:]), _ptr(taps), int(taps.shape[0] == batch and batch > 1), taps.shape[2],
            _i32x3(radius), _ptr(skip), self._stream(data),
        )
        return out

    def separable_conv3d_adjoint(self, grad: Tensor, taps: Tensor, radius: Sequence[int], skip: Tensor | None = None) -> Tensor:
        """Backward of ``separable_conv3d`` with respect to its data (``tio_separable_conv3d_adjoint``, ABI 16): the transposes
        of the three replicate-padded correlations in reverse order, float32 — what autograd derives through the reference's
        ``F.pad(mode="replicate")`` + grouped ``conv3d`` per axis (blur.py:157-252).  The transpose of a CLAMPED stencil folds the
        taps that were clamped onto a border voxel back onto it; rows flagged in *skip* pass their gradient through."""
        if grad.ndim != 5:
            raise ValueError("expected a (B, C, I, J, K) gradient")
        batch, channels = grad.shape[:2]
        grad = grad.detach().to(torch.float32).contiguous()
        taps = taps.to(device=grad.device, dtype=torch.float32).contiguous()
        if taps.ndim != 3 or taps.shape[1] != 3 or taps.shape[0] not in (1, batch):
            raise ValueError(f"taps must be (1|B, 3, stride), got {tuple(taps.shape)}")
        skip = self._flags(skip, batch, "skip")
        if skip is not None and skip.device != grad.device:
            skip = skip.to(grad.device)
        self._check("separable_conv3d_adjoint", grad, taps, skip)
        out = torch.empty_like(grad)
        tmp = torch.empty_like(grad) if sum(1 for r in radius if int(r) > 0) > 1 else None
        self._call(
            "separable_conv3d_adjoint", grad, _ptr(grad), _ptr(out), _ptr(tmp), batch, channels, _i32x3(grad.shape[2:]), _ptr(taps),
            int(taps.shape[0] == batch and batch > 1), taps.shape[2], _i32x3(radius), _ptr(skip), self._stream(grad),
        )
        return out

    def blur_fused(
        self,
        data: Tensor,
        taps: Tensor,
        radius: Sequence[int],
        *,
        bias_coarse: Tensor | None = None,
        noise: tuple | None = None,
    ) -> Tensor | None:
        """``Noise(Blur(BiasField(data)))`` in the passes of the separable stencil (``tio_blur_fused``).

        ``bias_coarse``: ``(B, C, si, sj, sk)`` float32 or ``None``; ``noise``: ``(mean, std,
        philox_seed)`` with scalar or ``(B,)`` tensors, or ``(mean, std, draws)`` with *draws* a float32
        device tensor of the data's shape holding one normal draw per element (the reference's seeded
        stream, ``HostNormalStream.randn_ahead``), or ``None``.  Returns ``None`` when this
        engine / these arguments have no fused form (nothing was launched): the caller then runs
        the three ops one after the other, which gives the same values.
        """
        if "blur_fused" not in self._fn or data.dtype != torch.float32 or data.ndim != 5:
            return None
        batch, channels = data.shape[:2]
        data = data.contiguous()
        taps = taps.to(torch.float32).contiguous()
        self._check("blur_fused", data, taps, bias_coarse)
        coarse_shape = None
        if bias_coarse is not None:
            bias_coarse = bias_coarse.to(torch.float32).contiguous()
            if bias_coarse.ndim != 5 or bias_coarse.shape[:2] != data.shape[:2]:
                raise ValueError("bias_coarse must be (B, C, si, sj, sk)")
            coarse_shape = _i32x3(bias_coarse.shape[2:])
        noise_on, mean_f, std_f, mean_t, std_t, batched, seed, base = 0, 0.0, 0.0, None, None, 0, 0, None
        if noise is not None:
            mean, std, seed = noise
            noise_on = 1
            if isinstance(seed, Tensor):  # explicit draws
                base, seed, noise_on = seed, 0, 2
                if base.dtype != torch.float32 or base.device != data.device or base.numel() != data.numel() or not base.is_contiguous():
                    raise ValueError("blur_fused: the draws must be a dense float32 tensor of the data's size on its device")
            batched = int(isinstance(mean, Tensor) or isinstance(std, Tensor))
            if batched:
                mean_t = torch.as_tensor(mean, dtype=torch.float32, device=data.device).expand(batch).contiguous()
                std_t = torch.as_tensor(std, dtype=torch.float32, device=data.device).expand(batch).contiguous()
            else:
                mean_f, std_f = float(mean), float(std)
        out = torch.empty_like(data)
        tmp = torch.empty((2, *data.shape), dtype=torch.float32, device=data.device)
        arguments = (
            _ptr(data), _ptr(out), _ptr(tmp), dtype_code(data.dtype), batch, channels, _i32x3(data.shape[2:]), _ptr(taps),
            int(taps.shape[0] == batch and batch > 1), taps.shape[2], _i32x3(radius), _ptr(bias_coarse), coarse_shape,
            noise_on, mean_f, std_f, _ptr(mean_t), _ptr(std_t), batched, int(seed) & (2**64 - 1), _ptr(base),
            int(_STENCIL_PRECISION == "fast"), self._stream(data),
        )
        if self.device_type == "cuda" and data.device.index != torch.cuda.current_device():
            with torch.cuda.device(data.device):
                status = self._fn["blur_fused"](*arguments)
        else:
            status = self._fn["blur_fused"](*arguments)
        if status == _abi.UNSUPPORTED_CONFIG:
            return None
        if status != _abi.OK:
            message = (self._fn["last_error"]() or b"").decode(errors="replace")
            raise EngineError(f"tio_blur_fused failed with status {status}: {message}")
        return out

    def bias_field_apply(
        self, data: Tensor, coarse: Tensor, *, divide: bool = False, skip: Tensor | None = None
    ) -> Tensor:
        """``data * exp(trilinear_upsample(coarse))`` (or ``/``); ``coarse`` is ``(B, C, si, sj, sk)``."""
        if data.ndim != 5 or coarse.ndim != 5 or coarse.shape[:2] != data.shape[:2]:
            raise ValueError("data and coarse must be 5-D with the same (B, C)")
        if data.dtype not in FLOAT_DTYPES:
            raise TypeError(f"bias_field_apply needs a floating dtype, got {data.dtype}")
        if _wants_grad(data):  # y = x * f (or x / f): the gradient is the same multiply applied to the incoming gradient
            result = self.bias_field_apply(data.detach(), coarse, divide=divide, skip=skip)
            return _AttachBackward.apply(data, result, lambda grad: self.bias_field_apply(grad.to(data.dtype), coarse, divide=divide, skip=skip))
        data = data.contiguous()
        coarse = coarse.to(torch.float32).contiguous()
        skip = self._flags(skip, data.shape[0], "skip")
        self._check("bias_field_apply", data, coarse, skip)
        out = torch.empty_like(data)
        self._call(
            "bias_field_apply", data, _ptr(data), _ptr(out), dtype_code(data.dtype), data.shape[0], data.shape[1],
            _i32x3(data.shape[2:]), _ptr(coarse), _i32x3(coarse.shape[2:]), int(bool(divide)), _ptr(skip),
            self._stream(data),
        )
        return out

    def add_noise(
        self,
        data: Tensor,
        mean: float | Tensor,
        std: float | Tensor,
        *,
        rician: bool = False,
        base1: Tensor | None = None,
        base2: Tensor | None = None,
        philox_seed: int = 0,
        keep: Tensor | None = None,
    ) -> Tensor:
        """``data + (mean + std * z)``; ``z`` from *base1*/*base2* or in-kernel Philox when ``None``."""
        if data.dtype not in FLOAT_DTYPES:
            raise TypeError(f"add_noise needs a floating dtype, got {data.dtype}")
        if _wants_grad(data):
            detached = data.detach()
            result = self.add_noise(detached, mean, std, rician=rician, base1=base1, base2=base2, philox_seed=philox_seed, keep=keep)
            if not rician:  # additive: dy/dx = 1 (gated-out rows are copies: 1 as well)
                return _AttachBackward.apply(data, result, lambda grad: grad)

            def rician_backward(grad):
                # y = sqrt((x + n1)^2 + n2^2): dy/dx = (x + n1) / y, with n1 the FIRST draw of the forward pass
                draws = base1 if base1 is not None else self.philox_normal(detached.shape, philox_seed, 0, detached.device)
                shape = (-1,) + (1,) * (detached.ndim - 1)
                mean_b = mean.to(detached.device).reshape(shape) if isinstance(mean, Tensor) else mean
                std_b = std.to(detached.device).reshape(shape) if isinstance(std, Tensor) else std
                numerator = detached.float() + (mean_b + std_b * draws)
                slope = torch.where(result != 0, numerator / result.float(), torch.zeros_like(numerator))
                if keep is not None:  # gated-out rows are exact copies of the input
                    rows = keep.to(detached.device).bool().reshape(shape)
                    slope = torch.where(rows, slope, torch.ones_like(slope))
                return (grad.float() * slope).to(data.dtype)

            return _AttachBackward.apply(data, result, rician_backward)
        data = data.contiguous()
        batch = data.shape[0]
        batched = isinstance(mean, Tensor) or isinstance(std, Tensor)
        mean_t = std_t = None
        if batched:
            mean_t = h2d(torch.as_tensor(mean, dtype=torch.float32), data.device).expand(batch).contiguous()
            std_t = h2d(torch.as_tensor(std, dtype=torch.float32), data.device).expand(batch).contiguous()
        for base in (base1, base2):
            if base is not None and (base.shape != data.shape or base.dtype != torch.float32):
                raise ValueError("base noise must be float32 and shaped like data")
        base1 = None if base1 is None else base1.contiguous()
        base2 = None if base2 is None else base2.contiguous()
        keep = self._flags(keep, batch, "keep")
        self._check("add_noise", data, mean_t, std_t, base1, base2, keep)
        out = torch.empty_like(data)
        n_per_element = data[0].numel() if batch else 0
        self._call(
            "add_noise", data, _ptr(data), _ptr(out), dtype_code(data.dtype), batch, n_per_element,
            0.0 if batched else float(mean), 0.0 if batched else float(std), _ptr(mean_t), _ptr(std_t),
            int(batched), int(bool(rician)), _ptr(base1), _ptr(base2), int(philox_seed) & (2**64 - 1),
            _ptr(keep), self._stream(data),
        )
        return out

    def philox_normal(self, shape: Sequence[int], seed: int, stream_id: int, device) -> Tensor:
        """Standard normals of the fast noise mode (Philox4x32-10 + Box-Muller)."""
        out = torch.empty(tuple(shape), dtype=torch.float32, device=device)
        self._check("philox_normal", out)
        self._call(
            "philox_normal", out, _ptr(out), out.numel(), int(seed) & (2**64 - 1), int(stream_id),
            self._stream(out),
        )
        return out

    def gamma_pow(self, data: Tensor, gamma: float | Tensor) -> Tensor:
        """``sign(data) * |data| ** gamma`` with a scalar or per-element ``(B,)`` exponent."""
        if data.dtype not in FLOAT_DTYPES:
            raise TypeError(f"gamma_pow needs a floating dtype, got {data.dtype}")
        if _wants_grad(data):  # y = sign(x) |x|^g: dy/dx = g |x|^(g - 1)
            detached = data.detach()
            result = self.gamma_pow(detached, gamma)

            def backward(grad):
                exponent = gamma.to(detached.device, torch.float32).reshape((-1,) + (1,) * (detached.ndim - 1)) if isinstance(gamma, Tensor) else float(gamma)
                return (grad.float() * exponent * detached.float().abs().pow(exponent - 1)).to(data.dtype)

            return _AttachBackward.apply(data, result, backward)
        data = data.contiguous()
        batch = data.shape[0]
        gamma_t = None
        if isinstance(gamma, Tensor):
            gamma_t = h2d(gamma.to(torch.float32), data.device).expand(batch).contiguous()
        self._check("gamma_pow", data, gamma_t)
        out = torch.empty_like(data)
        self._call(
            "gamma_pow", data, _ptr(data), _ptr(out), dtype_code(data.dtype), batch,
            data[0].numel() if batch else 0, 0.0 if gamma_t is not None else float(gamma), _ptr(gamma_t),
            int(gamma_t is not None), self._stream(data),
        )
        return out


    # -- F.interpolate users ---------------------------------------------------
    def interpolate3d(self, data: Tensor, out_shape: Sequence[int], mode: str) -> Tensor:
        """``F.interpolate(data.float(), size=out_shape, mode).to(data.dtype)``: ``"nearest"`` or ``"linear"`` (trilinear, align_corners)."""
        if data.ndim != 5:
            raise ValueError(f"expected a (B, C, I, J, K) tensor, got {tuple(data.shape)}")
        data = data.contiguous()
        self._check("interpolate3d", data)
        out_shape = tuple(int(s) for s in out_shape)
        out = torch.empty((*data.shape[:2], *out_shape), dtype=data.dtype, device=data.device)
        self._call(
            "interpolate3d", data, _ptr(data), _ptr(out), dtype_code(data.dtype), data.shape[0] * data.shape[1],
            _i32x3(data.shape[2:]), _i32x3(out_shape), INTERP_CODES[mode], self._stream(data),
        )
        return out

    def axis_gather_lerp(self, data: Tensor, axis: int, lower: Tensor, upper: Tensor | None = None,
                         weight: Tensor | None = None, active: Tensor | None = None) -> Tensor:
        """Per-element gather (``upper is None``) or two-point blend along one spatial axis; tables are ``(B, length)``."""
        if data.ndim != 5:
            raise ValueError(f"expected a (B, C, I, J, K) tensor, got {tuple(data.shape)}")
        data = data.contiguous()
        batch, length = data.shape[0], data.shape[2 + axis]

        def table(t, dtype):
            if t is None:
                return None
            t = h2d(t.to(dtype).contiguous(), data.device)
            if tuple(t.shape) != (batch, length):
                raise ValueError(f"index / weight tables must be (B, length) = {(batch, length)}, got {tuple(t.shape)}")
            return t

        lower, upper, weight = table(lower, torch.int32), table(upper, torch.int32), table(weight, torch.float32)
        active = self._flags(active, batch, "active")
        if active is not None:
            active = h2d(active, data.device)
        self._check("axis_gather_lerp", data, lower, upper, weight, active)
        out = torch.empty_like(data)
        self._call(
            "axis_gather_lerp", data, _ptr(data), _ptr(out), dtype_code(data.dtype), batch, data.shape[1], _i32x3(data.shape[2:]),
            int(axis), _ptr(lower), _ptr(upper), _ptr(weight), _ptr(active), self._stream(data),
        )
        return out

    def flip3d(self, data: Tensor, axes: Sequence[int] | None = None, per_element: Tensor | None = None) -> Tensor:
        """``torch.flip`` along spatial ``axes`` (0..2), or per element with a ``(B, 3)`` flag tensor."""
        if data.ndim != 5:
            raise ValueError(f"expected a (B, C, I, J, K) tensor, got {tuple(data.shape)}")
        if _wants_grad(data):  # a permutation: its transpose is itself
            result = self.flip3d(data.detach(), axes, per_element)
            return _AttachBackward.apply(data, result, lambda grad: self.flip3d(grad, axes, per_element))
        data = data.contiguous()
        mask = 0
        for axis in axes or ():
            if axis not in (0, 1, 2):
                raise ValueError(f"Axis must be 0, 1, or 2; got {axis}")
            mask |= 1 << axis
        flags = None
        if per_element is not None:
            if tuple(per_element.shape) != (data.shape[0], 3):
                raise ValueError(f"per-element flags must be (B, 3), got {tuple(per_element.shape)}")
            flags = h2d(per_element.to(torch.uint8).contiguous(), data.device)
        self._check("flip3d", data, flags)
        out = torch.empty_like(data)
        self._call("flip3d", data, _ptr(data), _ptr(out), dtype_code(data.dtype), data.shape[0], data.shape[1],
                   _i32x3(data.shape[2:]), mask, _ptr(flags), self._stream(data))
        return out

    def pad3d(self, data: Tensor, padding: Sequence[int], mode: str = "constant", fill: float = 0.0,
              fill_per_element: Tensor | None = None) -> Tensor:
        """``F.pad`` of the three spatial axes: ``padding = (i0, i1, j0, j1, k0, k1)``, ``mode`` as in ``F.pad``.

        ``fill_per_element``: ``(B,)`` constants (one per batch element, any dtype: cast to the data's).
        """
        if data.ndim != 5:
            raise ValueError(f"expected a (B, C, I, J, K) tensor, got {tuple(data.shape)}")
        code = {"constant": _abi.PAD_CONSTANT, "reflect": _abi.PAD_REFLECT, "replicate": _abi.PAD_REPLICATE, "circular": _abi.PAD_CIRCULAR}[mode]
        if _wants_grad(data) or _wants_grad(fill_per_element):
            if mode != "constant":
                raise EngineError(f"pad3d: mode {mode!r} has no backward in the {self.name} engine (constant / statistic padding does)")
            return _PadConstantFn.apply(data, fill_per_element, self, [int(p) for p in padding], float(fill))
        data = data.contiguous()
        padding = [int(p) for p in padding]
        if len(padding) != 6:
            raise ValueError("padding must have 6 values (i0, i1, j0, j1, k0, k1)")
        shape = [data.shape[2 + d] + padding[2 * d] + padding[2 * d + 1] for d in range(3)]
        fills = None
        if fill_per_element is not None:
            fills = fill_per_element.to(device=data.device, dtype=data.dtype).contiguous()
            if fills.numel() != data.shape[0]:
                raise ValueError("one fill value per batch element")
        self._check("pad3d", data, fills)
        out = torch.empty((*data.shape[:2], *shape), dtype=data.dtype, device=data.device)
        self._call("pad3d", data, _ptr(data), _ptr(out), dtype_code(data.dtype), data.shape[0], data.shape[1], _i32x3(data.shape[2:]),
                   (C.c_int32 * 6)(*padding), code, float(fill), _ptr(fills), self._stream(data))
        return out

    def unique_labels(self, data: Tensor) -> Tensor:
        """``torch.unique(data).double()`` (sorted) of a label map; a bitmap pass for 8- / 16-bit integers.

        One 4-byte read-back of the count, like ``torch.unique`` itself (the table's length is a host quantity).
        """
        if data.dtype not in (torch.uint8, torch.int8, torch.int16):
            return torch.unique(data).to(torch.float64)  # wide / floating label maps: ATen's sort
        data = data.contiguous()
        if data.data_ptr() % 16 != 0:
            data = data.clone()
        self._check("unique_labels", data)
        capacity = 256 if data.dtype != torch.int16 else 65536
        table = torch.empty(capacity, dtype=torch.float64, device=data.device)
        count = torch.empty(1, dtype=torch.int32, device=data.device)
        workspace = torch.empty(2048, dtype=torch.int32, device=data.device)
        self._call("unique_labels", data, _ptr(data), dtype_code(data.dtype), data.numel(), _ptr(table), _ptr(count), _ptr(workspace),
                   self._stream(data))
        return table[: int(count.item())]

    def kspace_segment_mix(self, segments: Sequence[Tensor], bounds: Sequence[int], out_dtype: torch.dtype,
                           active: Tensor | None = None) -> Tensor:
        """Motion's k-space composite (motion.py:334-372) of float32 ``(B, C, I, J, K)`` images.

        ``segments[0]`` is the still image, ``segments[s]`` the moved image whose spectrum fills
        the k-space planes ``[bounds[s], bounds[s + 1])`` along the first spatial axis.  Rows
        of inactive elements (``active[b] == 0``) are left unwritten.
        """
        first = segments[0]
        if first.ndim != 5:
            raise ValueError(f"expected (B, C, I, J, K) tensors, got {tuple(first.shape)}")
        if len(segments) > _abi.MAX_SEGMENTS:
            raise ValueError(f"at most {_abi.MAX_SEGMENTS - 1} motion transforms are supported, got {len(segments) - 1}")
        if len(bounds) != len(segments) + 1:
            raise ValueError("bounds must have one more entry than segments")
        segments = [s.contiguous() for s in segments]
        for s in segments:
            if s.dtype != torch.float32 or s.shape != first.shape:
                raise ValueError("segments must be float32 tensors of one shape")
        self._check("kspace_segment_mix", *segments, active)
        length = int(first.shape[2])
        table = self._mix_table(length, tuple(int(v) for v in bounds), first.device)
        flags = None if active is None else active.to(torch.uint8).contiguous()
        out = torch.empty(first.shape, dtype=out_dtype, device=first.device)
        pointers = (C.c_void_p * len(segments))(*[s.data_ptr() for s in segments])
        self._call("kspace_segment_mix", first, pointers, len(segments), (C.c_int32 * len(bounds))(*[int(v) for v in bounds]),
                   _ptr(table), _ptr(out), dtype_code(out_dtype), first.shape[0], first.shape[1], _i32x3(first.shape[2:]),
                   _ptr(flags), self._stream(first))
        return out

    def _mix_table(self, length: int, bounds: tuple, device) -> Tensor:
        """The ``(n_segments, I, I)`` band-pass table of ``tio_kspace_segment_mix`` on *device* (built once per shape)."""
        cache = self.__dict__.setdefault("_mix_tables", {})
        key = (length, bounds, str(device))
        table = cache.get(key)
        if table is None:
            host = torch.empty(len(bounds) - 1, length, length, dtype=torch.float32)
            status = self._fn["kspace_mix_table"](length, len(bounds) - 1, (C.c_int32 * len(bounds))(*bounds), C.c_void_p(host.data_ptr()))
            if status != _abi.OK:
                message = (self._fn["last_error"]() or b"").decode(errors="replace") if "last_error" in self._fn else ""
                raise EngineError(f"tio_kspace_mix_table failed with status {status}: {message}")
            if len(cache) >= 8:
                cache.clear()
            table = cache[key] = h2d(host, device) if self.device_type == "cuda" else host
        return table

    # -- feeding side -------------------------------------------------------
    def patch_accumulate(
        self,
        out: Tensor,
        weight_sum: Tensor | None,
        patches: Tensor,
        placements: Sequence[tuple[Sequence[int], Sequence[int], Sequence[int]]],
        mode: str,
        windows: Sequence[Tensor] | None = None,
    ) -> None:
        """Add ``patches`` ``(N, C, pi, pj, pk)`` into the ``(C, I, J, K)`` accumulators IN PLACE, in patch order.

        ``placements[n] = (dst_ini, src_ini, extent)``; ``mode`` is ``"crop"`` (element copy),
        ``"average"`` or ``"hann"`` (then ``windows`` = three float32 1-D windows on the device).
        """
        if patches.ndim != 5 or out.ndim != 4 or patches.shape[1] != out.shape[0]:
            raise ValueError(f"expected patches (N, C, i, j, k) and out (C, I, J, K), got {tuple(patches.shape)} / {tuple(out.shape)}")
        if patches.dtype != out.dtype or (weight_sum is not None and weight_sum.dtype != out.dtype):
            raise TypeError("patches and accumulators must share one dtype")
        if not out.is_contiguous() or (weight_sum is not None and not weight_sum.is_contiguous()):
            raise ValueError("accumulators must be contiguous")
        if len(placements) != patches.shape[0]:
            raise ValueError("one placement per patch")
        code = {"crop": _abi.OVERLAP_CROP, "average": _abi.OVERLAP_AVERAGE, "hann": _abi.OVERLAP_HANN}[mode]
        if code != _abi.OVERLAP_CROP and out.dtype not in FLOAT_DTYPES:
            raise TypeError(f"overlap_mode {mode!r} needs floating-point patches, got {out.dtype}")
        patches = patches.contiguous()
        window_tensors = [None, None, None]
        if code == _abi.OVERLAP_HANN:
            window_tensors = [w.to(device=out.device, dtype=torch.float32).contiguous() for w in windows]
        self._check("patch_accumulate", out, weight_sum, patches, *window_tensors)
        for start in range(0, patches.shape[0], _abi.MAX_PATCHES):
            chunk = patches[start : start + _abi.MAX_PATCHES]
            table = (_abi.PatchPlacement * chunk.shape[0])()
            for slot, (dst_ini, src_ini, extent) in enumerate(placements[start : start + chunk.shape[0]]):
                table[slot].dst_ini = _i32x3(dst_ini)
                table[slot].src_ini = _i32x3(src_ini)
                table[slot].extent = _i32x3(extent)
            self._call(
                "patch_accumulate", out, _ptr(out), _ptr(weight_sum), dtype_code(out.dtype), out.shape[0],
                _i32x3(out.shape[1:]), _ptr(chunk), chunk.shape[0], _i32x3(chunk.shape[2:]), table, code,
                _ptr(window_tensors[0]), _ptr(window_tensors[1]), _ptr(window_tensors[2]), self._stream(out),
            )


_ENGINE: Engine | None = None
_HIP_ENGINE: Engine | None = None


def hip_engine() -> Engine:
    """The engine over ``libtio_hip.so`` whatever `engine()` currently hands out (the tests swap that one for the CPU oracle):
    for callers that are bound to the HIP library themselves — the ``torch.ops.tio_hip`` custom ops' backward passes."""
    global _HIP_ENGINE
    if _HIP_ENGINE is None:
        from . import _lib  # noqa: PLC0415

        _, functions = _lib.load()
        _HIP_ENGINE = Engine(functions, "cuda", "hip")
    return _HIP_ENGINE


def engine() -> Engine:
    """The HIP engine (loads ``libtio_hip.so`` on first use; raises if unavailable)."""
    global _ENGINE
    if _ENGINE is None:
        from . import _lib  # noqa: PLC0415

        _, functions = _lib.load()
        _ENGINE = Engine(functions, "cuda", "hip")
    return _ENGINE
