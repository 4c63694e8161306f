"""Deferred elementwise neighbours of ``Blur`` on a device-resident ``ImagesBatch``.

``Compose([..., BiasField, Blur, Noise])`` is three trips through HBM in the reference (and
in the one-kernel-per-transform engine).  The bias field is a pointwise multiply *before*
the stencil and the noise a pointwise add *after* it, so they can ride on the stencil's
loads and stores (``tio_blur_fused``) — if the three transforms are allowed to meet.  They
meet here: ``BiasField`` and ``Blur`` do not launch anything when the data qualifies, they
queue a :class:`Pending` record on the batch; ``Noise`` (or the first reader of
``ImagesBatch.data``) flushes the queue with as few launches as possible.

Nothing observable changes: every transform still draws its parameters, gates and records
history at the usual moment; ``.data`` always returns finished values (reading it flushes);
the fused launch performs the same float32 operations in the same order as the three
separate ones (tests compare them bit for bit).  ``TIO_NO_LAZY_FUSION=1`` switches the
queueing off.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from dataclasses import field

import torch
from torch import Tensor


def enabled() -> bool:
    return os.environ.get("TIO_NO_LAZY_FUSION", "") in ("", "0")


@dataclass
class Pending:
    """What has been promised for a batch's tensor but not launched yet (in application order)."""

    bias_coarse: Tensor | None = None       # (B, C, si, sj, sk) float32, still on the HOST (flush uploads everything at once)
    blur: tuple | None = None               # (taps (n, 3, stride) float32 host tensor, radius [3])
    notes: list[str] = field(default_factory=list)

    def is_empty(self) -> bool:
        return self.bias_coarse is None and self.blur is None


def eligible(data: Tensor) -> bool:
    """Only float32 device tensors take the deferred route (everything else runs at once)."""
    return enabled() and data.is_cuda and data.dtype == torch.float32 and data.ndim == 5 and not data.requires_grad


def flush(data: Tensor, pending: Pending, *, noise: tuple | None = None) -> Tensor:
    """Run what is queued (optionally with a trailing noise stage) and return the finished tensor.

    ``noise``: ``(mean, std, source)`` — *source* a Philox seed word, a tensor of explicit draws shaped like the data, or a
    callable returning either (called once, after the parameter uploads)."""
    from .. import ops  # noqa: PLC0415

    engine = ops.engine()
    # the queued parameter blocks are still on the host: one staging copy for all of them (and the noise vectors)
    mean = std = None
    if noise is not None:
        mean, std = noise[0], noise[1]
    host_mean = mean if isinstance(mean, Tensor) and mean.device.type == "cpu" else None
    host_std = std if isinstance(std, Tensor) and std.device.type == "cpu" else None
    taps_host = pending.blur[0] if pending.blur is not None else None
    coarse_dev, taps_dev, mean_dev, std_dev = ops.h2d_packed([pending.bias_coarse, taps_host, host_mean, host_std], data.device)
    pending.bias_coarse = coarse_dev
    if pending.blur is not None:
        pending.blur = (taps_dev, pending.blur[1])
    if noise is not None:
        source = noise[2]
        if callable(source):
            # explicit draws that are still being prepared (the reference's stream: the host may have to wait for the plan of the
            # generator's state chain here) — asked for AFTER the uploads above have been enqueued, so that wait overlaps them
            source = source()
        noise = (mean_dev if host_mean is not None else mean, std_dev if host_std is not None else std, source)
    if pending.blur is not None:
        taps, radius = pending.blur
        fused = engine.blur_fused(data, taps, radius, bias_coarse=pending.bias_coarse, noise=noise)
        if fused is not None:
            return fused
    # no fused form for these arguments: the plain sequence, same values
    if pending.bias_coarse is not None:
        data = engine.bias_field_apply(data, pending.bias_coarse)
    if pending.blur is not None:
        taps, radius = pending.blur
        data = engine.separable_conv3d(data, taps, radius)
    if noise is not None:
        mean, std, seed = noise
        if isinstance(seed, Tensor):  # explicit draws (the reference's stream, made ahead): the same sum as its own pass
            data = engine.add_noise(data, mean, std, base1=seed.view(data.shape))
        else:
            data = engine.add_noise(data, mean, std, philox_seed=seed)
    return data
