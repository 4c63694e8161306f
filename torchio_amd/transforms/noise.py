"""``Noise`` on the HIP engine (mirror of reference ``transforms/intensity/noise.py``).

Two sources for the standard-normal draws:

``"reference"`` (default)
    exactly the reference: ``torch.randn(data.shape, generator=cpu_gen)`` from ONE
    CPU generator seeded with ``params["seed"]``, shared by the images in dict
    order (noise.py:108-116,177), copied to the device; bit-identical output.
``"philox"``
    the draws are generated inside the kernel (Philox4x32-10 + Box-Muller keyed by
    ``seed`` and the element index): no host RNG, no 64 MiB H2D per volume.  Same
    distribution, different stream — results are NOT reference-identical.

Select with :func:`set_noise_rng` (or ``TIO_NOISE_RNG``); the transform signature
is untouched so ``repr`` / history stay compatible.
"""
from __future__ import annotations

import os
from typing import Any

import torch
from torch import Tensor

from .. import ops
from ..data import _pending
from ..data.batch import SubjectsBatch
from .parameter_range import to_nonneg_range
from .parameter_range import to_range
from .transform import IntensityTransform

_NOISE_RNG = os.environ.get("TIO_NOISE_RNG", "reference")
_STREAMS_AHEAD: dict[int, Any] = {}  # id(params) -> HostNormalStream whose first plan a Compose started ahead (Noise._prefetch)


def set_noise_rng(mode: str) -> None:
    """``"reference"`` (seeded CPU mt19937 draws, parity) or ``"philox"`` (in-kernel draws)."""
    global _NOISE_RNG
    if mode not in ("reference", "philox"):
        raise ValueError('noise rng must be "reference" or "philox"')
    _NOISE_RNG = mode


def get_noise_rng() -> str:
    return _NOISE_RNG


def _reference_stream_images(transform, batch: SubjectsBatch):
    """The tensors the reference-identical stream would be drawn for on the device, or None (another mode / host images).
    (A module function: `Noise.apply_transform` also runs with the REFERENCE's transform instance as `self`, reference_binding.)"""
    images = transform._get_images(batch)
    if _NOISE_RNG != "reference" or not images or os.environ.get("TIO_HOST_RNG", "1") == "0":
        return None
    tensors = [img._data if hasattr(img, "_data") else img.data for img in images.values()]  # (shape / device: pending stages keep both)
    if all(t.is_cuda and ops.HostNormalStream.takes(t.shape) for t in tensors):
        return tensors
    return None


class Noise(IntensityTransform):
    """Additive Gaussian (or Rician) noise (noise.py:18-123)."""

    def __init__(self, *, mean=0.0, std=0.25, rician: bool = False, **kwargs: Any) -> None:
        super().__init__(**kwargs)
        self.mean = to_range(mean)
        self.std = to_nonneg_range(std)
        self.rician = rician

    @property
    def supports_per_instance_params(self) -> bool:
        return True

    @property
    def supports_per_instance_p(self) -> bool:
        return True

    @property
    def draws_ahead(self) -> bool:
        # parameters from the batch size alone; intensities change, geometry does not (a subclass that overrides either half speaks for itself)
        return type(self).make_params is Noise.make_params and type(self).apply_transform is Noise.apply_transform

    prefetch_is_threaded = True  # (Compose's draw-ahead road: `_prefetch` hands work to a native thread — it goes first)

    def make_params(self, batch: SubjectsBatch) -> dict[str, Any]:
        seed = int(torch.randint(0, 2**31, (1,)).item())  # seed FIRST, then mean, std (noise.py:75-80)
        n = self._resolve_n(batch)
        keep = self._keep_mask(batch, n)
        mean = self._mask_identity(self.mean.sample_1d(n), keep, identity=0.0)
        std = self._mask_identity(self.std.sample_1d(n), keep, identity=0.0)
        params = {
            "mean": self._serialize_param(mean),
            "std": self._serialize_param(std),
            "seed": seed,
            "rician": self.rician,
        }
        self._tag_batched(params, batch, n, keep, ["mean", "std"])
        return params

    def _prefetch(self, batch: SubjectsBatch, params: dict[str, Any]) -> None:
        """Draw-ahead road (Compose): the seed is known — start the plan of the stream's first image on the helper thread."""
        tensors = _reference_stream_images(self, batch)
        if not tensors or params.get("_keep") is not None:
            return
        first = tensors[0]
        if first.dtype not in (torch.float32, torch.float64) or not first.is_contiguous():
            return  # (another dtype is converted first: its draw keeps the count, but stay on the simple road)
        stream = ops.HostNormalStream(params["seed"])
        stream.prefetch_plan(first.numel(), first.device)
        # (the entry holds the parameters object itself — its id cannot be reused while the entry lives — and the seed
        # it was started for; `_abandon_prefetch`, called by Compose whatever happens to the children in between, drops an
        # entry nobody collected: no stale stream, running job or pinned buffer outlives a failed step)
        _STREAMS_AHEAD[id(params)] = (params, int(params["seed"]), stream)

    def _abandon_prefetch(self, params: dict[str, Any]) -> None:
        _STREAMS_AHEAD.pop(id(params), None)

    def apply_transform(self, batch: SubjectsBatch, params: dict[str, Any]) -> SubjectsBatch:
        mean, std, seed = params["mean"], params["std"], params["seed"]
        rician = params.get("rician", False)
        keep = params.get("_keep")
        generator = torch.Generator(device="cpu")
        generator.manual_seed(seed)
        engine = ops.engine()
        images = self._get_images(batch)
        # the reference-identical stream for device-resident images: the same draws as `torch.randn(..., generator=generator)`
        # below — made on the device from the host's plan of the state chain, or on all host cores (ops.HostNormalStream);
        # one stream object = the one generator of this call
        stream = None
        ahead = _STREAMS_AHEAD.pop(id(params), None)  # (started by `_prefetch`: its first plan is being computed, or done)
        if ahead is not None and ahead[0] is params and ahead[1] == int(seed):
            stream = ahead[2]
        if stream is None and _reference_stream_images(self, batch) is not None:
            stream = ops.HostNormalStream(seed)
        for index, img_batch in enumerate(images.values()):
            queue = getattr(img_batch, "_pending", None)  # foreign containers (reference_binding) never defer
            if (
                queue is not None and queue.blur is not None and _NOISE_RNG != "reference" and not rician and keep is None
                and _pending.eligible(img_batch._data)
            ):  # a Blur is still queued on this tensor: the noise rides on its stores
                device = img_batch._data.device
                # per-element vectors stay on the host: the flush uploads them with the queued bias / blur blocks
                mean_arg = torch.tensor(mean, dtype=torch.float32) if isinstance(mean, list) else mean
                std_arg = torch.tensor(std, dtype=torch.float32) if isinstance(std, list) else std
                # (the Philox draws as a kernel of their own on the draw stream, the stencil only adding them, were measured:
                # reading the draws costs the J + K pass what computing them does — 0.34 against 0.31 ms — and 2 V of traffic)
                img_batch._flush(noise=(mean_arg, std_arg, (index << 32) | int(seed)))
                continue
            if (
                queue is not None and queue.blur is not None and _NOISE_RNG == "reference" and stream is not None and not rician
                and keep is None and _pending.eligible(img_batch._data) and img_batch._data.is_contiguous()
            ):
                # the reference's own stream rides on the queued Blur's stores as well: the draws of this image are made on the
                # device's draw stream (next to the memory-bound kernels of the data stream), the sum costs no pass of its own
                target = img_batch._data
                if stream.can_draw_ahead(target.shape, target.device):
                    mean_arg = torch.tensor(mean, dtype=torch.float32) if isinstance(mean, list) else mean
                    std_arg = torch.tensor(std, dtype=torch.float32) if isinstance(std, list) else std

                    def draws(stream=stream, shape=target.shape, device=target.device):
                        ahead = stream.randn_ahead(shape, device)  # (None: the stream stands inside a group of 16 — the host road)
                        return ahead if ahead is not None else stream.randn(shape, device)

                    img_batch._flush(noise=(mean_arg, std_arg, draws))
                    continue
            data = img_batch.data
            # data + float32 noise promotes half / integer data to float32 (noise.py:119)
            work = data if data.dtype in (torch.float32, torch.float64) else data.float()
            device = data.device
            mean_arg = ops.h2d(torch.tensor(mean, dtype=torch.float32), device) if isinstance(mean, list) else mean
            std_arg = ops.h2d(torch.tensor(std, dtype=torch.float32), device) if isinstance(std, list) else std
            keep_arg = None if keep is None else ops.h2d(torch.tensor(keep, dtype=torch.uint8), device)
            if _NOISE_RNG == "reference":
                if stream is not None and not rician and keep_arg is None and work is data:
                    fused = stream.add_noise(work, mean_arg, std_arg)  # draws and sum in one kernel (large float32 images)
                    if fused is not None:
                        img_batch.data = fused
                        continue
                if stream is not None:
                    base1 = stream.randn(data.shape, device)
                    base2 = stream.randn(data.shape, device) if rician else None
                else:
                    base1 = torch.randn(data.shape, generator=generator).to(device)
                    base2 = torch.randn(data.shape, generator=generator).to(device) if rician else None
                img_batch.data = engine.add_noise(
                    work, mean_arg, std_arg, rician=rician, base1=base1, base2=base2, keep=keep_arg
                )
            else:
                img_batch.data = engine.add_noise(
                    work, mean_arg, std_arg, rician=rician, philox_seed=(index << 32) | int(seed), keep=keep_arg
                )
        return batch
