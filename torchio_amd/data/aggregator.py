"""Patch aggregator for dense inference, accumulating on the device.

Host-side mirror of reference ``src/torchio/data/aggregator.py`` (same constructor,
``add_batch`` / ``get_output``, overlap modes, errors).  The reference moves every model
output to the host (``tensor.cpu()``, aggregator.py:95) and adds it with one Python slice
assignment per patch; here the accumulators live where the model outputs live and one
``tio_patch_accumulate`` launch applies a whole batch of patches, in patch order, so the
sums carry the reference's rounding (``tests/test_gpu_aggregator.py``).  ``get_output``
returns a tensor on that device.
"""
from __future__ import annotations

import torch
from torch import Tensor

from .. import ops
from .patch import PatchLocation

_MODES = ("crop", "average", "hann")


class PatchAggregator:
    """Reassemble patches into a full volume (aggregator.py:13-74).

    Args:
        spatial_shape: spatial shape ``(I, J, K)`` of the sampled volume.
        overlap_mode: ``'crop'`` keeps the non-overlapping centres, ``'average'`` averages
            overlapping values, ``'hann'`` blends with a Hann window.
        patch_overlap: the overlap used during sampling (for ``'crop'``).
        output_shape: spatial shape of the aggregated volume when the model output is
            spatially smaller than its input patch; locations are scaled accordingly.
    """

    def __init__(self, spatial_shape, overlap_mode: str = "crop", patch_overlap=0, output_shape=None) -> None:
        if overlap_mode not in _MODES:
            raise ValueError(f"overlap_mode must be one of {_MODES}, got {overlap_mode!r}")
        self.input_spatial_shape = tuple(spatial_shape)
        self.overlap_mode = overlap_mode
        if isinstance(patch_overlap, int):
            patch_overlap = (patch_overlap, patch_overlap, patch_overlap)
        self.patch_overlap = tuple(patch_overlap)
        if output_shape is not None:
            self.spatial_shape = tuple(output_shape)
            self._scale = tuple(output_shape[d] / spatial_shape[d] for d in range(3))
        else:
            self.spatial_shape = tuple(spatial_shape)
            self._scale = (1.0, 1.0, 1.0)
        self._outputs: dict[str, Tensor] = {}
        self._counts: dict[str, Tensor] = {}
        self._hann_cache: dict[tuple, list[Tensor]] = {}

    # -- public API ----------------------------------------------------------------
    def add_batch(self, batch: Tensor | dict[str, Tensor], locations: list[PatchLocation]) -> None:
        """Add model outputs ``(B, C, I, J, K)`` (or a dict of them) at ``locations`` (aggregator.py:76-99)."""
        tensors = {"__default__": batch} if isinstance(batch, Tensor) else batch
        for key, tensor in tensors.items():
            if tensor.ndim != 5:
                raise ValueError(f"expected a 5D (B, C, I, J, K) tensor, got {tuple(tensor.shape)}")
            if len(locations) != tensor.shape[0]:
                raise ValueError(f"{len(locations)} locations for a batch of {tensor.shape[0]} patches")
            scaled = [loc.scaled(self._scale) if self._scale != (1.0, 1.0, 1.0) else loc for loc in locations]
            self._ensure_buffer(key, tensor)
            placements = [self._placement(tuple(tensor.shape[2:]), loc) for loc in scaled]
            windows = self._hann_windows(tuple(tensor.shape[2:]), tensor.device) if self.overlap_mode == "hann" else None
            ops.engine().patch_accumulate(
                self._outputs[key], self._counts.get(key), tensor.detach(), placements, self.overlap_mode, windows
            )

    def get_output(self, key: str | None = None) -> Tensor:
        """The aggregated ``(C, I, J, K)`` volume, on the device of the model outputs (aggregator.py:101-126)."""
        resolve_key = key if key is not None else "__default__"
        if resolve_key not in self._outputs:
            available = [k for k in self._outputs if k != "__default__"]
            raise KeyError(f"No output for key {key!r}. Available: {available}")
        output = self._outputs[resolve_key]
        if self.overlap_mode in ("average", "hann"):
            output = output / self._counts[resolve_key].clamp(min=1)
        return output

    # -- internals -----------------------------------------------------------------
    def _ensure_buffer(self, key: str, tensor: Tensor) -> None:
        if key in self._outputs:
            buffer = self._outputs[key]
            if buffer.dtype != tensor.dtype or buffer.device != tensor.device or buffer.shape[0] != tensor.shape[1]:
                raise ValueError(
                    f"output {key!r} was started with {buffer.shape[0]} channels of {buffer.dtype} on {buffer.device}; "
                    f"got {tensor.shape[1]} channels of {tensor.dtype} on {tensor.device}"
                )
            return
        shape = (tensor.shape[1], *self.spatial_shape)
        self._outputs[key] = torch.zeros(shape, dtype=tensor.dtype, device=tensor.device)
        if self.overlap_mode in ("average", "hann"):
            self._counts[key] = torch.zeros(shape, dtype=tensor.dtype, device=tensor.device)

    def _placement(self, patch_shape: tuple, location: PatchLocation):
        """(dst_ini, src_ini, extent) of one patch: the whole patch, or its trimmed centre for 'crop'."""
        if self.overlap_mode != "crop":
            if tuple(location.size) != patch_shape:
                raise ValueError(f"patch of shape {patch_shape} at a location of size {tuple(location.size)}")
            return (location.index_ini, (0, 0, 0), patch_shape)
        # aggregator.py:166-204: trim half the (scaled) overlap on every side that is not a volume border
        half = [round(self.patch_overlap[d] * self._scale[d]) // 2 for d in range(3)]
        ini, fin = list(location.index_ini), list(location.index_fin)
        crop_ini, crop_fin = [0, 0, 0], list(location.size)
        for d in range(3):
            if ini[d] > 0:
                ini[d] += half[d]
                crop_ini[d] += half[d]
            if fin[d] < self.spatial_shape[d]:
                fin[d] -= half[d]
                crop_fin[d] -= half[d]
        extent = [max(crop_fin[d] - crop_ini[d], 0) for d in range(3)]
        return (tuple(ini), tuple(crop_ini), tuple(extent))

    def _hann_windows(self, patch_shape: tuple, device) -> list[Tensor]:
        """The three 1-D factors of the reference's 3-D window (aggregator.py:237-245), on ``device``."""
        cache_key = (patch_shape, str(device))
        if cache_key not in self._hann_cache:
            self._hann_cache[cache_key] = [
                ops.h2d(torch.hann_window(size + 2, periodic=False)[1:-1].contiguous(), device) for size in patch_shape
            ]
        return self._hann_cache[cache_key]
