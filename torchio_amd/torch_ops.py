"""``torch.ops.tio_hip.*`` — the engine as PyTorch-ROCm custom ops.

    import torchio_amd.torch_ops          # loads csrc/libtio_torch_ops.so: TORCH_LIBRARY(tio_hip, ...)
    out, = torch.ops.tio_hip.resample3d([x], [1], mapping, None, [1, 1, 1], [1, 1, 1], x.shape[2:], True, [None])

The ops are registered in C++ (``csrc/torch_ops.cpp``) on top of the C ABI of ``libtio_hip.so``: they take and
return tensors, run on the current HIP stream, allocate their outputs and never synchronise.  The dispatcher sees
them, so they can be called from TorchScript-free C++ front ends and show up in profiler traces by name.  One
composite is added here in Python: ``gaussian_blur3d(x, sigma_vox)``, which builds the reference's normalised taps
(``transforms/blur.py``) and calls ``separable_conv3d``.

The transform classes of this package keep calling the C ABI through ``ctypes`` (``ops.py``): the marshalling cost
per call is the same order (a few microseconds), and the C ABI is the boundary that also serves non-PyTorch callers.
"""
from __future__ import annotations

import os

import numpy as np
import torch

_LIBRARY = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libtio_torch_ops.so")
_loaded = False


class TorchOpsError(RuntimeError):
    pass


def load() -> None:
    """Register the ``tio_hip`` op library with the dispatcher (idempotent); fails loudly when it is not built."""
    global _loaded
    if _loaded:
        return
    if not os.path.isfile(_LIBRARY):
        raise TorchOpsError(f"{_LIBRARY} is missing: run `make -C torchio_amd/csrc` (or `python -c 'import __graft_entry__ as g; g.build()'`)")
    torch.ops.load_library(_LIBRARY)
    _loaded = True


def gaussian_blur3d(x: torch.Tensor, sigma_vox) -> torch.Tensor:
    """``_gaussian_smooth(x, sigmas)`` (reference blur.py:129-154) on the custom ops: *sigma_vox* is ``(3,)`` or ``(B, 3)`` voxels."""
    from .transforms.blur import _stacked_gaussian_taps  # noqa: PLC0415

    load()
    sigmas = np.asarray(torch.as_tensor(sigma_vox, dtype=torch.float64).cpu().numpy(), dtype=np.float64)
    if np.all(sigmas <= 0):
        return x
    per_element = sigmas.ndim == 2 and not np.all(sigmas == sigmas[0])
    rows = sigmas if per_element else (sigmas[0] if sigmas.ndim == 2 else sigmas)[None]
    taps, radius, skip = _stacked_gaussian_taps(rows, per_element=per_element)
    skip_dev = None if skip is None else torch.from_numpy(skip).to(x.device)
    return torch.ops.tio_hip.separable_conv3d(x, taps.to(x.device), [int(r) for r in radius], skip_dev)


load()
