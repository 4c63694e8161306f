"""Loader for the in-tree HIP library ``csrc/libtio_hip.so``.

The product path has exactly one compute backend: the gfx950 kernels behind the
C ABI of ``include/tio_hip.h``.  There is no CPU fallback — if the library is
missing or does not export the declared symbols, every op raises.
"""
from __future__ import annotations

import ctypes
import os
import threading

from . import _abi

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIBRARY_PATH = os.path.join(_CSRC, "libtio_hip.so")

_lock = threading.Lock()
_lib = None
_functions = None


class HipLibraryError(RuntimeError):
    """The HIP extension is missing, stale or failed to load."""


def load():
    """Return ``(CDLL, {name: function})`` for libtio_hip.so, loading it once."""
    global _lib, _functions
    if _functions is not None:
        return _lib, _functions
    with _lock:
        if _functions is not None:
            return _lib, _functions
        if not os.path.isfile(LIBRARY_PATH):
            raise HipLibraryError(
                f"{LIBRARY_PATH} not found: build it with `make -C {_CSRC}` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
                "torchio_amd has no CPU fallback."
            )
        try:
            lib = ctypes.CDLL(LIBRARY_PATH)
        except OSError as error:
            raise HipLibraryError(f"cannot load {LIBRARY_PATH}: {error}") from error
        try:
            functions = _abi.bind(lib, "tio_", _abi.HIP_ONLY_PROTOTYPES)
        except AttributeError as error:
            raise HipLibraryError(f"{LIBRARY_PATH} is stale (missing symbol): {error}") from error
        version = functions["abi_version"]()
        if version != _abi.ABI_VERSION:
            raise HipLibraryError(
                f"{LIBRARY_PATH} has ABI version {version}, expected {_abi.ABI_VERSION}: rebuild it"
            )
        _lib, _functions = lib, functions
    return _lib, _functions
