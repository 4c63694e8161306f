"""Python handle on the CPU oracle (``oracle/libtio_oracle.so``).

TEST INFRASTRUCTURE ONLY: imported by ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` — never by ``torchio_amd``.  The oracle
exports the same C ABI as the HIP library (prefix ``tio_oracle_``, host pointers),
so it is driven through the same marshalling class on CPU tensors.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

from torchio_amd import _abi
from torchio_amd.ops import Engine

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBRARY_PATH = os.path.join(_HERE, "libtio_oracle.so")

_engine = None
_lib = None


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (no-op when the .so is up to date)."""
    source = os.path.join(_HERE, "tio_oracle.c")
    if force or not os.path.isfile(LIBRARY_PATH) or os.path.getmtime(LIBRARY_PATH) < os.path.getmtime(source):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True)
    return LIBRARY_PATH


def oracle_engine() -> Engine:
    """An :class:`Engine` over the CPU restatement (CPU tensors only)."""
    global _engine, _lib
    if _engine is None:
        if not os.path.isfile(LIBRARY_PATH):
            build()
        _lib = ctypes.CDLL(LIBRARY_PATH)
        functions = _abi.bind(_lib, "tio_oracle_")
        _engine = Engine(functions, "cpu", "oracle")
    return _engine


def num_threads() -> int:
    oracle_engine()
    _lib.tio_oracle_num_threads.restype = ctypes.c_int
    return int(_lib.tio_oracle_num_threads())
