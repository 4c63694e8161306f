"""The C-ABI shared library loads on a CPU-only box and exports what include/tio_hip.h declares."""
from __future__ import annotations

import ctypes
import os
import re

import pytest

from torchio_amd import _abi
from torchio_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "tio_hip.h")


def declared_functions() -> list[str]:
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tio_[a-z0-9_]+)\s*\(", text)))


def test_header_and_ctypes_table_agree():
    assert declared_functions() == sorted(_abi.HIP_SYMBOLS)


def test_library_loads_and_exports_every_declared_symbol():
    lib, functions = _lib.load()
    for name in declared_functions():
        assert hasattr(lib, name), f"libtio_hip.so does not export {name}"
    assert functions["abi_version"]() == _abi.ABI_VERSION
    assert isinstance(functions["device_count"](), int)
    assert functions["last_error"]() is not None


def test_oracle_exports_the_same_compute_entry_points():
    from oracle.oracle import LIBRARY_PATH
    from oracle.oracle import oracle_engine

    oracle_engine()
    lib = ctypes.CDLL(LIBRARY_PATH)
    for name in _abi.PROTOTYPES:
        assert hasattr(lib, "tio_oracle_" + name)


def test_argument_validation_without_a_gpu():
    """Pure host-side argument checks return error codes (no kernel is launched)."""
    _, functions = _lib.load()
    assert functions["resample3d"](None, 1, None, None) == -1
    assert b"null" in functions["last_error"]()
    geom = _abi.ResampleGeom()
    image = (_abi.ResampleImage * 1)()
    assert functions["resample3d"](ctypes.byref(geom), 99, image, None) == -1
    assert b"n_images" in functions["last_error"]()
    assert functions["gamma_pow"](None, None, 0, 1, 1, 1.0, None, 0, None) == -1
    assert functions["add_noise"](None, None, 0, 1, 1, 0.0, 1.0, None, None, 0, 0, None, None, 0, None, None) == -1


def test_kspace_mix_table_direct_sum_equals_the_closed_form_and_is_a_partition_of_unity():
    """Host helper of the HIP library (no GPU): W_s summed directly vs the oracle's Dirichlet-kernel form."""
    import numpy as np

    from oracle.oracle import LIBRARY_PATH
    from oracle.oracle import oracle_engine

    oracle_engine()
    oracle = ctypes.CDLL(LIBRARY_PATH)
    _, functions = _lib.load()
    for length, bounds in ((12, [0, 4, 8, 12]), (13, [0, 3, 6, 9, 13]), (1, [0, 0, 1]), (64, [0, 64]), (50, [0, 7, 7, 50])):
        n = len(bounds) - 1
        ours = np.empty((n, length, length), dtype=np.float32)
        theirs = np.empty_like(ours)
        c_bounds = (ctypes.c_int32 * len(bounds))(*bounds)
        assert functions["kspace_mix_table"](length, n, c_bounds, ours.ctypes.data_as(ctypes.c_void_p)) == 0
        oracle.tio_oracle_kspace_mix_table.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.c_void_p]
        assert oracle.tio_oracle_kspace_mix_table(length, n, c_bounds, theirs.ctypes.data_as(ctypes.c_void_p)) == 0
        np.testing.assert_allclose(ours, theirs, rtol=0, atol=2e-7)
        np.testing.assert_allclose(ours.astype(np.float64).sum(axis=0), np.eye(length), rtol=0, atol=1e-6)  # the slabs cover k-space once
        # against numpy's FFT: ifft(mask * fft(e_i')).real is column i' of W_s
        for s in range(n):
            mask = np.zeros(length)
            mask[bounds[s] : bounds[s + 1]] = 1
            expected = np.fft.ifft(mask[:, None] * np.fft.fft(np.eye(length), axis=0), axis=0).real  # [i, i']
            np.testing.assert_allclose(ours[s].T, expected, rtol=0, atol=2e-7)
    assert functions["kspace_mix_table"](8, 2, (ctypes.c_int32 * 3)(0, 4, 7), ours.ctypes.data_as(ctypes.c_void_p)) == -1
    assert b"bounds" in functions["last_error"]()
    assert functions["kspace_segment_mix"](None, 1, None, None, None, 0, 1, 1, None, None, None) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_functions", None)
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIBRARY_PATH", str(tmp_path / "libtio_hip.so"))
    with pytest.raises(_lib.HipLibraryError, match="no CPU fallback"):
        _lib.load()


def test_engine_refuses_cpu_tensors():
    import torch

    from torchio_amd import ops

    engine = ops.Engine(_lib.load()[1], "cuda", "hip")
    with pytest.raises(ops.EngineError, match="runs on cuda tensors"):
        engine.gamma_pow(torch.rand(1, 1, 2, 2, 2), 1.5)
