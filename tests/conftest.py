"""Shared pytest configuration.

Markers
    gpu  — needs a real MI355X (run by the driver with ``-m gpu``); everything
           else must pass on a CPU-only box.
"""
from __future__ import annotations

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)


def _reparse_library_switches_on_environment_writes():
    """libtio_hip.so parses its TIO_* switches once per process (``tio_reload_env`` parses them again).  The A/B tests flip
    such switches between calls through ``os.environ`` / ``monkeypatch.setenv``; both end in ``os.putenv`` /
    ``os.unsetenv``, so those two are wrapped here: a write to a TIO_* variable reloads the switches of a loaded library."""
    original_put, original_unset = os.putenv, os.unsetenv

    def reload_if_ours(key) -> None:
        name = key.decode() if isinstance(key, bytes) else str(key)
        if not name.startswith("TIO_"):
            return
        from torchio_amd import _lib

        if _lib._functions is not None:  # (never loads the library: CPU-only boxes may not be able to)
            _lib._functions["reload_env"]()

    def putenv(key, value):
        original_put(key, value)
        reload_if_ours(key)

    def unsetenv(key):
        original_unset(key)
        reload_if_ours(key)

    os.putenv, os.unsetenv = putenv, unsetenv


_reparse_library_switches_on_environment_writes()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X GPU")
    config.addinivalue_line("markers", "reference: test imports the reference from /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    """A plain ``pytest tests`` on a CPU-only box skips the gpu-marked tests instead of erroring in them."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (run with -m gpu on the GPU box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle engine (test infrastructure)."""
    from oracle.oracle import oracle_engine

    return oracle_engine()


@pytest.fixture(scope="session")
def hip():
    """The HIP engine; GPU tests fail loudly if it cannot be loaded."""
    import torch

    from torchio_amd import ops

    if not torch.cuda.is_available():
        pytest.skip("needs a real MI355X (run with -m gpu on the GPU box)")
    return ops.engine()
