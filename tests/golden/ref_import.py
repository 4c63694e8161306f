"""Import the UNMODIFIED reference TorchIO from /root/reference with stub modules.

Test/fixture infrastructure only.  The reference needs a few
non-hot-path third-party modules that are not installed in this image
(SimpleITK, nibabel, jaxtyping, loguru, humanize, tyro); none of them is used
by the augmentation hot path, so empty stubs are enough (one exception: ``Flip`` with anatomical
labels asks nibabel for the axis codes; the stub restates that published algorithm).  Nothing under
``torchio_amd/`` imports this file, and nothing here is available on the GPU
box (``/root/reference`` does not travel).
"""
from __future__ import annotations

import importlib.metadata
import os
import sys
import types

REFERENCE_SRC = "/root/reference/src"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_SRC, "torchio"))


def _stub(name: str, **attrs):
    mod = types.ModuleType(name)
    for key, value in attrs.items():
        setattr(mod, key, value)
    sys.modules[name] = mod
    return mod


class _Subscriptable:
    def __class_getitem__(cls, item):
        return cls


class _Logger:
    def __getattr__(self, name):
        return lambda *a, **k: None


def _aff2axcodes(affine, labels=(("L", "R"), ("P", "A"), ("I", "S")), tol=None):
    """nibabel 5.x ``orientations.aff2axcodes`` restated (``io_orientation`` + ``ornt2axcodes``): the closest
    world axis of every voxel axis after removing the zooms and projecting onto the nearest orthogonal matrix."""
    import numpy as np  # noqa: PLC0415

    rzs = np.asarray(affine, dtype=np.float64)[:3, :3]
    zooms = np.sqrt((rzs * rzs).sum(axis=0))
    zooms[zooms == 0] = 1
    rs = rzs / zooms
    p, s, qs = np.linalg.svd(rs, full_matrices=False)
    if tol is None:
        tol = s.max() * 3 * np.finfo(s.dtype).eps
    keep = s > tol
    r = p[:, keep] @ qs[keep]
    codes = []
    for in_ax in range(3):
        col = r[:, in_ax]
        if np.allclose(col, 0):
            codes.append(None)
            continue
        out_ax = int(np.argmax(np.abs(col)))
        codes.append(labels[out_ax][0 if col[out_ax] < 0 else 1])
        r[out_ax, :] = 0
    return tuple(codes)


def import_reference():
    """Return the reference ``torchio`` module (imported once, cached)."""
    if "torchio" in sys.modules and getattr(sys.modules["torchio"], "_tio_ref", False):
        return sys.modules["torchio"]
    if not reference_available():
        raise ImportError("reference tree not present")
    sys.dont_write_bytecode = True  # never write __pycache__ into the read-only tree
    if "SimpleITK" not in sys.modules:
        _stub("SimpleITK", Image=type("Image", (), {}), ImageFileReader=type("ImageFileReader", (), {}))
    if "nibabel" not in sys.modules:
        nib = _stub("nibabel", Nifti1Image=type("Nifti1Image", (), {}))
        nib.spatialimages = _stub("nibabel.spatialimages", SpatialImage=type("SpatialImage", (), {}))
        nib.orientations = _stub("nibabel.orientations", aff2axcodes=_aff2axcodes)
    if "jaxtyping" not in sys.modules:
        names = ["Float", "Int", "Bool", "Shaped", "Num", "UInt8", "Integer", "Real", "Inexact", "Array"]
        _stub("jaxtyping", **{n: type(n, (_Subscriptable,), {}) for n in names})
    if "loguru" not in sys.modules:
        _stub("loguru", logger=_Logger())
    if "humanize" not in sys.modules:
        _stub("humanize", naturalsize=lambda n, *a, **k: f"{n} B")
    if "tyro" not in sys.modules:
        tyro = _stub("tyro")
        tyro.conf = _stub("tyro.conf")
        tyro.extras = _stub("tyro.extras")
    real_version = importlib.metadata.version

    def _version(name):
        if name == "torchio":
            return "2.0.0a2"
        return real_version(name)

    importlib.metadata.version = _version
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    import torchio  # noqa: PLC0415

    torchio._tio_ref = True
    return torchio
