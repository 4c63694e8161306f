"""ctypes description of the C ABI declared in ``include/tio_hip.h``.

Only types and prototypes live here — no library is loaded.  ``bind(lib, prefix)``
attaches ``argtypes``/``restype`` to every entry point of a loaded library whose
symbols start with ``prefix`` (``"tio_"`` for ``libtio_hip.so``).
"""
from __future__ import annotations

import ctypes as C

ABI_VERSION = 16
MAX_IMAGES = 8

# tio_status
OK = 0
UNSUPPORTED_CONFIG = -5  # fused form not available for these arguments; nothing was launched

# tio_dtype (values fixed by include/tio_hip.h)
F32, F64, F16, BF16, U8, I8, I16, I32, I64 = range(9)
# tio_interp
NEAREST, LINEAR, LABEL_PV, LINEAR_ADJOINT, QUADRATIC, CUBIC = 0, 1, 2, 3, 4, 5
BSPLINE4, BSPLINE5, BSPLINE6, BSPLINE7 = 6, 7, 8, 9  # B-spline orders 4 - 7 ("fourth" ... "seventh")
# tio_pad_mode
PAD_CONSTANT, PAD_REFLECT, PAD_REPLICATE, PAD_CIRCULAR = 0, 1, 2, 3
# tio_precision
PRECISION_EXACT, PRECISION_FAST, PRECISION_TIGHT = 0, 1, 2
GEOM_LARGE_BOXES = 1  # tio_resample_geom.flags (ABI 14): SOME bricks' boxes exceed the staging tile
GEOM_MOSTLY_LARGE_BOXES = 2  # (ABI 15): MOST do


class ResampleGeom(C.Structure):
    """``tio_resample_geom``."""

    _fields_ = [
        ("batch", C.c_int32),
        ("in_shape", C.c_int32 * 3),
        ("out_shape", C.c_int32 * 3),
        ("affine_first", C.c_int32),
        ("mapping_dev", C.c_void_p),
        ("mapping_batched", C.c_int32),
        ("control_points_dev", C.c_void_p),
        ("cp_batched", C.c_int32),
        ("cp_shape", C.c_int32 * 3),
        ("cp_skip_dev", C.c_void_p),
        ("passthrough_dev", C.c_void_p),
        ("in_spacing", C.c_float * 3),
        ("out_spacing", C.c_float * 3),
        ("norm_shape", C.c_int32 * 3),
        ("precision", C.c_int32),
        ("plan_dev", C.c_void_p),
        ("plan_bytes", C.c_int64),
        ("flags", C.c_int32),
    ]


class ResampleImage(C.Structure):
    """``tio_resample_image``."""

    _fields_ = [
        ("in_", C.c_void_p),
        ("out", C.c_void_p),
        ("channels", C.c_int32),
        ("dtype", C.c_int32),
        ("interp", C.c_int32),
        ("fill_dev", C.c_void_p),
        ("labels_dev", C.c_void_p),
        ("n_labels", C.c_int32),
        ("pad_label", C.c_double),
        ("out_min_dev", C.c_void_p),
    ]


class PatchPlacement(C.Structure):
    """``tio_patch_placement`` (host memory)."""

    _fields_ = [("dst_ini", C.c_int32 * 3), ("src_ini", C.c_int32 * 3), ("extent", C.c_int32 * 3)]


MAX_PATCHES = 32
MAX_SEGMENTS = 32  # TIO_MAX_SEGMENTS
# tio_overlap_mode
OVERLAP_CROP, OVERLAP_AVERAGE, OVERLAP_HANN = 0, 1, 2

_I32x3 = C.POINTER(C.c_int32)

#: name -> (restype, argtypes); names are given without the library prefix.
PROTOTYPES = {
    "resample3d": (C.c_int, [C.POINTER(ResampleGeom), C.c_int32, C.POINTER(ResampleImage), C.c_void_p]),
    "channel_min": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "separable_conv3d": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, _I32x3,
         C.c_void_p, C.c_int32, C.c_int32, _I32x3, C.c_void_p, C.c_void_p],
    ),
    "separable_conv3d_adjoint": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, _I32x3, C.c_void_p, C.c_int32, C.c_int32, _I32x3,
         C.c_void_p, C.c_void_p],
    ),
    "bias_field_apply": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, _I32x3, C.c_void_p, _I32x3,
         C.c_int32, C.c_void_p, C.c_void_p],
    ),
    "add_noise": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_float, C.c_float,
         C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64,
         C.c_void_p, C.c_void_p],
    ),
    "philox_normal": (C.c_int, [C.c_void_p, C.c_int64, C.c_uint64, C.c_int32, C.c_void_p]),
    "gamma_pow": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_float, C.c_void_p,
         C.c_int32, C.c_void_p],
    ),
    "patch_accumulate": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, _I32x3, C.c_void_p, C.c_int32, _I32x3,
         C.POINTER(PatchPlacement), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "bspline_prefilter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, _I32x3, C.c_int32, C.c_void_p]),
    "interpolate3d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, _I32x3, _I32x3, C.c_int32, C.c_void_p]),
    "axis_gather_lerp": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, _I32x3, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
         C.c_void_p, C.c_void_p],
    ),
    "flip3d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, _I32x3, C.c_int32, C.c_void_p, C.c_void_p]),
    "pad3d": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, _I32x3, C.POINTER(C.c_int32), C.c_int32, C.c_double,
         C.c_void_p, C.c_void_p],
    ),
    "unique_labels": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "kspace_segment_mix": (
        C.c_int,
        [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_int32), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
         _I32x3, C.c_void_p, C.c_void_p],
    ),
    "kspace_mix_table": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_void_p]),
    "abi_version": (C.c_int, []),
}

#: entry points only the HIP library has (not the CPU restatement)
HOST_MT_STATE_BYTES = 2688

HIP_ONLY_PROTOTYPES = {
    "last_error": (C.c_char_p, []),
    "device_count": (C.c_int, []),
    "reload_env": (None, []),
    "resample3d_plan_bytes": (C.c_int64, [C.POINTER(ResampleGeom)]),
    "resample3d_plan": (C.c_int, [C.POINTER(ResampleGeom), C.c_void_p, C.c_int64, C.c_void_p]),
    "host_mt19937_seed": (C.c_int, [C.c_void_p, C.c_uint64]),
    "host_mt19937_randn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]),
    "host_mt19937_plan_words": (C.c_int64, [C.c_int64]),
    "host_mt19937_plan": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_int32]),
    "host_mt19937_plan_begin": (C.c_int64, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32]),
    "host_mt19937_plan_end": (C.c_int, [C.c_int64, C.POINTER(C.c_int64)]),
    "mt19937_randn_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "host_mt19937_plan_prefix": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "host_mt19937_segment_polynomials": (C.c_int, [C.c_int64, C.c_int32, C.c_void_p]),
    "mt19937_device_snapshots": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "mt19937_add_noise_device": (
        C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "blur_fused": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, _I32x3, C.c_void_p, C.c_int32, C.c_int32,
         _I32x3, C.c_void_p, _I32x3, C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int32, C.c_uint64,
         C.c_void_p, C.c_int32, C.c_void_p],
    ),
}

#: every symbol include/tio_hip.h declares for libtio_hip.so
HIP_SYMBOLS = tuple("tio_" + n for n in (*PROTOTYPES, *HIP_ONLY_PROTOTYPES))


def bind(lib: C.CDLL, prefix: str, extra: dict | None = None) -> dict:
    """Attach prototypes to ``lib`` and return ``{short_name: function}``.

    Raises ``AttributeError`` naming the first missing symbol.
    """
    table = {}
    protos = dict(PROTOTYPES)
    if extra:
        protos.update(extra)
    for name, (restype, argtypes) in protos.items():
        fn = getattr(lib, prefix + name)
        fn.restype = restype
        fn.argtypes = argtypes
        table[name] = fn
    return table
