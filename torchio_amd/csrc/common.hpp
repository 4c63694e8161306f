// common.hpp — shared device/host helpers for libtio_hip.so (gfx950 only).
//
// Numerics contract: this library is compiled with -ffp-contract=off.  Wherever
// the reference's arithmetic fuses a multiply-add (MKL sgemm, ATen's AVX lerp)
// the code calls __builtin_fmaf explicitly; everywhere else a*b+c rounds twice,
// exactly like the ATen scalar kernels it reproduces.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tio_hip.h"

namespace tio {

// ---- error reporting (thread-local text, see tio_last_error) -----------------
void set_error(const char* fmt, ...);
int fail(int status, const char* fmt, ...);
int check_launch(const char* what);

// ---- environment switches (A/B experiments, tests): read ONCE per process -----
// Every TIO_* switch of the library, parsed at the first use and again whenever tio_reload_env() is called
// (tests and A/B tools change the environment between calls; the hot entry points never call getenv —
// 13 calls per tio_resample3d until round 4, none of them safe against a concurrent setenv).
struct EnvSwitches {
  int nearest_kernel = 1;        // TIO_NEAREST_KERNEL (0: nearest images stay with the other images)
  int has_nearest_eps = 0;       // TIO_NEAREST_EPS (calibration runs only)
  float nearest_eps = 0.0f;
  int resample_path = 0;         // TIO_RESAMPLE_PATH: 0 unset, 1 gather, 2 tile
  int tile_variant = 0;          // TIO_TILE_VARIANT
  int tile_lds_floats = 0;       // TIO_TILE_LDS_FLOATS
  int tile_ablate = 0;           // TIO_TILE_ABLATE (instrumented instantiations only)
  int resample_exact = 0;        // TIO_RESAMPLE_EXACT set: never the FAST kernels, and TIGHT launches interpolate in ATen's order (bit-exact)
  int fast_kernel = 0;           // TIO_FAST_KERNEL: 0 unset, 1 brick, 2 planned
  int planned_lean = 1;          // TIO_PLANNED_LEAN
  int dma_packed = 1;            // TIO_DMA_PACKED
  int exact_plan = -1;           // TIO_EXACT_PLAN: -1 unset, else its value
  int exact_lean = -1;           // TIO_EXACT_LEAN: -1 unset (large exact float32 launches take resample_lean_exact_kernel), 0 never, 2 small launches too
  int nearest_exact = 1;         // TIO_NEAREST_EXACT=0: label maps without a fill rule keep the FAST-line kernel of rounds 3 - 5 (A/B)
  int nearest_lds = -1;          // TIO_NEAREST_LDS: bytes of (unused) dynamic LDS per block of resample_nearest_exact_kernel — an occupancy knob (A/B; -1: the launch's own choice)
  int lean_pair = 1;             // TIO_LEAN_PAIR=0: one exact-coordinate launch per channel (until round 6; A/B)
  int lean_label = 1;            // TIO_LEAN_LABEL=0: a call's label channel never rides along the images' exact-coordinate launch (until round 6; A/B)
  int lean_multi = 1;            // TIO_LEAN_MULTI=0: no multi-pass bricks (boxes beyond the tile sample voxel by voxel, as until round 5: A/B)
  int lean_interleave = 1;       // TIO_LEAN_INTERLEAVE (0: the exact-coordinate kernel issues its whole box before phase A, A/B)
  int fast_fill_recheck = 1;     // TIO_FAST_FILL_RECHECK (0: the FAST fill rule decides alone, A/B)
  int conv_no_fuse = 0;          // TIO_CONV_NO_FUSE set
  int conv_ring = 0;             // TIO_CONV_RING set
  int march_segs = 0;            // TIO_MARCH_SEGS (0 unset)
  int march_order = -1;          // TIO_MARCH_ORDER (-1 unset)
  int min_blocks = 0;            // TIO_MIN_BLOCKS (0 unset)
};
const EnvSwitches& env_switches();

// ---- element type conversion: `.float()` on load, `.to(dtype)` on store -------
__device__ __forceinline__ float bf16_bits_to_float(uint16_t h) {
  return __uint_as_float(static_cast<uint32_t>(h) << 16);
}

__device__ __forceinline__ uint16_t float_to_bf16_bits(float f) {
  uint32_t x = __float_as_uint(f);
  if ((x & 0x7FFFFFFFu) > 0x7F800000u) return 0x7FC0u;  // NaN
  x += 0x7FFFu + ((x >> 16) & 1u);                       // round to nearest even
  return static_cast<uint16_t>(x >> 16);
}

template <int DT>
struct Elem;
template <>
struct Elem<TIO_F32> {
  using type = float;
  static __device__ __forceinline__ float load(const void* p, int64_t i) { return static_cast<const float*>(p)[i]; }
  static __device__ __forceinline__ void store(void* p, int64_t i, float v) { static_cast<float*>(p)[i] = v; }
};
template <>
struct Elem<TIO_F64> {
  using type = double;
  static __device__ __forceinline__ float load(const void* p, int64_t i) {
    return static_cast<float>(static_cast<const double*>(p)[i]);
  }
  static __device__ __forceinline__ void store(void* p, int64_t i, float v) {
    static_cast<double*>(p)[i] = static_cast<double>(v);
  }
};
template <>
struct Elem<TIO_F16> {
  using type = _Float16;
  static __device__ __forceinline__ float load(const void* p, int64_t i) {
    return static_cast<float>(static_cast<const _Float16*>(p)[i]);
  }
  static __device__ __forceinline__ void store(void* p, int64_t i, float v) {
    // The value is a finished float32 result that `.to(float16)` rounds a SECOND time.  Without the
    // barrier the backend folds a preceding fma into v_fma_mixlo_f16 (one rounding from the exact
    // product-sum), which differs from the reference in near-tie cases.
    asm volatile("" : "+v"(v));
    static_cast<_Float16*>(p)[i] = static_cast<_Float16>(v);  // v_cvt_f16_f32: RNE
  }
};
template <>
struct Elem<TIO_BF16> {
  using type = uint16_t;
  static __device__ __forceinline__ float load(const void* p, int64_t i) {
    return bf16_bits_to_float(static_cast<const uint16_t*>(p)[i]);
  }
  static __device__ __forceinline__ void store(void* p, int64_t i, float v) {
    static_cast<uint16_t*>(p)[i] = float_to_bf16_bits(v);
  }
};
#define TIO_INT_ELEM(CODE, T)                                                                  \
  template <>                                                                                  \
  struct Elem<CODE> {                                                                          \
    using type = T;                                                                            \
    static __device__ __forceinline__ float load(const void* p, int64_t i) {                   \
      return static_cast<float>(static_cast<const T*>(p)[i]);                                  \
    }                                                                                          \
    static __device__ __forceinline__ void store(void* p, int64_t i, float v) {                \
      static_cast<T*>(p)[i] = static_cast<T>(static_cast<int64_t>(v)); /* trunc toward 0 */    \
    }                                                                                          \
  };
TIO_INT_ELEM(TIO_U8, uint8_t)
TIO_INT_ELEM(TIO_I8, int8_t)
TIO_INT_ELEM(TIO_I16, int16_t)
TIO_INT_ELEM(TIO_I32, int32_t)
TIO_INT_ELEM(TIO_I64, int64_t)
#undef TIO_INT_ELEM

// Run-time dtype (uniform across a launch → scalar branch).
__device__ __forceinline__ float load_as_float(const void* p, int dtype, int64_t i) {
  switch (dtype) {
    case TIO_F32: return Elem<TIO_F32>::load(p, i);
    case TIO_F64: return Elem<TIO_F64>::load(p, i);
    case TIO_F16: return Elem<TIO_F16>::load(p, i);
    case TIO_BF16: return Elem<TIO_BF16>::load(p, i);
    case TIO_U8: return Elem<TIO_U8>::load(p, i);
    case TIO_I8: return Elem<TIO_I8>::load(p, i);
    case TIO_I16: return Elem<TIO_I16>::load(p, i);
    case TIO_I32: return Elem<TIO_I32>::load(p, i);
    default: return Elem<TIO_I64>::load(p, i);
  }
}

__device__ __forceinline__ void store_from_float(void* p, int dtype, int64_t i, float v) {
  switch (dtype) {
    case TIO_F32: Elem<TIO_F32>::store(p, i, v); break;
    case TIO_F64: Elem<TIO_F64>::store(p, i, v); break;
    case TIO_F16: Elem<TIO_F16>::store(p, i, v); break;
    case TIO_BF16: Elem<TIO_BF16>::store(p, i, v); break;
    case TIO_U8: Elem<TIO_U8>::store(p, i, v); break;
    case TIO_I8: Elem<TIO_I8>::store(p, i, v); break;
    case TIO_I16: Elem<TIO_I16>::store(p, i, v); break;
    case TIO_I32: Elem<TIO_I32>::store(p, i, v); break;
    default: Elem<TIO_I64>::store(p, i, v); break;
  }
}

inline __host__ __device__ int dtype_size(int dtype) {
  switch (dtype) {
    case TIO_F32: case TIO_I32: return 4;
    case TIO_F64: case TIO_I64: return 8;
    case TIO_F16: case TIO_BF16: case TIO_I16: return 2;
    case TIO_U8: case TIO_I8: return 1;
    default: return 0;
  }
}

inline bool is_float_dtype(int dtype) {
  return dtype == TIO_F32 || dtype == TIO_F64 || dtype == TIO_F16 || dtype == TIO_BF16;
}

// ---- ATen upsample_linear (align_corners=True) source index / lambdas ---------
struct Lerp1D {
  int i0, i1;
  float l0, l1;
};

__device__ __forceinline__ Lerp1D lerp_index(int o, int n_in, int n_out, float scale) {
  Lerp1D r;
  if (n_out == n_in) {  // scale factor 1: plain copy
    r.i0 = o; r.i1 = o; r.l0 = 1.0f; r.l1 = 0.0f;
    return r;
  }
  const float real = scale * static_cast<float>(o);
  int i0 = static_cast<int>(floorf(real));
  i0 = min(i0, n_in - 1);
  float l1 = real - static_cast<float>(i0);
  l1 = fminf(fmaxf(l1, 0.0f), 1.0f);
  r.i0 = i0;
  r.i1 = i0 + ((i0 < n_in - 1) ? 1 : 0);
  r.l1 = l1;
  r.l0 = 1.0f - l1;
  return r;
}

// area_pixel_compute_scale<float>(n_in, n_out, align_corners=true)
inline __host__ __device__ float lerp_scale(int n_in, int n_out) {
  return (n_out > 1) ? static_cast<float>(n_in - 1) / static_cast<float>(n_out - 1) : 0.0f;
}

// ATen Interpolate<n>::eval (`out = t0*w0; out += t1*w1`) as the AVX dispatch build
// contracts it: fma(t0, w0, t1*w1) — pinned bit-for-bit against F.interpolate.
__device__ __forceinline__ float lerp2(float t0, float w0, float t1, float w1) {
  return __builtin_fmaf(t0, w0, __fmul_rn(t1, w1));
}

// Bijective XCD-aware block remap (guide §5.5 T1): the hardware dispatches
// block b to XCD b % 8; give every XCD one contiguous chunk of the tile space so
// neighbouring tiles share that XCD's private L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblocks) {
  constexpr unsigned NX = 8;
  const unsigned q = nblocks / NX, r = nblocks % NX;
  const unsigned xcd = bid % NX, slot = bid / NX;
  const unsigned start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + slot;
}


// Per-channel minimum through ordered-integer keys (intensity.hip: tio_channel_min; resample_fast.hpp: the minimum of a
// resampler's own output, folded into its stores).  NaN maps to key 0 so it wins, matching torch.min's NaN propagation.
__device__ __forceinline__ uint32_t float_to_key(float f) {
  if (f != f) return 0u;
  const uint32_t bits = __float_as_uint(f);
  return (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(uint32_t key) {
  if (key == 0u) return __uint_as_float(0x7FC00000u);
  const uint32_t bits = (key & 0x80000000u) ? (key & 0x7FFFFFFFu) : ~key;
  return __uint_as_float(bits);
}
// `*cap` keys (all ones) followed by `*cap` tickets (zero), *cap >= entries, one pair of arrays per (device, stream, kind):
// set up once, every user restores what it touched before it ends.  kind 0 = tio_channel_min (ONE self-contained launch
// per call, so calls that share a stream are ordered by it), kind 1 = the folded minimum of a planned resampling launch
// (two kernels, enqueued under the plan lease of resample.hip).  A grown array never frees its predecessor: another host
// thread may have been handed that pointer and not launched with it yet (a few KiB per growth, a handful per process).
uint32_t* min_workspace(hipStream_t s, int entries, int* cap, int kind = 0);

}  // namespace tio
