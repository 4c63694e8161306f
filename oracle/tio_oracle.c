/*
 * tio_oracle.c — CPU restatement of TorchIO's augmentation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under torchio_amd/ links, imports or calls
 * this file; it is used by tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py as the checker / baseline.
 *
 * Parity status: PINNED — except the B-spline functions (spline_*, tio_oracle_bspline_prefilter and the TIO_QUADRATIC /
 * TIO_CUBIC branch of tio_oracle_resample3d): the reference delegates those orders to torch-interpol, which is neither
 * vendored under /root/reference nor installed here; they restate its published algorithm and are pinned against
 * scipy.ndimage instead (tests/test_bspline.py).  THEIR PARITY WITH THE REFERENCE IS UNPINNED.
 * Every other function below is checked bit-for-bit (integer /
 * nearest results) or to 1e-5 (float results) against outputs of the unmodified
 * reference (TorchIO 2.0.0a2 on torch 2.10.0 CPU kernels, run in the build
 * container by tests/golden/make_golden.py); the vectors live in tests/golden/.
 *
 * The arithmetic the reference delegates to PyTorch ATen (not vendored under
 * /root/reference) is restated here from its published source semantics and was
 * pinned empirically (tests/golden/README.md):
 *   - mm of (N,4)x(4,4) float32 (MKL sgemm)       == forward FMA chain over k
 *   - upsample_trilinear3d(align_corners=True)    == nested lerp, W innermost
 *   - grid_sampler_3d(bilinear|nearest, zeros, align_corners=True)
 *   - replication_pad3d + conv3d (cross-correlation)
 *
 * Not the reference's: the Philox normal stream (tio_oracle_philox_normal, the philox road of tio_oracle_add_noise and
 * tio_oracle_blur_fused) — the engine's own throughput-mode generator, defined HERE and in csrc/intensity.hip by the same
 * arithmetic (Philox4x32-10, Box-Muller with sin / cos of 2 pi u as an IEEE float32 polynomial, round 5) — and
 * geom.precision: this file always computes the reference's sequence (TIO_PRECISION_EXACT); the TIGHT and FAST modes of the
 * engine are held to it within their stated tolerances (tests/test_gpu_tight.py, tests/test_gpu_full_size.py).
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -mfma -ffp-contract=off).
 * Floating-point contraction MUST stay off: where the reference fuses
 * (BLAS FMA) this file calls fmaf() explicitly, everywhere else products and
 * sums round separately exactly like the ATen scalar code.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/tio_hip.h"

#if defined(_OPENMP)
#include <omp.h>
#endif

/* ------------------------------------------------------------------------ */
/* dtype helpers: `.float()` on load, `.to(dtype)` on store                   */
/* ------------------------------------------------------------------------ */
static inline float half_to_float(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1Fu;
  uint32_t man = h & 0x3FFu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else { /* subnormal */
      int e = -1;
      do {
        man <<= 1;
        e++;
      } while (!(man & 0x400u));
      man &= 0x3FFu;
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
    }
  } else if (exp == 31) {
    bits = sign | 0x7F800000u | (man << 13);
  } else {
    bits = sign | ((exp + 112u) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

static inline uint16_t float_to_half(float f) { /* round-to-nearest-even */
  uint32_t x;
  memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t ax = x & 0x7FFFFFFFu;
  if (ax >= 0x7F800000u) return (uint16_t)(sign | (ax > 0x7F800000u ? 0x7E00u : 0x7C00u));
  if (ax >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u); /* overflow → inf */
  if (ax < 0x33000001u) return (uint16_t)sign;               /* underflow → 0 */
  int32_t exp = (int32_t)(ax >> 23) - 127;
  uint32_t man = (ax & 0x7FFFFFu) | 0x800000u;
  int shift;
  uint32_t base;
  if (exp < -14) { /* subnormal half */
    shift = 13 + (-14 - exp);
    base = 0;
  } else {
    shift = 13;
    base = (uint32_t)(exp + 15) << 10;
    man &= 0x7FFFFFu;
  }
  uint32_t q = man >> shift;
  uint32_t rem = man & ((1u << shift) - 1u);
  uint32_t halfway = 1u << (shift - 1);
  if (rem > halfway || (rem == halfway && (q & 1u))) q++;
  return (uint16_t)(sign | (base + q));
}

static inline float bf16_to_float(uint16_t h) {
  uint32_t bits = (uint32_t)h << 16;
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

static inline uint16_t float_to_bf16(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  if ((x & 0x7FFFFFFFu) > 0x7F800000u) return 0x7FC0u;
  uint32_t lsb = (x >> 16) & 1u;
  x += 0x7FFFu + lsb;
  return (uint16_t)(x >> 16);
}

static inline float load_as_float(const void* p, int dtype, int64_t i) {
  switch (dtype) {
    case TIO_F32: return ((const float*)p)[i];
    case TIO_F64: return (float)((const double*)p)[i];
    case TIO_F16: return half_to_float(((const uint16_t*)p)[i]);
    case TIO_BF16: return bf16_to_float(((const uint16_t*)p)[i]);
    case TIO_U8: return (float)((const uint8_t*)p)[i];
    case TIO_I8: return (float)((const int8_t*)p)[i];
    case TIO_I16: return (float)((const int16_t*)p)[i];
    case TIO_I32: return (float)((const int32_t*)p)[i];
    case TIO_I64: return (float)((const int64_t*)p)[i];
    default: return 0.0f;
  }
}

static inline void store_from_float(void* p, int dtype, int64_t i, float v) {
  switch (dtype) {
    case TIO_F32: ((float*)p)[i] = v; break;
    case TIO_F64: ((double*)p)[i] = (double)v; break;
    case TIO_F16: ((uint16_t*)p)[i] = float_to_half(v); break;
    case TIO_BF16: ((uint16_t*)p)[i] = float_to_bf16(v); break;
    case TIO_U8: ((uint8_t*)p)[i] = (uint8_t)(int64_t)v; break; /* trunc toward zero */
    case TIO_I8: ((int8_t*)p)[i] = (int8_t)(int64_t)v; break;
    case TIO_I16: ((int16_t*)p)[i] = (int16_t)(int64_t)v; break;
    case TIO_I32: ((int32_t*)p)[i] = (int32_t)(int64_t)v; break;
    case TIO_I64: ((int64_t*)p)[i] = (int64_t)v; break;
    default: break;
  }
}

static inline double load_as_double(const void* p, int dtype, int64_t i) {
  switch (dtype) {
    case TIO_F64: return ((const double*)p)[i];
    case TIO_I32: return (double)((const int32_t*)p)[i];
    case TIO_I64: return (double)((const int64_t*)p)[i];
    default: return (double)load_as_float(p, dtype, i); /* exact: these types embed in float32 */
  }
}

static inline void store_from_double(void* p, int dtype, int64_t i, double v) {
  switch (dtype) {
    case TIO_F64: ((double*)p)[i] = v; break;
    case TIO_I32: ((int32_t*)p)[i] = (int32_t)(int64_t)v; break;
    case TIO_I64: ((int64_t*)p)[i] = (int64_t)v; break;
    default: store_from_float(p, dtype, i, (float)v); break;
  }
}

static inline size_t dtype_size(int dtype) {
  switch (dtype) {
    case TIO_F32: case TIO_I32: return 4;
    case TIO_F64: case TIO_I64: return 8;
    case TIO_F16: case TIO_BF16: case TIO_I16: return 2;
    case TIO_U8: case TIO_I8: return 1;
    default: return 0;
  }
}

/* ------------------------------------------------------------------------ */
/* ATen upsample_{tri}linear (align_corners=True) index / lambda             */
/*   used by spatial.py:2182-2187 and bias_field.py:333-338                   */
/* ------------------------------------------------------------------------ */
typedef struct {
  int32_t i0, i1;
  float l0, l1;
} lerp1d;

static inline lerp1d lerp_index(int32_t o, int32_t n_in, int32_t n_out) {
  lerp1d r;
  if (n_out == n_in) { /* scale factor 1: plain copy */
    r.i0 = o;
    r.i1 = o;
    r.l0 = 1.0f;
    r.l1 = 0.0f;
    return r;
  }
  float scale = (n_out > 1) ? (float)(n_in - 1) / (float)(n_out - 1) : 0.0f;
  float real = scale * (float)o;
  int32_t i0 = (int32_t)floorf(real);
  if (i0 > n_in - 1) i0 = n_in - 1;
  float l1 = real - (float)i0;
  if (l1 < 0.0f) l1 = 0.0f;
  if (l1 > 1.0f) l1 = 1.0f;
  r.i0 = i0;
  r.i1 = i0 + ((i0 < n_in - 1) ? 1 : 0);
  r.l1 = l1;
  r.l0 = 1.0f - l1;
  return r;
}

/* Nested linear interpolation exactly as ATen's Interpolate<n>::eval:
 * innermost = last (K / W) axis, `output = t0*w0; output += t1*w1`.
 * The dispatch build (AVX2/AVX512 flags, -ffp-contract=fast) contracts this to
 * fma(t0, w0, t1*w1) — pinned bit-for-bit against F.interpolate (the other
 * three candidates: separate roundings, fma(t1,w1,t0*w0), differ in ~25 % of
 * the voxels by 1 ulp). */
static inline float lerp2(float t0, float w0, float t1, float w1) {
  return fmaf(t0, w0, t1 * w1);
}

static inline float trilerp(const float* v, int64_t s_i, int64_t s_j, int64_t s_k,
                            lerp1d li, lerp1d lj, lerp1d lk) {
  const float* p00 = v + li.i0 * s_i + lj.i0 * s_j;
  const float* p01 = v + li.i0 * s_i + lj.i1 * s_j;
  const float* p10 = v + li.i1 * s_i + lj.i0 * s_j;
  const float* p11 = v + li.i1 * s_i + lj.i1 * s_j;
  float a00 = lerp2(p00[lk.i0 * s_k], lk.l0, p00[lk.i1 * s_k], lk.l1);
  float a01 = lerp2(p01[lk.i0 * s_k], lk.l0, p01[lk.i1 * s_k], lk.l1);
  float a10 = lerp2(p10[lk.i0 * s_k], lk.l0, p10[lk.i1 * s_k], lk.l1);
  float a11 = lerp2(p11[lk.i0 * s_k], lk.l0, p11[lk.i1 * s_k], lk.l1);
  float b0 = lerp2(a00, lj.l0, a01, lj.l1);
  float b1 = lerp2(a10, lj.l0, a11, lj.l1);
  return lerp2(b0, li.l0, b1, li.l1);
}

/* ------------------------------------------------------------------------ */
/* Coordinates: spatial.py:1504-1648 + ATen un-normalise                     */
/* ------------------------------------------------------------------------ */

/* [c, 1] @ M^T for one row of M — MKL sgemm rounding: forward FMA chain. */
static inline float affine_row(const float* m, float a, float b, float c) {
  float t = a * m[0];
  t = fmaf(b, m[1], t);
  t = fmaf(c, m[2], t);
  t = fmaf(1.0f, m[3], t);
  return t;
}

/* g = 2 v / max(S-1,1) - 1  (spatial.py:1638-1646), then ATen
 * grid_sampler_unnormalize(align_corners=True): ((g + 1) / 2) * (S - 1). */
static inline float normalise_roundtrip(float v, int32_t norm_size, int32_t size) {
  float denom = (float)((norm_size - 1 > 1) ? norm_size - 1 : 1); /* the grid's shape: the first image's */
  float g = 2.0f * v / denom - 1.0f;
  return ((g + 1.0f) / 2.0f) * (float)(size - 1); /* grid_sample un-normalises with the sampled image's own size */
}

static inline int in_bounds(float f, int32_t n) { return f >= 0.0f && f <= (float)(n - 1); }

/* ------------------------------------------------------------------------ */
/* "label" partial-volume mode, literally: spatial.py:1275-1389 (C == 1)      */
/* ------------------------------------------------------------------------ */
static int compare_doubles(const void* a, const void* b) {
  double x = *(const double*)a, y = *(const double*)b;
  return (x > y) - (x < y);
}

/* torch.unique(data): sorted distinct values of the whole tensor; caller frees */
static double* unique_labels(const void* data, int dtype, int64_t n, int32_t* count) {
  double* values = (double*)malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
  for (int64_t i = 0; i < n; i++) values[i] = load_as_double(data, dtype, i);
  qsort(values, (size_t)n, sizeof(double), compare_doubles);
  int64_t m = 0;
  for (int64_t i = 0; i < n; i++)
    if (m == 0 || values[i] != values[m - 1]) values[m++] = values[i];
  *count = (int32_t)m;
  return values;
}

/* ATen sum over a strided dimension: multi_row_sum (cpu/SumKernel.cpp, torch 2.10),
 * one row.  Four accumulators; level 0 is dumped upwards after every level_step terms. */
static float cascade_sum(const float* x, int64_t size) {
  int ceil_log2 = 1;
  if (size > 2) {
    ceil_log2 = 0;
    while (((int64_t)1 << ceil_log2) < size) ceil_log2++;
  }
  const int64_t level_power = (ceil_log2 / 4 > 4) ? ceil_log2 / 4 : 4;
  const int64_t level_step = (int64_t)1 << level_power;
  const int64_t level_mask = level_step - 1;
  float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  int64_t i = 0;
  for (; i + level_step <= size;) {
    for (int64_t j = 0; j < level_step; ++j, ++i) acc[0] += x[i];
    for (int j = 1; j < 4; ++j) {
      acc[j] += acc[j - 1];
      acc[j - 1] = 0.0f;
      const int64_t mask = level_mask << (j * level_power);
      if ((i & mask) != 0) break;
    }
  }
  for (; i < size; ++i) acc[0] += x[i];
  for (int j = 1; j < 4; ++j) acc[0] += acc[j];
  return acc[0];
}

/* exported for the unit test that pins cascade_sum against torch.sum */
float tio_oracle_cascade_sum(const float* x, int64_t size) { return cascade_sum(x, size); }

/* one voxel: one-hot channel vector -> grid_sample(linear, zeros) -> argmax / sum > 0.5 */
static void label_pv_voxel(const tio_resample_image* img, const double* labels, int32_t n_labels, float* channels,
                           int64_t base_in, int64_t out_index, const float w[8], const int ok[8],
                           const int64_t off[8]) {
  for (int32_t l = 0; l < n_labels; l++) channels[l] = 0.0f;
  for (int t = 0; t < 8; t++) {
    if (!ok[t]) continue; /* out-of-bounds taps are skipped by grid_sampler_3d */
    const double value = load_as_double(img->in, img->dtype, base_in + off[t]);
    for (int32_t l = 0; l < n_labels; l++) {
      const float one_hot = (value == labels[l]) ? 1.0f : 0.0f; /* (values == targets).float() */
      channels[l] += one_hot * w[t];
    }
  }
  int32_t winner = 0; /* argmax: first maximum */
  for (int32_t l = 1; l < n_labels; l++)
    if (channels[l] > channels[winner]) winner = l;
  const int in_bounds = cascade_sum(channels, n_labels) > 0.5f;
  store_from_double(img->out, img->dtype, out_index, in_bounds ? labels[winner] : img->pad_label);
}

/* ---- B-spline orders 2 / 3 (torch-interpol's published algorithm; see include/tio_hip.h: TIO_QUADRATIC) ---------
 * Restated from the package's documented behaviour, NOT from its source (it is not vendored by the reference and not
 * installed here): basis weights of Unser's B-splines, `dct2` (half-sample symmetric) index reflection, the
 * extrapolate=False mask with its 0.05-voxel tolerance.  Pinned against scipy.ndimage.map_coordinates /
 * spline_filter(mode="reflect") in tests/test_bspline.py.  PARITY WITH THE REFERENCE ITSELF IS UNPINNED. */
static inline int spline_reflect(int i, int n) { /* dct2: ... c b a | a b c ... */
  const int n2 = 2 * n;
  if (i < 0) i = -i - 1;
  i %= n2;
  return i >= n ? n2 - i - 1 : i;
}

static inline void spline_weights(float x, int order, int* low, float w[4]) {
  if (order == 2) {
    const float c = floorf(x + 0.5f); /* nearest node */
    const float t = x - c;            /* [-0.5, 0.5] */
    *low = (int)c - 1;
    w[0] = 0.5f * (0.5f - t) * (0.5f - t);
    w[1] = 0.75f - t * t;
    w[2] = 0.5f * (0.5f + t) * (0.5f + t);
    w[3] = 0.0f;
  } else {
    const float f = floorf(x);
    const float t = x - f; /* [0, 1) */
    const float u = 1.0f - t;
    *low = (int)f - 1;
    w[0] = u * u * u / 6.0f;
    w[1] = (t * t * (t - 2.0f) * 3.0f + 4.0f) / 6.0f;
    w[2] = (u * u * (u - 2.0f) * 3.0f + 4.0f) / 6.0f;
    w[3] = t * t * t / 6.0f;
  }
}

/* Orders 4 - 7: the order + 1 basis weights at x from the Cox - de Boor recursion of the uniform B-spline,
 *   N_1(s) = [0 <= s < 1],   N_m(s) = (s N_{m-1}(s) + (m - s) N_{m-1}(s - 1)) / (m - 1),
 * evaluated for the values v_j = N_m(tau + j) that are not zero (tau in [0, 1): the position inside the knot interval),
 * in float64 (every term is positive, nothing cancels), rounded to float32 at the end.  Tap k sits at low + k and weighs
 * v_{order - k}.  Odd orders: knots at the integers (tau = x - floor(x)); even orders: at the half-integers. */
static inline void spline_weights_high(float x, int order, int* low, float w[8]) {
  const int odd = order & 1;
  const float base = odd ? floorf(x) : floorf(x + 0.5f);
  const double tau = odd ? (double)x - (double)base : ((double)x - (double)base) + 0.5;
  *low = (int)base - (odd ? (order - 1) / 2 : order / 2);
  double v[8] = {1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int m = 2; m <= order + 1; m++) {        /* v^(m)_j from v^(m-1)_j and v^(m-1)_{j-1}, highest j first (in place) */
    const double inv = 1.0 / (double)(m - 1);
    for (int j = m - 1; j >= 0; j--) {
      const double same = j <= m - 2 ? v[j] : 0.0, below = j >= 1 ? v[j - 1] : 0.0;
      v[j] = ((tau + (double)j) * same + ((double)m - tau - (double)j) * below) * inv;
    }
  }
  for (int k = 0; k <= order; k++) w[k] = (float)v[order - k];
  for (int k = order + 1; k < 8; k++) w[k] = 0.0f;
}

static float spline_sample(const float* coef, int32_t I, int32_t J, int32_t K, float vi, float vj, float vk, int order) {
  const float tiny = 5e-2f;
  if (!(vi > -tiny && vi < (float)(I - 1) + tiny && vj > -tiny && vj < (float)(J - 1) + tiny && vk > -tiny && vk < (float)(K - 1) + tiny))
    return 0.0f; /* extrapolate=False (NaN coordinates land here too) */
  int li, lj, lk;
  float wi[8], wj[8], wk[8];
  if (order <= 3) {
    spline_weights(vi, order, &li, wi);
    spline_weights(vj, order, &lj, wj);
    spline_weights(vk, order, &lk, wk);
  } else {
    spline_weights_high(vi, order, &li, wi);
    spline_weights_high(vj, order, &lj, wj);
    spline_weights_high(vk, order, &lk, wk);
  }
  float val = 0.0f;
  for (int a = 0; a <= order; a++) {
    const int64_t ia = spline_reflect(li + a, I);
    for (int b = 0; b <= order; b++) {
      const int64_t jb = spline_reflect(lj + b, J);
      const float wab = wi[a] * wj[b];
      for (int c = 0; c <= order; c++) {
        const int64_t kc = spline_reflect(lk + c, K);
        const float wabc = wab * wk[c];
        val = val + wabc * coef[(ia * J + jb) * K + kc];
      }
    }
  }
  return val;
}

/* one line of the recursive prefilter, in place (stride in elements); float32 like the reference's data.float() */
/* the poles of the order's prefilter: the roots inside the unit circle of sum_k beta^n(k) z^k (largest first) */
static int spline_poles(int order, float z[3]) {
  switch (order) {
    case 2: z[0] = -0.17157287525380990f; return 1; /* sqrt(8) - 3 */
    case 3: z[0] = -0.26794919243112270f; return 1; /* sqrt(3) - 2 */
    case 4: z[0] = -0.36134122590022033f; z[1] = -0.013725429297339118f; return 2;
    case 5: z[0] = -0.43057534709997358f; z[1] = -0.043096288203264665f; return 2;
    case 6: z[0] = -0.48829458930304598f; z[1] = -0.081679271076237445f; z[2] = -0.0014141518083258169f; return 3;
    default: z[0] = -0.53528043079643883f; z[1] = -0.12255461519232658f; z[2] = -0.0091486948096082803f; return 3;
  }
}

static void spline_filter_pole(float* c, int64_t n, int64_t stride, float z);
static void spline_filter_line(float* c, int64_t n, int64_t stride, int order) {
  float poles[3];
  const int n_poles = spline_poles(order, poles);
  if (n < 2) return;
  for (int p = 0; p < n_poles; p++) spline_filter_pole(c, n, stride, poles[p]);
}

/* one pole: gain, causal pass from the closed form of the mirrored series, anticausal pass */
static void spline_filter_pole(float* c, int64_t n, int64_t stride, float z) {
  const float gain = (1.0f - z) * (1.0f - 1.0f / z);
  for (int64_t i = 0; i < n; i++) c[i * stride] = c[i * stride] * gain;
  /* causal initialisation for the half-sample symmetric extension (closed form of the infinite mirrored sum) */
  float z_i = z;
  float z_n = 1.0f; /* z^n by repeated multiplication: the same n products on every implementation (no pow) */
  for (int64_t i = 0; i < n; i++) z_n = z_n * z;
  const float c0 = c[0];
  float acc = c[0] + z_n * c[(n - 1) * stride];
  for (int64_t i = 1; i < n; i++) {
    acc = acc + z_i * (c[i * stride] + z_n * c[(n - 1 - i) * stride]);
    z_i = z_i * z;
  }
  acc = acc * (z / (1.0f - z_n * z_n));
  c[0] = acc + c0;
  for (int64_t i = 1; i < n; i++) c[i * stride] = c[i * stride] + z * c[(i - 1) * stride];
  c[(n - 1) * stride] = c[(n - 1) * stride] * (z / (z - 1.0f));
  for (int64_t i = n - 2; i >= 0; i--) c[i * stride] = z * (c[(i + 1) * stride] - c[i * stride]);
}

int tio_oracle_bspline_prefilter(const void* x, float* y, int32_t dtype, int64_t n_bc, const int32_t shape[3], int32_t order,
                                 void* stream) {
  (void)stream;
  if (!x || !y || !shape || order < 2 || order > 7 || dtype_size(dtype) == 0) return TIO_ERR_INVALID_ARGUMENT;
  const int64_t I = shape[0], J = shape[1], K = shape[2], n = I * J * K;
#pragma omp parallel for schedule(static)
  for (int64_t v = 0; v < n_bc; v++) {
    float* c = y + v * n;
    for (int64_t q = 0; q < n; q++) c[q] = load_as_float(x, dtype, v * n + q);
    for (int64_t j = 0; j < J; j++)
      for (int64_t k = 0; k < K; k++) spline_filter_line(c + j * K + k, I, J * K, order);
    for (int64_t i = 0; i < I; i++)
      for (int64_t k = 0; k < K; k++) spline_filter_line(c + i * J * K + k, J, K, order);
    for (int64_t i = 0; i < I; i++)
      for (int64_t j = 0; j < J; j++) spline_filter_line(c + (i * J + j) * K, K, 1, order);
  }
  return TIO_OK;
}

int tio_oracle_resample3d(const tio_resample_geom* g, int32_t n_images,
                          const tio_resample_image* images, void* stream) {
  (void)stream;
  if (!g || !images || n_images < 1 || n_images > TIO_MAX_IMAGES) return TIO_ERR_INVALID_ARGUMENT;
  const int32_t B = g->batch;
  const int32_t I = g->in_shape[0], J = g->in_shape[1], K = g->in_shape[2];
  const int32_t Io = g->out_shape[0], Jo = g->out_shape[1], Ko = g->out_shape[2];
  const int64_t n_in = (int64_t)I * J * K, n_out = (int64_t)Io * Jo * Ko;
  const int has_cp = g->control_points_dev != NULL;
  const int32_t ni = g->cp_shape[0], nj = g->cp_shape[1], nk = g->cp_shape[2];
  const float* spacing = g->affine_first ? g->in_spacing : g->out_spacing;

  /* "label" images: torch.unique over the whole batch tensor (spatial.py:1360), unless the
   * caller already supplies the table */
  double* own_labels[TIO_MAX_IMAGES] = {0};
  const double* label_table[TIO_MAX_IMAGES] = {0};
  int32_t label_count[TIO_MAX_IMAGES] = {0};
  int32_t max_labels = 1;
  for (int32_t im = 0; im < n_images; im++) {
    if (images[im].interp != TIO_LABEL_PV) continue;
    if (images[im].channels != 1) return TIO_ERR_INVALID_ARGUMENT;
    if (images[im].labels_dev != NULL && images[im].n_labels > 0) {
      label_table[im] = images[im].labels_dev;
      label_count[im] = images[im].n_labels;
    } else {
      own_labels[im] = unique_labels(images[im].in, images[im].dtype, (int64_t)B * n_in, &label_count[im]);
      label_table[im] = own_labels[im];
    }
    if (label_count[im] > max_labels) max_labels = label_count[im];
  }

#pragma omp parallel for collapse(2) schedule(static)
  for (int32_t b = 0; b < B; b++) {
    for (int32_t io = 0; io < Io; io++) {
      float* channels = (float*)malloc((size_t)max_labels * sizeof(float));
      const float* m = g->mapping_dev + (g->mapping_batched ? (int64_t)b * 12 : 0);
      const int pass = g->passthrough_dev && g->passthrough_dev[b];
      const int elastic = has_cp && !(g->cp_skip_dev && g->cp_skip_dev[b]);
      const float* cp = has_cp ? g->control_points_dev + (g->cp_batched ? (int64_t)b * ni * nj * nk * 3 : 0) : NULL;
      lerp1d li = {0, 0, 1.0f, 0.0f};
      if (elastic) li = lerp_index(io, ni, Io);
      for (int32_t jo = 0; jo < Jo; jo++) {
        lerp1d lj = {0, 0, 1.0f, 0.0f};
        if (elastic) lj = lerp_index(jo, nj, Jo);
        for (int32_t ko = 0; ko < Ko; ko++) {
          const int64_t o_idx = ((int64_t)io * Jo + jo) * Ko + ko;
          if (pass) { /* exact copy: spatial.py:1101-1106 */
            for (int32_t im = 0; im < n_images; im++) {
              const tio_resample_image* img = &images[im];
              size_t es = dtype_size(img->dtype);
              for (int32_t c = 0; c < img->channels; c++) {
                int64_t off = ((int64_t)b * img->channels + c) * n_out + o_idx;
                if (img->interp == TIO_LINEAR_ADJOINT) { /* backward of the copy: identity; `out` (the gradient) is only read */
                  ((float*)img->in)[off] += ((const float*)img->out)[off];
                  continue;
                }
                memcpy((char*)img->out + off * es, (const char*)img->in + off * es, es);
              }
            }
            continue;
          }
          float ci = (float)io, cj = (float)jo, ck = (float)ko;
          float vi, vj, vk;
          if (elastic) {
            lerp1d lk = lerp_index(ko, nk, Ko);
            /* field layout (ni, nj, nk, 3): component stride 1 */
            float di = trilerp(cp + 0, (int64_t)nj * nk * 3, (int64_t)nk * 3, 3, li, lj, lk);
            float dj = trilerp(cp + 1, (int64_t)nj * nk * 3, (int64_t)nk * 3, 3, li, lj, lk);
            float dk = trilerp(cp + 2, (int64_t)nj * nk * 3, (int64_t)nk * 3, 3, li, lj, lk);
            if (g->affine_first) { /* spatial.py:1570-1573 */
              vi = affine_row(m + 0, ci, cj, ck) + di / spacing[0];
              vj = affine_row(m + 4, ci, cj, ck) + dj / spacing[1];
              vk = affine_row(m + 8, ci, cj, ck) + dk / spacing[2];
            } else { /* spatial.py:1574-1577 */
              float ei = ci + di / spacing[0];
              float ej = cj + dj / spacing[1];
              float ek = ck + dk / spacing[2];
              vi = affine_row(m + 0, ei, ej, ek);
              vj = affine_row(m + 4, ei, ej, ek);
              vk = affine_row(m + 8, ei, ej, ek);
            }
          } else { /* spatial.py:1542-1543 */
            vi = affine_row(m + 0, ci, cj, ck);
            vj = affine_row(m + 4, ci, cj, ck);
            vk = affine_row(m + 8, ci, cj, ck);
          }
          /* torchio axis i ≡ grid x ≡ ATen W; j ≡ y ≡ H; k ≡ z ≡ D */
          const float x = normalise_roundtrip(vi, g->norm_shape[0] > 0 ? g->norm_shape[0] : I, I);
          const float y = normalise_roundtrip(vj, g->norm_shape[1] > 0 ? g->norm_shape[1] : J, J);
          const float z = normalise_roundtrip(vk, g->norm_shape[2] > 0 ? g->norm_shape[2] : K, K);

          /* trilinear corner indices and weights (ATen grid_sampler_3d) */
          const float x0 = floorf(x), y0 = floorf(y), z0 = floorf(z);
          const float x1 = x0 + 1.0f, y1 = y0 + 1.0f, z1 = z0 + 1.0f;
          float w[8];
          w[0] = (x1 - x) * (y1 - y) * (z1 - z); /* tnw: (x0,y0,z0) */
          w[1] = (x - x0) * (y1 - y) * (z1 - z); /* tne: (x1,y0,z0) */
          w[2] = (x1 - x) * (y - y0) * (z1 - z); /* tsw: (x0,y1,z0) */
          w[3] = (x - x0) * (y - y0) * (z1 - z); /* tse: (x1,y1,z0) */
          w[4] = (x1 - x) * (y1 - y) * (z - z0); /* bnw: (x0,y0,z1) */
          w[5] = (x - x0) * (y1 - y) * (z - z0); /* bne */
          w[6] = (x1 - x) * (y - y0) * (z - z0); /* bsw */
          w[7] = (x - x0) * (y - y0) * (z - z0); /* bse */
          int ok[8];
          int64_t off[8];
          for (int t = 0; t < 8; t++) {
            float fx = (t & 1) ? x1 : x0, fy = (t & 2) ? y1 : y0, fz = (t & 4) ? z1 : z0;
            ok[t] = in_bounds(fx, I) && in_bounds(fy, J) && in_bounds(fz, K);
            off[t] = ok[t] ? ((int64_t)fx * J + (int64_t)fy) * K + (int64_t)fz : 0;
          }
          float mask = 0.0f; /* grid_sample(ones): spatial.py:1721-1727 */
          for (int t = 0; t < 8; t++)
            if (ok[t]) mask += w[t];
          /* nearest: nearbyint, half-to-even */
          const float xn = nearbyintf(x), yn = nearbyintf(y), zn = nearbyintf(z);
          const int okn = in_bounds(xn, I) && in_bounds(yn, J) && in_bounds(zn, K);
          const int64_t offn = okn ? ((int64_t)xn * J + (int64_t)yn) * K + (int64_t)zn : 0;

          for (int32_t im = 0; im < n_images; im++) {
            const tio_resample_image* img = &images[im];
            if (img->interp == TIO_LABEL_PV) {
              label_pv_voxel(img, label_table[im], label_count[im], channels, (int64_t)b * n_in, (int64_t)b * n_out + o_idx,
                             w, ok, off);
              continue;
            }
            for (int32_t c = 0; c < img->channels; c++) {
              const int64_t base_in = ((int64_t)b * img->channels + c) * n_in;
              const int64_t base_out = ((int64_t)b * img->channels + c) * n_out;
              float val;
              if (TIO_BSPLINE_ORDER(img->interp) != 0) { /* coefficients in, float32 out */
                ((float*)img->out)[base_out + o_idx] =
                    spline_sample((const float*)img->in + base_in, I, J, K, vi, vj, vk, TIO_BSPLINE_ORDER(img->interp));
                continue;
              }
              if (img->interp == TIO_LINEAR_ADJOINT) { /* backward of TIO_LINEAR: d(val)/d(in[tap]) = w[tap] */
                if (img->fill_dev && !(mask > 0.5f)) continue; /* the fill was taken: no gradient */
                const float gv = ((const float*)img->out)[base_out + o_idx];
                float* acc = (float*)img->in + base_in;
                for (int t = 0; t < 8; t++) {
                  if (!ok[t]) continue;
                  const float add = gv * w[t];
#pragma omp atomic
                  acc[off[t]] += add;
                }
                continue;
              }
              if (img->interp == TIO_LINEAR) {
                val = 0.0f;
                for (int t = 0; t < 8; t++)
                  if (ok[t]) val += load_as_float(img->in, img->dtype, base_in + off[t]) * w[t];
              } else {
                val = okn ? load_as_float(img->in, img->dtype, base_in + offn) : 0.0f;
              }
              if (img->fill_dev) val = (mask > 0.5f) ? val : img->fill_dev[c];
              store_from_float(img->out, img->dtype, base_out + o_idx, val);
            }
          }
        }
      }
      free(channels);
    }
  }
  for (int32_t im = 0; im < n_images; im++) free(own_labels[im]);
  return TIO_OK;
}

/* spatial.py:2054-2060,2094-2095: float(tensor.min()) per channel of sample 0 */
int tio_oracle_channel_min(const void* x, int32_t dtype, int32_t channels,
                           int64_t n_spatial, float* out, void* stream) {
  (void)stream;
  for (int32_t c = 0; c < channels; c++) {
    float best = INFINITY;
    int has_nan = 0;
    for (int64_t i = 0; i < n_spatial; i++) {
      float v = load_as_float(x, dtype, (int64_t)c * n_spatial + i);
      if (v != v) has_nan = 1;
      if (v < best) best = v;
    }
    out[c] = has_nan ? NAN : best;
  }
  return TIO_OK;
}

/* ------------------------------------------------------------------------ */
/* Blur stencil: blur.py:157-252 (and spatial.py:1980-2031)                   */
/* ------------------------------------------------------------------------ */
static void conv_axis(const float* src, float* dst, const int32_t shape[3], int axis,
                      const float* taps, int32_t r) {
  const int32_t I = shape[0], J = shape[1], K = shape[2];
  const int64_t stride = axis == 0 ? (int64_t)J * K : (axis == 1 ? K : 1);
  const int32_t n = shape[axis];
#pragma omp parallel for schedule(static)
  for (int32_t i = 0; i < I; i++)
    for (int32_t j = 0; j < J; j++)
      for (int32_t k = 0; k < K; k++) {
        const int64_t idx = ((int64_t)i * J + j) * K + k;
        const int32_t p = axis == 0 ? i : (axis == 1 ? j : k);
        const int64_t line = idx - (int64_t)p * stride;
        float acc = 0.0f;
        for (int32_t t = 0; t <= 2 * r; t++) {
          int32_t q = p + t - r; /* replicate padding = clamp */
          q = q < 0 ? 0 : (q > n - 1 ? n - 1 : q);
          acc += taps[t] * src[line + (int64_t)q * stride];
        }
        dst[idx] = acc;
      }
}

int tio_oracle_separable_conv3d(const void* x, void* y, void* tmp, int32_t dtype,
                                int32_t batch, int32_t channels, const int32_t shape[3],
                                const float* taps, int32_t taps_batched,
                                int32_t tap_stride, const int32_t radius[3],
                                const uint8_t* skip, void* stream) {
  (void)stream;
  (void)tmp;
  const int64_t n = (int64_t)shape[0] * shape[1] * shape[2];
  float* a = (float*)__builtin_malloc((size_t)n * sizeof(float));
  float* bbuf = (float*)__builtin_malloc((size_t)n * sizeof(float));
  if (!a || !bbuf) return TIO_ERR_INVALID_ARGUMENT;
  const size_t es = dtype_size(dtype);
  for (int32_t b = 0; b < batch; b++)
    for (int32_t c = 0; c < channels; c++) {
      const int64_t base = ((int64_t)b * channels + c) * n;
      if (skip && skip[b]) {
        memcpy((char*)y + base * es, (const char*)x + base * es, (size_t)n * es);
        continue;
      }
      for (int64_t i = 0; i < n; i++) a[i] = load_as_float(x, dtype, base + i);
      const float* t = taps + (taps_batched ? (int64_t)b * 3 * tap_stride : 0);
      float* src = a;
      float* dst = bbuf;
      for (int axis = 0; axis < 3; axis++) {
        if (radius[axis] <= 0) continue;
        conv_axis(src, dst, shape, axis, t + (int64_t)axis * tap_stride, radius[axis]);
        float* sw = src;
        src = dst;
        dst = sw;
      }
      for (int64_t i = 0; i < n; i++) store_from_float(y, dtype, base + i, src[i]);
    }
  __builtin_free(a);
  __builtin_free(bbuf);
  return TIO_OK;
}

/* Transpose of conv_axis: every product of the forward pass goes back where its factor came from (the literal adjoint of
 * blur.py:157-252's F.pad(mode="replicate") + conv3d along one axis — border voxels collect the taps that were clamped onto them). */
static void conv_axis_adjoint(const float* gy, float* gx, const int32_t shape[3], int axis, const float* taps, int32_t r) {
  const int32_t I = shape[0], J = shape[1], K = shape[2];
  const int64_t stride = axis == 0 ? (int64_t)J * K : (axis == 1 ? K : 1);
  const int32_t n = shape[axis];
  const int64_t total = (int64_t)I * J * K;
  for (int64_t e = 0; e < total; e++) gx[e] = 0.0f;
  for (int32_t i = 0; i < I; i++)
    for (int32_t j = 0; j < J; j++)
      for (int32_t k = 0; k < K; k++) {
        const int64_t idx = ((int64_t)i * J + j) * K + k;
        const int32_t p = axis == 0 ? i : (axis == 1 ? j : k);
        const int64_t line = idx - (int64_t)p * stride;
        for (int32_t t = 0; t <= 2 * r; t++) {
          int32_t q = p + t - r;
          q = q < 0 ? 0 : (q > n - 1 ? n - 1 : q);
          gx[line + (int64_t)q * stride] += taps[t] * gy[idx];
        }
      }
}

/* Backward of tio_oracle_separable_conv3d with respect to x (float32): the axes' transposes in reverse order; skipped rows pass
 * their gradient through. */
int tio_oracle_separable_conv3d_adjoint(const float* gy, float* gx, float* tmp, int32_t batch, int32_t channels,
                                        const int32_t shape[3], const float* taps, int32_t taps_batched, int32_t tap_stride,
                                        const int32_t radius[3], const uint8_t* skip, void* stream) {
  (void)stream;
  (void)tmp;
  const int64_t n = (int64_t)shape[0] * shape[1] * shape[2];
  float* a = (float*)__builtin_malloc((size_t)n * sizeof(float));
  float* bbuf = (float*)__builtin_malloc((size_t)n * sizeof(float));
  if (!a || !bbuf) return TIO_ERR_INVALID_ARGUMENT;
  for (int32_t b = 0; b < batch; b++)
    for (int32_t c = 0; c < channels; c++) {
      const int64_t base = ((int64_t)b * channels + c) * n;
      if (skip && skip[b]) {
        memcpy(gx + base, gy + base, (size_t)n * sizeof(float));
        continue;
      }
      memcpy(a, gy + base, (size_t)n * sizeof(float));
      const float* t = taps + (taps_batched ? (int64_t)b * 3 * tap_stride : 0);
      float* src = a;
      float* dst = bbuf;
      for (int axis = 2; axis >= 0; axis--) {
        if (radius[axis] <= 0) continue;
        conv_axis_adjoint(src, dst, shape, axis, t + (int64_t)axis * tap_stride, radius[axis]);
        float* sw = src;
        src = dst;
        dst = sw;
      }
      memcpy(gx + base, src, (size_t)n * sizeof(float));
    }
  __builtin_free(a);
  __builtin_free(bbuf);
  return TIO_OK;
}

/* ------------------------------------------------------------------------ */
/* BiasField: bias_field.py:201-255, 296-341                                  */
/* ------------------------------------------------------------------------ */
int tio_oracle_bias_field_apply(const void* x, void* y, int32_t dtype, int32_t batch,
                                int32_t channels, const int32_t shape[3],
                                const float* coarse, const int32_t cs[3],
                                int32_t divide, const uint8_t* skip, void* stream) {
  (void)stream;
  const int32_t I = shape[0], J = shape[1], K = shape[2];
  const int64_t n = (int64_t)I * J * K, nc = (int64_t)cs[0] * cs[1] * cs[2];
  const size_t es = dtype_size(dtype);
#pragma omp parallel for collapse(2) schedule(static)
  for (int32_t bc = 0; bc < batch * channels; bc++) {
    for (int32_t i = 0; i < I; i++) {
      const int32_t b = bc / channels;
      const int64_t base = (int64_t)bc * n;
      if (skip && skip[b]) {
        memcpy((char*)y + (base + (int64_t)i * J * K) * es,
               (const char*)x + (base + (int64_t)i * J * K) * es, (size_t)J * K * es);
        continue;
      }
      const float* f = coarse + (int64_t)bc * nc;
      lerp1d li = lerp_index(i, cs[0], I);
      for (int32_t j = 0; j < J; j++) {
        lerp1d lj = lerp_index(j, cs[1], J);
        for (int32_t k = 0; k < K; k++) {
          lerp1d lk = lerp_index(k, cs[2], K);
          const int64_t idx = base + ((int64_t)i * J + j) * K + k;
          float field = expf(trilerp(f, (int64_t)cs[1] * cs[2], cs[2], 1, li, lj, lk));
          if (dtype == TIO_F64) { /* f64 data * f32 field promotes to f64 */
            double v = ((const double*)x)[idx];
            ((double*)y)[idx] = divide ? v / (double)field : v * (double)field;
          } else {
            float v = load_as_float(x, dtype, idx);
            store_from_float(y, dtype, idx, divide ? v / field : v * field);
          }
        }
      }
    }
  }
  return TIO_OK;
}

/* ------------------------------------------------------------------------ */
/* Philox4x32-10 standard normals for the fast noise mode (this engine's own   */
/* counter-based stream — NOT a reference algorithm; the reference draws from  */
/* a CPU mt19937, noise.py:177).  The GPU kernel follows this definition.      */
/* ------------------------------------------------------------------------ */
static inline void philox4x32_10(uint32_t ctr[4], uint32_t k0, uint32_t k1) {
  for (int round = 0; round < 10; round++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * ctr[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * ctr[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ ctr[1] ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ ctr[3] ^ k1;
    uint32_t n3 = (uint32_t)p0;
    ctr[0] = n0; ctr[1] = n1; ctr[2] = n2; ctr[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

/* four normals for counter block q of stream `stream_id` */
static inline void philox_normal4(uint64_t seed, int32_t stream_id, uint64_t q, float z[4]) {
  uint32_t ctr[4] = {(uint32_t)q, (uint32_t)(q >> 32), (uint32_t)stream_id, 0u};
  philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
  for (int h = 0; h < 2; h++) {
    float u1 = ((float)(ctr[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    float u2 = ((float)(ctr[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    float radius = sqrtf(-2.0f * logf(u1));
    /* sin / cos of 2 pi u2 by quadrant + Taylor polynomials in IEEE float32 (fmaf): the sequence of csrc/intensity.hip's
     * sincos_rev, bit for bit — 8.6e-8 absolute against the true value over all 2^24 arguments; libm's sinf / cosf on
     * the ROUNDED angle 2 pi u2 (the form of rounds 1 - 4) is 4e-7 away from it: the angle's own rounding. */
    float sn, cs;
    {
      const float t = u2 * 4.0f, q = rintf(t), f = t - q, w = f * f;
      float p = fmaf(0.00016044118478735982f, w, -0.004681754135318688f);
      p = fmaf(p, w, 0.07969262624616704f);
      p = fmaf(p, w, -0.6459640975062462f);
      p = fmaf(p, w, 1.5707963267948966f);
      const float s0 = p * f;
      float c = fmaf(0.0009192602748394263f, w, -0.020863480763352960f);
      c = fmaf(c, w, 0.25366950790104797f);
      c = fmaf(c, w, -1.2337005501361697f);
      c = fmaf(c, w, 1.0f);
      const int qi = (int)q & 3;
      const float a = (qi & 1) ? c : s0, b = (qi & 1) ? s0 : c;
      sn = (qi & 2) ? -a : a;
      cs = ((qi + 1) & 2) ? -b : b;
    }
    z[2 * h] = radius * cs;
    z[2 * h + 1] = radius * sn;
  }
}

int tio_oracle_philox_normal(float* out, int64_t n, uint64_t seed, int32_t stream_id, void* stream) {
  (void)stream;
#pragma omp parallel for schedule(static)
  for (int64_t q = 0; q < (n + 3) / 4; q++) {
    float z[4];
    philox_normal4(seed, stream_id, (uint64_t)q, z);
    for (int t = 0; t < 4; t++)
      if (4 * q + t < n) out[4 * q + t] = z[t];
  }
  return TIO_OK;
}

/* ------------------------------------------------------------------------ */
/* Noise: noise.py:98-123, 166-178                                            */
/* ------------------------------------------------------------------------ */
int tio_oracle_add_noise(const void* x, void* y, int32_t dtype, int32_t batch,
                         int64_t n_per_element, float mean, float std,
                         const float* mean_b, const float* std_b, int32_t params_batched,
                         int32_t rician, const float* base1, const float* base2,
                         uint64_t philox_seed, const uint8_t* keep, void* stream) {
  (void)stream;
  const size_t es = dtype_size(dtype);
  for (int32_t b = 0; b < batch; b++) {
    const float mu = params_batched ? mean_b[b] : mean;
    const float sd = params_batched ? std_b[b] : std;
    const int64_t base = (int64_t)b * n_per_element;
    if (keep && !keep[b]) { /* noise.py:126-146 */
      memcpy((char*)y + base * es, (const char*)x + base * es, (size_t)n_per_element * es);
      continue;
    }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n_per_element; i++) {
      const int64_t idx = base + i;
      float z1, z2 = 0.0f;
      if (base1) {
        z1 = base1[idx];
        if (rician) z2 = base2[idx];
      } else {
        float z[4];
        philox_normal4(philox_seed, 0, (uint64_t)(idx >> 2), z);
        z1 = z[idx & 3];
        if (rician) {
          philox_normal4(philox_seed, 1, (uint64_t)(idx >> 2), z);
          z2 = z[idx & 3];
        }
      }
      const float n1 = mu + sd * z1; /* noise.py:178 */
      if (dtype == TIO_F64) {
        double v = ((const double*)x)[idx];
        if (rician) {
          double n2 = (double)(mu + sd * z2);
          double s = v + (double)n1;
          ((double*)y)[idx] = sqrt(s * s + n2 * n2);
        } else {
          ((double*)y)[idx] = v + (double)n1;
        }
      } else {
        float v = load_as_float(x, dtype, idx);
        float r;
        if (rician) { /* noise.py:117 */
          float n2 = mu + sd * z2;
          float s = v + n1;
          r = sqrtf(s * s + n2 * n2);
        } else {
          r = v + n1; /* noise.py:119 */
        }
        store_from_float(y, dtype, idx, r);
      }
    }
  }
  return TIO_OK;
}

/* ------------------------------------------------------------------------ */
/* Gamma: gamma.py:90 — sign(x) * |x| ** gamma                                 */
/* ------------------------------------------------------------------------ */
int tio_oracle_gamma_pow(const void* x, void* y, int32_t dtype, int32_t batch,
                         int64_t n_per_element, float gamma, const float* gamma_b,
                         int32_t params_batched, void* stream) {
  (void)stream;
  for (int32_t b = 0; b < batch; b++) {
    const float gm = params_batched ? gamma_b[b] : gamma;
    const int64_t base = (int64_t)b * n_per_element;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n_per_element; i++) {
      const int64_t idx = base + i;
      if (dtype == TIO_F64) {
        double v = ((const double*)x)[idx];
        double s = (v > 0) - (v < 0);
        ((double*)y)[idx] = s * pow(fabs(v), (double)gm);
      } else {
        float v = load_as_float(x, dtype, idx);
        float s = (float)((v > 0.0f) - (v < 0.0f));
        store_from_float(y, dtype, idx, s * powf(fabsf(v), gm));
      }
    }
  }
  return TIO_OK;
}

/* ------------------------------------------------------------------------ */
/* PatchAggregator.add_batch: aggregator.py:76-232, one patch after the other  */
/* ------------------------------------------------------------------------ */
static inline float round_storage_f(float v, int dtype) {
  if (dtype == TIO_F16) return half_to_float(float_to_half(v));
  if (dtype == TIO_BF16) return bf16_to_float(float_to_bf16(v));
  return v;
}

int tio_oracle_patch_accumulate(void* out, void* weight_sum, int32_t dtype, int32_t channels,
                                const int32_t vol_shape[3], const void* patches, int32_t n_patches,
                                const int32_t patch_shape[3], const tio_patch_placement* placements,
                                int32_t mode, const float* window_i, const float* window_j,
                                const float* window_k, void* stream) {
  (void)stream;
  if (!out || !patches || !placements || n_patches < 0 || n_patches > TIO_MAX_PATCHES) return TIO_ERR_INVALID_ARGUMENT;
  const size_t es = dtype_size(dtype);
  if (es == 0) return TIO_ERR_UNSUPPORTED_DTYPE;
  const int is_float = dtype == TIO_F32 || dtype == TIO_F64 || dtype == TIO_F16 || dtype == TIO_BF16;
  if (mode != TIO_OVERLAP_CROP && (!is_float || !weight_sum)) return TIO_ERR_UNSUPPORTED_DTYPE;
  const int64_t vol_n = (int64_t)vol_shape[0] * vol_shape[1] * vol_shape[2];
  const int64_t patch_n = (int64_t)patch_shape[0] * patch_shape[1] * patch_shape[2];
  for (int32_t p = 0; p < n_patches; p++) { /* for idx, loc in enumerate(locations): _add_patch */
    const tio_patch_placement* q = &placements[p];
    for (int32_t c = 0; c < channels; c++)
      for (int32_t di = 0; di < q->extent[0]; di++)
        for (int32_t dj = 0; dj < q->extent[1]; dj++)
          for (int32_t dk = 0; dk < q->extent[2]; dk++) {
            const int32_t si = q->src_ini[0] + di, sj = q->src_ini[1] + dj, sk = q->src_ini[2] + dk;
            const int64_t p_idx = ((int64_t)p * channels + c) * patch_n + ((int64_t)si * patch_shape[1] + sj) * patch_shape[2] + sk;
            const int64_t v_idx = (int64_t)c * vol_n +
                                  ((int64_t)(q->dst_ini[0] + di) * vol_shape[1] + (q->dst_ini[1] + dj)) * vol_shape[2] +
                                  (q->dst_ini[2] + dk);
            if (mode == TIO_OVERLAP_CROP) { /* outputs[...] = cropped */
              memcpy((char*)out + v_idx * es, (const char*)patches + p_idx * es, es);
            } else if (dtype == TIO_F64) {
              double* o = (double*)out + v_idx;
              double* n = (double*)weight_sum + v_idx;
              const double value = ((const double*)patches)[p_idx];
              if (mode == TIO_OVERLAP_AVERAGE) {
                *o += value;
                *n += 1.0;
              } else {
                const double w = (double)((window_i[si] * window_j[sj]) * window_k[sk]);
                *o += value * w;
                *n += w;
              }
            } else { /* float32 arithmetic, rounded to the storage dtype by the in-place op */
              const float value = load_as_float(patches, dtype, p_idx);
              float o = load_as_float(out, dtype, v_idx), n = load_as_float(weight_sum, dtype, v_idx);
              if (mode == TIO_OVERLAP_AVERAGE) {
                o = round_storage_f(o + value, dtype);
                n = round_storage_f(n + 1.0f, dtype);
              } else {
                const float w = (window_i[si] * window_j[sj]) * window_k[sk];
                o = round_storage_f(o + value * w, dtype);
                n = round_storage_f(n + w, dtype);
              }
              store_from_float(out, dtype, v_idx, o);
              store_from_float(weight_sum, dtype, v_idx, n);
            }
          }
  }
  return TIO_OK;
}

/* ------------------------------------------------------------------------ */
/* F.interpolate users: resize.py:57-82, anisotropy.py:128-392                */
/* ------------------------------------------------------------------------ */
int tio_oracle_interpolate3d(const void* x, void* y, int32_t dtype, int64_t n_bc, const int32_t in_shape[3],
                             const int32_t out_shape[3], int32_t mode, void* stream) {
  (void)stream;
  const size_t es = dtype_size(dtype);
  if (es == 0) return TIO_ERR_UNSUPPORTED_DTYPE;
  const int64_t n_in = (int64_t)in_shape[0] * in_shape[1] * in_shape[2];
  const int64_t n_out = (int64_t)out_shape[0] * out_shape[1] * out_shape[2];
#pragma omp parallel for collapse(2) schedule(static)
  for (int64_t bc = 0; bc < n_bc; bc++)
    for (int32_t i = 0; i < out_shape[0]; i++)
      for (int32_t j = 0; j < out_shape[1]; j++)
        for (int32_t k = 0; k < out_shape[2]; k++) {
          const int64_t o = bc * n_out + ((int64_t)i * out_shape[1] + j) * out_shape[2] + k;
          if (mode == TIO_NEAREST) { /* nearest_neighbor_compute_source_index: floor(dst * scale), scale = in / out in float */
            int32_t src[3];
            const int32_t dst[3] = {i, j, k};
            for (int d = 0; d < 3; d++) {
              const float scale = (float)in_shape[d] / (float)out_shape[d];
              int32_t v = (int32_t)floorf((float)dst[d] * scale);
              src[d] = v < in_shape[d] - 1 ? v : in_shape[d] - 1;
            }
            const int64_t from = bc * n_in + ((int64_t)src[0] * in_shape[1] + src[1]) * in_shape[2] + src[2];
            memcpy((char*)y + o * es, (const char*)x + from * es, es);
            continue;
          }
          const lerp1d li = lerp_index(i, in_shape[0], out_shape[0]);
          const lerp1d lj = lerp_index(j, in_shape[1], out_shape[1]);
          const lerp1d lk = lerp_index(k, in_shape[2], out_shape[2]);
          float corner[2][2][2];
          for (int a = 0; a < 2; a++)
            for (int b = 0; b < 2; b++)
              for (int c = 0; c < 2; c++)
                corner[a][b][c] = load_as_float(x, dtype, bc * n_in + ((int64_t)(a ? li.i1 : li.i0) * in_shape[1] + (b ? lj.i1 : lj.i0)) * in_shape[2] + (c ? lk.i1 : lk.i0));
          const float a00 = lerp2(corner[0][0][0], lk.l0, corner[0][0][1], lk.l1);
          const float a01 = lerp2(corner[0][1][0], lk.l0, corner[0][1][1], lk.l1);
          const float a10 = lerp2(corner[1][0][0], lk.l0, corner[1][0][1], lk.l1);
          const float a11 = lerp2(corner[1][1][0], lk.l0, corner[1][1][1], lk.l1);
          store_from_float(y, dtype, o, lerp2(lerp2(a00, lj.l0, a01, lj.l1), li.l0, lerp2(a10, lj.l0, a11, lj.l1), li.l1));
        }
  return TIO_OK;
}

int tio_oracle_axis_gather_lerp(const void* x, void* y, int32_t dtype, int32_t batch, int32_t channels, const int32_t shape[3],
                                int32_t axis, const int32_t* lower, const int32_t* upper, const float* weight,
                                const uint8_t* active, void* stream) {
  (void)stream;
  const size_t es = dtype_size(dtype);
  if (es == 0) return TIO_ERR_UNSUPPORTED_DTYPE;
  if (axis < 0 || axis > 2) return TIO_ERR_INVALID_ARGUMENT;
  const int64_t n = (int64_t)shape[0] * shape[1] * shape[2];
  const int32_t length = shape[axis];
  const int64_t stride = axis == 0 ? (int64_t)shape[1] * shape[2] : (axis == 1 ? shape[2] : 1);
  for (int32_t b = 0; b < batch; b++)
    for (int32_t c = 0; c < channels; c++) {
      const int64_t base = ((int64_t)b * channels + c) * n;
      if (active && !active[b]) {
        memcpy((char*)y + base * es, (const char*)x + base * es, (size_t)n * es);
        continue;
      }
      for (int64_t r = 0; r < n; r++) {
        const int32_t k = (int32_t)(r % shape[2]), j = (int32_t)((r / shape[2]) % shape[1]), i = (int32_t)(r / ((int64_t)shape[1] * shape[2]));
        const int32_t p = axis == 0 ? i : (axis == 1 ? j : k);
        const int64_t line = base + r - (int64_t)p * stride;
        const int32_t lo = lower[(int64_t)b * length + p];
        if (!upper) { /* torch.gather(data.float(), ...).to(dtype) */
          memcpy((char*)y + (base + r) * es, (const char*)x + (line + lo * stride) * es, es);
          continue;
        }
        const int32_t hi = upper[(int64_t)b * length + p];
        const float w = weight[(int64_t)b * length + p];
        const float one_minus = 1.0f - w;
        const float lower_term = load_as_float(x, dtype, line + lo * stride) * one_minus;
        const float upper_term = load_as_float(x, dtype, line + hi * stride) * w;
        store_from_float(y, dtype, base + r, lower_term + upper_term);
      }
    }
  return TIO_OK;
}

/* flip.py:182-236 */
int tio_oracle_flip3d(const void* x, void* y, int32_t dtype, int32_t batch, int32_t channels, const int32_t shape[3],
                      int32_t axes_mask, const uint8_t* flags, void* stream) {
  (void)stream;
  const size_t es = dtype_size(dtype);
  if (es == 0) return TIO_ERR_UNSUPPORTED_DTYPE;
  const int64_t n = (int64_t)shape[0] * shape[1] * shape[2];
  for (int32_t b = 0; b < batch; b++) {
    const int mask = flags ? ((flags[b * 3] ? 1 : 0) | (flags[b * 3 + 1] ? 2 : 0) | (flags[b * 3 + 2] ? 4 : 0)) : axes_mask;
    for (int32_t c = 0; c < channels; c++) {
      const int64_t base = ((int64_t)b * channels + c) * n;
      for (int32_t i = 0; i < shape[0]; i++)
        for (int32_t j = 0; j < shape[1]; j++)
          for (int32_t k = 0; k < shape[2]; k++) {
            const int32_t si = (mask & 1) ? shape[0] - 1 - i : i, sj = (mask & 2) ? shape[1] - 1 - j : j, sk = (mask & 4) ? shape[2] - 1 - k : k;
            memcpy((char*)y + (base + ((int64_t)i * shape[1] + j) * shape[2] + k) * es,
                   (const char*)x + (base + ((int64_t)si * shape[1] + sj) * shape[2] + sk) * es, es);
          }
    }
  }
  return TIO_OK;
}

/* _padding.py:62-104 (F.pad semantics) */
static int pad_source_index(int q, int n, int mode) {
  if (mode == TIO_PAD_REPLICATE) return q < 0 ? 0 : (q > n - 1 ? n - 1 : q);
  if (mode == TIO_PAD_CIRCULAR) {
    q %= n;
    return q < 0 ? q + n : q;
  }
  if (n == 1) return 0;
  const int period = 2 * (n - 1);
  q %= period;
  if (q < 0) q += period;
  return q < n ? q : period - q;
}

int tio_oracle_pad3d(const void* x, void* y, int32_t dtype, int32_t batch, int32_t channels, const int32_t in_shape[3],
                     const int32_t padding[6], int32_t mode, double fill, const void* fill_per_element, void* stream) {
  (void)stream;
  const size_t es = dtype_size(dtype);
  if (es == 0) return TIO_ERR_UNSUPPORTED_DTYPE;
  int32_t out[3];
  for (int d = 0; d < 3; d++) out[d] = in_shape[d] + padding[2 * d] + padding[2 * d + 1];
  const int64_t n_in = (int64_t)in_shape[0] * in_shape[1] * in_shape[2], n_out = (int64_t)out[0] * out[1] * out[2];
  for (int32_t b = 0; b < batch; b++)
    for (int32_t c = 0; c < channels; c++) {
      const int64_t bc = (int64_t)b * channels + c;
      for (int32_t i = 0; i < out[0]; i++)
        for (int32_t j = 0; j < out[1]; j++)
          for (int32_t k = 0; k < out[2]; k++) {
            int32_t si = i - padding[0], sj = j - padding[2], sk = k - padding[4];
            const int inside = si >= 0 && si < in_shape[0] && sj >= 0 && sj < in_shape[1] && sk >= 0 && sk < in_shape[2];
            const int64_t o = bc * n_out + ((int64_t)i * out[1] + j) * out[2] + k;
            if (!inside && mode == TIO_PAD_CONSTANT) {
              if (fill_per_element) memcpy((char*)y + o * es, (const char*)fill_per_element + (size_t)b * es, es);
              else store_from_double(y, dtype, o, fill);
              continue;
            }
            if (!inside) {
              si = pad_source_index(si, in_shape[0], mode);
              sj = pad_source_index(sj, in_shape[1], mode);
              sk = pad_source_index(sk, in_shape[2], mode);
            }
            memcpy((char*)y + o * es, (const char*)x + (bc * n_in + ((int64_t)si * in_shape[1] + sj) * in_shape[2] + sk) * es, es);
          }
    }
  return TIO_OK;
}

/* ---- Motion: k-space compositing (transforms/intensity/motion.py:334-390) ------------------
 * The reference's algorithm, literally: 3-D DFT of the still image, planes [bounds[s], bounds[s+1])
 * along the first spatial axis overwritten with the same planes of the moved image's DFT, inverse
 * 3-D DFT, real part.  The transforms are plain O(n^2)-per-line DFTs in float64 (test sizes only);
 * the reference runs complex64 FFTs, so the two agree to float32 rounding, not bit for bit. */
static void dft_axis(double* re, double* im, const int32_t shape[3], int axis, int inverse) {
  const int len = shape[axis];
  if (len == 1) return;
  const int64_t stride = axis == 0 ? (int64_t)shape[1] * shape[2] : (axis == 1 ? shape[2] : 1);
  const int64_t total = (int64_t)shape[0] * shape[1] * shape[2];
  double* cs = (double*)malloc(sizeof(double) * 2 * (size_t)len);
  for (int t = 0; t < len; t++) {
    cs[2 * t] = cos(2.0 * M_PI * t / len);
    cs[2 * t + 1] = (inverse ? 1.0 : -1.0) * sin(2.0 * M_PI * t / len);
  }
  double* line = (double*)malloc(sizeof(double) * 2 * (size_t)len);
  for (int64_t base = 0; base < total; base++) {
    if ((base / stride) % len != 0) continue; /* first element of every line along `axis` */
    for (int f = 0; f < len; f++) {
      double sr = 0.0, si = 0.0;
      for (int t = 0; t < len; t++) {
        const int64_t w = ((int64_t)f * t) % len;
        const double xr = re[base + t * stride], xi = im[base + t * stride];
        sr += xr * cs[2 * w] - xi * cs[2 * w + 1];
        si += xr * cs[2 * w + 1] + xi * cs[2 * w];
      }
      line[2 * f] = sr;
      line[2 * f + 1] = si;
    }
    for (int f = 0; f < len; f++) {
      re[base + f * stride] = inverse ? line[2 * f] / len : line[2 * f];
      im[base + f * stride] = inverse ? line[2 * f + 1] / len : line[2 * f + 1];
    }
  }
  free(line);
  free(cs);
}

int tio_oracle_kspace_segment_mix(const void* const* segments, int32_t n_segments, const int32_t* bounds, const float* mix,
                                  void* out, int32_t dtype, int32_t batch, int32_t channels, const int32_t shape[3],
                                  const uint8_t* active, void* stream) {
  (void)stream;
  (void)mix; /* the table is the HIP library's formulation; the oracle follows the reference's FFT route */
  if (segments == NULL || bounds == NULL || shape == NULL) return TIO_ERR_INVALID_ARGUMENT;
  if (n_segments < 1 || n_segments > TIO_MAX_SEGMENTS || bounds[0] != 0 || bounds[n_segments] != shape[0]) return TIO_ERR_INVALID_ARGUMENT;
  if (dtype_size(dtype) == 0) return TIO_ERR_UNSUPPORTED_DTYPE;
  const int64_t total = (int64_t)shape[0] * shape[1] * shape[2], plane = (int64_t)shape[1] * shape[2];
  double* buffers = (double*)malloc(sizeof(double) * 4 * (size_t)total);
  double *sr = buffers, *si = buffers + total, *mr = buffers + 2 * total, *mi = buffers + 3 * total;
  for (int64_t bc = 0; bc < (int64_t)batch * channels; bc++) {
    if (active != NULL && active[bc / channels] == 0) continue;
    for (int s = 0; s < n_segments; s++) {
      const float* x = (const float*)segments[s] + bc * total;
      double *r = s == 0 ? sr : mr, *i = s == 0 ? si : mi;
      for (int64_t v = 0; v < total; v++) { r[v] = (double)x[v]; i[v] = 0.0; }
      for (int axis = 0; axis < 3; axis++) dft_axis(r, i, shape, axis, 0);
      if (s > 0) { /* spectrum[:, :, start:end] = moved_spectrum[:, :, start:end] */
        memcpy(sr + bounds[s] * plane, mr + bounds[s] * plane, sizeof(double) * (size_t)((bounds[s + 1] - bounds[s]) * plane));
        memcpy(si + bounds[s] * plane, mi + bounds[s] * plane, sizeof(double) * (size_t)((bounds[s + 1] - bounds[s]) * plane));
      }
    }
    for (int axis = 0; axis < 3; axis++) dft_axis(sr, si, shape, axis, 1);
    for (int64_t v = 0; v < total; v++) store_from_double(out, dtype, bc * total + v, (double)(float)sr[v]); /* .real (float32) .to(dtype) */
  }
  free(buffers);
  return TIO_OK;
}

/* The table of the GEMM formulation, restated independently (closed form of the cosine sum:
 * Dirichlet kernel) so that the tests can hold the HIP library's direct summation against it. */
int tio_oracle_kspace_mix_table(int32_t length, int32_t n_segments, const int32_t* bounds, float* table) {
  if (bounds == NULL || table == NULL || length < 1 || n_segments < 1) return TIO_ERR_INVALID_ARGUMENT;
  if (bounds[0] != 0 || bounds[n_segments] != length) return TIO_ERR_INVALID_ARGUMENT;
  for (int s = 0; s < n_segments; s++) {
    const int lo = bounds[s], count = bounds[s + 1] - bounds[s];
    if (count < 0) return TIO_ERR_INVALID_ARGUMENT;
    for (int d = 0; d < length; d++) {
      /* sum_{f=lo}^{lo+count-1} cos(f t) = sin(count t / 2) / sin(t / 2) * cos((2 lo + count - 1) t / 2), t = 2 pi d / I */
      double value;
      if (d == 0) {
        value = (double)count;
      } else {
        const double t = 2.0 * M_PI * d / length;
        value = sin(count * t / 2.0) / sin(t / 2.0) * cos((2.0 * lo + count - 1.0) * t / 2.0);
      }
      const float w = (float)(value / length);
      for (int ip = 0; ip < length; ip++) table[(int64_t)s * length * length + (int64_t)ip * length + (ip + d) % length] = w;
    }
  }
  return TIO_OK;
}

/* torch.unique(data) for the label mode's table (spatial.py:1360): sorted distinct values */
int tio_oracle_unique_labels(const void* x, int32_t dtype, int64_t n, double* table, int32_t* count, void* workspace, void* stream) {
  (void)workspace;
  (void)stream;
  if (dtype != TIO_U8 && dtype != TIO_I8 && dtype != TIO_I16) return TIO_ERR_UNSUPPORTED_DTYPE;
  if (table == NULL || count == NULL) return TIO_ERR_INVALID_ARGUMENT;
  int32_t found = 0;
  double* values = unique_labels(x, dtype, n, &found);
  for (int32_t i = 0; i < found; i++) table[i] = values[i];
  free(values);
  *count = found;
  return TIO_OK;
}

int tio_oracle_abi_version(void) { return TIO_ABI_VERSION; }

int tio_oracle_num_threads(void) {
#if defined(_OPENMP)
  return omp_get_max_threads();
#else
  return 1;
#endif
}
