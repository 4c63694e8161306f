// intensity.hip — intensity-path kernels for gfx950:
//   tio_separable_conv3d  (Blur / antialias stencil;  blur.py:157-252)
//   tio_bias_field_apply  (BiasField;                 bias_field.py:201-341)
//   tio_add_noise         (Noise;                     noise.py:98-178)
//   tio_philox_normal     (the fast-mode normal stream of tio_add_noise)
//   tio_gamma_pow         (Gamma;                     gamma.py:80-142)
//   tio_channel_min       (default_pad_value="minimum"; spatial.py:2054-2095)
// All are HBM-bound elementwise / stencil passes: one read + one write of the
// volume per op (the separable stencil: per active axis).  No MFMA.
#include <stdlib.h>

#include <algorithm>

#include <mutex>
#include <vector>

#include "common.hpp"

namespace tio {

constexpr int kBlock = 256;

// =============================================================================
// Philox4x32-10 + Box-Muller (fast noise mode; definition in oracle/tio_oracle.c)
// =============================================================================
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int round = 0; round < 10; round++) {
    const uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * c[0];
    const uint64_t p1 = static_cast<uint64_t>(0xCD9E8D57u) * c[2];
    const uint32_t n0 = static_cast<uint32_t>(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = static_cast<uint32_t>(p1);
    const uint32_t n2 = static_cast<uint32_t>(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = static_cast<uint32_t>(p0);
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

// sin and cos of 2 pi u, u in (0, 1), in plain IEEE float32 operations (the oracle runs the SAME sequence: bit-identical).
// Round 5: the hardware units (v_sin_f32 / v_cos_f32, argument in revolutions) are good to ~1e-6 absolute; times the radius
// and the noise's std that is a few 1e-7 of the intensity range — the size of the north-star bar READ PER VOXEL at a voxel
// whose value the noise has brought close to zero (1e-4 x 1e-3 range): scripts/r5_headline_error_budget.py measured the whole
// pipeline without Noise at 4.5e-7 of that bar and with it at 1.02, whatever the resampler and the stencil did.  Quadrant
// q = rint(4 u), f = 4 u - q in [-1/2, 1/2] (both exact), Taylor polynomials of sin / cos(pi/2 f) through f^9 / f^8
// (truncation 1.7e-9 / 2.5e-8): 8.6e-8 absolute against the true value over all 2^24 arguments.
__device__ __forceinline__ void sincos_rev(float u, float& sn, float& cs) {
  const float t = __fmul_rn(u, 4.0f);
  const float q = rintf(t);
  const float f = __fsub_rn(t, q);
  const float w = __fmul_rn(f, f);
  float p = __builtin_fmaf(0.00016044118478735982f, w, -0.004681754135318688f);
  p = __builtin_fmaf(p, w, 0.07969262624616704f);
  p = __builtin_fmaf(p, w, -0.6459640975062462f);
  p = __builtin_fmaf(p, w, 1.5707963267948966f);
  const float s0 = __fmul_rn(p, f);
  float c = __builtin_fmaf(0.0009192602748394263f, w, -0.020863480763352960f);
  c = __builtin_fmaf(c, w, 0.25366950790104797f);
  c = __builtin_fmaf(c, w, -1.2337005501361697f);
  c = __builtin_fmaf(c, w, 1.0f);
  const int qi = static_cast<int>(q) & 3;
  const float a = (qi & 1) ? c : s0, b = (qi & 1) ? s0 : c;
  sn = (qi & 2) ? -a : a;
  cs = ((qi + 1) & 2) ? -b : b;
}

__device__ __forceinline__ void philox_normal4(uint64_t seed, int stream_id, uint64_t q, float z[4]) {
  uint32_t c[4] = {static_cast<uint32_t>(q), static_cast<uint32_t>(q >> 32), static_cast<uint32_t>(stream_id), 0u};
  philox4x32_10(c, static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const float u1 = __fmul_rn(__fadd_rn(static_cast<float>(c[2 * h] >> 8), 0.5f), 1.0f / 16777216.0f);
    const float u2 = __fmul_rn(__fadd_rn(static_cast<float>(c[2 * h + 1] >> 8), 0.5f), 1.0f / 16777216.0f);
    // -2 ln(u1) through the hardware log2 / sqrt units (u1 in (0, 1): no special cases; one ulp each: the radius is
    // within ~2e-7 relative of libm's)
    const float nl = __fmul_rn(-1.3862943611198906f, __builtin_amdgcn_logf(u1));  // -2 ln 2 * log2(u1)
    const float radius = __builtin_amdgcn_sqrtf(fmaxf(nl, 0.0f));
    float cs, sn;
    sincos_rev(u2, sn, cs);
    z[2 * h] = __fmul_rn(radius, cs);
    z[2 * h + 1] = __fmul_rn(radius, sn);
  }
}

// =============================================================================
// Separable cross-correlation with replicate padding
// =============================================================================
// One pass = one axis.  Lanes always run along K (contiguous), so every global
// access is a coalesced row segment.  A block stages the line segment it needs
// (+ replicate-clamped halo) in LDS once and every thread then reads its
// 2r+1 taps from LDS in tap order — the accumulation order of the oracle — so
// each input element is fetched from HBM/L2 ~once per pass instead of 2r+1 times.
//   axis I / J : tile = kConvLine outputs along the axis x 64 lanes along K
//   axis K     : tile = 4 rows x 256 outputs along K (one wave per row)
constexpr int kConvLine = 32;   // outputs along the stencil axis per block (axes I, J)
constexpr int kConvKSpan = 256; // outputs along K per wave (axis K)
constexpr int kMaxRadius = 48;  // LDS budget: (32 + 96) * 64 * 4 B = 32 KiB

struct ConvArgs {
  const void* src;
  void* dst;
  const void* x_orig;  // the original input (skip rows are copied from it bit-exactly)
  const float* taps;
  const uint8_t* skip;
  int I, J, K;
  int channels;
  int axis, radius;
  int taps_batched, tap_stride;
  int orig_dtype;
  int last_pass;
  int tiles_a;  // tiles along the stencil axis (axes I, J) / row groups (axis K)
  int bcs;      // B * C (register-window marching kernel: strips are enumerated per wave)
  int radius_k;  // > 0: the J pass also applies the K taps to every row it produces (fused J+K)
  // tio_blur_fused: BiasField folded into the loads of the I pass, Noise into the stores of the last pass
  const float* bias_coarse;          // (B, C, ci, cj, ck) or nullptr
  int bias_ci, bias_cj, bias_ck;
  float bias_si, bias_sj, bias_sk;   // ATen lerp scales of the coarse grid
  int noise_on, noise_batched;
  float noise_mean, noise_std;
  const float* noise_mean_b;
  const float* noise_std_b;
  uint64_t noise_seed;
  const float* noise_base;  // noise_on == 2: the normal draws of every element, laid out like the data (the reference's seeded stream)
  // tio_blur_fused(fast_math = 1): the taps of the marching kernel accumulate with fused multiply-adds (one rounding per
  // tap instead of the reference's two: results within float rounding, ~1e-7 relative — the J+K pass is bound by vector
  // instructions, not by memory, and the taps are two thirds of them)
  int fma;
};

template <int SRC_DT, int DST_DT>
__global__ __launch_bounds__(kBlock) void conv_line_kernel(const ConvArgs a) {
  // axes I and J.  grid: x = K tiles (64), y = tiles along the axis, z = other axis * (B*C)
  extern __shared__ __attribute__((aligned(16))) float s_mem[];
  const int r = a.radius, ntaps = 2 * r + 1;
  float* s_taps = s_mem;                 // ntaps
  float* s_tile = s_mem + ((ntaps + 3) & ~3);  // (kConvLine + 2r) x 64
  const int n_other = a.axis == 0 ? a.J : a.I;  // size of the non-stencil, non-K axis
  const int other = blockIdx.z % n_other;
  const int bc = blockIdx.z / n_other;
  const int b = bc / a.channels;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave index in an SGPR
  const int k = blockIdx.x * 64 + lane;
  const int n = a.axis == 0 ? a.I : a.J;        // length of the stencil axis
  const int p0 = blockIdx.y * kConvLine;         // first output position along the axis
  const int64_t n_spatial = static_cast<int64_t>(a.I) * a.J * a.K;
  const int64_t base = static_cast<int64_t>(bc) * n_spatial;
  const int64_t stride = a.axis == 0 ? static_cast<int64_t>(a.J) * a.K : a.K;
  const int64_t other_stride = a.axis == 0 ? a.K : static_cast<int64_t>(a.J) * a.K;
  const int64_t line = base + other * other_stride + k;  // element (axis pos 0, other, k)
  const bool active = k < a.K;

  if (a.skip != nullptr && a.skip[b] != 0) {
    // rows with no blur are restored bit-exactly (blur.py:249-251): emitted unchanged by the last pass
    if (a.last_pass && active) {
      const int es = dtype_size(a.orig_dtype);
      for (int q = wave; q < kConvLine && p0 + q < n; q += kBlock / 64) {
        const int64_t e = line + static_cast<int64_t>(p0 + q) * stride;
        const char* s = static_cast<const char*>(a.x_orig) + e * es;
        char* d = static_cast<char*>(a.dst) + e * es;
        for (int c = 0; c < es; c++) d[c] = s[c];
      }
    }
    return;
  }
  const float* t = a.taps + (a.taps_batched ? static_cast<int64_t>(b) * 3 * a.tap_stride : 0) +
                   static_cast<int64_t>(a.axis) * a.tap_stride;
  for (int i = threadIdx.x; i < ntaps; i += kBlock) s_taps[i] = t[i];
  const int rows = min(kConvLine, n - p0) + 2 * r;
  if (active) {
    for (int q = wave; q < rows; q += kBlock / 64) {
      const int pos = min(max(p0 + q - r, 0), n - 1);  // replicate padding == clamp
      s_tile[q * 64 + lane] = Elem<SRC_DT>::load(a.src, line + static_cast<int64_t>(pos) * stride);
    }
  }
  __syncthreads();
  if (!active) return;
  for (int q = wave; q < kConvLine && p0 + q < n; q += kBlock / 64) {
    float acc = 0.0f;
    const float* col = s_tile + q * 64 + lane;
    for (int tt = 0; tt < ntaps; tt++) acc = __fadd_rn(acc, __fmul_rn(s_taps[tt], col[tt * 64]));
    Elem<DST_DT>::store(a.dst, line + static_cast<int64_t>(p0 + q) * stride, acc);
  }
}

template <int SRC_DT, int DST_DT>
__global__ __launch_bounds__(kBlock) void conv_k_kernel(const ConvArgs a) {
  // axis K.  grid: x = K tiles (256), y = J tiles (4 rows, one per wave), z = I * (B*C)
  extern __shared__ __attribute__((aligned(16))) float s_mem[];
  const int r = a.radius, ntaps = 2 * r + 1;
  float* s_taps = s_mem;
  const int pitch = kConvKSpan + 2 * r;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave index in an SGPR
  float* s_row = s_mem + ((ntaps + 3) & ~3) + wave * pitch;
  const int i = blockIdx.z % a.I;
  const int bc = blockIdx.z / a.I;
  const int b = bc / a.channels;
  const int j = blockIdx.y * (kBlock / 64) + wave;
  const int k0 = blockIdx.x * kConvKSpan;
  const int64_t n_spatial = static_cast<int64_t>(a.I) * a.J * a.K;
  const int64_t row = static_cast<int64_t>(bc) * n_spatial + (static_cast<int64_t>(i) * a.J + j) * a.K;
  const bool active = j < a.J;

  if (a.skip != nullptr && a.skip[b] != 0) {
    if (a.last_pass && active) {
      const int es = dtype_size(a.orig_dtype);
      for (int q = lane; q < kConvKSpan && k0 + q < a.K; q += 64) {
        const int64_t e = row + k0 + q;
        const char* s = static_cast<const char*>(a.x_orig) + e * es;
        char* d = static_cast<char*>(a.dst) + e * es;
        for (int c = 0; c < es; c++) d[c] = s[c];
      }
    }
    return;
  }
  const float* t = a.taps + (a.taps_batched ? static_cast<int64_t>(b) * 3 * a.tap_stride : 0) + 2 * a.tap_stride;
  for (int q = threadIdx.x; q < ntaps; q += kBlock) s_taps[q] = t[q];
  const int span = min(kConvKSpan, a.K - k0);
  if (active) {
    for (int q = lane; q < span + 2 * r; q += 64) {
      const int pos = min(max(k0 + q - r, 0), a.K - 1);
      s_row[q] = Elem<SRC_DT>::load(a.src, row + pos);
    }
  }
  __syncthreads();
  if (!active) return;
  for (int q = lane; q < span; q += 64) {
    float acc = 0.0f;
    for (int tt = 0; tt < ntaps; tt++) acc = __fadd_rn(acc, __fmul_rn(s_taps[tt], s_row[q + tt]));
    Elem<DST_DT>::store(a.dst, row + k0 + q, acc);
  }
}

// ---- float32 fast paths: 16-byte global accesses ------------------------------------
// Same arithmetic (tap order, separate multiply and add) as the generic kernels above;
// used when source and destination are float32, K % 4 == 0, pointers are 16-byte
// aligned and the radius is small enough for the LDS tile.
constexpr int kConvMaxRadiusV4 = 16;  // (32 + 2*16) rows x 1 KiB = 64 KiB of LDS

constexpr int kConvStep = 16;  // output rows per marching step (axes I, J)

template <bool FUSE_K, bool PRE_BIAS, int POST_NOISE>
__global__ __launch_bounds__(kBlock) void conv_line_v4_kernel(const ConvArgs a) {
  // axes I and J.  grid: x = K tiles (256 = 64 lanes x float4), y = segments along the axis, z = other axis * (B*C).
  // A block marches along the stencil axis with a ring of kConvStep + 2r + kConvStep rows in
  // LDS: every input row is fetched exactly once (no halo re-reads between steps) and the
  // 16-byte loads of step s+1 are in flight while step s is computed.
  extern __shared__ __attribute__((aligned(16))) float s_mem[];
  const int r = a.radius, ntaps = 2 * r + 1;
  const int ring = 2 * kConvStep + 2 * r;
  float* s_taps = s_mem;                                                   // ntaps
  float4* s_ring = reinterpret_cast<float4*>(s_mem + ((ntaps + 3) & ~3));  // ring x 64 float4
  // fused J+K: every wave owns one staged row (8 + 256 + 8 floats) behind the ring
  float* s_krow = reinterpret_cast<float*>(s_ring + ring * 64) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * 272;
  const int n_other = a.axis == 0 ? a.J : a.I;
  const int other = blockIdx.z % n_other;
  const int bc = blockIdx.z / n_other;
  const int b = bc / a.channels;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave index in an SGPR
  const int k = blockIdx.x * 256 + 4 * lane;
  const int n = a.axis == 0 ? a.I : a.J;
  const int seg = (n + gridDim.y - 1) / gridDim.y;                 // outputs per segment (multiple of kConvStep except the last)
  const int p_begin = blockIdx.y * seg, p_end = min(p_begin + seg, n);
  const int64_t n_spatial = static_cast<int64_t>(a.I) * a.J * a.K;
  const int64_t stride = a.axis == 0 ? static_cast<int64_t>(a.J) * a.K : a.K;
  const int64_t other_stride = a.axis == 0 ? a.K : static_cast<int64_t>(a.J) * a.K;
  const int64_t line = static_cast<int64_t>(bc) * n_spatial + other * other_stride + k;
  const bool active = k < a.K;
  const float* src = static_cast<const float*>(a.src);
  float* dst = static_cast<float*>(a.dst);
  if (p_begin >= p_end) return;

  if (a.skip != nullptr && a.skip[b] != 0) {  // rows with no blur: emitted unchanged by the last pass
    if (a.last_pass && active) {
      const float* orig = static_cast<const float*>(a.x_orig);
      for (int p = p_begin + wave; p < p_end; p += kBlock / 64) {
        const int64_t e = line + static_cast<int64_t>(p) * stride;
        *reinterpret_cast<float4*>(dst + e) = *reinterpret_cast<const float4*>(orig + e);
      }
    }
    return;
  }
  const float* t = a.taps + (a.taps_batched ? static_cast<int64_t>(b) * 3 * a.tap_stride : 0) +
                   static_cast<int64_t>(a.axis) * a.tap_stride;
  typedef __attribute__((address_space(4))) const float* const_float_ptr;  // taps are read-only for the whole launch
  const_float_ptr tc = (const_float_ptr)(t);
  const_float_ptr tk = (const_float_ptr)(a.taps + (a.taps_batched ? static_cast<int64_t>(b) * 3 * a.tap_stride : 0) + 2 * a.tap_stride);
  (void)s_taps; (void)tk; (void)s_krow;
  float tkw[17];  // the K taps of the fused stage, once, at their window positions (see conv_march_kernel)
  (void)tkw;
  if constexpr (FUSE_K) {
    const int rk0 = a.radius_k;
#pragma unroll
    for (int jj = 0; jj < 17; jj++) {
      const int tt = jj - (8 - rk0);
      tkw[jj] = (tt >= 0 && tt <= 2 * rk0) ? tk[tt] : 0.0f;
    }
  }
  float noise_mu = a.noise_mean, noise_sd = a.noise_std;  // scalar loads, once (see conv_march_kernel)
  if constexpr (POST_NOISE != 0) {
    if (a.noise_batched) {
      noise_mu = ((const_float_ptr)a.noise_mean_b)[b];
      noise_sd = ((const_float_ptr)a.noise_std_b)[b];
    }
  }
  constexpr int RW = kConvStep / (kBlock / 64);  // rows per wave per step (4)
  // PRE_BIAS (I pass of tio_blur_fused): every row is multiplied by exp(trilinear(coarse)) on its
  // way into the ring — the arithmetic of bias_kernel (K- and J-lerps of a coarse plane are
  // invariants of the lane's four k positions, redone only when the row enters a new cell)
  Lerp1D b_lk[4];
  Lerp1D b_lj{0, 0, 1.0f, 0.0f};
  float b_p0[4] = {0.f, 0.f, 0.f, 0.f}, b_p1[4] = {0.f, 0.f, 0.f, 0.f};
  int b_cur0 = -1, b_cur1 = -1;
  const float* b_fg = nullptr;
  if constexpr (PRE_BIAS) {
    b_fg = a.bias_coarse + static_cast<int64_t>(bc) * (a.bias_ci * a.bias_cj * a.bias_ck);
    b_lj = lerp_index(other, a.bias_cj, a.J, a.bias_sj);
#pragma unroll
    for (int e = 0; e < 4; e++) b_lk[e] = lerp_index(min(k + e, a.K - 1), a.bias_ck, a.K, a.bias_sk);
  }
  auto bias_row = [&](float4 v, int pos) -> float4 {
    if constexpr (PRE_BIAS) {
      const Lerp1D li = lerp_index(pos, a.bias_ci, a.I, a.bias_si);
      const int s_i = a.bias_cj * a.bias_ck, s_j = a.bias_ck;
      auto plane = [&](int ii, float (&out)[4]) {
        const float* r0 = b_fg + ii * s_i + b_lj.i0 * s_j;
        const float* r1 = b_fg + ii * s_i + b_lj.i1 * s_j;
#pragma unroll
        for (int e = 0; e < 4; e++)
          out[e] = lerp2(lerp2(r0[b_lk[e].i0], b_lk[e].l0, r0[b_lk[e].i1], b_lk[e].l1), b_lj.l0,
                         lerp2(r1[b_lk[e].i0], b_lk[e].l0, r1[b_lk[e].i1], b_lk[e].l1), b_lj.l1);
      };
      if (li.i0 != b_cur0) {
        if (li.i0 == b_cur1) {
#pragma unroll
          for (int e = 0; e < 4; e++) b_p0[e] = b_p1[e];
        } else {
          plane(li.i0, b_p0);
        }
        b_cur0 = li.i0;
      }
      if (li.i1 != b_cur1) {
        if (li.i1 == b_cur0) {
#pragma unroll
          for (int e = 0; e < 4; e++) b_p1[e] = b_p0[e];
        } else {
          plane(li.i1, b_p1);
        }
        b_cur1 = li.i1;
      }
      v.x = __fmul_rn(v.x, expf(lerp2(b_p0[0], li.l0, b_p1[0], li.l1)));  // bias_field.py:341, :130
      v.y = __fmul_rn(v.y, expf(lerp2(b_p0[1], li.l0, b_p1[1], li.l1)));
      v.z = __fmul_rn(v.z, expf(lerp2(b_p0[2], li.l0, b_p1[2], li.l1)));
      v.w = __fmul_rn(v.w, expf(lerp2(b_p0[3], li.l0, b_p1[3], li.l1)));
    }
    return v;
  };
  // logical row index q counts from p_begin - r; ring slot = q mod ring (tracked incrementally)
  // prologue: rows q in [0, kConvStep + 2r) for the first step
  if (active) {
    for (int q = wave; q < kConvStep + 2 * r; q += kBlock / 64) {
      const int pos = min(max(p_begin - r + q, 0), n - 1);  // replicate padding == clamp
      s_ring[q * 64 + lane] = bias_row(*reinterpret_cast<const float4*>(src + line + static_cast<int64_t>(pos) * stride), pos);
    }
  }
  __syncthreads();
  int slot0 = 0;  // ring slot of logical row (step start - r)
  // rows of step s+1 (pre1) are written to the ring at the end of step s; rows of step s+2
  // (pre2) are requested at the start of step s — two steps of 16-byte loads in flight per lane
  typedef float v4f __attribute__((ext_vector_type(4)));  // native vector type: stays in registers (HIP's float4 struct arrays did not)
  v4f pre1[RW], pre2[RW];
#define TIO_FETCH_ROWS(DSTV, P_FIRST) /* rows P_FIRST + r + wave*RW + u, replicate-clamped */        \
  _Pragma("unroll") for (int u = 0; u < RW; u++) {                                                  \
    const int pos_ = min(max((P_FIRST) + r + wave * RW + u, 0), n - 1);                             \
    DSTV[u] = *reinterpret_cast<const v4f*>(src + line + static_cast<int64_t>(pos_) * stride);     \
  }
  if (active && p_begin + kConvStep < p_end) { TIO_FETCH_ROWS(pre1, p_begin + kConvStep) }
  for (int p0 = p_begin; p0 < p_end; p0 += kConvStep) {
    const bool more = p0 + kConvStep < p_end;
    if (active && p0 + 2 * kConvStep < p_end) { TIO_FETCH_ROWS(pre2, p0 + 2 * kConvStep) }
    if (active) {
#pragma unroll
      for (int u = 0; u < RW; u++) {
        const int o = wave * RW + u;  // output row inside the step
        if (p0 + o < p_end) {
          float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
          int slot = slot0 + o;
          if (slot >= ring) slot -= ring;
          for (int tt = 0; tt < ntaps; tt++) {
            const float w = tc[tt];  // scalar load (constant address space), not an LDS read per lane
            const float4 v = s_ring[slot * 64 + lane];
            acc.x = __fadd_rn(acc.x, __fmul_rn(w, v.x));
            acc.y = __fadd_rn(acc.y, __fmul_rn(w, v.y));
            acc.z = __fadd_rn(acc.z, __fmul_rn(w, v.z));
            acc.w = __fadd_rn(acc.w, __fmul_rn(w, v.w));
            slot = slot + 1 == ring ? 0 : slot + 1;
          }
          if constexpr (FUSE_K) {
            // the row this wave just produced (its 256 K positions are spread over the 64 lanes)
            // goes through the register-window K filter of conv_k_v4_kernel before it is stored:
            // the float32 rounding between the J and the K pass is kept, only the HBM round trip goes
            const int rk = a.radius_k;
            const float edge_l = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc.x), 0));
            const float edge_r = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc.w), a.K / 4 - 1));
            *reinterpret_cast<float4*>(s_krow + 8 + 4 * lane) = acc;
            // replicate padding over the WHOLE halo (8 positions a side), not only the radius: the fast taps run over a tier of
            // radii with zero-padded taps, and 0 * (whatever LDS held before) is NaN when that happens to be non-finite
            // (round 5: tests/test_gpu_ops_parity.py::test_fused_jk_stage_every_k_radius failed behind a test that left Inf there)
            (void)rk;
            for (int h = lane; h < 16; h += a.K / 4) {  // only the K/4 lanes that own data are active here
              if (h < 8) s_krow[h] = edge_l;                   // left
              else s_krow[8 + a.K + (h - 8)] = edge_r;         // right
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const float4* rv = reinterpret_cast<const float4*>(s_krow + 8) + lane;
            float w[20];
#pragma unroll
            for (int d = 0; d < 5; d++) {
              const float4 c = rv[d - 2];
              w[4 * d] = c.x; w[4 * d + 1] = c.y; w[4 * d + 2] = c.z; w[4 * d + 3] = c.w;
            }
            float4 out = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
            for (int jj = 0; jj < 17; jj++) {
              if (jj >= 8 - rk && jj <= 8 + rk) {
                const float tw = tkw[jj];
                out.x = __fadd_rn(out.x, __fmul_rn(tw, w[jj]));
                out.y = __fadd_rn(out.y, __fmul_rn(tw, w[jj + 1]));
                out.z = __fadd_rn(out.z, __fmul_rn(tw, w[jj + 2]));
                out.w = __fadd_rn(out.w, __fmul_rn(tw, w[jj + 3]));
              }
            }
            acc = out;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if constexpr (POST_NOISE != 0) {
              // the arithmetic of noise_kernel's 16-byte path: same Philox block (global element
              // index >> 2), same mean + std * z, same add
              const int64_t e0 = line + static_cast<int64_t>(p0 + o) * stride;  // element index of acc.x in the tensor
              float z[4];
              if constexpr (POST_NOISE == 2) {  // explicit draws (tio_add_noise with base1_dev): the reference's own stream
                const float4 zb = *reinterpret_cast<const float4*>(a.noise_base + e0);
                z[0] = zb.x; z[1] = zb.y; z[2] = zb.z; z[3] = zb.w;
              } else {
                philox_normal4(a.noise_seed, 0, static_cast<uint64_t>(e0 >> 2), z);
              }
              const float mu = noise_mu, sd = noise_sd;
              acc.x = __fadd_rn(acc.x, __fadd_rn(mu, __fmul_rn(sd, z[0])));
              acc.y = __fadd_rn(acc.y, __fadd_rn(mu, __fmul_rn(sd, z[1])));
              acc.z = __fadd_rn(acc.z, __fadd_rn(mu, __fmul_rn(sd, z[2])));
              acc.w = __fadd_rn(acc.w, __fadd_rn(mu, __fmul_rn(sd, z[3])));
            }
          }
          *reinterpret_cast<float4*>(dst + line + static_cast<int64_t>(p0 + o) * stride) = acc;
        }
      }
    }
    if (more) {
      // the new rows overwrite slots that were last read one step ago (ring = 2 steps + 2r)
      if (active) {
#pragma unroll
        for (int u = 0; u < RW; u++) {
          int slot = slot0 + kConvStep + 2 * r + wave * RW + u;
          if (slot >= ring) slot -= ring;
          if (slot >= ring) slot -= ring;
          if constexpr (PRE_BIAS) {
            const int pos = min(max(p0 + kConvStep + r + wave * RW + u, 0), n - 1);
            s_ring[slot * 64 + lane] = bias_row(make_float4(pre1[u].x, pre1[u].y, pre1[u].z, pre1[u].w), pos);
          } else {
            *reinterpret_cast<v4f*>(&s_ring[slot * 64 + lane]) = pre1[u];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < RW; u++) pre1[u] = pre2[u];
      slot0 += kConvStep;
      if (slot0 >= ring) slot0 -= ring;
      __syncthreads();
    }
  }
}

#undef TIO_FETCH_ROWS

// ---- float32 fast path for radii <= 8: the stencil window lives in registers ---------------
// Along I or J a lane only ever needs ITS OWN column's history, so nothing has to be shared:
// every wave marches alone down one (other, b, c) strip of 256 K positions (one float4 per
// lane) with the last 2R+1 rows in VGPRs.  The marching loop is unrolled by the window length,
// which turns the rotating window into compile-time register names (no moves, no LDS ring, no
// block barrier); two rows of 16-byte loads are in flight per lane.  Arithmetic and tap order
// are those of the generic kernels.  LDS is only used by the fused K stage (one row per wave).
constexpr int kMarchMaxRadius = 8;
#ifndef TIO_MARCH_AHEAD
#define TIO_MARCH_AHEAD 2
#endif
constexpr int kMarchAhead = TIO_MARCH_AHEAD;

// (the fused J + K instantiations of the usual radii are held to 128 registers — four waves per SIMD: with explicit draws the
// R = 6 one needed 130, one wave per SIMD less)
// FMA (tio_blur_fused(fast_math = 1)) is a template parameter: as a block-uniform branch both forms of every tap sat in the
// W-times unrolled loop — 12 600 lines of assembly, more than the instruction cache holds.
template <int R, bool FUSE_K, bool PRE_BIAS, int POST_NOISE, bool FMA>
__global__ __launch_bounds__(kBlock, (FUSE_K && !PRE_BIAS && R <= 6) ? 4 : 1) void conv_march_kernel(const ConvArgs a) {
  constexpr int W = 2 * R + 1;
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(4))) const float* const_float_ptr;
  extern __shared__ __attribute__((aligned(16))) float s_mem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* s_krow = s_mem + wave * 272;  // fused J+K: 8 + 256 + 8 floats per wave
  (void)s_krow;
  const int n_other = a.axis == 0 ? a.J : a.I;
  // block order (tiles_a): 1 = strips along grid.x (consecutive blocks work on adjacent strips of the
  // same segment: their rows are adjacent in memory), 0 = strips along grid.z
  const int strip_block = a.tiles_a ? blockIdx.x : blockIdx.z;
  const int k_tile = a.tiles_a ? blockIdx.z : blockIdx.x;
  const int strip = strip_block * (kBlock / 64) + wave;
  if (strip >= n_other * a.bcs) return;  // waves are independent: no barrier anywhere below
  const int other = strip % n_other;
  const int bc = strip / n_other;
  const int b = bc / a.channels;
  const int k = k_tile * 256 + 4 * lane;
  const int n = a.axis == 0 ? a.I : a.J;
  const int seg = (n + gridDim.y - 1) / gridDim.y;
  const int p_begin = blockIdx.y * seg, p_end = min(p_begin + seg, n);
  const int64_t n_spatial = static_cast<int64_t>(a.I) * a.J * a.K;
  const int64_t stride = a.axis == 0 ? static_cast<int64_t>(a.J) * a.K : a.K;
  const int64_t other_stride = a.axis == 0 ? a.K : static_cast<int64_t>(a.J) * a.K;
  const int64_t line = static_cast<int64_t>(bc) * n_spatial + other * other_stride + k;
  const float* src = static_cast<const float*>(a.src);
  float* dst = static_cast<float*>(a.dst);
  if (p_begin >= p_end || k >= a.K) return;

  if (a.skip != nullptr && a.skip[b] != 0) {  // rows with no blur: emitted unchanged by the last pass
    if (a.last_pass) {
      const float* orig = static_cast<const float*>(a.x_orig);
      for (int p = p_begin; p < p_end; p++) {
        const int64_t e = line + static_cast<int64_t>(p) * stride;
        *reinterpret_cast<v4f*>(dst + e) = *reinterpret_cast<const v4f*>(orig + e);
      }
    }
    return;
  }
  const int64_t tap_base = a.taps_batched ? static_cast<int64_t>(b) * 3 * a.tap_stride : 0;
  const_float_ptr tc = (const_float_ptr)(a.taps + tap_base + static_cast<int64_t>(a.axis) * a.tap_stride);
  const_float_ptr tk = (const_float_ptr)(a.taps + tap_base + 2 * a.tap_stride);
  (void)tk;
  float tw[W];  // scalar registers
#pragma unroll
  for (int t = 0; t < W; t++) tw[t] = tc[t];
  // The K taps as well, ONCE, at the positions the register window uses them (tap t of radius rk sits at jj = 8 - rk + t;
  // zeros beyond the radius).  Until round 4 every tap was a scalar load INSIDE the marching loop — its address depends on the
  // run-time radius — followed by s_waitcnt lgkmcnt(0): thirteen serialised round trips through the scalar cache per row.
  float tkw[17];
  (void)tkw;
  if constexpr (FUSE_K) {
    const int rk0 = a.radius_k;
#pragma unroll
    for (int jj = 0; jj < 17; jj++) {
      const int t = jj - (8 - rk0);
      tkw[jj] = (t >= 0 && t <= 2 * rk0) ? tk[t] : 0.0f;
    }
  }
  // noise parameters of this strip's element, once, through the scalar cache: as plain global loads inside
  // the marching loop they sit behind a branch, and the compiler's wait at the join is vmcnt(0) - it
  // drained the two prefetched rows (and the previous store) on every row
  float noise_mu = a.noise_mean, noise_sd = a.noise_std;
  if constexpr (POST_NOISE != 0) {
    if (a.noise_batched) {
      noise_mu = ((const_float_ptr)a.noise_mean_b)[b];
      noise_sd = ((const_float_ptr)a.noise_std_b)[b];
    }
  }

  // PRE_BIAS: the arithmetic of bias_kernel, coarse planes cached while the row stays in a cell
  Lerp1D b_lk[4];
  Lerp1D b_lj{0, 0, 1.0f, 0.0f};
  float b_p0[4] = {0.f, 0.f, 0.f, 0.f}, b_p1[4] = {0.f, 0.f, 0.f, 0.f};
  int b_cur0 = -1, b_cur1 = -1;
  const float* b_fg = nullptr;
  if constexpr (PRE_BIAS) {
    b_fg = a.bias_coarse + static_cast<int64_t>(bc) * (a.bias_ci * a.bias_cj * a.bias_ck);
    b_lj = lerp_index(other, a.bias_cj, a.J, a.bias_sj);
#pragma unroll
    for (int e = 0; e < 4; e++) b_lk[e] = lerp_index(min(k + e, a.K - 1), a.bias_ck, a.K, a.bias_sk);
  }
  auto bias_row = [&](v4f v, int pos) -> v4f {
    if constexpr (PRE_BIAS) {
      const Lerp1D li = lerp_index(pos, a.bias_ci, a.I, a.bias_si);
      const int s_i = a.bias_cj * a.bias_ck, s_j = a.bias_ck;
      auto plane = [&](int ii, float (&out)[4]) {
        const float* r0 = b_fg + ii * s_i + b_lj.i0 * s_j;
        const float* r1 = b_fg + ii * s_i + b_lj.i1 * s_j;
#pragma unroll
        for (int e = 0; e < 4; e++)
          out[e] = lerp2(lerp2(r0[b_lk[e].i0], b_lk[e].l0, r0[b_lk[e].i1], b_lk[e].l1), b_lj.l0,
                         lerp2(r1[b_lk[e].i0], b_lk[e].l0, r1[b_lk[e].i1], b_lk[e].l1), b_lj.l1);
      };
      if (li.i0 != b_cur0) {
        if (li.i0 == b_cur1) {
#pragma unroll
          for (int e = 0; e < 4; e++) b_p0[e] = b_p1[e];
        } else {
          plane(li.i0, b_p0);
        }
        b_cur0 = li.i0;
      }
      if (li.i1 != b_cur1) {
        if (li.i1 == b_cur0) {
#pragma unroll
          for (int e = 0; e < 4; e++) b_p1[e] = b_p0[e];
        } else {
          plane(li.i1, b_p1);
        }
        b_cur1 = li.i1;
      }
      v.x = __fmul_rn(v.x, expf(lerp2(b_p0[0], li.l0, b_p1[0], li.l1)));  // bias_field.py:341, :130
      v.y = __fmul_rn(v.y, expf(lerp2(b_p0[1], li.l0, b_p1[1], li.l1)));
      v.z = __fmul_rn(v.z, expf(lerp2(b_p0[2], li.l0, b_p1[2], li.l1)));
      v.w = __fmul_rn(v.w, expf(lerp2(b_p0[3], li.l0, b_p1[3], li.l1)));
    }
    return v;
  };
#define TIO_ROW_POS(P) min(max((P), 0), n - 1) /* replicate padding == clamp */
#define TIO_ROW_LOAD(P) (*reinterpret_cast<const v4f*>(src + line + static_cast<int64_t>(TIO_ROW_POS(P)) * stride))
  v4f win[W];
#pragma unroll
  for (int t = 0; t < 2 * R; t++) win[t] = bias_row(TIO_ROW_LOAD(p_begin - R + t), TIO_ROW_POS(p_begin - R + t));
  win[2 * R] = win[0];
  // kMarchAhead rows of 16-byte loads in flight per lane (4 measured no faster than 2: the passes
  // are not latency bound)
  v4f nxt[kMarchAhead];
#pragma unroll
  for (int d = 0; d < kMarchAhead; d++) nxt[d] = TIO_ROW_LOAD(p_begin + R + d);
  for (int p0 = p_begin; p0 < p_end; p0 += W) {
#pragma unroll
    for (int u = 0; u < W; u++) {
      const int p = p0 + u;
      if (p >= p_end) return;  // wave uniform
      __builtin_amdgcn_sched_barrier(0);  // keep the rows apart: no hoisting of later rows' work into this one
      // the newest row (p + R) replaces the oldest one; the window of output p is slots u .. u + 2R (mod W)
      win[(2 * R + u) % W] = bias_row(nxt[0], TIO_ROW_POS(p + R));
#pragma unroll
      for (int d = 0; d + 1 < kMarchAhead; d++) nxt[d] = nxt[d + 1];
      // explicit draws of this output row: requested BEFORE the row that is loaded ahead (vector loads return in order: the
      // wait in front of the sum then leaves the newer row load in flight), used a whole window of taps later
      v4f zrow = {0.0f, 0.0f, 0.0f, 0.0f};
      if constexpr (POST_NOISE == 2) zrow = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(a.noise_base + line + static_cast<int64_t>(p) * stride));
      (void)zrow;
      nxt[kMarchAhead - 1] = TIO_ROW_LOAD(p + R + kMarchAhead);
      float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if constexpr (FMA) {
#pragma unroll
        for (int t = 0; t < W; t++) {
          const v4f v = win[(u + t) % W];
          acc.x = __builtin_fmaf(tw[t], v.x, acc.x);
          acc.y = __builtin_fmaf(tw[t], v.y, acc.y);
          acc.z = __builtin_fmaf(tw[t], v.z, acc.z);
          acc.w = __builtin_fmaf(tw[t], v.w, acc.w);
        }
      } else {
#pragma unroll
        for (int t = 0; t < W; t++) {
          const v4f v = win[(u + t) % W];
          acc.x = __fadd_rn(acc.x, __fmul_rn(tw[t], v.x));
          acc.y = __fadd_rn(acc.y, __fmul_rn(tw[t], v.y));
          acc.z = __fadd_rn(acc.z, __fmul_rn(tw[t], v.z));
          acc.w = __fadd_rn(acc.w, __fmul_rn(tw[t], v.w));
        }
      }
      if constexpr (FUSE_K) {
        // the register-window K filter of conv_k_v4_kernel on the row this wave just produced
        const int rk = a.radius_k;
        const float edge_l = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc.x), 0));
        const float edge_r = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc.w), a.K / 4 - 1));
        *reinterpret_cast<float4*>(s_krow + 8 + 4 * lane) = acc;
        // replicate padding over the WHOLE halo (8 positions a side), not only the radius: the fast taps below run over a tier of
        // radii with zero-padded taps, and 0 * (whatever LDS held before) is NaN when that happens to be non-finite (round 5:
        // test_fused_jk_stage_every_k_radius failed for the radii inside a tier behind a test that had left Inf in LDS)
        for (int h = lane; h < 16; h += a.K / 4) {  // only the K/4 lanes that own data are active here
          if (h < 8) s_krow[h] = edge_l;                   // left
          else s_krow[8 + a.K + (h - 8)] = edge_r;         // right
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const float4* rv = reinterpret_cast<const float4*>(s_krow + 8) + lane;
        float w[20];
#pragma unroll
        for (int d = 0; d < 5; d++) {
          const float4 c = rv[d - 2];
          w[4 * d] = c.x; w[4 * d + 1] = c.y; w[4 * d + 2] = c.z; w[4 * d + 3] = c.w;
        }
        float4 out = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if constexpr (FMA) {
          // fast taps: three branch-free tiers (radius <= 4 / <= 6 / <= 8) over the preloaded, zero-padded taps — a tap
          // beyond the radius is fma(0, w, out) = out for every finite w (the fast mode's contract is float rounding on finite
          // data; the exact mode below keeps the skipped taps skipped)
#define TIO_K_TIER(T)                                              \
  _Pragma("unroll") for (int jj = 8 - (T); jj <= 8 + (T); jj++) { \
    const float tv = tkw[jj];                                      \
    out.x = __builtin_fmaf(tv, w[jj], out.x);                      \
    out.y = __builtin_fmaf(tv, w[jj + 1], out.y);                  \
    out.z = __builtin_fmaf(tv, w[jj + 2], out.z);                  \
    out.w = __builtin_fmaf(tv, w[jj + 3], out.w);                  \
  }
          if (rk <= 4) { TIO_K_TIER(4) } else if (rk <= 6) { TIO_K_TIER(6) } else { TIO_K_TIER(8) }
#undef TIO_K_TIER
        } else {
          // exact taps: the taps taken and their order are the oracle's (a skipped tap stays skipped: 0 * w is not nothing for
          // a non-finite w).  Tiers of two radii — the inner taps of a tier unconditional, only its outermost pair (radii
          // 6 / 8) or the taps beyond |d| = 1 (radii <= 4) behind a scalar branch: 2 - 6 branches per row instead of 17.
#define TIO_K_TAP(JJ)                                       \
  {                                                         \
    const float tv = tkw[JJ];                               \
    out.x = __fadd_rn(out.x, __fmul_rn(tv, w[(JJ)]));       \
    out.y = __fadd_rn(out.y, __fmul_rn(tv, w[(JJ) + 1]));   \
    out.z = __fadd_rn(out.z, __fmul_rn(tv, w[(JJ) + 2]));   \
    out.w = __fadd_rn(out.w, __fmul_rn(tv, w[(JJ) + 3]));   \
  }
          if (rk <= 4) {
#pragma unroll
            for (int jj = 4; jj <= 12; jj++) {
              if (jj >= 7 && jj <= 9) TIO_K_TAP(jj)
              else if (jj >= 8 - rk && jj <= 8 + rk) TIO_K_TAP(jj)
            }
          } else if (rk <= 6) {
            if (rk == 6) TIO_K_TAP(2)
#pragma unroll
            for (int jj = 3; jj <= 13; jj++) TIO_K_TAP(jj)
            if (rk == 6) TIO_K_TAP(14)
          } else {
            if (rk == 8) TIO_K_TAP(0)
#pragma unroll
            for (int jj = 1; jj <= 15; jj++) TIO_K_TAP(jj)
            if (rk == 8) TIO_K_TAP(16)
          }
#undef TIO_K_TAP
        }
        acc = out;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if constexpr (POST_NOISE != 0) {  // the arithmetic of noise_kernel's 16-byte path
          const int64_t e0 = line + static_cast<int64_t>(p) * stride;
          float z[4];
          if constexpr (POST_NOISE == 2) {
            z[0] = zrow.x; z[1] = zrow.y; z[2] = zrow.z; z[3] = zrow.w;
          } else {
            philox_normal4(a.noise_seed, 0, static_cast<uint64_t>(e0 >> 2), z);
          }
          const float mu = noise_mu, sd = noise_sd;
          acc.x = __fadd_rn(acc.x, __fadd_rn(mu, __fmul_rn(sd, z[0])));
          acc.y = __fadd_rn(acc.y, __fadd_rn(mu, __fmul_rn(sd, z[1])));
          acc.z = __fadd_rn(acc.z, __fadd_rn(mu, __fmul_rn(sd, z[2])));
          acc.w = __fadd_rn(acc.w, __fadd_rn(mu, __fmul_rn(sd, z[3])));
        }
      }
      *reinterpret_cast<float4*>(dst + line + static_cast<int64_t>(p) * stride) = acc;
    }
  }
#undef TIO_ROW_LOAD
#undef TIO_ROW_POS
}

constexpr int kConvKRows = 16;  // rows per wave per block (axis K)

__global__ __launch_bounds__(kBlock) void conv_k_v4_kernel(const ConvArgs a) {
  // axis K.  grid: x = K tiles (256), y = J groups (4 waves x kConvKRows rows), z = I * (B*C).
  // Each wave walks kConvKRows rows: one 16-byte load per lane per row (the next row's load
  // is in flight while the current one is filtered), taps applied to lane-consecutive
  // positions (conflict-free LDS reads), results transposed through LDS so that the store
  // is one 16-byte access per lane as well.
  extern __shared__ __attribute__((aligned(16))) float s_mem[];
  const int r = a.radius, ntaps = 2 * r + 1;
  const int r4 = max((r + 3) & ~3, 8);           // aligned offset of the main part inside a staged row (>= 8: window path)
  const int pitch = kConvKSpan + 2 * r4;         // floats per wave row (multiple of 4)
  float* s_taps = s_mem;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave index in an SGPR
  float* s_row = s_mem + ((ntaps + 3) & ~3) + wave * (pitch + kConvKSpan);
  float* s_out = s_row + pitch;
  const int i = blockIdx.z % a.I;
  const int bc = blockIdx.z / a.I;
  const int b = bc / a.channels;
  const int j0 = (blockIdx.y * (kBlock / 64) + wave) * kConvKRows;
  const int k0 = blockIdx.x * kConvKSpan;
  const int64_t n_spatial = static_cast<int64_t>(a.I) * a.J * a.K;
  const int64_t plane = static_cast<int64_t>(bc) * n_spatial + static_cast<int64_t>(i) * a.J * a.K;
  const float* src = static_cast<const float*>(a.src);
  float* dst = static_cast<float*>(a.dst);
  const int kk = k0 + 4 * lane;                  // this lane's 4 consecutive positions
  const bool lane_in = kk < a.K;                 // K % 4 == 0: all four or none
  const int j_end = min(j0 + kConvKRows, a.J);

  if (a.skip != nullptr && a.skip[b] != 0) {
    if (a.last_pass && lane_in) {
      const float* orig = static_cast<const float*>(a.x_orig);
      for (int j = j0; j < j_end; j++)
        *reinterpret_cast<float4*>(dst + plane + static_cast<int64_t>(j) * a.K + kk) =
            *reinterpret_cast<const float4*>(orig + plane + static_cast<int64_t>(j) * a.K + kk);
    }
    return;
  }
  const float* t = a.taps + (a.taps_batched ? static_cast<int64_t>(b) * 3 * a.tap_stride : 0) + 2 * a.tap_stride;
  typedef __attribute__((address_space(4))) const float* const_float_ptr;  // taps are read-only for the whole launch
  const_float_ptr tc = (const_float_ptr)(t);  // scalar loads; every wave works on its own LDS rows, no block barrier
  (void)s_taps;
  // the taps of the register-window path, once, at the window positions that use them (see conv_march_kernel)
  float tkw[17];
#pragma unroll
  for (int jj = 0; jj < 17; jj++) {
    const int tt = jj - (8 - r);
    tkw[jj] = (r <= 8 && tt >= 0 && tt <= 2 * r) ? tc[tt] : 0.0f;
  }
  const int span = min(kConvKSpan, a.K - k0);
  const float* in = s_row + r4 - r;
  // halo: lane q < 2r owns one replicate-clamped position outside the span (r on each side);
  // it is fetched together with the row, one row ahead, so no load sits on the critical path
  const int h_off = lane < r ? lane - r : span + (lane - r);
  const int h_pos = min(max(k0 + h_off, 0), a.K - 1);
  const bool h_in = lane < 2 * r;  // r <= kConvMaxRadiusV4 = 16: at most 32 halo lanes
  constexpr int G = 4;  // rows per group: the next group's loads are in flight while this one is filtered
  float4 cur[G], nxt[G];
  float hcur[G], hnxt[G];
#pragma unroll
  for (int u = 0; u < G; u++) {
    cur[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    hcur[u] = 0.0f;
    if (j0 + u < j_end) {
      const int64_t row = plane + static_cast<int64_t>(j0 + u) * a.K;
      if (lane_in) cur[u] = *reinterpret_cast<const float4*>(src + row + kk);
      if (h_in) hcur[u] = src[row + h_pos];
    }
  }
  for (int jg = j0; jg < j_end; jg += G) {
#pragma unroll
    for (int u = 0; u < G; u++) {
      nxt[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      hnxt[u] = 0.0f;
      if (jg + G + u < j_end) {
        const int64_t row = plane + static_cast<int64_t>(jg + G + u) * a.K;
        if (lane_in) nxt[u] = *reinterpret_cast<const float4*>(src + row + kk);
        if (h_in) hnxt[u] = src[row + h_pos];
      }
    }
#pragma unroll
    for (int u = 0; u < G; u++) {
      const int j = jg + u;
      if (j < j_end) {
        const int64_t row = plane + static_cast<int64_t>(j) * a.K;
        if (lane_in) *reinterpret_cast<float4*>(s_row + r4 + 4 * lane) = cur[u];
        if (h_in) s_row[r4 + h_off] = hcur[u];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // same wave: LDS operations complete in order
        __builtin_amdgcn_wave_barrier();
        if (r <= 8) {
          // register window: the lane's 4 outputs need positions 4l-8 .. 4l+11 = five aligned
          // 16-byte LDS reads (instead of one 4-byte read per tap and output); taps outside the
          // radius are skipped by a scalar branch, so the accumulation order is the oracle's
          const float4* rv = reinterpret_cast<const float4*>(s_row + r4) + lane;
          float w[20];
#pragma unroll
          for (int d = 0; d < 5; d++) {
            const float4 c = rv[d - 2];
            w[4 * d] = c.x; w[4 * d + 1] = c.y; w[4 * d + 2] = c.z; w[4 * d + 3] = c.w;
          }
          float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
          for (int jj = 0; jj < 17; jj++) {
            if (jj >= 8 - r && jj <= 8 + r) {
              const float tw = tkw[jj];
              acc.x = __fadd_rn(acc.x, __fmul_rn(tw, w[jj]));
              acc.y = __fadd_rn(acc.y, __fmul_rn(tw, w[jj + 1]));
              acc.z = __fadd_rn(acc.z, __fmul_rn(tw, w[jj + 2]));
              acc.w = __fadd_rn(acc.w, __fmul_rn(tw, w[jj + 3]));
            }
          }
          if (lane_in) *reinterpret_cast<float4*>(dst + row + kk) = acc;
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          continue;
        }
        for (int q = lane; q < span; q += 64) {
          float acc = 0.0f;
          for (int tt = 0; tt < ntaps; tt++) acc = __fadd_rn(acc, __fmul_rn(tc[tt], in[q + tt]));
          s_out[q] = acc;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane_in) *reinterpret_cast<float4*>(dst + row + kk) = *reinterpret_cast<const float4*>(s_out + 4 * lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
#pragma unroll
    for (int u = 0; u < G; u++) {
      cur[u] = nxt[u];
      hcur[u] = hnxt[u];
    }
  }
}

struct ConvFuse {  // optional pointwise stages of tio_blur_fused
  const float* bias_coarse = nullptr;
  int bias_shape[3] = {0, 0, 0};
  int noise_on = 0, noise_batched = 0;
  float noise_mean = 0.0f, noise_std = 0.0f;
  const float* noise_mean_b = nullptr;
  const float* noise_std_b = nullptr;
  uint64_t noise_seed = 0;
  const float* noise_base = nullptr;
  int fma = 0;
  bool any() const { return bias_coarse != nullptr || noise_on != 0; }
};

template <int DT>
static int launch_conv(const void* x, void* y, float* tmp0, float* tmp1, int32_t batch, int32_t channels,
                       const int32_t shape[3], const float* taps, int taps_batched, int tap_stride,
                       const int32_t radius[3], const uint8_t* skip, hipStream_t stream, const ConvFuse& fuse = ConvFuse()) {
  int active[3], n_active = 0;
  for (int ax = 0; ax < 3; ax++)
    if (radius[ax] > 0) active[n_active++] = ax;
  const void* src = x;
  // J and K can share one pass when the whole K row sits in one 256-wide tile and both radii
  // are small: the J kernel filters every row it produces along K before storing it
  const bool fuse_jk = DT == TIO_F32 && radius[1] > 0 && radius[2] > 0 && radius[1] <= kConvMaxRadiusV4 && radius[2] <= 8 &&
                       shape[2] <= 256 && (shape[2] & 3) == 0 && !env_switches().conv_no_fuse;
  if (fuse.any()) {
    // the pointwise stages ride on the marching I pass (loads) and the fused J+K pass (stores)
    const bool ok = DT == TIO_F32 && n_active == 3 && fuse_jk && radius[0] <= kConvMaxRadiusV4 && skip == nullptr &&
                    ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(tmp0)) & 15) == 0;
    if (!ok) return TIO_ERR_UNSUPPORTED_CONFIG;
  }
  for (int s = 0; s < n_active; s++) {
    const int axis = active[s];
    if (axis == 2 && fuse_jk) break;  // done by the J pass
    const bool first = s == 0, last = s == n_active - 1 || (axis == 1 && fuse_jk);
    void* dst = last ? y : static_cast<void*>((s % 2 == 0) ? tmp0 : tmp1);
    ConvArgs a{};
    a.src = src; a.dst = dst; a.x_orig = x; a.taps = taps; a.skip = skip;
    a.I = shape[0]; a.J = shape[1]; a.K = shape[2];
    a.channels = channels; a.axis = axis; a.radius = radius[axis];
    a.taps_batched = taps_batched; a.tap_stride = tap_stride; a.orig_dtype = DT; a.last_pass = last ? 1 : 0;
    const int ntaps = 2 * radius[axis] + 1;
    const int bcs = batch * channels;
    dim3 grid;
    size_t lds;
    if (axis == 2) {
      grid = dim3((shape[2] + kConvKSpan - 1) / kConvKSpan, (shape[1] + 3) / 4, static_cast<unsigned>(shape[0]) * bcs);
      lds = (((ntaps + 3) & ~3) + 4 * (kConvKSpan + 2 * radius[axis])) * sizeof(float);
    } else {
      const int n = shape[axis], other = axis == 0 ? shape[1] : shape[0];
      grid = dim3((shape[2] + 63) / 64, (n + kConvLine - 1) / kConvLine, static_cast<unsigned>(other) * bcs);
      lds = (((ntaps + 3) & ~3) + (kConvLine + 2 * radius[axis]) * 64) * sizeof(float);
    }
    if (grid.z > 65535u || grid.y > 65535u) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_separable_conv3d: volume too large for one launch");
#define TIO_CONV_LAUNCH(S, D)                                                                      \
  do {                                                                                             \
    if (axis == 2) hipLaunchKernelGGL((conv_k_kernel<S, D>), grid, dim3(kBlock), lds, stream, a);  \
    else hipLaunchKernelGGL((conv_line_kernel<S, D>), grid, dim3(kBlock), lds, stream, a);         \
  } while (0)
    const bool f32_pass = (first ? DT == TIO_F32 : true) && (last ? DT == TIO_F32 : true);
    const bool aligned = ((shape[2] & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) |
                                                     reinterpret_cast<uintptr_t>(x)) & 15) == 0;
    if (f32_pass && aligned && radius[axis] <= kConvMaxRadiusV4) {  // 16-byte access fast paths
      if (axis == 2) {
        const int r4 = std::max((radius[axis] + 3) & ~3, 8);
        lds = (((ntaps + 3) & ~3) + 4 * (2 * kConvKSpan + 2 * r4)) * sizeof(float);
        grid.y = static_cast<unsigned>((shape[1] + 4 * kConvKRows - 1) / (4 * kConvKRows));
        hipLaunchKernelGGL(conv_k_v4_kernel, grid, dim3(kBlock), lds, stream, a);
      } else {
        // one or two marching segments per line: enough blocks to fill the chip, halo re-read only at the cut
        const int n = shape[axis], other = axis == 0 ? shape[1] : shape[0];
        const int64_t lines = static_cast<int64_t>((shape[2] + 255) / 256) * other * bcs;
        int segs = lines >= 4096 ? 1 : (lines >= 2048 ? 2 : 4);
        int seg_len = ((n + segs - 1) / segs + kConvStep - 1) / kConvStep * kConvStep;
        segs = (n + seg_len - 1) / seg_len;
        grid.x = static_cast<unsigned>((shape[2] + 255) / 256);
        grid.y = static_cast<unsigned>(segs);
        const bool fused = axis == 1 && fuse_jk;
        a.radius_k = fused ? radius[2] : 0;
        const bool pre_bias = axis == 0 && fuse.bias_coarse != nullptr;
        const bool post_noise = fused && fuse.noise_on != 0;
        const bool noise_base = post_noise && fuse.noise_on == 2;
        if (pre_bias) {
          a.bias_coarse = fuse.bias_coarse;
          a.bias_ci = fuse.bias_shape[0]; a.bias_cj = fuse.bias_shape[1]; a.bias_ck = fuse.bias_shape[2];
          a.bias_si = lerp_scale(a.bias_ci, shape[0]); a.bias_sj = lerp_scale(a.bias_cj, shape[1]); a.bias_sk = lerp_scale(a.bias_ck, shape[2]);
        }
        if (post_noise) {
          a.noise_on = fuse.noise_on; a.noise_base = fuse.noise_base; a.noise_batched = fuse.noise_batched; a.noise_mean = fuse.noise_mean; a.noise_std = fuse.noise_std;
          a.noise_mean_b = fuse.noise_mean_b; a.noise_std_b = fuse.noise_std_b; a.noise_seed = fuse.noise_seed;
        }
        // (the bias variant needs > 240 VGPRs beyond radius 6: the LDS ring kernel is the better choice there)
        if (radius[axis] <= (pre_bias ? 6 : kMarchMaxRadius) && !env_switches().conv_ring) {
          // register-window marching: one strip per wave, enough segments for >= 8 waves per SIMD
          a.bcs = bcs;
          a.fma = fuse.fma;
          const int64_t strips = lines;
          int want = static_cast<int>((8192 + strips - 1) / strips);
          want = std::max(1, std::min(want, std::max(1, n / 32)));
          if (env_switches().march_segs > 0) want = env_switches().march_segs;  // experiments
          const int len = (n + want - 1) / want;
          grid.y = static_cast<unsigned>((n + len - 1) / len);
          grid.z = static_cast<unsigned>((static_cast<int64_t>(other) * bcs + kBlock / 64 - 1) / (kBlock / 64));
          lds = fused ? 4 * 272 * sizeof(float) : 0;
          a.tiles_a = 0;
          if (env_switches().march_order >= 0) a.tiles_a = env_switches().march_order != 0;  // experiments
          if (a.tiles_a) std::swap(grid.x, grid.z);
#define TIO_MARCH_VARIANT_F(RR, FM)                                                                                                  \
  {                                                                                                                                  \
    if (fused && noise_base) hipLaunchKernelGGL((conv_march_kernel<RR, true, false, 2, FM>), grid, dim3(kBlock), lds, stream, a);     \
    else if (fused && post_noise) hipLaunchKernelGGL((conv_march_kernel<RR, true, false, 1, FM>), grid, dim3(kBlock), lds, stream, a); \
    else if (fused) hipLaunchKernelGGL((conv_march_kernel<RR, true, false, 0, FM>), grid, dim3(kBlock), lds, stream, a);              \
    else if (pre_bias) hipLaunchKernelGGL((conv_march_kernel<RR, false, true, 0, FM>), grid, dim3(kBlock), lds, stream, a);           \
    else hipLaunchKernelGGL((conv_march_kernel<RR, false, false, 0, FM>), grid, dim3(kBlock), lds, stream, a);                        \
  }
#define TIO_MARCH_VARIANT(RR)                                                                              \
  {                                                                                                        \
    if (fuse.fma) { TIO_MARCH_VARIANT_F(RR, true) } else { TIO_MARCH_VARIANT_F(RR, false) }                                      \
  }
          switch (radius[axis]) {
            case 1: TIO_MARCH_VARIANT(1) break;
            case 2: TIO_MARCH_VARIANT(2) break;
            case 3: TIO_MARCH_VARIANT(3) break;
            case 4: TIO_MARCH_VARIANT(4) break;
            case 5: TIO_MARCH_VARIANT(5) break;
            case 6: TIO_MARCH_VARIANT(6) break;
            case 7: TIO_MARCH_VARIANT(7) break;
            default: TIO_MARCH_VARIANT(8) break;
          }
#undef TIO_MARCH_VARIANT
#undef TIO_MARCH_VARIANT_F
          src = dst;
          continue;
        }
        lds = (((ntaps + 3) & ~3) + (2 * kConvStep + 2 * radius[axis]) * 256 + (fused ? 4 * 272 : 0)) * sizeof(float);
#define TIO_LINE_LAUNCH(FK, PB, PN)                                                                                   \
  {                                                                                                                   \
    auto kern = conv_line_v4_kernel<FK, PB, PN>;                                                                      \
    if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                               static_cast<int>(lds)) != hipSuccess)                                  \
      return fail(TIO_ERR_LAUNCH, "tio_separable_conv3d: cannot reserve %zu bytes of LDS", lds);                       \
    hipLaunchKernelGGL(kern, grid, dim3(kBlock), lds, stream, a);                                                      \
  }
        if (fused && noise_base) TIO_LINE_LAUNCH(true, false, 2)
        else if (fused && post_noise) TIO_LINE_LAUNCH(true, false, 1)
        else if (fused) TIO_LINE_LAUNCH(true, false, 0)
        else if (pre_bias) TIO_LINE_LAUNCH(false, true, 0)
        else TIO_LINE_LAUNCH(false, false, 0)
#undef TIO_LINE_LAUNCH
      }
      src = dst;
      continue;
    }
    if (first && last) TIO_CONV_LAUNCH(DT, DT);
    else if (first) TIO_CONV_LAUNCH(DT, TIO_F32);
    else if (last) TIO_CONV_LAUNCH(TIO_F32, DT);
    else TIO_CONV_LAUNCH(TIO_F32, TIO_F32);
#undef TIO_CONV_LAUNCH
    src = dst;
  }
  return check_launch("tio_separable_conv3d");
}

// =============================================================================
// BiasField
// =============================================================================
constexpr int kMaxCoarseLds = 8192;  // floats (32 KiB)

// grid: x = K tiles (64 lanes), y = J tiles (4 rows, one per wave), z = I tiles * (B*C);
// each thread walks kBiasTileI slabs so the j/k lerp terms are computed once.
constexpr int kBiasTileI = 8;

template <int DT>
__global__ __launch_bounds__(kBlock) void bias_kernel(const void* __restrict__ x, void* __restrict__ y,
                                                      int channels, int I, int J, int K,
                                                      const float* __restrict__ coarse, int ci, int cj, int ck,
                                                      float scale_i, float scale_j, float scale_k, int divide,
                                                      const uint8_t* __restrict__ skip, int tiles_i) {
  extern __shared__ __attribute__((aligned(16))) float s_coarse[];
  const int it = blockIdx.z % tiles_i;
  const int bc = blockIdx.z / tiles_i;
  const int b = bc / channels;
  const int64_t n = static_cast<int64_t>(I) * J * K;
  const int64_t base = static_cast<int64_t>(bc) * n;
  const bool skipped = skip != nullptr && skip[b] != 0;
  const int nc = ci * cj * ck;
  const float* fg = coarse + static_cast<int64_t>(bc) * nc;
  const bool in_lds = !skipped && nc <= kMaxCoarseLds;
  if (in_lds) {
    for (int t = threadIdx.x; t < nc; t += blockDim.x) s_coarse[t] = fg[t];
    __syncthreads();
  }
  const int k = blockIdx.x * 64 + (threadIdx.x & 63);
  const int j = blockIdx.y * (kBlock / 64) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (k >= K || j >= J) return;
  const int i_begin = it * kBiasTileI, i_end = min(i_begin + kBiasTileI, I);
  using T = typename Elem<DT>::type;
  const int64_t row = base + static_cast<int64_t>(j) * K + k;
  const int64_t slab = static_cast<int64_t>(J) * K;
  if (skipped) {  // std == 0 rows restored bit-exactly (bias_field.py:247-253)
    for (int i = i_begin; i < i_end; i++) static_cast<T*>(y)[row + i * slab] = static_cast<const T*>(x)[row + i * slab];
    return;
  }
  const Lerp1D lj = lerp_index(j, cj, J, scale_j);
  const Lerp1D lk = lerp_index(k, ck, K, scale_k);
  const int s_i = cj * ck, s_j = ck;
  // all loads of the i-tile are issued before the first use: 8 independent HBM requests in flight per lane
  float v[kBiasTileI];
  double vd[DT == TIO_F64 ? kBiasTileI : 1];
#pragma unroll
  for (int u = 0; u < kBiasTileI; u++) {
    const int i = i_begin + u;
    if (i < i_end) {
      if constexpr (DT == TIO_F64) vd[u] = static_cast<const double*>(x)[row + i * slab];
      else v[u] = Elem<DT>::load(x, row + i * slab);
    }
  }
  // The K- and J-lerps of a coarse plane are invariants of this thread's (j, k) column
  // (ATen nests K innermost, then J, then I), so they are redone only when the walk
  // along I enters a new coarse cell — a block-uniform event.
  auto coarse_plane = [&](int ii) -> float {
    const int o0 = ii * s_i + lj.i0 * s_j, o1 = ii * s_i + lj.i1 * s_j;
    float c00, c01, c10, c11;
    if (in_lds) {
      c00 = s_coarse[o0 + lk.i0]; c01 = s_coarse[o0 + lk.i1]; c10 = s_coarse[o1 + lk.i0]; c11 = s_coarse[o1 + lk.i1];
    } else {
      c00 = fg[o0 + lk.i0]; c01 = fg[o0 + lk.i1]; c10 = fg[o1 + lk.i0]; c11 = fg[o1 + lk.i1];
    }
    return lerp2(lerp2(c00, lk.l0, c01, lk.l1), lj.l0, lerp2(c10, lk.l0, c11, lk.l1), lj.l1);
  };
  int cur0 = -1, cur1 = -1;
  float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
  for (int u = 0; u < kBiasTileI; u++) {
    const int i = i_begin + u;
    if (i >= i_end) break;
    const Lerp1D li = lerp_index(i, ci, I, scale_i);
    if (li.i0 != cur0) {
      p0 = (li.i0 == cur1) ? p1 : coarse_plane(li.i0);
      cur0 = li.i0;
    }
    if (li.i1 != cur1) {
      p1 = (li.i1 == cur0) ? p0 : coarse_plane(li.i1);
      cur1 = li.i1;
    }
    const float field = expf(lerp2(p0, li.l0, p1, li.l1));  // bias_field.py:341
    const int64_t idx = row + i * slab;
    if constexpr (DT == TIO_F64) {  // f64 data (x) f32 field promotes to f64
      static_cast<double*>(y)[idx] = divide ? vd[u] / static_cast<double>(field) : vd[u] * static_cast<double>(field);
    } else {
      Elem<DT>::store(y, idx, divide ? __fdiv_rn(v[u], field) : __fmul_rn(v[u], field));
    }
  }
}

__global__ __launch_bounds__(kBlock) void philox_normal_kernel(float* __restrict__ out, int64_t n, uint64_t seed,
                                                               int stream_id) {
  const int64_t q = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (4 * q >= n) return;
  float z[4];
  philox_normal4(seed, stream_id, static_cast<uint64_t>(q), z);
  if (4 * q + 3 < n) {
    *reinterpret_cast<float4*>(out + 4 * q) = make_float4(z[0], z[1], z[2], z[3]);
  } else {
    for (int t = 0; t < 4; t++)
      if (4 * q + t < n) out[4 * q + t] = z[t];
  }
}

// =============================================================================
// Noise: thread = 4 consecutive elements (one Philox block / one 16-B access)
// =============================================================================
template <int DT>
__global__ __launch_bounds__(kBlock) void noise_kernel(const void* __restrict__ x, void* __restrict__ y,
                                                       int64_t n_per_element, float mean, float std,
                                                       const float* __restrict__ mean_b,
                                                       const float* __restrict__ std_b, int params_batched,
                                                       int rician, const float* __restrict__ base1,
                                                       const float* __restrict__ base2, uint64_t seed,
                                                       const uint8_t* __restrict__ keep) {
  const int b = blockIdx.y;
  const int64_t q = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t e0 = 4 * q;
  if (e0 >= n_per_element) return;
  const int64_t base = static_cast<int64_t>(b) * n_per_element;
  const float mu = params_batched ? mean_b[b] : mean;
  const float sd = params_batched ? std_b[b] : std;
  const bool kept = keep == nullptr || keep[b] != 0;
  using T = typename Elem<DT>::type;
  const int cnt = static_cast<int>(min(static_cast<int64_t>(4), n_per_element - e0));
  if (!kept) {  // gated-out rows restored bit-exactly (noise.py:126-146)
    for (int t = 0; t < cnt; t++) static_cast<T*>(y)[base + e0 + t] = static_cast<const T*>(x)[base + e0 + t];
    return;
  }
  float z1[4], z2[4] = {0.f, 0.f, 0.f, 0.f};
  if (base1 != nullptr) {
    for (int t = 0; t < cnt; t++) {
      z1[t] = base1[base + e0 + t];
      if (rician) z2[t] = base2[base + e0 + t];
    }
  } else {
    // Philox blocks are indexed by the GLOBAL element index (base + e0) >> 2 so
    // that the stream equals tio_philox_normal over the whole tensor; rows are
    // 4-aligned whenever n_per_element % 4 == 0, otherwise take the slow path.
    if (((base + e0) & 3) == 0) {
      philox_normal4(seed, 0, static_cast<uint64_t>((base + e0) >> 2), z1);
      if (rician) philox_normal4(seed, 1, static_cast<uint64_t>((base + e0) >> 2), z2);
    } else {
      for (int t = 0; t < cnt; t++) {
        float z[4];
        const int64_t g = base + e0 + t;
        philox_normal4(seed, 0, static_cast<uint64_t>(g >> 2), z);
        z1[t] = z[g & 3];
        if (rician) {
          philox_normal4(seed, 1, static_cast<uint64_t>(g >> 2), z);
          z2[t] = z[g & 3];
        }
      }
    }
  }
  if constexpr (DT == TIO_F32) {
    // whole, 16-byte aligned quad: one vector load and one vector store per thread
    if (cnt == 4 && (((base + e0) & 3) == 0) && !rician &&
        (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0)) {
      const float4 v = *reinterpret_cast<const float4*>(static_cast<const float*>(x) + base + e0);
      float4 o;
      o.x = __fadd_rn(v.x, __fadd_rn(mu, __fmul_rn(sd, z1[0])));  // noise.py:178, :119
      o.y = __fadd_rn(v.y, __fadd_rn(mu, __fmul_rn(sd, z1[1])));
      o.z = __fadd_rn(v.z, __fadd_rn(mu, __fmul_rn(sd, z1[2])));
      o.w = __fadd_rn(v.w, __fadd_rn(mu, __fmul_rn(sd, z1[3])));
      *reinterpret_cast<float4*>(static_cast<float*>(y) + base + e0) = o;
      return;
    }
  }
  for (int t = 0; t < cnt; t++) {
    const int64_t idx = base + e0 + t;
    const float n1 = __fadd_rn(mu, __fmul_rn(sd, z1[t]));  // noise.py:178
    if constexpr (DT == TIO_F64) {
      const double v = static_cast<const double*>(x)[idx];
      if (rician) {
        const double n2 = static_cast<double>(__fadd_rn(mu, __fmul_rn(sd, z2[t])));
        const double s = v + static_cast<double>(n1);
        static_cast<double*>(y)[idx] = sqrt(s * s + n2 * n2);
      } else {
        static_cast<double*>(y)[idx] = v + static_cast<double>(n1);
      }
    } else {
      const float v = Elem<DT>::load(x, idx);
      float r;
      if (rician) {  // noise.py:117
        const float n2 = __fadd_rn(mu, __fmul_rn(sd, z2[t]));
        const float s = __fadd_rn(v, n1);
        r = sqrtf(__fadd_rn(__fmul_rn(s, s), __fmul_rn(n2, n2)));
      } else {
        r = __fadd_rn(v, n1);  // noise.py:119
      }
      Elem<DT>::store(y, idx, r);
    }
  }
}

// =============================================================================
// Gamma
// =============================================================================
template <int DT>
__global__ __launch_bounds__(kBlock) void gamma_kernel(const void* __restrict__ x, void* __restrict__ y,
                                                       int64_t n_per_element, float gamma,
                                                       const float* __restrict__ gamma_b, int params_batched) {
  const int b = blockIdx.y;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_per_element) return;
  const int64_t idx = static_cast<int64_t>(b) * n_per_element + i;
  const float gm = params_batched ? gamma_b[b] : gamma;
  if constexpr (DT == TIO_F64) {
    const double v = static_cast<const double*>(x)[idx];
    const double s = static_cast<double>((v > 0) - (v < 0));
    static_cast<double*>(y)[idx] = s * pow(fabs(v), static_cast<double>(gm));
  } else {
    const float v = Elem<DT>::load(x, idx);
    const float s = static_cast<float>((v > 0.0f) - (v < 0.0f));
    Elem<DT>::store(y, idx, __fmul_rn(s, powf(fabsf(v), gm)));  // gamma.py:90
  }
}

// =============================================================================
// Per-channel minimum of the first batch element (device-resident result)
// =============================================================================
// (float_to_key / key_to_float: common.hpp)

// keys (all ones) + tickets (zero) of tio_channel_min, one pair of arrays per (device, stream): set up once, every
// launch restores them.  Calls on one stream are ordered; different streams get different arrays.
uint32_t* min_workspace(hipStream_t s, int entries, int* cap_out, int kind) {
  struct Slot { int device; hipStream_t stream; int kind; uint32_t* ptr; int cap; };
  static std::mutex mu;
  static std::vector<Slot> slots;
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  Slot* found = nullptr;
  for (Slot& sl : slots)
    if (sl.device == device && sl.stream == s && sl.kind == kind) found = &sl;
  if (found != nullptr && found->cap >= entries) {
    *cap_out = found->cap;
    return found->ptr;
  }
  const int cap = entries < 1024 ? 1024 : entries;
  uint32_t* ptr = nullptr;
  if (hipMalloc(&ptr, sizeof(uint32_t) * 2 * cap) != hipSuccess) return nullptr;
  if (hipMemset(ptr, 0xFF, sizeof(uint32_t) * cap) != hipSuccess || hipMemset(ptr + cap, 0, sizeof(uint32_t) * cap) != hipSuccess ||
      hipDeviceSynchronize() != hipSuccess) {  // (the memsets run on the null stream: finished before any launch on `s` reads them)
    (void)hipFree(ptr);
    return nullptr;
  }
  if (found != nullptr) {
    // the predecessor is NOT freed: a concurrent caller on this stream may hold it and launch with it after we return
    found->ptr = ptr; found->cap = cap;
  } else {
    slots.push_back(Slot{device, s, kind, ptr, cap});
  }
  *cap_out = cap;
  return ptr;
}

// One launch: every block folds its minimum into keys[c] (ordered-integer image of the float), takes a ticket, and the
// block that draws the last ticket of its channel decodes the result into out[c] and leaves keys[c] / tickets[c] as it
// found them (all ones / zero) for the next call on this stream — no init launch before, no decode launch after.
template <int DT>
__global__ __launch_bounds__(kBlock) void min_reduce_kernel(const void* __restrict__ x, int64_t n_spatial,
                                                            uint32_t* __restrict__ keys, uint32_t* __restrict__ tickets,
                                                            float* __restrict__ out) {
  const int c = blockIdx.y;
  const int64_t base = static_cast<int64_t>(c) * n_spatial;
  uint32_t best = 0xFFFFFFFFu;
  bool vectorised = false;
  if constexpr (DT == TIO_F32) {
    // 16-byte loads, four independent ones in flight per lane (a 64 MiB channel in ~15 us)
    const float* p = static_cast<const float*>(x) + base;
    if ((n_spatial & 3) == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
      vectorised = true;
      const int64_t n4 = n_spatial >> 2, step = static_cast<int64_t>(gridDim.x) * blockDim.x;
      const float4* p4 = reinterpret_cast<const float4*>(p);
      int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
      for (; i + 7 * step < n4; i += 8 * step) {  // (eight loads in flight per lane: what a 512^3 volume needs to keep HBM busy from 2 - 4 blocks per CU)
        const float4 a0 = p4[i], a1 = p4[i + step], a2 = p4[i + 2 * step], a3 = p4[i + 3 * step];
        const float4 a4 = p4[i + 4 * step], a5 = p4[i + 5 * step], a6 = p4[i + 6 * step], a7 = p4[i + 7 * step];
        best = min(best, min(min(float_to_key(a0.x), float_to_key(a0.y)), min(float_to_key(a0.z), float_to_key(a0.w))));
        best = min(best, min(min(float_to_key(a1.x), float_to_key(a1.y)), min(float_to_key(a1.z), float_to_key(a1.w))));
        best = min(best, min(min(float_to_key(a2.x), float_to_key(a2.y)), min(float_to_key(a2.z), float_to_key(a2.w))));
        best = min(best, min(min(float_to_key(a3.x), float_to_key(a3.y)), min(float_to_key(a3.z), float_to_key(a3.w))));
        best = min(best, min(min(float_to_key(a4.x), float_to_key(a4.y)), min(float_to_key(a4.z), float_to_key(a4.w))));
        best = min(best, min(min(float_to_key(a5.x), float_to_key(a5.y)), min(float_to_key(a5.z), float_to_key(a5.w))));
        best = min(best, min(min(float_to_key(a6.x), float_to_key(a6.y)), min(float_to_key(a6.z), float_to_key(a6.w))));
        best = min(best, min(min(float_to_key(a7.x), float_to_key(a7.y)), min(float_to_key(a7.z), float_to_key(a7.w))));
      }
      for (; i + 3 * step < n4; i += 4 * step) {
        const float4 a0 = p4[i], a1 = p4[i + step], a2 = p4[i + 2 * step], a3 = p4[i + 3 * step];
        best = min(best, min(min(float_to_key(a0.x), float_to_key(a0.y)), min(float_to_key(a0.z), float_to_key(a0.w))));
        best = min(best, min(min(float_to_key(a1.x), float_to_key(a1.y)), min(float_to_key(a1.z), float_to_key(a1.w))));
        best = min(best, min(min(float_to_key(a2.x), float_to_key(a2.y)), min(float_to_key(a2.z), float_to_key(a2.w))));
        best = min(best, min(min(float_to_key(a3.x), float_to_key(a3.y)), min(float_to_key(a3.z), float_to_key(a3.w))));
      }
      for (; i < n4; i += step) {
        const float4 a0 = p4[i];
        best = min(best, min(min(float_to_key(a0.x), float_to_key(a0.y)), min(float_to_key(a0.z), float_to_key(a0.w))));
      }
    }
  }
  if (!vectorised) {
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n_spatial;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
      best = min(best, float_to_key(Elem<DT>::load(x, base + i)));
    }
  }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) best = min(best, static_cast<uint32_t>(__shfl_xor(static_cast<int>(best), s)));
  __shared__ uint32_t s_best[kBlock / 64];
  if ((threadIdx.x & 63) == 0) s_best[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t m = s_best[0];
    for (int w = 1; w < kBlock / 64; w++) m = min(m, s_best[w]);
    // (no fence: the ticket is drawn after the minimum's atomic has returned, both at agent scope — a `__threadfence()`
    // per block is an L2 write-back each and made this kernel 50 % slower)
    const uint32_t seen = __hip_atomic_fetch_min(&keys[c], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::"v"(seen) : "memory");
    if (__hip_atomic_fetch_add(&tickets[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {  // all the others have published
      out[c] = key_to_float(__hip_atomic_exchange(&keys[c], 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      __hip_atomic_exchange(&tickets[c], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// dtype dispatch helpers --------------------------------------------------------
#define TIO_DISPATCH_FLOAT(DTYPE, MACRO)                \
  switch (DTYPE) {                                      \
    case TIO_F32: MACRO(TIO_F32); break;                \
    case TIO_F64: MACRO(TIO_F64); break;                \
    case TIO_F16: MACRO(TIO_F16); break;                \
    case TIO_BF16: MACRO(TIO_BF16); break;              \
    default: break;                                     \
  }

#define TIO_DISPATCH_ALL(DTYPE, MACRO)                  \
  switch (DTYPE) {                                      \
    case TIO_F32: MACRO(TIO_F32); break;                \
    case TIO_F64: MACRO(TIO_F64); break;                \
    case TIO_F16: MACRO(TIO_F16); break;                \
    case TIO_BF16: MACRO(TIO_BF16); break;              \
    case TIO_U8: MACRO(TIO_U8); break;                  \
    case TIO_I8: MACRO(TIO_I8); break;                  \
    case TIO_I16: MACRO(TIO_I16); break;                \
    case TIO_I32: MACRO(TIO_I32); break;                \
    case TIO_I64: MACRO(TIO_I64); break;                \
    default: break;                                     \
  }

}  // namespace tio

using namespace tio;

extern "C" int tio_separable_conv3d(const void* x, void* y, void* tmp, int32_t dtype, int32_t batch,
                                    int32_t channels, const int32_t shape[3], const float* taps_dev,
                                    int32_t taps_batched, int32_t tap_stride, const int32_t radius[3],
                                    const uint8_t* skip_dev, void* stream) {
  if (batch == 0) return TIO_OK;  // an empty batch has no data pointers to speak of
  if (x == nullptr || y == nullptr || shape == nullptr || radius == nullptr)
    return fail(TIO_ERR_INVALID_ARGUMENT, "tio_separable_conv3d: null argument");
  if (!is_float_dtype(dtype)) return fail(TIO_ERR_UNSUPPORTED_DTYPE, "tio_separable_conv3d: dtype %d", dtype);
  if (batch < 0 || channels < 1 || shape[0] < 1 || shape[1] < 1 || shape[2] < 1)
    return fail(TIO_ERR_INVALID_ARGUMENT, "tio_separable_conv3d: bad shape");
  int n_active = 0;
  for (int a = 0; a < 3; a++) {
    if (radius[a] < 0 || 2 * radius[a] + 1 > tap_stride)
      return fail(TIO_ERR_INVALID_ARGUMENT, "tio_separable_conv3d: radius[%d]=%d does not fit tap_stride=%d", a,
                  radius[a], tap_stride);
    if (radius[a] > kMaxRadius) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_separable_conv3d: radius[%d]=%d exceeds %d", a, radius[a], kMaxRadius);
    if (radius[a] > 0) n_active++;
  }
  if (n_active > 0 && taps_dev == nullptr) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_separable_conv3d: null taps");
  const int64_t n = static_cast<int64_t>(shape[0]) * shape[1] * shape[2];
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (batch == 0) return TIO_OK;
  if (n_active == 0) {  // every sigma <= 0: identity (blur.py:144-145)
    if (hipMemcpyAsync(y, x, static_cast<size_t>(batch) * channels * n * dtype_size(dtype), hipMemcpyDeviceToDevice, s) !=
        hipSuccess)
      return fail(TIO_ERR_LAUNCH, "tio_separable_conv3d: copy failed");
    return TIO_OK;
  }
  if (n_active > 1 && tmp == nullptr)
    return fail(TIO_ERR_INVALID_ARGUMENT, "tio_separable_conv3d: tmp is required when more than one axis is active");
  float* tmp0 = static_cast<float*>(tmp);
  float* tmp1 = tmp0 != nullptr ? tmp0 + static_cast<int64_t>(batch) * channels * n : nullptr;
#define TIO_CONV(DT) \
  return launch_conv<DT>(x, y, tmp0, tmp1, batch, channels, shape, taps_dev, taps_batched, tap_stride, radius, skip_dev, s)
  TIO_DISPATCH_FLOAT(dtype, TIO_CONV)
#undef TIO_CONV
  return fail(TIO_ERR_UNSUPPORTED_DTYPE, "tio_separable_conv3d: dtype %d", dtype);
}

// =============================================================================
// Backward of the replicate-padded separable correlation (ABI 16)
// =============================================================================
// y[p] = sum_t w[t] x[clamp(p + t - r)] along one axis.  Its transpose as a GATHER (one thread per element of gx, lanes along K):
// gx[m] = sum_p gy[p] W(p, m), W(p, m) = the taps t with clamp(p + t - r) == m — for every p within r of m the one tap
// t = m - p + r, and on the two border voxels also every tap that was clamped onto them: m == 0 collects t < r - p of
// p = 0 .. r - 1, m == n - 1 collects t > n - 1 - p + r of p = n - r .. n - 1.  Off the hot path (nobody trains through a
// Gaussian blur in the pipeline of BASELINE.json): plain global loads, float32 accumulation in ascending p.
struct ConvAdjointArgs {
  const float* gy;
  float* gx;
  const float* taps;  // (1|B, 3, tap_stride)
  const uint8_t* skip;
  int64_t n_spatial, n_total, stride;
  int channels, extent, r, axis, tap_stride, taps_batched, J, K;
};

__global__ __launch_bounds__(kBlock) void conv_axis_adjoint_kernel(const ConvAdjointArgs a) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (e >= a.n_total) return;
  const int b = static_cast<int>(e / (a.n_spatial * a.channels));
  if (a.skip != nullptr && a.skip[b] != 0) {
    a.gx[e] = a.gy[e];
    return;
  }
  const int64_t v = e % a.n_spatial;
  const int k = static_cast<int>(v % a.K), j = static_cast<int>((v / a.K) % a.J), i = static_cast<int>(v / (static_cast<int64_t>(a.K) * a.J));
  const int m = a.axis == 0 ? i : (a.axis == 1 ? j : k);
  const int n = a.extent, r = a.r;
  const float* w = a.taps + (a.taps_batched ? static_cast<int64_t>(b) * 3 * a.tap_stride : 0) + static_cast<int64_t>(a.axis) * a.tap_stride;
  const float* line = a.gy + (e - static_cast<int64_t>(m) * a.stride);
  float acc = 0.0f;
  const int lo = max(0, m - r), hi = min(n - 1, m + r);
  for (int p = lo; p <= hi; p++) acc = __builtin_fmaf(w[m - p + r], line[static_cast<int64_t>(p) * a.stride], acc);
  if (m == 0) {  // taps clamped onto the first voxel
    for (int p = 0; p <= min(r - 1, n - 1); p++) {
      float clamped = 0.0f;
      for (int t = 0; t < r - p; t++) clamped += w[t];
      acc = __builtin_fmaf(clamped, line[static_cast<int64_t>(p) * a.stride], acc);
    }
  }
  if (m == n - 1) {  // ... onto the last
    for (int p = max(0, n - r); p <= n - 1; p++) {
      float clamped = 0.0f;
      for (int t = n - p + r; t <= 2 * r; t++) clamped += w[t];
      acc = __builtin_fmaf(clamped, line[static_cast<int64_t>(p) * a.stride], acc);
    }
  }
  a.gx[e] = acc;
}

extern "C" int tio_separable_conv3d_adjoint(const float* gy, float* gx, float* tmp, int32_t batch, int32_t channels,
                                            const int32_t shape[3], const float* taps_dev, int32_t taps_batched,
                                            int32_t tap_stride, const int32_t radius[3], const uint8_t* skip_dev, void* stream) {
  if (batch == 0) return TIO_OK;
  if (gy == nullptr || gx == nullptr || shape == nullptr || radius == nullptr)
    return fail(TIO_ERR_INVALID_ARGUMENT, "tio_separable_conv3d_adjoint: null argument");
  if (batch < 0 || channels < 1 || shape[0] < 1 || shape[1] < 1 || shape[2] < 1)
    return fail(TIO_ERR_INVALID_ARGUMENT, "tio_separable_conv3d_adjoint: bad shape");
  int n_active = 0;
  for (int a = 0; a < 3; a++) {
    if (radius[a] < 0 || 2 * radius[a] + 1 > tap_stride)
      return fail(TIO_ERR_INVALID_ARGUMENT, "tio_separable_conv3d_adjoint: radius[%d]=%d does not fit tap_stride=%d", a, radius[a], tap_stride);
    if (radius[a] > 0) n_active++;
  }
  if (n_active > 0 && taps_dev == nullptr) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_separable_conv3d_adjoint: null taps");
  if (n_active > 1 && tmp == nullptr)
    return fail(TIO_ERR_INVALID_ARGUMENT, "tio_separable_conv3d_adjoint: tmp is required when more than one axis is active");
  const int64_t n = static_cast<int64_t>(shape[0]) * shape[1] * shape[2];
  const int64_t total = static_cast<int64_t>(batch) * channels * n;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n_active == 0) {
    if (hipMemcpyAsync(gx, gy, static_cast<size_t>(total) * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess)
      return fail(TIO_ERR_LAUNCH, "tio_separable_conv3d_adjoint: copy failed");
    return TIO_OK;
  }
  if ((total + kBlock - 1) / kBlock > 0x7FFFFFFF) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_separable_conv3d_adjoint: tensor too large");
  // the forward pass runs I, J, K: the transposes run K, J, I; the last one writes gx
  const float* src = gy;
  int pass = 0;
  for (int axis = 2; axis >= 0; axis--) {
    if (radius[axis] <= 0) continue;
    float* dst = ((n_active - 1 - pass) % 2 == 0) ? gx : tmp;
    ConvAdjointArgs a;
    a.gy = src; a.gx = dst; a.taps = taps_dev; a.skip = skip_dev;
    a.n_spatial = n; a.n_total = total;
    a.stride = axis == 0 ? static_cast<int64_t>(shape[1]) * shape[2] : (axis == 1 ? shape[2] : 1);
    a.channels = channels; a.extent = shape[axis]; a.r = radius[axis]; a.axis = axis; a.tap_stride = tap_stride;
    a.taps_batched = taps_batched; a.J = shape[1]; a.K = shape[2];
    hipLaunchKernelGGL(conv_axis_adjoint_kernel, dim3(static_cast<unsigned>((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, a);
    src = dst;
    pass++;
  }
  return check_launch("tio_separable_conv3d_adjoint");
}

extern "C" int tio_blur_fused(const void* x, void* y, void* tmp, int32_t dtype, int32_t batch, int32_t channels,
                              const int32_t shape[3], const float* taps_dev, int32_t taps_batched, int32_t tap_stride,
                              const int32_t radius[3], const float* bias_coarse_dev, const int32_t bias_coarse_shape[3],
                              int32_t noise_on, float noise_mean, float noise_std, const float* noise_mean_dev,
                              const float* noise_std_dev, int32_t noise_batched, uint64_t philox_seed, const float* noise_base_dev,
                              int32_t fast_math, void* stream) {
  if (batch == 0) return TIO_OK;
  if (x == nullptr || y == nullptr || tmp == nullptr || shape == nullptr || radius == nullptr || taps_dev == nullptr)
    return fail(TIO_ERR_INVALID_ARGUMENT, "tio_blur_fused: null argument");
  if (dtype != TIO_F32) return TIO_ERR_UNSUPPORTED_CONFIG;
  if (batch < 0 || channels < 1) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_blur_fused: bad batch/channels");
  for (int d = 0; d < 3; d++) {
    if (shape[d] < 1) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_blur_fused: shapes must be >= 1");
    if (radius[d] < 0 || radius[d] > kMaxRadius || 2 * radius[d] + 1 > tap_stride)
      return fail(TIO_ERR_INVALID_ARGUMENT, "tio_blur_fused: bad radius");
  }
  if (bias_coarse_dev != nullptr) {
    if (bias_coarse_shape == nullptr) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_blur_fused: null coarse shape");
    for (int d = 0; d < 3; d++)
      if (bias_coarse_shape[d] < 1) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_blur_fused: bad coarse shape");
  }
  if (noise_on && noise_batched && (noise_mean_dev == nullptr || noise_std_dev == nullptr))
    return fail(TIO_ERR_INVALID_ARGUMENT, "tio_blur_fused: batched noise needs mean_dev and std_dev");
  if (noise_on < 0 || noise_on > 2) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_blur_fused: noise_on must be 0, 1 or 2");
  if (noise_on == 2 && noise_base_dev == nullptr) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_blur_fused: noise_on == 2 needs noise_base_dev");
  if (noise_on == 2 && (reinterpret_cast<uintptr_t>(noise_base_dev) & 15) != 0) return TIO_ERR_UNSUPPORTED_CONFIG;
  if (batch == 0) return TIO_OK;
  const int64_t n = static_cast<int64_t>(shape[0]) * shape[1] * shape[2];
  ConvFuse fuse;
  fuse.bias_coarse = bias_coarse_dev;
  if (bias_coarse_dev != nullptr)
    for (int d = 0; d < 3; d++) fuse.bias_shape[d] = bias_coarse_shape[d];
  fuse.noise_on = noise_on; fuse.noise_batched = noise_batched; fuse.noise_mean = noise_mean; fuse.noise_std = noise_std;
  fuse.noise_mean_b = noise_mean_dev; fuse.noise_std_b = noise_std_dev; fuse.noise_seed = philox_seed;
  fuse.noise_base = noise_on == 2 ? noise_base_dev : nullptr;
  fuse.fma = fast_math != 0;
  float* tmp0 = static_cast<float*>(tmp);
  float* tmp1 = tmp0 + static_cast<int64_t>(batch) * channels * n;
  const int status = launch_conv<TIO_F32>(x, y, tmp0, tmp1, batch, channels, shape, taps_dev, taps_batched, tap_stride, radius, nullptr,
                                          static_cast<hipStream_t>(stream), fuse);
  return status;
}

extern "C" int tio_bias_field_apply(const void* x, void* y, int32_t dtype, int32_t batch, int32_t channels,
                                    const int32_t shape[3], const float* coarse_dev,
                                    const int32_t coarse_shape[3], int32_t divide, const uint8_t* skip_dev,
                                    void* stream) {
  if (batch == 0) return TIO_OK;
  if (x == nullptr || y == nullptr || shape == nullptr || coarse_dev == nullptr || coarse_shape == nullptr)
    return fail(TIO_ERR_INVALID_ARGUMENT, "tio_bias_field_apply: null argument");
  if (!is_float_dtype(dtype)) return fail(TIO_ERR_UNSUPPORTED_DTYPE, "tio_bias_field_apply: dtype %d", dtype);
  for (int d = 0; d < 3; d++)
    if (shape[d] < 1 || coarse_shape[d] < 1) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_bias_field_apply: bad shape");
  if (batch < 0 || channels < 1) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_bias_field_apply: bad batch/channels");
  if (batch == 0) return TIO_OK;
  const int nc = coarse_shape[0] * coarse_shape[1] * coarse_shape[2];
  const size_t lds = nc <= kMaxCoarseLds ? static_cast<size_t>(nc) * sizeof(float) : 0;
  const int tiles_i = (shape[0] + kBiasTileI - 1) / kBiasTileI;
  const dim3 grid(static_cast<unsigned>((shape[2] + 63) / 64), static_cast<unsigned>((shape[1] + 3) / 4),
                  static_cast<unsigned>(tiles_i) * batch * channels);
  if (grid.z > 65535u || grid.y > 65535u) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_bias_field_apply: volume too large for one launch");
#define TIO_BIAS(DT)                                                                                              \
  hipLaunchKernelGGL((bias_kernel<DT>), grid, dim3(kBlock), lds, static_cast<hipStream_t>(stream), x, y, channels, \
                     shape[0], shape[1], shape[2], coarse_dev, coarse_shape[0], coarse_shape[1], coarse_shape[2], \
                     lerp_scale(coarse_shape[0], shape[0]), lerp_scale(coarse_shape[1], shape[1]),                \
                     lerp_scale(coarse_shape[2], shape[2]), divide, skip_dev, tiles_i)
  TIO_DISPATCH_FLOAT(dtype, TIO_BIAS)
#undef TIO_BIAS
  return check_launch("tio_bias_field_apply");
}

extern "C" int tio_add_noise(const void* x, void* y, int32_t dtype, int32_t batch, int64_t n_per_element,
                             float mean, float std, const float* mean_dev, const float* std_dev,
                             int32_t params_batched, int32_t rician, const float* base1_dev,
                             const float* base2_dev, uint64_t philox_seed, const uint8_t* keep_dev, void* stream) {
  if (batch == 0 || n_per_element == 0) return TIO_OK;
  if (x == nullptr || y == nullptr) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_add_noise: null argument");
  if (!is_float_dtype(dtype)) return fail(TIO_ERR_UNSUPPORTED_DTYPE, "tio_add_noise: dtype %d", dtype);
  if (batch < 0 || n_per_element < 0) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_add_noise: negative size");
  if (params_batched && (mean_dev == nullptr || std_dev == nullptr))
    return fail(TIO_ERR_INVALID_ARGUMENT, "tio_add_noise: params_batched needs mean_dev and std_dev");
  if (base1_dev != nullptr && rician && base2_dev == nullptr)
    return fail(TIO_ERR_INVALID_ARGUMENT, "tio_add_noise: rician with base1_dev needs base2_dev");
  if (batch == 0 || n_per_element == 0) return TIO_OK;
  const int64_t quads = (n_per_element + 3) / 4;
  const dim3 grid(static_cast<unsigned>((quads + kBlock - 1) / kBlock), static_cast<unsigned>(batch));
#define TIO_NOISE(DT)                                                                                          \
  hipLaunchKernelGGL((noise_kernel<DT>), grid, dim3(kBlock), 0, static_cast<hipStream_t>(stream), x, y,        \
                     n_per_element, mean, std, mean_dev, std_dev, params_batched, rician, base1_dev, base2_dev, \
                     philox_seed, keep_dev)
  TIO_DISPATCH_FLOAT(dtype, TIO_NOISE)
#undef TIO_NOISE
  return check_launch("tio_add_noise");
}

extern "C" int tio_philox_normal(float* out_dev, int64_t n, uint64_t philox_seed, int32_t stream_id, void* stream) {
  if (out_dev == nullptr || n < 0) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_philox_normal: bad argument");
  if (n == 0) return TIO_OK;
  const int64_t quads = (n + 3) / 4;
  hipLaunchKernelGGL(philox_normal_kernel, dim3(static_cast<unsigned>((quads + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), out_dev, n, philox_seed, stream_id);
  return check_launch("tio_philox_normal");
}

extern "C" int tio_gamma_pow(const void* x, void* y, int32_t dtype, int32_t batch, int64_t n_per_element,
                             float gamma, const float* gamma_dev, int32_t params_batched, void* stream) {
  if (batch == 0 || n_per_element == 0) return TIO_OK;
  if (x == nullptr || y == nullptr) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_gamma_pow: null argument");
  if (!is_float_dtype(dtype)) return fail(TIO_ERR_UNSUPPORTED_DTYPE, "tio_gamma_pow: dtype %d", dtype);
  if (batch < 0 || n_per_element < 0) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_gamma_pow: negative size");
  if (params_batched && gamma_dev == nullptr) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_gamma_pow: null gamma_dev");
  if (batch == 0 || n_per_element == 0) return TIO_OK;
  const dim3 grid(static_cast<unsigned>((n_per_element + kBlock - 1) / kBlock), static_cast<unsigned>(batch));
#define TIO_GAMMA(DT)                                                                                      \
  hipLaunchKernelGGL((gamma_kernel<DT>), grid, dim3(kBlock), 0, static_cast<hipStream_t>(stream), x, y,    \
                     n_per_element, gamma, gamma_dev, params_batched)
  TIO_DISPATCH_FLOAT(dtype, TIO_GAMMA)
#undef TIO_GAMMA
  return check_launch("tio_gamma_pow");
}

extern "C" int tio_channel_min(const void* x, int32_t dtype, int32_t channels, int64_t n_spatial, float* out_dev,
                               void* stream) {
  if (x == nullptr || out_dev == nullptr) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_channel_min: null argument");
  if (dtype_size(dtype) == 0) return fail(TIO_ERR_UNSUPPORTED_DTYPE, "tio_channel_min: dtype %d", dtype);
  if (channels < 1 || n_spatial < 1) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_channel_min: empty input");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int ws_cap = 0;
  uint32_t* keys = min_workspace(s, channels, &ws_cap);
  if (keys == nullptr) return fail(TIO_ERR_LAUNCH, "tio_channel_min: cannot allocate the reduction workspace");
  uint32_t* tickets = keys + ws_cap;
  const int64_t want = (n_spatial + kBlock - 1) / kBlock;
  int64_t cap = 512;  // two blocks per CU: measured 18 us per 64 MiB channel (2048 blocks: 34 us, the atomics and block tails add up)
  if (env_switches().min_blocks > 0) cap = env_switches().min_blocks;  // experiments
  const unsigned gx = static_cast<unsigned>(want < cap ? want : cap);
#define TIO_MIN(DT) \
  hipLaunchKernelGGL((min_reduce_kernel<DT>), dim3(gx, static_cast<unsigned>(channels)), dim3(kBlock), 0, s, x, n_spatial, keys, tickets, out_dev)
  TIO_DISPATCH_ALL(dtype, TIO_MIN)
#undef TIO_MIN
  return check_launch("tio_channel_min");
}
