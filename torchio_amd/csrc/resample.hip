// resample.hip — fused spatial resampling for gfx950 (tio_resample3d).
//
// One launch replaces the reference's grid construction + two grid_sample calls
// (SURVEY.md §2.2 K1–K7; reference spatial.py:1504-1648, 1695-1731, 2171-2189):
// every output voxel computes its source coordinate in registers — affine 3x4
// as the same forward-FMA chain MKL's sgemm produces, optional trilinear lookup
// of the elastic control points staged in LDS, the redundant normalise /
// un-normalise round trip of F.grid_sample — then gathers 8 taps (or 1 for
// nearest), accumulates the in-bounds weight mask in the same order as ATen and
// applies `mask > 0.5 ? value : fill`.  No (I,J,K,3) grid ever reaches HBM.
//
// HBM-bound by design (algorithmic traffic = read input once + write output
// once); no MFMA: this is a gather/stencil op.  What actually limits it is the
// per-voxel float32 instruction count needed to stay bit-identical with the
// reference, so the kernel is organised around (1) latency hiding — one block
// walks kTileI output slabs so the control points are staged once per 2048
// voxels and many independent gathers are in flight, (2) a wave-uniform
// "interior" path (all 8 taps of all 64 lanes in bounds: no predication, no
// mask arithmetic — the mask is exactly > 0.5 there), (3) IEEE-exact division
// by reciprocal + two FMA refinements instead of the 12-instruction expansion.
#include <functional>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>
#include <algorithm>

#include "common.hpp"

namespace tio {

struct ImgArgs {
  const void* in;
  void* out;
  const float* fill;  // nullptr → no mask step
  int channels;
  int dtype;
  int interp;
  // TIO_LABEL_PV only: sorted label table (may be null), its length, the pad label
  const double* labels;
  int n_labels;
  double pad_label;
  // the folded minimum (planned FAST launches): per-channel result, and this image's first key in the workspace
  float* out_min;
  uint32_t* min_keys;
};

struct ResampleArgs {
  int B;
  int I, J, K;
  int Io, Jo, Ko;
  int affine_first;
  const float* mapping;
  int mapping_batched;
  const float* cp;
  int cp_batched;
  int ni, nj, nk;
  const uint8_t* cp_skip;
  const uint8_t* passthrough;
  float sp[3], rsp[3];              // mm → voxel spacing and its float32 reciprocal
  int unit_spacing;                 // all three spacings == 1.0f: d / 1 is the identity
  float scale_i, scale_j, scale_k;  // ATen lerp scales of the control grid
  float den[3], rden[3];            // max(S-1,1) per input axis and reciprocal
  float size_m1[3];                 // S-1 per input axis
  float dh[3], rdh[3], half_h[3];   // den/2, its reciprocal, (S-1)/2: the folded normalise round trip
  int short_div;                    // every den <= 8192: one division refinement is exact
  int any_linear, any_nearest;      // which coordinate products are needed at all
  int n_images;
  ImgArgs img[TIO_MAX_IMAGES];
  int tiles_k, tiles_j, tiles_i;
  unsigned magic_k, magic_j, magic_i;  // multiply-high reciprocals of the tile counts (tile kernel)
  int cp_lds;    // floats of LDS reserved for the control points (0 = read them from global)
  int tile_cap;  // floats of LDS available for one staged input brick (tile kernel)
  int ablate;    // profiling only (TIO_TILE_ABLATE): 1 = no staging, 2 = no sampling, 4 = trivial coordinates
  int dma_packed;  // planned bricks: DMA instructions cover rows across x-plane boundaries (A/B: TIO_DMA_PACKED=0)
  int any_fill;      // an image of the launch has a fill rule
  int plan_multi;    // plan_bricks_kernel: bricks whose box exceeds the tile get pass boxes over halves / quarters of their planes (the exact-coordinate lean kernel reads them)
  int fill_recheck;  // FAST kernels: voxels whose in-bounds weight is within a margin of 1/2 take the exact chain's decision (A/B: TIO_FAST_FILL_RECHECK=0)
};

constexpr int kTileI = 8;          // output slabs walked by one block
constexpr int kRowsPerBlock = 4;   // one wave per output row (jo)
constexpr int kLanes = 64;         // contiguous ko per wave → coalesced stores
constexpr int kMaxCpLds = 6144;    // floats of control points staged in LDS (24 KiB)
constexpr int kLdsFloatsPerCU = 40960;   // 160 KiB
constexpr int kPlannedMinBricks = 12288;  // below: one kernel with in-kernel boxes (the plan costs a launch)
constexpr int kTileMinCap = 6144;       // the per-voxel fallback parks 8 planes x 3 coordinates x 256 threads there
constexpr int kTileBlocksPerCU = 3;       // resident blocks the default LDS budget is sized for

// IEEE-754 correctly rounded n / d from r = RN(1/d): q0 = RN(n r), two Markstein
// refinements (each: exact remainder by FMA, correction by FMA).  Checked
// bit-for-bit against the hardware division on 1.9e9 operands (integer
// divisors 1..2100, spacings in [0.05, 20], exact-multiple neighbourhoods).
__device__ __forceinline__ float exact_div(float n, float d, float r) {
  float q = __fmul_rn(n, r);
  float e = __builtin_fmaf(-d, q, n);
  q = __builtin_fmaf(e, r, q);
  e = __builtin_fmaf(-d, q, n);
  return __builtin_fmaf(e, r, q);
}

// g = 2 v / max(S-1,1) - 1 (spatial.py:1638-1646) followed by ATen's
// grid_sampler_unnormalize(align_corners=True): ((g + 1) / 2) * (S - 1).
__device__ __forceinline__ float normalise_roundtrip(float v, float den, float rden, float size_m1) {
  const float g = __fsub_rn(exact_div(__fmul_rn(2.0f, v), den, rden), 1.0f);
  return __fmul_rn(__fmul_rn(__fadd_rn(g, 1.0f), 0.5f), size_m1);
}

// trilinear lookup of the three displacement components from the (ni,nj,nk,3)
// field; nesting and rounding exactly as ATen's upsample_trilinear3d (K innermost)
struct Disp {
  float i, j, k;
};

__device__ __forceinline__ Disp cp_trilerp3(const float* __restrict__ cp, int s_i, int s_j, const Lerp1D& li,
                                            const Lerp1D& lj, const Lerp1D& lk) {
  const float* p00 = cp + li.i0 * s_i + lj.i0 * s_j;
  const float* p01 = cp + li.i0 * s_i + lj.i1 * s_j;
  const float* p10 = cp + li.i1 * s_i + lj.i0 * s_j;
  const float* p11 = cp + li.i1 * s_i + lj.i1 * s_j;
  const int k0 = lk.i0 * 3, k1 = lk.i1 * 3;
  float r[3];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float a00 = lerp2(p00[k0 + c], lk.l0, p00[k1 + c], lk.l1);
    const float a01 = lerp2(p01[k0 + c], lk.l0, p01[k1 + c], lk.l1);
    const float a10 = lerp2(p10[k0 + c], lk.l0, p10[k1 + c], lk.l1);
    const float a11 = lerp2(p11[k0 + c], lk.l0, p11[k1 + c], lk.l1);
    const float b0 = lerp2(a00, lj.l0, a01, lj.l1);
    const float b1 = lerp2(a10, lj.l0, a11, lj.l1);
    r[c] = lerp2(b0, li.l0, b1, li.l1);
  }
  return Disp{r[0], r[1], r[2]};
}

// ---- sampling --------------------------------------------------------------------
// Interior path: every tap of every lane of the wave is in bounds, so there is no
// predication, no mask arithmetic (the in-bounds weight sum is 1 up to rounding,
// always > 0.5) and the 8 offsets are base + launch constants.
template <int DT>
__device__ __forceinline__ void sample_interior(const ImgArgs& g, int b, int64_t n_in, int64_t n_out, int64_t o_idx,
                                                const float (&w)[8], int base, int dJK, int dK, int offn) {
  for (int c = 0; c < g.channels; c++) {
    const int64_t bc = static_cast<int64_t>(b) * g.channels + c;
    const typename Elem<DT>::type* p = static_cast<const typename Elem<DT>::type*>(g.in) + bc * n_in;
    float val;
    if (g.interp == TIO_LINEAR_ADJOINT) {  // backward of TIO_LINEAR: scatter g * w to the 8 taps (all in bounds here)
      if constexpr (DT == TIO_F32) {
        const float gv = static_cast<const float*>(g.out)[bc * n_out + o_idx];
        float* q = const_cast<float*>(static_cast<const float*>(g.in)) + bc * n_in + base;
        unsafeAtomicAdd(q, gv * w[0]); unsafeAtomicAdd(q + dJK, gv * w[1]);
        unsafeAtomicAdd(q + dK, gv * w[2]); unsafeAtomicAdd(q + dJK + dK, gv * w[3]);
        unsafeAtomicAdd(q + 1, gv * w[4]); unsafeAtomicAdd(q + dJK + 1, gv * w[5]);
        unsafeAtomicAdd(q + dK + 1, gv * w[6]); unsafeAtomicAdd(q + dJK + dK + 1, gv * w[7]);
      }
      continue;
    }
    if (g.interp == TIO_LINEAR) {
      const typename Elem<DT>::type* q = p + base;
      const float v0 = Elem<DT>::load(q, 0), v1 = Elem<DT>::load(q, dJK);
      const float v2 = Elem<DT>::load(q, dK), v3 = Elem<DT>::load(q, dJK + dK);
      const float v4 = Elem<DT>::load(q, 1), v5 = Elem<DT>::load(q, dJK + 1);
      const float v6 = Elem<DT>::load(q, dK + 1), v7 = Elem<DT>::load(q, dJK + dK + 1);
      val = __fadd_rn(0.0f, __fmul_rn(v0, w[0]));  // keep ATen's `0 + v*w` (sign of zero)
      val = __fadd_rn(val, __fmul_rn(v1, w[1]));
      val = __fadd_rn(val, __fmul_rn(v2, w[2]));
      val = __fadd_rn(val, __fmul_rn(v3, w[3]));
      val = __fadd_rn(val, __fmul_rn(v4, w[4]));
      val = __fadd_rn(val, __fmul_rn(v5, w[5]));
      val = __fadd_rn(val, __fmul_rn(v6, w[6]));
      val = __fadd_rn(val, __fmul_rn(v7, w[7]));
    } else {
      val = Elem<DT>::load(p, offn);
    }
    Elem<DT>::store(g.out, bc * n_out + o_idx, val);
  }
}

// Boundary path: per-tap bounds, zero padding, in-bounds weight mask and fill, in
// exactly ATen's accumulation order (tnw,tne,tsw,tse,bnw,bne,bsw,bse).
template <int DT>
__device__ __forceinline__ void sample_boundary(const ImgArgs& g, int b, int64_t n_in, int64_t n_out, int64_t o_idx,
                                                const float (&w)[8], const int (&off)[8], unsigned okbits, float mask,
                                                int offn, bool okn) {
  for (int c = 0; c < g.channels; c++) {
    const int64_t bc = static_cast<int64_t>(b) * g.channels + c;
    const typename Elem<DT>::type* p = static_cast<const typename Elem<DT>::type*>(g.in) + bc * n_in;
    float val;
    if (g.interp == TIO_LINEAR_ADJOINT) {  // backward of TIO_LINEAR with per-tap bounds; no gradient where the fill was taken
      if constexpr (DT == TIO_F32) {
        if (g.fill == nullptr || mask > 0.5f) {
          const float gv = static_cast<const float*>(g.out)[bc * n_out + o_idx];
          float* q = const_cast<float*>(static_cast<const float*>(g.in)) + bc * n_in;
#pragma unroll
          for (int k = 0; k < 8; k++)
            if ((okbits >> k) & 1u) unsafeAtomicAdd(q + off[k], gv * w[k]);
        }
      }
      continue;
    }
    if (g.interp == TIO_LINEAR) {
      val = 0.0f;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const float v = Elem<DT>::load(p, off[k]);
        const float next = __fadd_rn(val, __fmul_rn(v, w[k]));
        val = ((okbits >> k) & 1u) ? next : val;
      }
    } else {
      const float v = Elem<DT>::load(p, offn);
      val = okn ? v : 0.0f;
    }
    if (g.fill != nullptr) val = (mask > 0.5f) ? val : g.fill[c];
    Elem<DT>::store(g.out, bc * n_out + o_idx, val);
  }
}

}  // namespace tio
#include "resample_label.hpp"
namespace tio {

// DTMODE selects which element types a kernel instantiation can sample, so that the
// common launches do not pay registers for the rare ones: 0 = float32 only,
// 1 = float32 + {int16, uint8, int32} (intensity + label maps), 2 = every tio_dtype.
#define TIO_DISPATCH_IMAGE(DTMODE, DTYPE, CALL)                 \
  if constexpr (DTMODE == 0) {                                  \
    CALL(TIO_F32);                                              \
  } else {                                                      \
    switch (DTYPE) { /* uniform → scalar branch */              \
      case TIO_F32: CALL(TIO_F32); break;                       \
      case TIO_I16: CALL(TIO_I16); break;                       \
      case TIO_U8: CALL(TIO_U8); break;                         \
      case TIO_I32: CALL(TIO_I32); break;                       \
      default:                                                  \
        if constexpr (DTMODE == 2) {                            \
          switch (DTYPE) {                                      \
            case TIO_I64: CALL(TIO_I64); break;                 \
            case TIO_F16: CALL(TIO_F16); break;                 \
            case TIO_BF16: CALL(TIO_BF16); break;               \
            case TIO_F64: CALL(TIO_F64); break;                 \
            default: CALL(TIO_I8); break;                       \
          }                                                     \
        }                                                       \
        break;                                                  \
    }                                                           \
  }

// LABEL_PV = true: the same coordinates, every image resampled in the "label"
// partial-volume mode (resample_label.hpp); launched separately so that the intensity
// kernels carry none of its registers.
// B-spline orders 2 / 3 (include/tio_hip.h: TIO_QUADRATIC): same operations, same order as oracle/tio_oracle.c
__device__ __forceinline__ int spline_reflect(int i, int n) {
  const int n2 = 2 * n;
  if (i < 0) i = -i - 1;
  i %= n2;
  return i >= n ? n2 - i - 1 : i;
}

__device__ __forceinline__ void spline_weights(float x, int order, int& low, float (&w)[4]) {
  if (order == 2) {
    const float c = floorf(__fadd_rn(x, 0.5f));
    const float t = __fsub_rn(x, c);
    low = static_cast<int>(c) - 1;
    const float m = __fsub_rn(0.5f, t), p = __fadd_rn(0.5f, t);
    w[0] = __fmul_rn(__fmul_rn(0.5f, m), m);
    w[1] = __fsub_rn(0.75f, __fmul_rn(t, t));
    w[2] = __fmul_rn(__fmul_rn(0.5f, p), p);
    w[3] = 0.0f;
  } else {
    const float f = floorf(x);
    const float t = __fsub_rn(x, f);
    const float u = __fsub_rn(1.0f, t);
    low = static_cast<int>(f) - 1;
    w[0] = __fdiv_rn(__fmul_rn(__fmul_rn(u, u), u), 6.0f);
    w[1] = __fdiv_rn(__fadd_rn(__fmul_rn(__fmul_rn(__fmul_rn(t, t), __fsub_rn(t, 2.0f)), 3.0f), 4.0f), 6.0f);
    w[2] = __fdiv_rn(__fadd_rn(__fmul_rn(__fmul_rn(__fmul_rn(u, u), __fsub_rn(u, 2.0f)), 3.0f), 4.0f), 6.0f);
    w[3] = __fdiv_rn(__fmul_rn(__fmul_rn(t, t), t), 6.0f);
  }
}

// Orders 4 - 7 (oracle/tio_oracle.c: spline_weights_high, operation for operation): the Cox - de Boor recursion of the
// uniform B-spline in float64, v_j = N_m(tau + j), every term positive; tap k at low + k weighs v_{order - k}.
template <int ORDER>
__device__ __forceinline__ void spline_weights_high(float x, int& low, float (&w)[ORDER + 1]) {
  constexpr bool odd = (ORDER & 1) != 0;
  const float base = odd ? floorf(x) : floorf(__fadd_rn(x, 0.5f));
  const double tau = odd ? __dsub_rn(static_cast<double>(x), static_cast<double>(base))
                         : __dadd_rn(__dsub_rn(static_cast<double>(x), static_cast<double>(base)), 0.5);
  low = static_cast<int>(base) - (odd ? (ORDER - 1) / 2 : ORDER / 2);
  double v[ORDER + 1];
  v[0] = 1.0;
#pragma unroll
  for (int j = 1; j <= ORDER; j++) v[j] = 0.0;
#pragma unroll
  for (int m = 2; m <= ORDER + 1; m++) {
    const double inv = __ddiv_rn(1.0, static_cast<double>(m - 1));
#pragma unroll
    for (int j = ORDER; j >= 0; j--) {
      if (j <= m - 1) {
        const double same = j <= m - 2 ? v[j] : 0.0, below = j >= 1 ? v[j - 1] : 0.0;
        v[j] = __dmul_rn(__dadd_rn(__dmul_rn(__dadd_rn(tau, static_cast<double>(j)), same),
                                   __dmul_rn(__dsub_rn(__dsub_rn(static_cast<double>(m), tau), static_cast<double>(j)), below)), inv);
      }
    }
  }
#pragma unroll
  for (int k = 0; k <= ORDER; k++) w[k] = static_cast<float>(v[ORDER - k]);
}

template <int ORDER>
__device__ __forceinline__ float spline_sample_high(const float* __restrict__ coef, int I, int J, int K, float vi, float vj, float vk) {
  const float tiny = 5e-2f;
  if (!((vi > -tiny) & (vi < __fadd_rn(static_cast<float>(I - 1), tiny)) & (vj > -tiny) & (vj < __fadd_rn(static_cast<float>(J - 1), tiny)) &
        (vk > -tiny) & (vk < __fadd_rn(static_cast<float>(K - 1), tiny))))
    return 0.0f;
  int li, lj, lk;
  float wi[ORDER + 1], wj[ORDER + 1], wk[ORDER + 1];
  spline_weights_high<ORDER>(vi, li, wi);
  spline_weights_high<ORDER>(vj, lj, wj);
  spline_weights_high<ORDER>(vk, lk, wk);
  int kc[ORDER + 1];
#pragma unroll
  for (int r = 0; r <= ORDER; r++) kc[r] = spline_reflect(lk + r, K);
  float val = 0.0f;
#pragma unroll 1
  for (int p = 0; p <= ORDER; p++) {
    const int64_t ia = spline_reflect(li + p, I);
#pragma unroll 1
    for (int q = 0; q <= ORDER; q++) {
      const int64_t jb = spline_reflect(lj + q, J);
      // (wi / wj indexed by loop counters that are not unrolled: selected from registers, no scratch)
      float wp = wi[0], wq = wj[0];
#pragma unroll
      for (int e = 1; e <= ORDER; e++) { wp = p == e ? wi[e] : wp; wq = q == e ? wj[e] : wq; }
      const float wab = __fmul_rn(wp, wq);
      const float* row = coef + (ia * J + jb) * K;
#pragma unroll
      for (int r = 0; r <= ORDER; r++) {
        const float wabc = __fmul_rn(wab, wk[r]);
        val = __fadd_rn(val, __fmul_rn(wabc, row[kc[r]]));
      }
    }
  }
  return val;
}

__device__ __forceinline__ float spline_sample(const float* __restrict__ coef, int I, int J, int K, float vi, float vj, float vk, int order) {
  if (order == 4) return spline_sample_high<4>(coef, I, J, K, vi, vj, vk);
  if (order == 5) return spline_sample_high<5>(coef, I, J, K, vi, vj, vk);
  if (order == 6) return spline_sample_high<6>(coef, I, J, K, vi, vj, vk);
  if (order == 7) return spline_sample_high<7>(coef, I, J, K, vi, vj, vk);
  const float tiny = 5e-2f;
  if (!((vi > -tiny) & (vi < __fadd_rn(static_cast<float>(I - 1), tiny)) & (vj > -tiny) & (vj < __fadd_rn(static_cast<float>(J - 1), tiny)) &
        (vk > -tiny) & (vk < __fadd_rn(static_cast<float>(K - 1), tiny))))
    return 0.0f;
  int li, lj, lk;
  float wi[4], wj[4], wk[4];
  spline_weights(vi, order, li, wi);
  spline_weights(vj, order, lj, wj);
  spline_weights(vk, order, lk, wk);
  float val = 0.0f;
  for (int p = 0; p <= order; p++) {
    const int64_t ia = spline_reflect(li + p, I);
    for (int q = 0; q <= order; q++) {
      const int64_t jb = spline_reflect(lj + q, J);
      const float wab = __fmul_rn(wi[p], wj[q]);
      for (int r = 0; r <= order; r++) {
        const int64_t kc = spline_reflect(lk + r, K);
        const float wabc = __fmul_rn(wab, wk[r]);
        val = __fadd_rn(val, __fmul_rn(wabc, coef[(ia * J + jb) * K + kc]));
      }
    }
  }
  return val;
}

// MODE: 0 = nearest / trilinear images, 1 = "label" partial-volume images, 2 = B-spline images (coefficients in, float32 out)
template <bool ELASTIC_POSSIBLE, int DTMODE, int MODE = 0>
__global__ __launch_bounds__(kRowsPerBlock* kLanes) void resample_kernel(const ResampleArgs a) {
  constexpr bool LABEL_PV = MODE == 1;
  extern __shared__ __attribute__((aligned(16))) float s_cp[];

  // tile decode: XCD-contiguous chunks of (b, it, jt, kt), kt fastest
  const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
  const int kt = tile % a.tiles_k;
  const unsigned t1 = tile / a.tiles_k;
  const int jt = t1 % a.tiles_j;
  const unsigned t2 = t1 / a.tiles_j;
  const int it = t2 % a.tiles_i;
  const int b = t2 / a.tiles_i;

  const int lane = threadIdx.x & (kLanes - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kLanes);
  const int jo = jt * kRowsPerBlock + wave;
  const int ko = kt * kLanes + lane;
  const int i_begin = it * kTileI;
  const int i_end = min(i_begin + kTileI, a.Io);

  const bool pass = a.passthrough != nullptr && a.passthrough[b] != 0;
  bool elastic = false;
  bool cp_in_lds = false;
  const float* cp_global = nullptr;
  if constexpr (ELASTIC_POSSIBLE) {
    elastic = !(a.cp_skip != nullptr && a.cp_skip[b] != 0);
    if (elastic && !pass) {
      const int n_cp = a.ni * a.nj * a.nk * 3;
      cp_global = a.cp + (a.cp_batched ? static_cast<int64_t>(b) * n_cp : 0);
      if (n_cp <= kMaxCpLds) {
        for (int t = threadIdx.x; t < n_cp; t += blockDim.x) s_cp[t] = cp_global[t];
        __syncthreads();
        cp_in_lds = true;
      }
    }
  }
  if (jo >= a.Jo || ko >= a.Ko) return;

  const int64_t n_in = static_cast<int64_t>(a.I) * a.J * a.K;
  const int64_t n_out = static_cast<int64_t>(a.Io) * a.Jo * a.Ko;
  const int64_t row = static_cast<int64_t>(jo) * a.Ko + ko;
  const int64_t slab = static_cast<int64_t>(a.Jo) * a.Ko;

  if (pass) {  // gated-out element: bit-exact copy (spatial.py:1101-1106)
    for (int io = i_begin; io < i_end; io++) {
      const int64_t o_idx = io * slab + row;
      for (int im = 0; im < a.n_images; im++) {
        const ImgArgs& g = a.img[im];
        const int es = dtype_size(g.dtype);
        for (int c = 0; c < g.channels; c++) {
          const int64_t off = (static_cast<int64_t>(b) * g.channels + c) * n_out + o_idx;
          if (g.interp == TIO_LINEAR_ADJOINT) {
            // backward of the bit-exact copy: the identity.  `in` is the gradient accumulator, `out` the incoming
            // gradient, which is only ever READ in this mode (every voxel is visited once: no atomic needed)
            float* acc = const_cast<float*>(static_cast<const float*>(g.in));
            acc[off] = __fadd_rn(acc[off], static_cast<const float*>(g.out)[off]);
            continue;
          }
          const char* s = static_cast<const char*>(g.in) + off * es;
          char* d = static_cast<char*>(g.out) + off * es;
          for (int e = 0; e < es; e++) d[e] = s[e];
        }
      }
    }
    return;
  }

  const float* m = a.mapping + (a.mapping_batched ? b * 12 : 0);
  const float m00 = m[0], m01 = m[1], m02 = m[2], m03 = m[3];
  const float m10 = m[4], m11 = m[5], m12 = m[6], m13 = m[7];
  const float m20 = m[8], m21 = m[9], m22 = m[10], m23 = m[11];
  const float cj = static_cast<float>(jo), ck = static_cast<float>(ko);

  // k- and j-dependent control-grid lerp terms are loop invariants
  Lerp1D lj{0, 0, 1.0f, 0.0f}, lk{0, 0, 1.0f, 0.0f};
  int s_i = 0, s_j = 0;
  if constexpr (ELASTIC_POSSIBLE) {
    if (elastic) {
      lj = lerp_index(jo, a.nj, a.Jo, a.scale_j);
      lk = lerp_index(ko, a.nk, a.Ko, a.scale_k);
      s_i = a.nj * a.nk * 3;
      s_j = a.nk * 3;
    }
  }
  const int dJK = a.J * a.K, dK = a.K;
  const float hx = a.size_m1[0], hy = a.size_m1[1], hz = a.size_m1[2];

  for (int io = i_begin; io < i_end; io++) {
    const float ci = static_cast<float>(io);
    float vi, vj, vk;
    // [c,1] @ M^T per row: a*m0, then fma(b,m1,.), fma(c,m2,.), fma(1,m3,.) — the
    // rounding sequence of MKL sgemm (K = 4), pinned against the reference.
#define TIO_AFFINE_ROW(M0, M1, M2, M3, A, B, C) \
  __builtin_fmaf(1.0f, M3, __builtin_fmaf(C, M2, __builtin_fmaf(B, M1, __fmul_rn(A, M0))))
    bool done = false;
    if constexpr (ELASTIC_POSSIBLE) {
      if (elastic) {
        const Lerp1D li = lerp_index(io, a.ni, a.Io, a.scale_i);
        const Disp d = cp_in_lds ? cp_trilerp3(s_cp, s_i, s_j, li, lj, lk) : cp_trilerp3(cp_global, s_i, s_j, li, lj, lk);
        float di = d.i, dj = d.j, dk = d.k;
        if (!a.unit_spacing) {  // mm → voxels: true division by the spacing
          di = exact_div(di, a.sp[0], a.rsp[0]);
          dj = exact_div(dj, a.sp[1], a.rsp[1]);
          dk = exact_div(dk, a.sp[2], a.rsp[2]);
        }
        if (a.affine_first) {  // spatial.py:1570-1573
          vi = __fadd_rn(TIO_AFFINE_ROW(m00, m01, m02, m03, ci, cj, ck), di);
          vj = __fadd_rn(TIO_AFFINE_ROW(m10, m11, m12, m13, ci, cj, ck), dj);
          vk = __fadd_rn(TIO_AFFINE_ROW(m20, m21, m22, m23, ci, cj, ck), dk);
        } else {  // spatial.py:1574-1577
          const float ei = __fadd_rn(ci, di), ej = __fadd_rn(cj, dj), ek = __fadd_rn(ck, dk);
          vi = TIO_AFFINE_ROW(m00, m01, m02, m03, ei, ej, ek);
          vj = TIO_AFFINE_ROW(m10, m11, m12, m13, ei, ej, ek);
          vk = TIO_AFFINE_ROW(m20, m21, m22, m23, ei, ej, ek);
        }
        done = true;
      }
    }
    if (!done) {  // spatial.py:1542-1543
      vi = TIO_AFFINE_ROW(m00, m01, m02, m03, ci, cj, ck);
      vj = TIO_AFFINE_ROW(m10, m11, m12, m13, ci, cj, ck);
      vk = TIO_AFFINE_ROW(m20, m21, m22, m23, ci, cj, ck);
    }
#undef TIO_AFFINE_ROW
    if constexpr (MODE == 2) {  // grid_pull takes the voxel coordinates as they are (spatial.py:1749-1760)
      const int64_t o_idx = io * slab + row;
      for (int im = 0; im < a.n_images; im++) {
        const ImgArgs& g = a.img[im];
        for (int c = 0; c < g.channels; c++) {
          const int64_t bc = static_cast<int64_t>(b) * g.channels + c;
          static_cast<float*>(g.out)[bc * n_out + o_idx] =
              spline_sample(static_cast<const float*>(g.in) + bc * n_in, a.I, a.J, a.K, vi, vj, vk, TIO_BSPLINE_ORDER(g.interp));
        }
      }
      continue;
    }
    // torchio axis i ≡ grid x ≡ ATen W ; j ≡ y ≡ H ; k ≡ z ≡ D
    const float x = normalise_roundtrip(vi, a.den[0], a.rden[0], hx);
    const float y = normalise_roundtrip(vj, a.den[1], a.rden[1], hy);
    const float z = normalise_roundtrip(vk, a.den[2], a.rden[2], hz);

    // ATen grid_sampler_3d corner weights (computed even for nearest data when a fill
    // mask is needed: the mask is always trilinear, spatial.py:1722-1727)
    const float x0 = floorf(x), y0 = floorf(y), z0 = floorf(z);
    const float x1 = x0 + 1.0f, y1 = y0 + 1.0f, z1 = z0 + 1.0f;
    float w[8];
    bool interior = true;
    if (a.any_linear) {
      const float wx0 = x1 - x, wx1 = x - x0;
      const float wy0 = y1 - y, wy1 = y - y0;
      const float wz0 = z1 - z, wz1 = z - z0;
      w[0] = __fmul_rn(__fmul_rn(wx0, wy0), wz0);
      w[1] = __fmul_rn(__fmul_rn(wx1, wy0), wz0);
      w[2] = __fmul_rn(__fmul_rn(wx0, wy1), wz0);
      w[3] = __fmul_rn(__fmul_rn(wx1, wy1), wz0);
      w[4] = __fmul_rn(__fmul_rn(wx0, wy0), wz1);
      w[5] = __fmul_rn(__fmul_rn(wx1, wy0), wz1);
      w[6] = __fmul_rn(__fmul_rn(wx0, wy1), wz1);
      w[7] = __fmul_rn(__fmul_rn(wx1, wy1), wz1);
      // x0 ∈ [0, S-2] ⇔ x0 and x1 both in bounds (integral floats; NaN fails)
      interior = (x0 >= 0.0f) & (x0 <= hx - 1.0f) & (y0 >= 0.0f) & (y0 <= hy - 1.0f) & (z0 >= 0.0f) & (z0 <= hz - 1.0f);
    }
    const int64_t o_idx = io * slab + row;
    if constexpr (LABEL_PV) {
      for (int im = 0; im < a.n_images; im++) {
        const ImgArgs& g = a.img[im];
        LabelSite site;
        site.in = g.in; site.out = g.out; site.labels = g.labels; site.pad_label = g.pad_label;
        site.in_base = static_cast<int64_t>(b) * n_in; site.out_index = static_cast<int64_t>(b) * n_out + o_idx;
        site.dtype = g.dtype; site.n_labels = g.n_labels; site.J = a.J; site.K = a.K;
        site.hx = hx; site.hy = hy; site.hz = hz;
        label_pv_voxel(site, x, y, z);
      }
      continue;
    }
    // nearest: nearbyint = round half to even (v_rndne_f32)
    const float xn = rintf(x), yn = rintf(y), zn = rintf(z);
    bool okn = true;
    if (a.any_nearest) {
      okn = (xn >= 0.0f) & (xn <= hx) & (yn >= 0.0f) & (yn <= hy) & (zn >= 0.0f) & (zn <= hz);
      interior = interior & okn;
    }

    if (__builtin_amdgcn_ballot_w64(!interior) == 0) {  // wave-uniform: whole wave interior
      const int base = (static_cast<int>(x0) * a.J + static_cast<int>(y0)) * a.K + static_cast<int>(z0);
      const int offn = (static_cast<int>(xn) * a.J + static_cast<int>(yn)) * a.K + static_cast<int>(zn);
      for (int im = 0; im < a.n_images; im++) {
        const ImgArgs& g = a.img[im];
#define TIO_CALL(DT) sample_interior<DT>(g, b, n_in, n_out, o_idx, w, base, dJK, dK, offn)
        TIO_DISPATCH_IMAGE(DTMODE, g.dtype, TIO_CALL)
#undef TIO_CALL
      }
    } else {
      const bool bx0 = (x0 >= 0.0f) & (x0 <= hx), bx1 = (x1 >= 0.0f) & (x1 <= hx);
      const bool by0 = (y0 >= 0.0f) & (y0 <= hy), by1 = (y1 >= 0.0f) & (y1 <= hy);
      const bool bz0 = (z0 >= 0.0f) & (z0 <= hz), bz1 = (z1 >= 0.0f) & (z1 <= hz);
      // clamp before the int conversion so that far-away coordinates stay defined
      const int ix0 = static_cast<int>(fminf(fmaxf(x0, 0.0f), hx)), ix1 = static_cast<int>(fminf(fmaxf(x1, 0.0f), hx));
      const int iy0 = static_cast<int>(fminf(fmaxf(y0, 0.0f), hy)), iy1 = static_cast<int>(fminf(fmaxf(y1, 0.0f), hy));
      const int iz0 = static_cast<int>(fminf(fmaxf(z0, 0.0f), hz)), iz1 = static_cast<int>(fminf(fmaxf(z1, 0.0f), hz));
      int off[8];
      unsigned okbits = 0;
      float mask = 0.0f;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const bool ok = ((k & 1) ? bx1 : bx0) & ((k & 2) ? by1 : by0) & ((k & 4) ? bz1 : bz0);
        off[k] = (((k & 1) ? ix1 : ix0) * a.J + ((k & 2) ? iy1 : iy0)) * a.K + ((k & 4) ? iz1 : iz0);
        okbits |= ok ? (1u << k) : 0u;
        if (a.any_linear) {
          const float next = __fadd_rn(mask, w[k]);  // same order as ATen's accumulation
          mask = ok ? next : mask;
        }
      }
      const int offn = (static_cast<int>(fminf(fmaxf(xn, 0.0f), hx)) * a.J + static_cast<int>(fminf(fmaxf(yn, 0.0f), hy))) * a.K +
                       static_cast<int>(fminf(fmaxf(zn, 0.0f), hz));
      for (int im = 0; im < a.n_images; im++) {
        const ImgArgs& g = a.img[im];
#define TIO_CALL(DT) sample_boundary<DT>(g, b, n_in, n_out, o_idx, w, off, okbits, mask, offn, okn)
        TIO_DISPATCH_IMAGE(DTMODE, g.dtype, TIO_CALL)
#undef TIO_CALL
      }
    }
  }
}

}  // namespace tio

#include "resample_exact_chain.hpp"
#include "resample_tile.hpp"
#include "resample_fast.hpp"
#include "resample_lean_exact.hpp"
#include "resample_nearest.hpp"

// Device scratch for the brick plan of a planned launch (resample_fast.hpp): one buffer per (device, stream), grown on
// demand and kept.  A planned launch is a PAIR of kernels on the caller's stream — plan_bricks_kernel writes the buffer,
// the sampling kernel reads it — so the buffer is LEASED: the lease holds the slot's mutex from the lookup until both
// kernels are enqueued.  Two host threads that share a stream (the reference's Queue workers on the default stream,
// data/queue.py:119-123; ctypes drops the GIL around these calls) therefore enqueue planA, sampleA, planB, sampleB and
// never planA, planB, sampleA; the stream then orders the pairs on the device.  Growing (hipStreamSynchronize + hipFree +
// hipMalloc) also happens under the lease, i.e. while no other call holds a pointer it has not launched with yet.
namespace {
struct PlanSlot {
  int device;
  hipStream_t stream;
  int* ptr = nullptr;
  size_t cap = 0;
  std::mutex busy;
};
struct PlanLease {
  std::unique_lock<std::mutex> hold;
  int* ptr = nullptr;
};
}  // namespace

static PlanLease plan_workspace(hipStream_t s, size_t bytes) {
  static std::mutex registry_mu;
  static std::vector<PlanSlot*> registry;  // slots are never destroyed: their addresses (and mutexes) stay valid
  PlanLease lease;
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return lease;
  PlanSlot* slot = nullptr;
  {
    std::lock_guard<std::mutex> lock(registry_mu);
    for (PlanSlot* sl : registry)
      if (sl->device == device && sl->stream == s) slot = sl;
    if (slot == nullptr) {
      slot = new PlanSlot();
      slot->device = device; slot->stream = s;
      registry.push_back(slot);
    }
  }
  lease.hold = std::unique_lock<std::mutex>(slot->busy);
  if (slot->cap < bytes) {
    if (slot->ptr != nullptr) {
      (void)hipStreamSynchronize(s);  // the previous (smaller) plan of this stream may still be read
      (void)hipFree(slot->ptr);
      slot->ptr = nullptr; slot->cap = 0;
    }
    if (hipMalloc(&slot->ptr, bytes) != hipSuccess) { slot->ptr = nullptr; return lease; }
    // (the first tio::kPlanHeaderInts ints: the multi-pass bricks' list header — zero between launches, see resample_lean_exact_kernel)
    if (hipMemsetAsync(slot->ptr, 0, tio::kPlanHeaderInts * sizeof(int), s) != hipSuccess) { (void)hipFree(slot->ptr); slot->ptr = nullptr; return lease; }
    slot->cap = bytes;
  }
  lease.ptr = slot->ptr;
  return lease;
}

// `folded` comes back true when the launch itself produced every requested out_min_dev (planned FAST bricks)
// `mode`: kPlanNone — the call itself; kPlanQuery — only *plan_bytes (the plan this geometry's launch would start from; 0: none);
// kPlanOnly — enqueue the planning kernel into plan_out and return (tio_resample3d_plan).  The two plan modes run the same
// decisions as the call, with one float32 trilinear image standing in for the caller's (the planner never reads an image).
enum { kPlanNone = 0, kPlanQuery = 1, kPlanOnly = 2 };
static int resample3d_impl(const tio_resample_geom* geom, int32_t n_images, const tio_resample_image* images, void* stream, bool* folded,
                           int mode = kPlanNone, int* plan_out = nullptr, int64_t plan_out_bytes = 0, int64_t* plan_bytes = nullptr) {
  using namespace tio;
  *folded = false;
  if (plan_bytes != nullptr) *plan_bytes = 0;
  tio_resample_image stand_in{};
  if (mode != kPlanNone) {
    static float aligned_dummy[4] __attribute__((aligned(16)));  // never dereferenced: the plan modes return before any sampling launch
    stand_in.in = aligned_dummy; stand_in.out = aligned_dummy; stand_in.channels = 1; stand_in.dtype = TIO_F32; stand_in.interp = TIO_LINEAR;
    images = &stand_in; n_images = 1;
  }
  if (geom == nullptr || images == nullptr) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_resample3d: null argument");
  if (n_images < 1 || n_images > TIO_MAX_IMAGES)
    return fail(TIO_ERR_INVALID_ARGUMENT, "tio_resample3d: n_images=%d not in [1, %d]", n_images, TIO_MAX_IMAGES);
  if (geom->batch < 0) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_resample3d: negative batch");
  if (geom->batch == 0) return TIO_OK;  // an empty batch has no data pointers to speak of
  if (geom->mapping_dev == nullptr) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_resample3d: mapping_dev is null");
  for (int d = 0; d < 3; d++) {
    if (geom->in_shape[d] < 1 || geom->out_shape[d] < 1)
      return fail(TIO_ERR_INVALID_ARGUMENT, "tio_resample3d: shapes must be >= 1");
  }
  const int64_t n_in = static_cast<int64_t>(geom->in_shape[0]) * geom->in_shape[1] * geom->in_shape[2];
  const int64_t n_out = static_cast<int64_t>(geom->out_shape[0]) * geom->out_shape[1] * geom->out_shape[2];
  if (n_in >= (1LL << 31) || n_out >= (1LL << 31))
    return fail(TIO_ERR_INVALID_ARGUMENT, "tio_resample3d: more than 2^31 voxels per channel");
  if (geom->passthrough_dev != nullptr && n_in != n_out)
    return fail(TIO_ERR_INVALID_ARGUMENT, "tio_resample3d: passthrough needs in_shape == out_shape");
  if (geom->control_points_dev != nullptr) {
    for (int d = 0; d < 3; d++)
      if (geom->cp_shape[d] < 1) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_resample3d: bad cp_shape");
  }

  ResampleArgs a{};
  a.B = geom->batch;
  a.I = geom->in_shape[0]; a.J = geom->in_shape[1]; a.K = geom->in_shape[2];
  a.Io = geom->out_shape[0]; a.Jo = geom->out_shape[1]; a.Ko = geom->out_shape[2];
  a.affine_first = geom->affine_first;
  a.mapping = geom->mapping_dev;
  a.mapping_batched = geom->mapping_batched;
  a.cp = geom->control_points_dev;
  a.cp_batched = geom->cp_batched;
  a.ni = geom->cp_shape[0]; a.nj = geom->cp_shape[1]; a.nk = geom->cp_shape[2];
  a.cp_skip = geom->cp_skip_dev;
  a.passthrough = geom->passthrough_dev;
  const float* sp = geom->affine_first ? geom->in_spacing : geom->out_spacing;
  a.unit_spacing = 1;
  a.short_div = 1;
  for (int d = 0; d < 3; d++) {
    a.sp[d] = sp[d];
    a.rsp[d] = 1.0f / sp[d];
    if (sp[d] != 1.0f) a.unit_spacing = 0;
    const int size = geom->in_shape[d];
    const int norm = geom->norm_shape[d] > 0 ? geom->norm_shape[d] : size;  // the grid's normalisation (first image's shape)
    a.den[d] = static_cast<float>(norm - 1 > 1 ? norm - 1 : 1);
    a.rden[d] = 1.0f / a.den[d];
    a.size_m1[d] = static_cast<float>(size - 1);
    a.dh[d] = 0.5f * a.den[d];
    a.rdh[d] = 1.0f / a.dh[d];
    a.half_h[d] = 0.5f * a.size_m1[d];
    if (a.den[d] > 8192.0f) a.short_div = 0;
    if (geom->norm_shape[d] < 0) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_resample3d: negative norm_shape");
  }
  if (a.cp != nullptr) {
    if (geom->control_points_dev != nullptr && !(sp[0] > 0.0f && sp[1] > 0.0f && sp[2] > 0.0f))
      return fail(TIO_ERR_INVALID_ARGUMENT, "tio_resample3d: spacing must be positive");
    a.scale_i = lerp_scale(a.ni, a.Io);
    a.scale_j = lerp_scale(a.nj, a.Jo);
    a.scale_k = lerp_scale(a.nk, a.Ko);
  }
  // "label" partial-volume images go through their own launch of the gather kernel (same
  // coordinates); everything else shares one launch of the brick / gather kernel
  ResampleArgs pv = a;
  ResampleArgs spl = a;  // B-spline images (TIO_QUADRATIC / TIO_CUBIC): their own launch of the gather kernel as well
  // nearest images without a fill rule (label maps): their own kernel (resample_nearest.hpp), bit-identical to the exact
  // chain whatever the precision mode of the call; TIO_NEAREST_KERNEL=0 keeps them with the other images (A/B)
  NearestArgs nn{};
  const EnvSwitches& env = env_switches();  // (parsed once per process / tio_reload_env(): no getenv on this road)
  const bool nn_enabled = env.nearest_kernel != 0 &&
                          static_cast<int64_t>(a.I) * a.J <= (1LL << 24) && a.K < (1 << 24);
  a.n_images = 0;
  pv.n_images = 0;
  pv.any_linear = 1;
  spl.n_images = 0;
  int dtmode = 0;
  bool any_adjoint = false;
  for (int i = 0; i < n_images; i++) {
    const tio_resample_image& s = images[i];
    if (s.in == nullptr || s.out == nullptr || s.channels < 1)
      return fail(TIO_ERR_INVALID_ARGUMENT, "tio_resample3d: image %d has null data or no channels", i);
    if (dtype_size(s.dtype) == 0) return fail(TIO_ERR_UNSUPPORTED_DTYPE, "tio_resample3d: image %d dtype %d", i, s.dtype);
    if (s.interp != TIO_NEAREST && s.interp != TIO_LINEAR && s.interp != TIO_LABEL_PV && s.interp != TIO_LINEAR_ADJOINT &&
        TIO_BSPLINE_ORDER(s.interp) == 0)
      return fail(TIO_ERR_INVALID_ARGUMENT, "tio_resample3d: image %d interp %d", i, s.interp);
    if (TIO_BSPLINE_ORDER(s.interp) != 0) {
      if (s.dtype != TIO_F32)
        return fail(TIO_ERR_UNSUPPORTED_DTYPE, "tio_resample3d: image %d: B-spline images hold float32 coefficients (tio_bspline_prefilter)", i);
      spl.img[spl.n_images++] = ImgArgs{s.in, s.out, nullptr, s.channels, s.dtype, s.interp, nullptr, 0, 0.0, nullptr, nullptr};
      continue;
    }
    if (s.interp == TIO_LINEAR_ADJOINT) {
      if (s.dtype != TIO_F32) return fail(TIO_ERR_UNSUPPORTED_DTYPE, "tio_resample3d: image %d: the adjoint works on float32 gradients", i);
      any_adjoint = true;
    }
    if (s.interp == TIO_LABEL_PV) {
      if (s.channels != 1)
        return fail(TIO_ERR_INVALID_ARGUMENT, "tio_resample3d: image %d: TIO_LABEL_PV needs channels == 1, got %d", i, s.channels);
      if (s.n_labels < 0 || (s.n_labels > 0 && s.labels_dev == nullptr))
        return fail(TIO_ERR_INVALID_ARGUMENT, "tio_resample3d: image %d: bad label table", i);
      pv.img[pv.n_images++] = ImgArgs{s.in, s.out, nullptr, 1, s.dtype, s.interp, s.labels_dev, s.n_labels, s.pad_label, nullptr, nullptr};
      continue;
    }
    if (nn_enabled && s.interp == TIO_NEAREST && s.out_min_dev == nullptr) {  // (with or without a fill rule: round 4)
      nn.img[nn.n_images++] = NearestImg{s.in, s.out, s.channels, dtype_size(s.dtype), s.fill_dev, s.dtype};
      if (s.fill_dev != nullptr) nn.any_fill = 1;
      continue;
    }
    a.img[a.n_images++] = ImgArgs{s.in, s.out, s.fill_dev, s.channels, s.dtype, s.interp, nullptr, 0, 0.0, s.out_min_dev, nullptr};
    // the in-bounds weight mask needs the trilinear weights even for nearest data (spatial.py:1722-1727)
    if (s.interp == TIO_LINEAR || s.interp == TIO_LINEAR_ADJOINT || s.fill_dev != nullptr) a.any_linear = 1;
    if (s.interp == TIO_NEAREST) a.any_nearest = 1;
    const int need = s.dtype == TIO_F32 ? 0 : ((s.dtype == TIO_I16 || s.dtype == TIO_U8 || s.dtype == TIO_I32) ? 1 : 2);
    dtmode = need > dtmode ? need : dtmode;
  }
  if (a.B == 0) return TIO_OK;

  hipStream_t s = static_cast<hipStream_t>(stream);
  const int n_cp = a.cp != nullptr ? a.ni * a.nj * a.nk * 3 : 0;

  // round 6: ONE plain label channel of a call that also samples float32 images may ride along the images' last exact-coordinate
  // launch (resample_lean_exact_label_kernel) instead of taking its own kernel; whatever road the images take in the end — every
  // return below — the label map is sampled: if nobody has taken it along, its own launch goes out when this scope is left
  struct DeferredLaunch {
    std::function<void()> launch;
    ~DeferredLaunch() { if (launch) launch(); }
  } label_rides;
  int label_es = 0;
  if (nn.n_images > 0) {
    nn.B = a.B; nn.I = a.I; nn.J = a.J; nn.K = a.K; nn.Io = a.Io; nn.Jo = a.Jo; nn.Ko = a.Ko; nn.affine_first = a.affine_first;
    nn.mapping = a.mapping; nn.cp = a.cp; nn.cp_skip = a.cp_skip; nn.passthrough = a.passthrough;
    nn.mapping_batched = a.mapping_batched; nn.cp_batched = a.cp_batched; nn.ni = a.ni; nn.nj = a.nj; nn.nk = a.nk;
    nn.unit_spacing = a.unit_spacing;
    nn.scale_i = a.scale_i; nn.scale_j = a.scale_j; nn.scale_k = a.scale_k;
    for (int d = 0; d < 3; d++) {
      nn.sp[d] = a.sp[d]; nn.rsp[d] = a.rsp[d]; nn.den[d] = a.den[d]; nn.rden[d] = a.rden[d]; nn.size_m1[d] = a.size_m1[d];
      nn.ratio[d] = a.half_h[d] / a.dh[d];
      nn.dh[d] = a.dh[d]; nn.rdh[d] = a.rdh[d]; nn.half_h[d] = a.half_h[d];
    }
    const bool nn_rows = nn.Ko >= 48;  // bricks of 16 x 4 x 64 (a wave = one output row) unless the volume is narrower than that
    const int nn_tj = nn_rows ? 4 : 16, nn_tk = nn_rows ? 64 : 16;
    nn.tiles_k = (nn.Ko + nn_tk - 1) / nn_tk; nn.tiles_j = (nn.Jo + nn_tj - 1) / nn_tj; nn.tiles_i = (nn.Io + 15) / 16;
    nn.magic_k = nn.tiles_k > 1 ? 0xFFFFFFFFu / nn.tiles_k + 1u : 0u;
    nn.magic_j = nn.tiles_j > 1 ? 0xFFFFFFFFu / nn.tiles_j + 1u : 0u;
    nn.magic_i = nn.tiles_i > 1 ? 0xFFFFFFFFu / nn.tiles_i + 1u : 0u;
    const int64_t blocks = static_cast<int64_t>(nn.B) * nn.tiles_i * nn.tiles_j * nn.tiles_k;
    if (blocks >= (1LL << 31)) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_resample3d: grid too large");
    nn.eps = kNearestEps;
    if (env.has_nearest_eps) nn.eps = env.nearest_eps;  // (calibration runs only)
    const dim3 grid(static_cast<unsigned>(blocks)), block(256);
    // round 6: without a fill rule the reference's own coordinates plane by plane (resample_nearest_exact_kernel; its gate is the
    // float exact-coordinate kernel's: the short division, unit spacing under control points, rows of at least 48 voxels)
    const bool nn_exact = env.nearest_exact != 0 && nn.any_fill == 0 && nn_rows && a.short_div != 0 && (a.cp == nullptr || a.unit_spacing != 0);
    for (int es = 1; es <= 8; es *= 2) {  // one launch per element size present
      bool present = false;
      for (int i = 0; i < nn.n_images; i++) present = present || nn.img[i].es == es;
      if (!present) continue;
      // ... and its tail's: 24-bit offsets and buffer loads (I J <= 2^24, every channel below 2^32 bytes)
      const int64_t n_in_e = static_cast<int64_t>(nn.I) * nn.J * nn.K, n_out_e = static_cast<int64_t>(nn.Io) * nn.Jo * nn.Ko;
      const bool nn_narrow = static_cast<int64_t>(nn.I) * nn.J <= (1 << 24) && static_cast<int64_t>(nn.K) * es < (1 << 24) &&
                             n_in_e * es < (int64_t{1} << 32) - 16 && n_out_e * es < (int64_t{1} << 32) - 16;
      if (nn_exact && nn_narrow && env.lean_label != 0 && nn.n_images == 1 && nn.img[0].channels == 1 && es <= 4 && a.n_images > 0 && pv.n_images == 0 &&
          spl.n_images == 0 && !any_adjoint) {
        const unsigned nn_lds = env.nearest_lds >= 0 ? static_cast<unsigned>(env.nearest_lds) : (nn.cp == nullptr ? 52000u : 0u);
        label_es = es;
        label_rides.launch = [nn, grid, block, nn_lds, s, es]() {
#define TIO_NN_EXACT(ES)                                                                                                \
  if (nn.cp != nullptr) hipLaunchKernelGGL((resample_nearest_exact_kernel<true, ES>), grid, block, nn_lds, s, nn);           \
  else hipLaunchKernelGGL((resample_nearest_exact_kernel<false, ES>), grid, block, nn_lds, s, nn);
          if (es == 1) { TIO_NN_EXACT(1) } else if (es == 2) { TIO_NN_EXACT(2) } else { TIO_NN_EXACT(4) }
#undef TIO_NN_EXACT
        };
        continue;
      }
      if (nn_exact && nn_narrow) {
        // resident blocks per CU through UNUSED dynamic LDS: without control points the kernel holds 61 registers (eight blocks per CU),
        // and eight blocks' slanted input footprints evict one another's cache lines — three blocks per CU measured 0.236 -> 0.210 ms
        // (int16) and 0.344 -> 0.286 (int32) on 8 x 256^3 at the bench's ranges, uint8 0.172 -> 0.177; with control points (95 registers,
        // five blocks) the launch is arithmetic bound and loses from four blocks down (profiles/r06_labels.md).  TIO_NEAREST_LDS: A/B
        const unsigned nn_lds = env.nearest_lds >= 0 ? static_cast<unsigned>(env.nearest_lds) : (nn.cp == nullptr ? 52000u : 0u);
#define TIO_NN_EXACT(ES)                                                                                                \
  if (nn.cp != nullptr) hipLaunchKernelGGL((resample_nearest_exact_kernel<true, ES>), grid, block, nn_lds, s, nn);           \
  else hipLaunchKernelGGL((resample_nearest_exact_kernel<false, ES>), grid, block, nn_lds, s, nn);
        if (es == 1) { TIO_NN_EXACT(1) } else if (es == 2) { TIO_NN_EXACT(2) } else if (es == 4) { TIO_NN_EXACT(4) } else { TIO_NN_EXACT(8) }
#undef TIO_NN_EXACT
        continue;
      }
#define TIO_NN_LAUNCH_SHAPE(ES, TJ, TK)                                                                                 \
  if (nn.cp != nullptr) hipLaunchKernelGGL((resample_nearest_kernel<true, ES, TJ, TK, 16>), grid, block, 0, s, nn);     \
  else hipLaunchKernelGGL((resample_nearest_kernel<false, ES, TJ, TK, 16>), grid, block, 0, s, nn);
#define TIO_NN_LAUNCH(ES)                                                                                               \
  if (nn_rows) { TIO_NN_LAUNCH_SHAPE(ES, 4, 64) } else { TIO_NN_LAUNCH_SHAPE(ES, 16, 16) }
      if (es == 1) { TIO_NN_LAUNCH(1) } else if (es == 2) { TIO_NN_LAUNCH(2) } else if (es == 4) { TIO_NN_LAUNCH(4) } else { TIO_NN_LAUNCH(8) }
#undef TIO_NN_LAUNCH
#undef TIO_NN_LAUNCH_SHAPE
    }
    if (a.n_images == 0 && pv.n_images == 0 && spl.n_images == 0) return check_launch("tio_resample3d");
  }
  if (pv.n_images > 0) {
    pv.tiles_k = (pv.Ko + kLanes - 1) / kLanes;
    pv.tiles_j = (pv.Jo + kRowsPerBlock - 1) / kRowsPerBlock;
    pv.tiles_i = (pv.Io + kTileI - 1) / kTileI;
    const int64_t blocks = static_cast<int64_t>(pv.B) * pv.tiles_i * pv.tiles_j * pv.tiles_k;
    if (blocks >= (1LL << 31)) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_resample3d: grid too large");
    const dim3 grid(static_cast<unsigned>(blocks)), block(kRowsPerBlock * kLanes);
    const size_t lds = (n_cp > 0 && n_cp <= kMaxCpLds) ? static_cast<size_t>(n_cp) * sizeof(float) : 0;
    if (pv.cp != nullptr)
      hipLaunchKernelGGL((resample_kernel<true, 2, 1>), grid, block, lds, s, pv);
    else
      hipLaunchKernelGGL((resample_kernel<false, 2, 1>), grid, block, lds, s, pv);
    if (a.n_images == 0 && spl.n_images == 0) return check_launch("tio_resample3d");
  }
  if (spl.n_images > 0) {
    spl.tiles_k = (spl.Ko + kLanes - 1) / kLanes;
    spl.tiles_j = (spl.Jo + kRowsPerBlock - 1) / kRowsPerBlock;
    spl.tiles_i = (spl.Io + kTileI - 1) / kTileI;
    const int64_t blocks = static_cast<int64_t>(spl.B) * spl.tiles_i * spl.tiles_j * spl.tiles_k;
    if (blocks >= (1LL << 31)) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_resample3d: grid too large");
    if (static_cast<int64_t>(spl.I) * spl.J * spl.K >= (1LL << 31)) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_resample3d: volume too large");
    const dim3 grid(static_cast<unsigned>(blocks)), block(kRowsPerBlock * kLanes);
    const size_t lds = (n_cp > 0 && n_cp <= kMaxCpLds) ? static_cast<size_t>(n_cp) * sizeof(float) : 0;
    if (spl.cp != nullptr)
      hipLaunchKernelGGL((resample_kernel<true, 0, 2>), grid, block, lds, s, spl);
    else
      hipLaunchKernelGGL((resample_kernel<false, 0, 2>), grid, block, lds, s, spl);
    if (a.n_images == 0) return check_launch("tio_resample3d");
  }

  // Path: LDS-staged bricks whenever a trilinear image is present (the 8-tap gather is
  // what the staging removes); pure nearest launches keep the one-load gather kernel.
  // TIO_RESAMPLE_PATH=gather|tile overrides (A/B tests compare the two bit for bit).
  bool use_tile = a.any_linear != 0 && !any_adjoint;  // the adjoint scatters to global memory: gather kernel
  if (env.resample_path == 1) use_tile = false;
  if (env.resample_path == 2) use_tile = true;
  if (any_adjoint) use_tile = false;
  if (static_cast<int64_t>(a.Jo) * a.Ko * 8 >= (1LL << 31)) use_tile = false;  // 32-bit byte offsets inside one output plane
  if (n_in >= (1LL << 30)) use_tile = false;  // 32-bit byte offsets inside one input channel (f32 brick DMA)
  if (use_tile) {
    a.fill_recheck = env.fast_fill_recheck;
    a.any_fill = 0;
    for (int i = 0; i < a.n_images; i++) a.any_fill |= a.img[i].fill != nullptr ? 1 : 0;
    const int variant = env.tile_variant;
    int cap = env.tile_lds_floats;
    a.ablate = env.tile_ablate;
    a.cp_lds = (n_cp > 0 && n_cp <= kMaxCpLds) ? ((n_cp + 3) & ~3) : 0;
    // default brick budget: whatever lets kTileBlocksPerCU blocks share the CU's 160 KiB
    // (minus 2 KiB: the hardware allocates LDS in granules, an exact third does not fit three times)
    const int bpc = variant == 3 ? 2 : ((variant == 2 || variant == 4) ? 4 : kTileBlocksPerCU);  // resident blocks the variant is built for
    if (cap <= 0) cap = kLdsFloatsPerCU / bpc - 512 - a.cp_lds - kTileRedInts;
    const int max_cap = kLdsFloatsPerCU - a.cp_lds - kTileRedInts;
    a.tile_cap = cap < kTileMinCap ? kTileMinCap : (cap > max_cap ? max_cap : cap);
    const size_t lds = static_cast<size_t>(a.cp_lds + kTileRedInts + a.tile_cap) * sizeof(float);
#define TIO_TILE_LAUNCH(EL, DM, TI, TJ, TK, OCC)                                                              \
  {                                                                                                      \
    a.tiles_k = (a.Ko + TK - 1) / TK;                                                                    \
    a.tiles_j = (a.Jo + TJ - 1) / TJ;                                                                    \
    a.tiles_i = (a.Io + TI - 1) / TI;                                                                    \
    a.magic_k = a.tiles_k > 1 ? 0xFFFFFFFFu / a.tiles_k + 1u : 0u;                                       \
    a.magic_j = a.tiles_j > 1 ? 0xFFFFFFFFu / a.tiles_j + 1u : 0u;                                       \
    a.magic_i = a.tiles_i > 1 ? 0xFFFFFFFFu / a.tiles_i + 1u : 0u;                                       \
    const int64_t blocks = static_cast<int64_t>(a.B) * a.tiles_i * a.tiles_j * a.tiles_k;                \
    if (blocks >= (1LL << 31)) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_resample3d: grid too large");  \
    auto kernel = resample_tile_kernel<EL, DM, TI, TJ, TK, OCC>;                                            \
    if (lds > 48 * 1024) {                                                                               \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, \
                              static_cast<int>(lds)) != hipSuccess)                                      \
        return fail(TIO_ERR_LAUNCH, "tio_resample3d: cannot reserve %zu bytes of LDS", lds);             \
    }                                                                                                    \
    hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(blocks)), dim3(TJ* TK), lds, s, a,             \
                       (TI == 16 && TJ == 16 && TK == 16) ? plan_exact : static_cast<const int*>(nullptr)); \
  }
#define TIO_TILE_SHAPE(TI, TJ, TK, OCC)                                                  \
  {                                                                                 \
    if (a.cp != nullptr) {                                                          \
      if (dtmode == 0) TIO_TILE_LAUNCH(true, 0, TI, TJ, TK, OCC)                        \
      else if (dtmode == 1) TIO_TILE_LAUNCH(true, 1, TI, TJ, TK, OCC)                   \
      else TIO_TILE_LAUNCH(true, 2, TI, TJ, TK, OCC)                                    \
    } else {                                                                        \
      if (dtmode == 0) TIO_TILE_LAUNCH(false, 0, TI, TJ, TK, OCC)                       \
      else if (dtmode == 1) TIO_TILE_LAUNCH(false, 1, TI, TJ, TK, OCC)                  \
      else TIO_TILE_LAUNCH(false, 2, TI, TJ, TK, OCC)                                   \
    }                                                                               \
  }
#define TIO_TILE_SHAPE_F32(TI, TJ, TK, OCC) /* experimental shapes: float32 launches only */ \
  {                                                                                 \
    if (dtmode != 0) {                                                              \
      TIO_TILE_SHAPE(16, 16, 16, 3)                                                 \
    } else if (a.cp != nullptr) {                                                   \
      TIO_TILE_LAUNCH(true, 0, TI, TJ, TK, OCC)                                         \
    } else {                                                                        \
      TIO_TILE_LAUNCH(false, 0, TI, TJ, TK, OCC)                                        \
    }                                                                               \
  }
    // fast intensity path: float32 trilinear images only (nearest / label images need the exact coordinates)
    const bool fast = geom->precision == TIO_PRECISION_FAST && dtmode == 0 && !a.any_nearest && variant == 0 &&
                      !env.resample_exact;
    // Round 5: the lean planned kernel with the reference's own coordinates (resample_lean_exact.hpp).  TIO_PRECISION_TIGHT:
    // fused interpolation on exact coordinates / taps / fill decisions; TIO_PRECISION_EXACT: ATen's interpolation order too,
    // bit-identical to the brick kernel below (TIO_EXACT_LEAN=0 keeps large exact launches on the brick kernel, =2 sends
    // small ones to the lean kernel as well: A/B and tests).  Float32 trilinear images only, divisors the short division is
    // proven for, unit spacing whenever a displacement is divided by it; everything else runs the exact brick kernel.
    const bool tight = geom->precision == TIO_PRECISION_TIGHT && !env.resample_exact;  // (TIO_RESAMPLE_EXACT: the A/B switch forces ATen's interpolation order too)
    const bool lean_exact = !fast && (tight || ((geom->precision == TIO_PRECISION_EXACT || geom->precision == TIO_PRECISION_TIGHT) && env.exact_lean != 0)) && dtmode == 0 &&
                            !a.any_nearest && variant == 0 && a.ablate == 0 && a.short_div != 0 && (a.cp == nullptr || a.unit_spacing != 0);
    if (fast || lean_exact) {
      a.tiles_k = (a.Ko + 15) / 16; a.tiles_j = (a.Jo + 15) / 16; a.tiles_i = (a.Io + 15) / 16;
      a.magic_k = a.tiles_k > 1 ? 0xFFFFFFFFu / a.tiles_k + 1u : 0u;
      a.magic_j = a.tiles_j > 1 ? 0xFFFFFFFFu / a.tiles_j + 1u : 0u;
      a.magic_i = a.tiles_i > 1 ? 0xFFFFFFFFu / a.tiles_i + 1u : 0u;
      const int64_t blocks = static_cast<int64_t>(a.B) * a.tiles_i * a.tiles_j * a.tiles_k;
      if (blocks >= (1LL << 31)) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_resample3d: grid too large");
      // Planned bricks (resample_fast.hpp): a one-thread-per-brick planning kernel, then one block per brick that starts
      // from its 64-byte descriptor.  Needs 16-byte rows for the LDS-DMA and control cells at least a brick wide (the
      // box comes from <= 27 vertices); everything else — and TIO_FAST_KERNEL=brick, the A/B switch — runs the brick
      // kernel's FAST instantiation with its in-kernel boxes.
      // (small launches keep the single-kernel road: the planning kernel and the gap before the second launch cost
      // ~10-15 us, more than the planned bricks save below ~12 k bricks; TIO_FAST_KERNEL=planned forces them)
      const bool force_planned = lean_exact ? (env.fast_kernel == 2 || env.exact_lean == 2) : env.fast_kernel == 2;
      bool planned = (lean_exact || env.fast_kernel != 1) && (a.K & 3) == 0 && (blocks >= kPlannedMinBricks || force_planned) && blocks < (1LL << 26);
      for (int i = 0; i < a.n_images; i++) planned = planned && (reinterpret_cast<uintptr_t>(a.img[i].in) & 15) == 0;
      // the caller expects boxes beyond the staging tile (tio_hip.h: TIO_GEOM_LARGE_BOXES): a planned brick whose box does not fit
      // samples voxel by voxel from global memory, the brick kernels below split it into passes over its planes
      // (round 6: the exact-coordinate lean road stages such bricks in passes itself — the hint only sends FAST launches elsewhere)
      if ((geom->flags & (TIO_GEOM_LARGE_BOXES | TIO_GEOM_MOSTLY_LARGE_BOXES)) != 0 && !force_planned && !lean_exact) planned = false;
      if (planned && a.cp != nullptr) {
        const int n_ctl[3] = {a.ni, a.nj, a.nk}, n_vox[3] = {a.Io, a.Jo, a.Ko};
        for (int d = 0; d < 3; d++)
          if (n_ctl[d] > 2 && (n_vox[d] - 1) < 16 * (n_ctl[d] - 1)) planned = false;
      }
      if (planned) {
        // one single-channel image (what a FAST intensity launch almost always is): the lean kernel (resample_fast.hpp),
        // whose bricks may be 8 planes thick (half the tile: twice the blocks per CU)
        const bool lean = env.planned_lean != 0 || lean_exact;
        const int64_t items64 = blocks;
        // the largest tile three blocks of which fit a CU: LDS is handed out in granules of 1 280 bytes here (measured: 13 440
        // floats keep three blocks resident, 13 568 drop to two — profiles/r04_tile_cap.log); 300 floats more than the round-3
        // value, which is 1.5 % of a fused affine + elastic launch (fewer bricks on the per-voxel road)
        int cap_p = (kLdsFloatsPerCU / kTileBlocksPerCU) / 320 * 320;
        if (env.tile_lds_floats > 0) cap_p = env.tile_lds_floats;
        if (cap_p < kTileMinCap) cap_p = kTileMinCap;
        if (cap_p > kLdsFloatsPerCU) cap_p = kLdsFloatsPerCU;
        a.tile_cap = cap_p;
        a.cp_lds = 0;
        const size_t lds_p = static_cast<size_t>(cap_p) * sizeof(float);
        const int n_items = static_cast<int>(items64);
        // round 3: DMA instructions that cover rows across x-plane boundaries (resample_fast.hpp: stream_stage_packed);
        // TIO_DMA_PACKED=0 switches them off in the general kernel (A/B)
        a.dma_packed = env.dma_packed;
        // (the exact-coordinate kernel stages bricks whose box exceeds the tile in passes over their planes: pass boxes behind the descriptors)
        // — on the hint of a caller who holds the mappings (TIO_GEOM_LARGE_BOXES): the second kernel sits BEHIND the first on the stream,
        // ~6 - 15 us that a launch without such bricks should not pay (measured +0 ... +3.7 % on the bench's launches when it was
        // unconditional: profiles/r06_resample.md); without the hint such bricks sample voxel by voxel, as until round 5
        // (TIO_GEOM_MOSTLY_LARGE_BOXES: plan_multi = 2 — every block of ONE launch runs the body with the pass switches, nothing is listed)
        a.plan_multi = (lean_exact && env.lean_multi != 0) ? ((geom->flags & TIO_GEOM_MOSTLY_LARGE_BOXES) != 0 ? 2 : ((geom->flags & TIO_GEOM_LARGE_BOXES) != 0 ? 1 : 0)) : 0;
        // [header: kPlanHeaderInts ints] [B x 16 floats] [n_items descriptors] ( [n_items x 4 pass boxes] [the list of multi-pass bricks] )
        const size_t plan_list_at = static_cast<size_t>(a.B) * 16 + static_cast<size_t>(n_items) * (kDescInts + kPassInts * kPassesPerBrick);  // (ints behind the header)
        const size_t plan_need = (kPlanHeaderInts + (a.plan_multi ? plan_list_at + ((static_cast<size_t>(n_items) + 3) & ~static_cast<size_t>(3))
                                                                    : static_cast<size_t>(a.B) * 16 + static_cast<size_t>(n_items) * kDescInts)) * sizeof(int);
        if (mode == kPlanQuery) { *plan_bytes = static_cast<int64_t>(plan_need); return TIO_OK; }
        PlanLease lease;
        int* plan = nullptr;
        bool planned_ahead = false;
        if (mode == kPlanOnly) {
          if (plan_out == nullptr || plan_out_bytes < static_cast<int64_t>(plan_need) || (reinterpret_cast<uintptr_t>(plan_out) & 15) != 0)
            return fail(TIO_ERR_INVALID_ARGUMENT, "tio_resample3d_plan: the plan needs %zu bytes, 16-byte aligned", plan_need);
          plan = plan_out + kPlanHeaderInts;
        } else if (geom->plan_dev != nullptr && geom->plan_bytes >= static_cast<int64_t>(plan_need) &&
                   (reinterpret_cast<uintptr_t>(geom->plan_dev) & 15) == 0) {
          plan = static_cast<int*>(const_cast<void*>(geom->plan_dev)) + kPlanHeaderInts;  // made ahead by tio_resample3d_plan: no planning kernel on this stream
          planned_ahead = true;
        } else {
          lease = plan_workspace(s, plan_need);
          if (lease.ptr == nullptr) return fail(TIO_ERR_LAUNCH, "tio_resample3d: cannot allocate the brick plan");
          plan = lease.ptr + kPlanHeaderInts;  // (the lease is released when this function returns: after both kernels are enqueued)
        }
        if (!planned_ahead) {
          const int plan_lanes = plan_group(a.cp != nullptr);  // lanes per brick
          const int plan_threads = n_items * plan_lanes > a.B ? n_items * plan_lanes : a.B;
          const dim3 plan_grid((plan_threads + 255) / 256);
          // the header of the multi-pass bricks' list (length, cursor, done count) is zero between launches: the leased workspace is
          // zeroed when it is allocated and the last walker of a launch leaves zeros; a caller's buffer is zeroed here
          if (a.plan_multi && mode == kPlanOnly && hipMemsetAsync(plan - kPlanHeaderInts, 0, kPlanHeaderInts * sizeof(int), s) != hipSuccess)
            return fail(TIO_ERR_LAUNCH, "tio_resample3d_plan: cannot reset the brick plan");
          if (a.cp != nullptr) hipLaunchKernelGGL((plan_bricks_kernel<true, 16, 16, 16>), plan_grid, dim3(256), 0, s, a, plan, n_items);
          else hipLaunchKernelGGL((plan_bricks_kernel<false, 16, 16, 16>), plan_grid, dim3(256), 0, s, a, plan, n_items);
        }
        if (mode == kPlanOnly) { *plan_bytes = static_cast<int64_t>(plan_need); return check_launch("tio_resample3d_plan"); }
        // the folded minimum: kMinSlots keys per channel that asked for it (common.hpp: min_workspace, all ones between
        // launches), finished by min_finish_kernel behind the sampling kernel
        uint32_t* min_keys = nullptr;
        MinOuts min_outs{};
        int min_channels = 0;
        {
          for (int i = 0; i < a.n_images; i++) min_channels += a.img[i].out_min != nullptr ? a.img[i].channels : 0;
          if (min_channels > 0 && min_channels <= kMinChannels && pv.n_images == 0) {
            int cap = 0;
            min_keys = min_workspace(s, min_channels * kMinSlots, &cap, /*kind=*/1);  // its own array: tio_channel_min may be enqueued between this launch's two kernels
            if (min_keys == nullptr) return fail(TIO_ERR_LAUNCH, "tio_resample3d: cannot allocate the reduction workspace");
            int slot = 0;
            for (int i = 0; i < a.n_images; i++)
              if (a.img[i].out_min != nullptr) {
                a.img[i].min_keys = min_keys + slot * kMinSlots;
                for (int c = 0; c < a.img[i].channels; c++) min_outs.p[slot + c] = a.img[i].out_min + c;
                slot += a.img[i].channels;
              }
            *folded = true;
          } else {
            min_channels = 0;
            for (int i = 0; i < a.n_images; i++) a.img[i].out_min = nullptr;
          }
        }
        auto launch_planned = [&](auto kernel) -> int {
          if (lds_p > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                      static_cast<int>(lds_p)) != hipSuccess)
            return fail(TIO_ERR_LAUNCH, "tio_resample3d: cannot reserve %zu bytes of LDS", lds_p);
          hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(n_items)), dim3(256), lds_p, s, a, static_cast<const int*>(plan));
          if (min_channels > 0)
            hipLaunchKernelGGL(min_finish_kernel, dim3(static_cast<unsigned>(min_channels)), dim3(kMinSlots), 0, s, min_keys, min_outs, min_channels);
          return check_launch("tio_resample3d");
        };
        if (lean) {
          LeanArgs la{};
          la.plan = plan; la.cp = a.cp;
          la.I = a.I; la.J = a.J; la.K = a.K; la.Io = a.Io; la.Jo = a.Jo; la.Ko = a.Ko;
          la.B = a.B; la.n_items = n_items;
          la.bricks_per_element = static_cast<unsigned>(a.tiles_i) * a.tiles_j * a.tiles_k;
          la.bpe_magic = la.bricks_per_element > 1 ? 0xFFFFFFFFu / la.bricks_per_element + 1u : 0u;
          la.ni = a.ni; la.nj = a.nj; la.nk = a.nk; la.cp_batched = a.cp_batched;
          la.sci = a.scale_i; la.scj = a.scale_j; la.sck = a.scale_k;
          for (int e = 0; e < 3; e++) la.dsc[e] = a.rsp[e] * (a.affine_first ? a.half_h[e] / a.dh[e] : 1.0f);
          la.hx = a.size_m1[0]; la.hy = a.size_m1[1]; la.hz = a.size_m1[2];
          la.affine_first = a.affine_first; la.ablate = a.ablate;
          la.mapping = a.mapping; la.mapping_batched = a.mapping_batched; la.unit_spacing = a.unit_spacing; la.fill_recheck = a.fill_recheck;
          for (int e = 0; e < 3; e++) { la.sp[e] = a.sp[e]; la.rsp[e] = a.rsp[e]; la.den[e] = a.den[e]; la.rden[e] = a.rden[e]; }
          for (int e = 0; e < 3; e++) { la.dh[e] = a.dh[e]; la.rdh[e] = a.rdh[e]; la.half_h[e] = a.half_h[e]; }
          la.interleave = env.lean_interleave;
          la.tile_floats = cap_p;
          auto kernel = a.cp != nullptr ? resample_planned_lean_kernel<true, 16, 16, 16, 3> : resample_planned_lean_kernel<false, 16, 16, 16, 3>;
          if (lean_exact) {  // the reference's coordinates; `tight`: fused lerps, else ATen's order (bit-identical to the brick kernel)
            if (tight) {
              if (min_channels > 0) kernel = a.cp != nullptr ? resample_lean_exact_kernel<true, false, 3, true> : resample_lean_exact_kernel<false, false, 3, true>;
              else kernel = a.cp != nullptr ? resample_lean_exact_kernel<true, false, 3> : resample_lean_exact_kernel<false, false, 3>;
            } else {
              if (min_channels > 0) kernel = a.cp != nullptr ? resample_lean_exact_kernel<true, true, 3, true> : resample_lean_exact_kernel<false, true, 3, true>;
              else kernel = a.cp != nullptr ? resample_lean_exact_kernel<true, true, 3> : resample_lean_exact_kernel<false, true, 3>;
            }
          } else if (la.ablate != 0)  // TIO_TILE_ABLATE: the instrumented instantiation (experiments only)
            kernel = a.cp != nullptr ? resample_planned_lean_kernel<true, 16, 16, 16, 3, true> : resample_planned_lean_kernel<false, 16, 16, 16, 3, true>;
          else if (min_channels > 0)  // the folded minimum: the instantiation whose element-0 bricks track what they store
            kernel = a.cp != nullptr ? resample_planned_lean_kernel<true, 16, 16, 16, 3, false, true> : resample_planned_lean_kernel<false, 16, 16, 16, 3, false, true>;
          const size_t lds_launch = lds_p;
          const unsigned grid_launch = static_cast<unsigned>(n_items), block_launch = 256;
          // ... and, behind an exact-coordinate launch, the kernel whose blocks walk the planner's list of multi-pass bricks (boxes beyond
          // the tile: staged in halves / quarters of their planes) — three per CU, leaving at once when the list is empty
          auto kernel_multi = a.cp != nullptr ? resample_lean_exact_multi_kernel<true, false> : resample_lean_exact_multi_kernel<false, false>;
          if (a.plan_multi == 1) {
            if (tight) {
              if (min_channels > 0) kernel_multi = a.cp != nullptr ? resample_lean_exact_multi_kernel<true, false, true> : resample_lean_exact_multi_kernel<false, false, true>;
            } else if (min_channels > 0) {
              kernel_multi = a.cp != nullptr ? resample_lean_exact_multi_kernel<true, true, true> : resample_lean_exact_multi_kernel<false, true, true>;
            } else {
              kernel_multi = a.cp != nullptr ? resample_lean_exact_multi_kernel<true, true> : resample_lean_exact_multi_kernel<false, true>;
            }
          }
          if (a.plan_multi == 2) {  // most bricks need passes: ONE launch of the body that knows them
            if (tight) kernel = min_channels > 0 ? (a.cp != nullptr ? resample_lean_exact_all_kernel<true, false, true> : resample_lean_exact_all_kernel<false, false, true>)
                                                 : (a.cp != nullptr ? resample_lean_exact_all_kernel<true, false> : resample_lean_exact_all_kernel<false, false>);
            else kernel = min_channels > 0 ? (a.cp != nullptr ? resample_lean_exact_all_kernel<true, true, true> : resample_lean_exact_all_kernel<false, true, true>)
                                           : (a.cp != nullptr ? resample_lean_exact_all_kernel<true, true> : resample_lean_exact_all_kernel<false, true>);
          }
          const bool walk_list = a.plan_multi == 1;
          const unsigned grid_multi = static_cast<unsigned>(std::min<int64_t>(n_items, 3 * 256));
          if (lds_launch > 48 * 1024 && (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                             static_cast<int>(lds_launch)) != hipSuccess ||
                                         (walk_list && hipFuncSetAttribute(reinterpret_cast<const void*>(kernel_multi), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                                              static_cast<int>(lds_launch)) != hipSuccess)))
            return fail(TIO_ERR_LAUNCH, "tio_resample3d: cannot reserve %zu bytes of LDS", lds_launch);
          int launches_left = 0;
          for (int i = 0; i < a.n_images; i++) launches_left += a.img[i].channels;
          // round 6: two channels per launch (resample_lean_exact_pair_kernel: one descriptor round trip, one set of control planes, ONE
          // coordinate chain for both) where the launch is an exact-coordinate one without multi-pass bricks and without a folded
          // minimum — a subject's float32 images share their geometry — and the call's label channel, if one waits (label_rides), with the
          // last of them (resample_lean_exact_label_kernel).  TIO_LEAN_PAIR=0: one launch per channel, TIO_LEAN_LABEL=0: the label map's own kernel (A/B)
          const bool label_here = static_cast<bool>(label_rides.launch) && lean_exact && a.plan_multi == 0 && min_channels == 0 && a.passthrough == nn.passthrough;
          if (lean_exact && a.plan_multi == 0 && min_channels == 0 && ((launches_left >= 2 && env.lean_pair != 0) || label_here)) {
            const bool pairs = launches_left >= 2 && env.lean_pair != 0;
            auto kernel_pair = tight ? (a.cp != nullptr ? resample_lean_exact_pair_kernel<true, false> : resample_lean_exact_pair_kernel<false, false>)
                                     : (a.cp != nullptr ? resample_lean_exact_pair_kernel<true, true> : resample_lean_exact_pair_kernel<false, true>);
            auto kernel_label_pair = tight ? (a.cp != nullptr ? resample_lean_exact_label_kernel<true, false, true> : resample_lean_exact_label_kernel<false, false, true>)
                                           : (a.cp != nullptr ? resample_lean_exact_label_kernel<true, true, true> : resample_lean_exact_label_kernel<false, true, true>);
            auto kernel_label_one = tight ? (a.cp != nullptr ? resample_lean_exact_label_kernel<true, false, false> : resample_lean_exact_label_kernel<false, false, false>)
                                          : (a.cp != nullptr ? resample_lean_exact_label_kernel<true, true, false> : resample_lean_exact_label_kernel<false, true, false>);
            if (lds_launch > 48 * 1024 &&
                (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel_pair), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_launch)) != hipSuccess ||
                 (label_here && (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel_label_pair), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_launch)) != hipSuccess ||
                                 hipFuncSetAttribute(reinterpret_cast<const void*>(kernel_label_one), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_launch)) != hipSuccess))))
              return fail(TIO_ERR_LAUNCH, "tio_resample3d: cannot reserve %zu bytes of LDS", lds_launch);
            if (label_here) { la.lab_in = nn.img[0].in; la.lab_out = nn.img[0].out; la.lab_es = label_es; }
            bool have_first = false;
            for (int i = 0; i < a.n_images; i++) {
              const ImgArgs& g = a.img[i];
              for (int c = 0; c < g.channels; c++) {
                const float* in_c = static_cast<const float*>(g.in) + static_cast<int64_t>(c) * n_in;
                float* out_c = static_cast<float*>(g.out) + static_cast<int64_t>(c) * n_out;
                const float* fill_c = g.fill != nullptr ? g.fill + c : nullptr;
                const int64_t in_stride = static_cast<int64_t>(g.channels) * n_in, out_stride = static_cast<int64_t>(g.channels) * n_out;
                --launches_left;
                if (pairs && !have_first && launches_left > 0) {  // (an odd channel out is launched alone, below)
                  la.in = in_c; la.out = out_c; la.fill = fill_c; la.in_stride = in_stride; la.out_stride = out_stride;
                  have_first = true;
                  continue;
                }
                la.last_use = launches_left == 0;
                la.min_keys = nullptr;
                const bool with_label = label_here && launches_left == 0;
                if (have_first) {
                  la.in2 = in_c; la.out2 = out_c; la.fill2 = fill_c; la.in_stride2 = in_stride; la.out_stride2 = out_stride;
                  if (with_label) hipLaunchKernelGGL(kernel_label_pair, dim3(grid_launch), dim3(block_launch), lds_launch, s, la);
                  else hipLaunchKernelGGL(kernel_pair, dim3(grid_launch), dim3(block_launch), lds_launch, s, la);
                  have_first = false;
                } else {
                  la.in = in_c; la.out = out_c; la.fill = fill_c; la.in_stride = in_stride; la.out_stride = out_stride;
                  if (with_label) hipLaunchKernelGGL(kernel_label_one, dim3(grid_launch), dim3(block_launch), lds_launch, s, la);
                  else hipLaunchKernelGGL(kernel, dim3(grid_launch), dim3(block_launch), lds_launch, s, la);
                }
              }
            }
            if (label_here) label_rides.launch = nullptr;  // (taken along)
            return check_launch("tio_resample3d");
          }
          // one plan, one launch per channel of every image (the geometry, hence the plan, is shared)
          for (int i = 0; i < a.n_images; i++) {
            const ImgArgs& g = a.img[i];
            la.in_stride = static_cast<int64_t>(g.channels) * n_in;
            la.out_stride = static_cast<int64_t>(g.channels) * n_out;
            for (int c = 0; c < g.channels; c++) {
              la.in = static_cast<const float*>(g.in) + static_cast<int64_t>(c) * n_in;
              la.out = static_cast<float*>(g.out) + static_cast<int64_t>(c) * n_out;
              la.fill = g.fill != nullptr ? g.fill + c : nullptr;
              la.min_keys = (min_channels > 0 && g.min_keys != nullptr) ? g.min_keys + c * kMinSlots : nullptr;
              la.last_use = --launches_left == 0;
              hipLaunchKernelGGL(kernel, dim3(grid_launch), dim3(block_launch), lds_launch, s, la);
              if (walk_list) hipLaunchKernelGGL(kernel_multi, dim3(grid_multi), dim3(block_launch), lds_launch, s, la);
            }
          }
          if (min_channels > 0)
            hipLaunchKernelGGL(min_finish_kernel, dim3(static_cast<unsigned>(min_channels)), dim3(kMinSlots), 0, s, min_keys, min_outs, min_channels);
          return check_launch("tio_resample3d");
        }
        if (a.cp != nullptr) return launch_planned(resample_planned_kernel<true, 16, 16, 16, 3>);
        return launch_planned(resample_planned_kernel<false, 16, 16, 16, 3>);
      }
      // (a TIGHT / EXACT launch the lean kernel does not take — too small, unaligned — falls through to the exact brick kernel)
      if (fast) {
        if (mode != kPlanNone) return TIO_OK;  // a FAST launch of in-kernel boxes: no plan
        auto launch_fast = [&](auto kernel) -> int {
          if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     static_cast<int>(lds)) != hipSuccess)
            return fail(TIO_ERR_LAUNCH, "tio_resample3d: cannot reserve %zu bytes of LDS", lds);
          hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), lds, s, a, static_cast<const int*>(nullptr));
          return check_launch("tio_resample3d");
        };
        if (a.cp != nullptr) return launch_fast(resample_tile_kernel<true, 0, 16, 16, 16, 3, true>);
        return launch_fast(resample_tile_kernel<false, 0, 16, 16, 16, 3, true>);
      }
    }
    // Large affine-only exact launches of 16^3 bricks are planned too (resample_tile.hpp: the planned box only decides what
    // is staged; the corner evaluation and its reductions leave the head of every block); TIO_EXACT_PLAN=0 switches it off,
    // =2 forces it for small launches (A/B, tests).
    const int* plan_exact = nullptr;
    PlanLease exact_lease;  // held until the brick kernel is enqueued (end of this function)
    {
      // Measured (8 x 256^3): affine 0.497 -> 0.478 ms; elastic launches LOSE (0.588 -> 0.610: 27 vertices with their
      // control-point reads per brick cost the planner more than the brick kernel's own reduction), and so do small ones.
      // (Round 3, with the 32-lanes-per-brick planner: elastic 0.604 -> 0.622 ms, affine + elastic 0.666 -> 0.662: the exact
      // elastic kernel is bound by its coordinate chain, not by what precedes it — profiles/r03_exp23_native.log.)
      const bool want = variant == 0 && a.ablate == 0 && a.cp == nullptr && env.exact_plan != 0 &&
                        (static_cast<int64_t>(a.B) * ((a.Io + 15) / 16) * ((a.Jo + 15) / 16) * ((a.Ko + 15) / 16) >= kPlannedMinBricks ||
                         env.exact_plan == 2);
      if (want) {
        a.tiles_k = (a.Ko + 15) / 16; a.tiles_j = (a.Jo + 15) / 16; a.tiles_i = (a.Io + 15) / 16;
        a.magic_k = a.tiles_k > 1 ? 0xFFFFFFFFu / a.tiles_k + 1u : 0u;
        a.magic_j = a.tiles_j > 1 ? 0xFFFFFFFFu / a.tiles_j + 1u : 0u;
        a.magic_i = a.tiles_i > 1 ? 0xFFFFFFFFu / a.tiles_i + 1u : 0u;
        const int64_t items64 = static_cast<int64_t>(a.B) * a.tiles_i * a.tiles_j * a.tiles_k;
        if (items64 < (1LL << 26)) {
          const int n_items = static_cast<int>(items64);
          // (every plan starts behind kPlanHeaderInts ints — the list header of the exact-coordinate lean road, which shares the leased
          // workspace of the stream with this one and must find it zero)
          const size_t plan_need = (kPlanHeaderInts + static_cast<size_t>(a.B) * 16 + static_cast<size_t>(n_items) * kDescInts) * sizeof(int);
          if (mode == kPlanQuery) { *plan_bytes = static_cast<int64_t>(plan_need); return TIO_OK; }
          int* plan = nullptr;
          bool planned_ahead = false;
          if (mode == kPlanOnly) {
            if (plan_out == nullptr || plan_out_bytes < static_cast<int64_t>(plan_need) || (reinterpret_cast<uintptr_t>(plan_out) & 15) != 0)
              return fail(TIO_ERR_INVALID_ARGUMENT, "tio_resample3d_plan: the plan needs %zu bytes, 16-byte aligned", plan_need);
            plan = plan_out + kPlanHeaderInts;
          } else if (geom->plan_dev != nullptr && geom->plan_bytes >= static_cast<int64_t>(plan_need) &&
                     (reinterpret_cast<uintptr_t>(geom->plan_dev) & 15) == 0) {
            plan = static_cast<int*>(const_cast<void*>(geom->plan_dev)) + kPlanHeaderInts;
            planned_ahead = true;
          } else {
            exact_lease = plan_workspace(s, plan_need);
            if (exact_lease.ptr == nullptr) return fail(TIO_ERR_LAUNCH, "tio_resample3d: cannot allocate the brick plan");
            plan = exact_lease.ptr + kPlanHeaderInts;
          }
          if (!planned_ahead) {
            const int plan_lanes = plan_group(a.cp != nullptr);  // lanes per brick
            const int plan_threads = n_items * plan_lanes > a.B ? n_items * plan_lanes : a.B;
            const dim3 plan_grid((plan_threads + 255) / 256);
            if (a.cp != nullptr) hipLaunchKernelGGL((plan_bricks_kernel<true, 16, 16, 16>), plan_grid, dim3(256), 0, s, a, plan, n_items);
            else hipLaunchKernelGGL((plan_bricks_kernel<false, 16, 16, 16>), plan_grid, dim3(256), 0, s, a, plan, n_items);
          }
          if (mode == kPlanOnly) { *plan_bytes = static_cast<int64_t>(plan_need); return check_launch("tio_resample3d_plan"); }
          plan_exact = plan;
        }
      }
    }
    if (mode != kPlanNone) return TIO_OK;  // bricks with in-kernel boxes: no plan
    switch (variant) {
      case 1: TIO_TILE_SHAPE_F32(16, 8, 32, 3) break;
      case 2: TIO_TILE_SHAPE_F32(8, 8, 32, 4) break;
      case 3: TIO_TILE_SHAPE_F32(8, 16, 32, 2) break;   /* 512 threads, 2 blocks per CU */
      case 4: TIO_TILE_SHAPE_F32(8, 16, 16, 4) break;
      default: TIO_TILE_SHAPE(16, 16, 16, 3) break;
    }
#undef TIO_TILE_SHAPE_F32
#undef TIO_TILE_SHAPE
#undef TIO_TILE_LAUNCH
    return check_launch("tio_resample3d");
  }

  if (mode != kPlanNone) return TIO_OK;  // the gather kernel: no plan
  a.tiles_k = (a.Ko + kLanes - 1) / kLanes;
  a.tiles_j = (a.Jo + kRowsPerBlock - 1) / kRowsPerBlock;
  a.tiles_i = (a.Io + kTileI - 1) / kTileI;
  const int64_t blocks = static_cast<int64_t>(a.B) * a.tiles_i * a.tiles_j * a.tiles_k;
  if (blocks >= (1LL << 31)) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_resample3d: grid too large");
  const dim3 grid(static_cast<unsigned>(blocks)), block(kRowsPerBlock * kLanes);
  const size_t lds = (n_cp > 0 && n_cp <= kMaxCpLds) ? static_cast<size_t>(n_cp) * sizeof(float) : 0;
#define TIO_LAUNCH(EL, DM) hipLaunchKernelGGL((resample_kernel<EL, DM>), grid, block, lds, s, a)
  if (a.cp != nullptr) {
    if (dtmode == 0) TIO_LAUNCH(true, 0); else if (dtmode == 1) TIO_LAUNCH(true, 1); else TIO_LAUNCH(true, 2);
  } else {
    if (dtmode == 0) TIO_LAUNCH(false, 0); else if (dtmode == 1) TIO_LAUNCH(false, 1); else TIO_LAUNCH(false, 2);
  }
#undef TIO_LAUNCH
  return check_launch("tio_resample3d");
}

extern "C" int64_t tio_resample3d_plan_bytes(const tio_resample_geom* geom) {
  bool folded = false;
  int64_t bytes = 0;
  if (geom == nullptr || geom->batch < 1) return 0;
  const int status = resample3d_impl(geom, 0, nullptr, nullptr, &folded, kPlanQuery, nullptr, 0, &bytes);
  return status == TIO_OK ? bytes : 0;
}

extern "C" int tio_resample3d_plan(const tio_resample_geom* geom, void* plan_dev, int64_t plan_bytes, void* stream) {
  bool folded = false;
  int64_t bytes = 0;
  if (geom == nullptr || geom->batch < 1) return tio::fail(TIO_ERR_INVALID_ARGUMENT, "tio_resample3d_plan: no geometry / empty batch");
  const int status = resample3d_impl(geom, 0, nullptr, stream, &folded, kPlanOnly, static_cast<int*>(plan_dev), plan_bytes, &bytes);
  if (status == TIO_OK && bytes == 0) return tio::fail(TIO_ERR_INVALID_ARGUMENT, "tio_resample3d_plan: this geometry's launch does not start from a plan");
  return status;
}

extern "C" int tio_resample3d(const tio_resample_geom* geom, int32_t n_images, const tio_resample_image* images, void* stream) {
  bool folded = false;
  const int status = resample3d_impl(geom, n_images, images, stream, &folded);
  if (status != TIO_OK || folded || geom == nullptr || images == nullptr || geom->batch < 1) return status;
  // out_min_dev of launches that did not fold it into their stores: the plain reduction over what was just written
  const int64_t n_out = static_cast<int64_t>(geom->out_shape[0]) * geom->out_shape[1] * geom->out_shape[2];
  for (int i = 0; i < n_images; i++) {
    const tio_resample_image& im = images[i];
    if (im.out_min_dev == nullptr || im.interp == TIO_LINEAR_ADJOINT) continue;
    const int st = tio_channel_min(im.out, im.dtype, im.channels, n_out, im.out_min_dev, stream);
    if (st != TIO_OK) return st;
  }
  return TIO_OK;
}

