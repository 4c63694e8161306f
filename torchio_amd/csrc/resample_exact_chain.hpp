// resample_exact_chain.hpp — the reference's float32 coordinate chain and fill decision for ONE voxel, operation for
// operation: what the kernels that decide with cheaper arithmetic fall back to where the cheap answer is not safe.
//
//   * resample_nearest.hpp: a label voxel whose FAST coordinate lies within a margin of a half-integer;
//   * the FAST float kernels (resample_fast.hpp, resample_tile_kernel<..., FAST = true>): a voxel whose in-bounds weight
//     ("mask", spatial.py:1719-1728: F.grid_sample of a volume of ones, `mask > 0.5 ? value : fill`) is within a margin
//     of 1/2.  The FAST coordinates are within ~1e-4 voxel of the reference's, which moves a VALUE by less than the
//     1e-4 tolerance — but flips the comparison for the few hundred voxels per 256^3 volume whose weight sits that
//     close to the threshold, and a flipped voxel differs by (value - fill), which a Blur then smears over its
//     neighbourhood (1 142 of 50 M voxels beyond 1e-4).  Same scheme as the label kernel: the
//     cheap line DECIDES wherever |mask - 1/2| > margin; the sampling loops only RECORD the planes of a column that
//     fall inside the margin (one bit each), and a tail behind the loops re-decides those voxels with the exact chain
//     and overwrites what the loop stored (same thread, same address: program order).  The tail is behind the loops
//     because inlined into them the exact chain's registers add to the hot loop's (a first version: 95 -> 157 VGPRs
//     and 27 spilled SGPRs in the lean kernel); behind them the kernel needs the larger of the two, not the sum.
//
// Included by resample.hip after the helpers of the exact kernels (exact_div, normalise_roundtrip, cp_trilerp3).
#pragma once

namespace tio {

// the launch constants the exact chain reads (resample_kernel's ResampleArgs and the nearest kernel's NearestArgs carry
// fields of the same names; the lean kernel fills this from its own argument block)
struct ExactChainArgs {
  int ni, nj, nk, Io, unit_spacing, affine_first;
  float scale_i;
  float sp[3], rsp[3], den[3], rden[3], size_m1[3];
};

// The exact coordinate chain of ONE voxel — the sequence of resample_kernel (resample.hip), operation for operation.
template <bool ELASTIC_POSSIBLE, typename Args>
__device__ __forceinline__ void exact_voxel_coords(const Args& a, const float (&m)[12], bool elastic, const float* __restrict__ cp,
                                                   const Lerp1D& lj, const Lerp1D& lk, int io, float cj, float ck, float& x, float& y,
                                                   float& z) {
  const float ci = static_cast<float>(io);
  float vi = 0.0f, vj = 0.0f, vk = 0.0f;
#define TIO_AFFINE_ROW(M0, M1, M2, M3, A, B, C) \
  __builtin_fmaf(1.0f, M3, __builtin_fmaf(C, M2, __builtin_fmaf(B, M1, __fmul_rn(A, M0))))
  bool done = false;
  if constexpr (ELASTIC_POSSIBLE) {
    if (elastic) {
      const Lerp1D li = lerp_index(io, a.ni, a.Io, a.scale_i);
      const Disp d = cp_trilerp3(cp, a.nj * a.nk * 3, a.nk * 3, li, lj, lk);
      float di = d.i, dj = d.j, dk = d.k;
      if (!a.unit_spacing) {
        di = exact_div(di, a.sp[0], a.rsp[0]);
        dj = exact_div(dj, a.sp[1], a.rsp[1]);
        dk = exact_div(dk, a.sp[2], a.rsp[2]);
      }
      if (a.affine_first) {
        vi = __fadd_rn(TIO_AFFINE_ROW(m[0], m[1], m[2], m[3], ci, cj, ck), di);
        vj = __fadd_rn(TIO_AFFINE_ROW(m[4], m[5], m[6], m[7], ci, cj, ck), dj);
        vk = __fadd_rn(TIO_AFFINE_ROW(m[8], m[9], m[10], m[11], ci, cj, ck), dk);
      } else {
        const float ei = __fadd_rn(ci, di), ej = __fadd_rn(cj, dj), ek = __fadd_rn(ck, dk);
        vi = TIO_AFFINE_ROW(m[0], m[1], m[2], m[3], ei, ej, ek);
        vj = TIO_AFFINE_ROW(m[4], m[5], m[6], m[7], ei, ej, ek);
        vk = TIO_AFFINE_ROW(m[8], m[9], m[10], m[11], ei, ej, ek);
      }
      done = true;
    }
  }
  if (!done) {
    vi = TIO_AFFINE_ROW(m[0], m[1], m[2], m[3], ci, cj, ck);
    vj = TIO_AFFINE_ROW(m[4], m[5], m[6], m[7], ci, cj, ck);
    vk = TIO_AFFINE_ROW(m[8], m[9], m[10], m[11], ci, cj, ck);
  }
#undef TIO_AFFINE_ROW
  x = normalise_roundtrip(vi, a.den[0], a.rden[0], a.size_m1[0]);
  y = normalise_roundtrip(vj, a.den[1], a.rden[1], a.size_m1[1]);
  z = normalise_roundtrip(vk, a.den[2], a.rden[2], a.size_m1[2]);
}

// The in-bounds weight of the reference's fill rule at the exact coordinate: ATen's grid_sampler_3d corner weights of the
// taps inside the volume, accumulated in its tap order (the sequence of resample_kernel / tile_mask).
__device__ __forceinline__ float exact_fill_mask(float x, float y, float z, float hx, float hy, float hz) {
  const float x0 = floorf(x), y0 = floorf(y), z0 = floorf(z);
  const float x1 = x0 + 1.0f, y1 = y0 + 1.0f, z1 = z0 + 1.0f;
  const float wx0 = x1 - x, wx1 = x - x0, wy0 = y1 - y, wy1 = y - y0, wz0 = z1 - z, wz1 = z - z0;
  const bool ox0 = (x0 >= 0.0f) & (x0 <= hx), ox1 = (x1 >= 0.0f) & (x1 <= hx);
  const bool oy0 = (y0 >= 0.0f) & (y0 <= hy), oy1 = (y1 >= 0.0f) & (y1 <= hy);
  const bool oz0 = (z0 >= 0.0f) & (z0 <= hz), oz1 = (z1 >= 0.0f) & (z1 <= hz);
  float mask = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const float w = __fmul_rn(__fmul_rn((k & 1) ? wx1 : wx0, (k & 2) ? wy1 : wy0), (k & 4) ? wz1 : wz0);
    const bool ok = ((k & 1) ? ox1 : ox0) & ((k & 2) ? oy1 : oy0) & ((k & 4) ? oz1 : oz0);
    const float next = __fadd_rn(mask, w);
    mask = ok ? next : mask;
  }
  return mask;
}

// The voxel's value at the exact coordinate, straight from global memory (zero padding, nested fma lerps: the FAST
// interpolant): what a re-decided voxel stores when the exact chain says "keep".
__device__ __forceinline__ float exact_chain_sample(const float* __restrict__ chan, int I, int J, int K, float x, float y, float z) {
  if (!(fabsf(x) <= 1e30f) | !(fabsf(y) <= 1e30f) | !(fabsf(z) <= 1e30f)) return 0.0f;
  const float x0 = floorf(x), y0 = floorf(y), z0 = floorf(z);
  const float fx = x - x0, fy = y - y0, fz = z - z0;
  const float cx = fminf(fmaxf(x0, -2.0f), static_cast<float>(I)), cy = fminf(fmaxf(y0, -2.0f), static_cast<float>(J)),
              cz = fminf(fmaxf(z0, -2.0f), static_cast<float>(K));
  const int ix = static_cast<int>(cx), iy = static_cast<int>(cy), iz = static_cast<int>(cz);
  float v[8];
#pragma unroll
  for (int t = 0; t < 8; t++) {  // t = dx + 2 dy + 4 dz
    const int px = ix + (t & 1), py = iy + ((t >> 1) & 1), pz = iz + (t >> 2);
    const bool ok = (static_cast<unsigned>(px) < static_cast<unsigned>(I)) & (static_cast<unsigned>(py) < static_cast<unsigned>(J)) &
                    (static_cast<unsigned>(pz) < static_cast<unsigned>(K)) & (cx == x0) & (cy == y0) & (cz == z0);
    v[t] = ok ? chan[(static_cast<int64_t>(px) * J + py) * K + pz] : 0.0f;
  }
  const float a00 = __builtin_fmaf(fz, v[4] - v[0], v[0]), a10 = __builtin_fmaf(fz, v[5] - v[1], v[1]);
  const float a01 = __builtin_fmaf(fz, v[6] - v[2], v[2]), a11 = __builtin_fmaf(fz, v[7] - v[3], v[3]);
  const float b0 = __builtin_fmaf(fy, a01 - a00, a00), b1 = __builtin_fmaf(fy, a11 - a10, a10);
  return __builtin_fmaf(fx, b1 - b0, b0);
}

// How far the FAST and the exact in-bounds weight of a voxel can differ: each is a product of three per-axis factors in
// [0, 1], piecewise linear in the coordinate with slope <= 1, so |d mask| <= |dx| + |dy| + |dz|; the two coordinate chains
// differ per axis by less than 1e-7 (row + S + |x|) (measured for the label kernel: profiles/r03_exp20_nearest_margin_scan.log;
// row = sum_c |m_rc| S_out,c + |m_r3| bounds every partial sum of the reference's matmul row).  A voxel near the
// threshold has |x| <= S + 1, an elastic displacement is bounded by |affine part| + |x|.  kFillEps leaves a factor of ten.
constexpr float kFillEps = 1e-6f;
__device__ __forceinline__ float fast_fill_margin(const float (&m)[12], float so_i, float so_j, float so_k, float s_x, float s_y, float s_z) {
  float sum = 0.0f;
#pragma unroll
  for (int r = 0; r < 3; r++)
    sum += 2.0f * (fabsf(m[4 * r]) * so_i + fabsf(m[4 * r + 1]) * so_j + fabsf(m[4 * r + 2]) * so_k + fabsf(m[4 * r + 3]));
  sum += 2.0f * (s_x + s_y + s_z) + 48.0f;
  // NaN / Inf / absurd mappings: an infinite margin — every voxel the FAST mask looks at is re-decided by the exact chain
  return (sum <= 1e30f) ? kFillEps * sum : __builtin_inff();
}

// The tail of a FAST kernel with a fill rule: the planes of this thread's column whose in-bounds weight came within the
// margin of 1/2 (bits of `unsure`, bit t = plane u0 + t) are re-decided by the reference's exact chain; the voxel is stored
// again — the value sampled at the exact coordinate from global memory, or the fill value.  A wave gets here only when one
// of its lanes has such a plane (a few hundred voxels of a 256^3 volume, all on the volume's surface).
template <bool ELASTIC_POSSIBLE, typename Args>
__device__ __forceinline__ void fast_fill_tail(unsigned unsure, const Args& ea, const float* __restrict__ mapping_b, bool elastic, const float* cp,
                                               const Lerp1D& lj, const Lerp1D& lk, int u0, float cj, float ck, const float* __restrict__ in_chan, int I,
                                               int J, int K, float hx, float hy, float hz, float fillv, char* out_chan, int64_t slab_b, unsigned urow,
                                               uint32_t* kmin = nullptr) {
  if (__builtin_amdgcn_ballot_w64(unsure != 0u) == 0ull) return;
  float mm[12];
#pragma unroll
  for (int q = 0; q < 12; q++) mm[q] = mapping_b[q];
#pragma unroll 1
  while (unsure != 0u) {
    const int t = __builtin_ctz(unsure);
    unsure &= unsure - 1u;
    float x, y, z;
    exact_voxel_coords<ELASTIC_POSSIBLE>(ea, mm, elastic, cp, lj, lk, u0 + t, cj, ck, x, y, z);
    const bool keep = exact_fill_mask(x, y, z, hx, hy, hz) > 0.5f;
    const float val = keep ? exact_chain_sample(in_chan, I, J, K, x, y, z) : fillv;
    *reinterpret_cast<float*>(out_chan + static_cast<int64_t>(u0 + t) * slab_b + urow) = val;
    if (kmin != nullptr) *kmin = min(*kmin, float_to_key(val));
  }
}

}  // namespace tio
