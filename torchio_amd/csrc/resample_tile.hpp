// resample_tile.hpp — LDS-staged brick variant of the fused resampler (included by
// resample.hip; shares its argument block and the float32 coordinate chain).
//
// Why: the gather kernel issues 8 wave-wide dword gathers per output voxel; under a
// rotation every one of them touches ~10 cache lines, so the texture-address path
// (not HBM) bounds it at ~1.1 TB/s.  Here one block owns an output brick of
// TI x TJ x TK voxels (TK contiguous along K):
//   A  every thread walks TI output planes of its (j, k) column and keeps the final
//      sampling coordinates (x, y, z) in registers — the same float32 chain as the
//      gather kernel; the elastic lookup collapses to one lerp per component because
//      the k- and j-lerps of a control plane are invariant along the thread's column;
//   B  the block reduces the brick's input bounding box (floor coordinates + 1,
//      clamped to the volume plus a one-voxel zero apron, z aligned to 4) with DPP
//      wave reductions and stages it in LDS with coalesced 16-byte loads — every input
//      line is fetched once per brick instead of once per tap.  A brick whose box does
//      not fit the LDS budget is processed in 2 or 4 passes over its planes;
//   C  the 8 taps come from LDS (ds_read2_b32 pairs along z), accumulate in ATen's
//      order and are stored as TK-contiguous row segments.
// Nearest images, non-finite geometry and boxes that do not fit even per quarter fall
// back to a per-voxel global gather with identical arithmetic.  Output is bit-identical
// to the gather kernel and to the CPU oracle (tests/native/resample_bench.cpp).
//
// Instruction budget matters more than bytes here (measured on MI355X,
// tests/native/valu_rates.cpp): float32 add/mul/fma and integer add issue at one
// wave-instruction per ~2.6 cycles per SIMD, everything else (floor, cvt, min/max,
// shifts, integer multiplies, selects, DPP) at ~4.3, ds_bpermute shuffles at ~25.  Hence
// float address arithmetic, endpoint bounding boxes for monotone (affine-only) columns,
// DPP instead of shuffles, and no per-voxel selects.
#pragma once

namespace tio {

// Hide a value's provenance from the optimiser: without this LLVM hoists the weight /
// index arithmetic of all TI voxels out of the pass, image and channel loops and spills
// hundreds of registers.  Not volatile (the scheduler may move it freely) but tied to a
// loop-varying scalar so it cannot leave the loop it is written in.
#define TIO_OPAQUE3(A, B, C, DEP) asm("" : "+v"(A), "+v"(B), "+v"(C) : "s"(DEP))

constexpr int kTileRedInts = 512;    // LDS ints: 7 reduction slots x (8 ints x <= 8 waves), then 4 pass boxes x 16
constexpr int kTileBoxBase = 448;
constexpr int kTileStashPlanes = 4;
#ifndef TILE_GROUP
#define TILE_GROUP 4
#endif  // planes parked in the brick area by the per-voxel fallback

__device__ __forceinline__ unsigned fastdiv(unsigned n, unsigned magic, unsigned d) {
  return d == 1 ? n : __umulhi(n, magic);  // exact for n * d < 2^32
}
__device__ __forceinline__ unsigned fastdiv_magic(unsigned d) { return d > 1 ? 0xFFFFFFFFu / d + 1u : 0u; }
// The same with one correction step: exact for every n, d < 2^31 (q d <= n + d cannot wrap).  magic = ceil(2^32 / d), so the multiply-high is
// floor(n / d) or one more (the excess n e / (d 2^32), e < d, is below 1); what decodes a block index uses this form —
// block counts reach 2^31 and the lean kernel divides by the bricks of a whole batch element (B bpe^2 >= 2^32 for one
// 672^3 volume: ADVICE r3, the last brick of element 0 went to element 1).  Scalar operands: three SALU instructions.
__device__ __forceinline__ unsigned fastdiv_exact(unsigned n, unsigned magic, unsigned d) {
  if (d == 1) return n;
  unsigned q = __umulhi(n, magic);
  if (q * d > n) q--;
  return q;
}

// wave-wide max in 4 DPP steps + 4 readlanes (validated by tests/native/dpp_check.cpp)
__device__ __forceinline__ int wave_max_i32(int v) {
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false));  // row_half_mirror
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false));  // row_mirror
  const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
  const int r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
  return max(max(r0, r1), max(r2, r3));
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  // unsigned order of v = signed order of v ^ 0x80000000; complementing reverses it without overflow: min = ~max(~.)
  const int flipped = static_cast<int>((~v) ^ 0x80000000u);
  return ~(static_cast<uint32_t>(wave_max_i32(flipped)) ^ 0x80000000u);
}

// block-wide max of N <= 8 values; every call uses its own LDS slot, so one barrier per call
template <int NW, int N>
__device__ __forceinline__ void block_max6(int (&r)[N], int* s_red, int slot, int wave, int lane) {
  static_assert(N <= 8, "a slot holds 8 ints per wave");
#pragma unroll
  for (int q = 0; q < N; q++) r[q] = wave_max_i32(r[q]);
  int* s = s_red + slot * (NW * 8);
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < N; q++) s[wave * 8 + q] = r[q];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < N; q++) {
    int v = s[q];
#pragma unroll
    for (int w = 1; w < NW; w++) v = max(v, s[w * 8 + q]);
    r[q] = __builtin_amdgcn_readfirstlane(v);
  }
}

// g = 2 v / max(S-1,1) - 1 then ATen's un-normalise ((g + 1) / 2) * (S - 1), with the two
// exact scalings folded into the constants: v / (den/2) is the same real quotient as
// 2v / den, and (t * 0.5) * h == t * (0.5 h) because t * 0.5 is exact.  short_div: one
// Markstein refinement is already correctly rounded for every divisor k/2, k <= 8192
// (exhaustive over all mantissas: tests/native/divtest.c).
template <bool SHORT>
__device__ __forceinline__ float normalise_roundtrip_folded(float v, float dh, float rdh, float half_h) {
  float q = __fmul_rn(v, rdh);
  float e = __builtin_fmaf(-dh, q, v);
  q = __builtin_fmaf(e, rdh, q);
  if constexpr (!SHORT) {
    e = __builtin_fmaf(-dh, q, v);
    q = __builtin_fmaf(e, rdh, q);
  }
  const float g = __fsub_rn(q, 1.0f);
  return __fmul_rn(__fadd_rn(g, 1.0f), half_h);
}
// block-uniform choice at run time (the compiler turns it into a select over both results;
// the hot loops branch once outside instead and call the template)
__device__ __forceinline__ float normalise_roundtrip_folded(float v, float dh, float rdh, float half_h, bool short_div) {
  return short_div ? normalise_roundtrip_folded<true>(v, dh, rdh, half_h) : normalise_roundtrip_folded<false>(v, dh, rdh, half_h);
}

// one control plane of the displacement field for this thread's (j, k) column:
// the two inner lerps of ATen's upsample_trilinear3d (K innermost, then J)
__device__ __forceinline__ void cp_plane(const float* __restrict__ cp, int ii, int s_i, int s_j, const Lerp1D& lj,
                                         const Lerp1D& lk, float (&P)[3]) {
  const float* p0 = cp + ii * s_i + lj.i0 * s_j;
  const float* p1 = cp + ii * s_i + lj.i1 * s_j;
  const int k0 = lk.i0 * 3, k1 = lk.i1 * 3;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float a0 = lerp2(p0[k0 + c], lk.l0, p0[k1 + c], lk.l1);
    const float a1 = lerp2(p1[k0 + c], lk.l0, p1[k1 + c], lk.l1);
    P[c] = lerp2(a0, lj.l0, a1, lj.l1);
  }
}

// dtype access by kernel mode: 0 = float32 only, 1 = float32/int16/uint8/int32, 2 = any
template <int DTMODE>
__device__ __forceinline__ float load_mode(const void* p, int dtype, int64_t i) {
  if constexpr (DTMODE == 0) {
    return Elem<TIO_F32>::load(p, i);
  } else if constexpr (DTMODE == 1) {
    switch (dtype) {
      case TIO_F32: return Elem<TIO_F32>::load(p, i);
      case TIO_I16: return Elem<TIO_I16>::load(p, i);
      case TIO_U8: return Elem<TIO_U8>::load(p, i);
      default: return Elem<TIO_I32>::load(p, i);
    }
  } else {
    return load_as_float(p, dtype, i);
  }
}

template <int DTMODE>
__device__ __forceinline__ void store_mode(void* p, int dtype, int64_t i, float v) {
  if constexpr (DTMODE == 0) {
    Elem<TIO_F32>::store(p, i, v);
  } else if constexpr (DTMODE == 1) {
    switch (dtype) {
      case TIO_F32: Elem<TIO_F32>::store(p, i, v); break;
      case TIO_I16: Elem<TIO_I16>::store(p, i, v); break;
      case TIO_U8: Elem<TIO_U8>::store(p, i, v); break;
      default: Elem<TIO_I32>::store(p, i, v); break;
    }
  } else {
    store_from_float(p, dtype, i, v);
  }
}

// store through a block-uniform base pointer + 32-bit byte offset (the launcher guarantees
// that one output plane is below 2 GiB): the address stays "SGPR base + VGPR offset", no
// 64-bit vector arithmetic per voxel
template <int DTMODE>
__device__ __forceinline__ void store_at(char* base, int dtype, unsigned byte_off, float v) {
  if constexpr (DTMODE == 0) {
    *reinterpret_cast<float*>(base + byte_off) = v;
  } else {
    store_mode<DTMODE>(base + byte_off, dtype, 0, v);
  }
}

// Full ATen semantics for ONE voxel of one image from global memory (all channels):
// per-tap bounds, zero padding, in-bounds weight mask, fill — the arithmetic of the
// gather kernel's boundary path.  `interior` (block uniform) promises that every tap
// is in bounds (only used to shorten the nearest path).
template <int DTMODE>
__device__ __forceinline__ void gather_voxel(const ImgArgs& g, const ResampleArgs& a, int b, int64_t n_in, int64_t n_out,
                                             int o_idx, float x, float y, float z, bool interior) {
  const float hx = a.size_m1[0], hy = a.size_m1[1], hz = a.size_m1[2];
  const bool linear = g.interp == TIO_LINEAR;
  const bool need_w = linear | (g.fill != nullptr);
  const int es = dtype_size(g.dtype);
  float w[8];
  int off[8];
  unsigned okbits = 0xFFu;
  float mask = 1.0f;
  if (need_w && !(interior && !linear)) {
    const float x0 = floorf(x), y0 = floorf(y), z0 = floorf(z);
    const float x1 = x0 + 1.0f, y1 = y0 + 1.0f, z1 = z0 + 1.0f;
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
    const bool bx0 = (x0 >= 0.0f) & (x0 <= hx), bx1 = (x1 >= 0.0f) & (x1 <= hx);
    const bool by0 = (y0 >= 0.0f) & (y0 <= hy), by1 = (y1 >= 0.0f) & (y1 <= hy);
    const bool bz0 = (z0 >= 0.0f) & (z0 <= hz), bz1 = (z1 >= 0.0f) & (z1 <= hz);
    const int ix0 = static_cast<int>(fminf(fmaxf(x0, 0.0f), hx)), ix1 = static_cast<int>(fminf(fmaxf(x1, 0.0f), hx));
    const int iy0 = static_cast<int>(fminf(fmaxf(y0, 0.0f), hy)), iy1 = static_cast<int>(fminf(fmaxf(y1, 0.0f), hy));
    const int iz0 = static_cast<int>(fminf(fmaxf(z0, 0.0f), hz)), iz1 = static_cast<int>(fminf(fmaxf(z1, 0.0f), hz));
    okbits = 0;
    mask = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const bool ok = ((k & 1) ? bx1 : bx0) & ((k & 2) ? by1 : by0) & ((k & 4) ? bz1 : bz0);
      off[k] = (((k & 1) ? ix1 : ix0) * a.J + ((k & 2) ? iy1 : iy0)) * a.K + ((k & 4) ? iz1 : iz0);
      okbits |= ok ? (1u << k) : 0u;
      const float next = __fadd_rn(mask, w[k]);  // same order as ATen's accumulation
      mask = ok ? next : mask;
    }
  }
  int offn = 0;
  bool okn = true;
  if (!linear) {  // nearbyint = round half to even (v_rndne_f32)
    const float xn = rintf(x), yn = rintf(y), zn = rintf(z);
    if (interior) {
      offn = (static_cast<int>(xn) * a.J + static_cast<int>(yn)) * a.K + static_cast<int>(zn);
    } else {
      okn = (xn >= 0.0f) & (xn <= hx) & (yn >= 0.0f) & (yn <= hy) & (zn >= 0.0f) & (zn <= hz);
      offn = (static_cast<int>(fminf(fmaxf(xn, 0.0f), hx)) * a.J + static_cast<int>(fminf(fmaxf(yn, 0.0f), hy))) * a.K +
             static_cast<int>(fminf(fmaxf(zn, 0.0f), hz));
    }
  }
  for (int c = 0; c < g.channels; c++) {
    const int64_t bc = static_cast<int64_t>(b) * g.channels + c;
    const char* p = static_cast<const char*>(g.in) + bc * n_in * es;
    float val;
    if (linear) {
      val = 0.0f;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const float v = load_mode<DTMODE>(p, g.dtype, off[k]);
        const float next = __fadd_rn(val, __fmul_rn(v, w[k]));
        val = ((okbits >> k) & 1u) ? next : val;
      }
    } else {
      const float v = load_mode<DTMODE>(p, g.dtype, offn);
      val = okn ? v : 0.0f;
    }
    if (g.fill != nullptr) val = (mask > 0.5f) ? val : g.fill[c];
    store_mode<DTMODE>(g.out, g.dtype, bc * n_out + o_idx, val);
  }
}

// ---- the staged box of one pass (block uniform, kept in SGPRs) -------------------------
// The brick plan written by plan_bricks_kernel (resample_fast.hpp): 16 dwords per brick after 16 floats per batch element
enum : int { kDescInts = 16, kDescStaged = 0, kDescOutside = 1, kDescSlow = 2, kDescGated = 3 };

struct TileBox {
  int bx0, bx1, by0, by1, zlo, zhi, za;  // inclusive tap range per axis; za = zlo aligned down to 4
  int Lx, Ly, Lz;                        // staged extent (Lz a multiple of 4, rows dense)
  int interior, outside, fits;
};

__device__ __forceinline__ TileBox make_box(const int (&r)[6], const ResampleArgs& a, bool weird) {
  TileBox bx;
  const int xmin = -r[0], xmax = r[1], ymin = -r[2], ymax = r[3], zmin = -r[4], zmax = r[5];
  // every tap in bounds ⇔ first taps ≥ 0 and second taps ≤ S-1
  bx.interior = (xmin >= 0) & (xmax + 1 <= a.I - 1) & (ymin >= 0) & (ymax + 1 <= a.J - 1) & (zmin >= 0) &
                (zmax + 1 <= a.K - 1) & !weird;
  // every tap of every voxel out of bounds (second taps < 0 or first taps > S-1 on some axis)
  bx.outside = ((xmax + 1 < 0) | (xmin > a.I - 1) | (ymax + 1 < 0) | (ymin > a.J - 1) | (zmax + 1 < 0) | (zmin > a.K - 1)) & !weird;
  // staged box: every tap of every voxel, NOT clipped to the volume — cells outside are
  // staged as zeros, so boundary bricks address the brick exactly like interior ones (no
  // per-tap clamping; adding +0 terms never changes ATen's partial sums).  A box whose
  // tracked range hit the +-kTileFar clamp is either `outside` or wider than 4096 → !fits.
  bx.bx0 = xmin; bx.bx1 = xmax + 1;
  bx.by0 = ymin; bx.by1 = ymax + 1;
  bx.zlo = zmin; bx.zhi = zmax + 1;
  bx.za = bx.zlo & ~3;
  bx.Lx = bx.bx1 - bx.bx0 + 1; bx.Ly = bx.by1 - bx.by0 + 1; bx.Lz = ((bx.zhi + 4) & ~3) - bx.za;
  bx.fits = !weird && (static_cast<int64_t>(bx.Lx) * bx.Ly * bx.Lz <= static_cast<int64_t>(a.tile_cap)) && (bx.Lx <= 4096) &&
            (bx.Ly <= 4096);
  return bx;
}

// floor + clamp (coordinates may be astronomically far away) of a float bound pair →
// the two ints the reduction maximises: -min first tap (≥ -2) and max first tap (≤ S)
constexpr float kTileFar = 8192.0f;  // first taps are tracked in [-kTileFar, S + kTileFar]; beyond → never "fits"

__device__ __forceinline__ void bound_ints(float lo, float hi, float cap, int& neg_lo, int& pos_hi) {
  neg_lo = -static_cast<int>(fminf(fmaxf(floorf(lo), -kTileFar), cap + kTileFar));
  pos_hi = static_cast<int>(fminf(fmaxf(floorf(hi), -kTileFar), cap + kTileFar));
}

__device__ __forceinline__ void store_box(int* s, const TileBox& b) {
  s[0] = b.bx0; s[1] = b.bx1; s[2] = b.by0; s[3] = b.by1; s[4] = b.zlo; s[5] = b.zhi; s[6] = b.za;
  s[7] = b.Lx; s[8] = b.Ly; s[9] = b.Lz; s[10] = b.interior; s[11] = b.outside; s[12] = b.fits;
}
__device__ __forceinline__ TileBox load_box(const int* s) {
  TileBox b;
#define TIO_RFL(I) __builtin_amdgcn_readfirstlane(s[I])
  b.bx0 = TIO_RFL(0); b.bx1 = TIO_RFL(1); b.by0 = TIO_RFL(2); b.by1 = TIO_RFL(3); b.zlo = TIO_RFL(4); b.zhi = TIO_RFL(5);
  b.za = TIO_RFL(6); b.Lx = TIO_RFL(7); b.Ly = TIO_RFL(8); b.Lz = TIO_RFL(9); b.interior = TIO_RFL(10);
  b.outside = TIO_RFL(11); b.fits = TIO_RFL(12);
#undef TIO_RFL
  return b;
}

// ---- stage the box in LDS: dense rows of Lz floats, positions outside the volume are 0 --
// float32, K % 4 == 0, 16-byte aligned base: one 16-byte chunk per lane, LPR lanes per
// row (a power of two ≥ Lz/4), row coordinates advanced incrementally (no divisions in
// the loop).
template <int NT>
__device__ __forceinline__ void stage_brick_f32x4(float* __restrict__ tile, const float* __restrict__ src, int tid,
                                                  const TileBox& bx, int I, int J, int K) {
  const int cpr = bx.Lz >> 2;  // 16-byte chunks per row
  if (cpr > 64) {              // absurdly long rows: plain strided loop
    const unsigned total = static_cast<unsigned>(bx.Lx * bx.Ly) * cpr;
    const unsigned m_cpr = fastdiv_magic(cpr), m_ly = fastdiv_magic(bx.Ly);
    for (unsigned id = tid; id < total; id += NT) {
      const unsigned row = fastdiv(id, m_cpr, cpr), ch = id - row * cpr;
      const unsigned xr = fastdiv(row, m_ly, bx.Ly), yr = row - xr * bx.Ly;
      const int gx = bx.bx0 + static_cast<int>(xr), gy = bx.by0 + static_cast<int>(yr), gz = bx.za + 4 * static_cast<int>(ch);
      float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if ((static_cast<unsigned>(gx) < static_cast<unsigned>(I)) & (static_cast<unsigned>(gy) < static_cast<unsigned>(J)) &
          (static_cast<unsigned>(gz) < static_cast<unsigned>(K)))
        v = *reinterpret_cast<const float4*>(src + (gx * J + gy) * K + gz);
      *reinterpret_cast<float4*>(tile + 4 * id) = v;
    }
    return;
  }
  const int lpr_log = cpr <= 4 ? 2 : (cpr <= 8 ? 3 : (cpr <= 16 ? 4 : (cpr <= 32 ? 5 : 6)));
  const int rows = bx.Lx * bx.Ly;
  const int rpi = NT >> lpr_log;  // rows per block iteration
  const int ch = tid & ((1 << lpr_log) - 1);
  int r = tid >> lpr_log;
  const int gz = bx.za + 4 * ch;
  const bool ch_ok = (ch < cpr) & (static_cast<unsigned>(gz) < static_cast<unsigned>(K));
  const unsigned m_ly = fastdiv_magic(bx.Ly);
  int xr = static_cast<int>(fastdiv(r, m_ly, bx.Ly));
  int yr = r - xr * bx.Ly;
  const int dx = static_cast<int>(fastdiv(rpi, m_ly, bx.Ly)), dy = rpi - dx * bx.Ly;  // uniform row step
  int lds = r * bx.Lz + 4 * ch;
  const int lds_step = rpi * bx.Lz;
  // All of a thread's chunks are requested before the first one is written to LDS: one
  // memory round trip per brick instead of one per group (U * rpi rows cover the box in a
  // single sweep for every box that fits the default LDS budget).
  constexpr int U = 10;
  while (r < rows) {  // NB: r differs between lanes by < rpi, the loop is exec-masked at the tail
    float4 v[U];
    const int r0 = r, lds0 = lds;
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int gx = bx.bx0 + xr, gy = bx.by0 + yr;
      v[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if ((r < rows) & ch_ok & (static_cast<unsigned>(gx) < static_cast<unsigned>(I)) & (static_cast<unsigned>(gy) < static_cast<unsigned>(J)))
        v[u] = *reinterpret_cast<const float4*>(src + (gx * J + gy) * K + gz);
      r += rpi; lds += lds_step;
      xr += dx; yr += dy;
      if (yr >= bx.Ly) { yr -= bx.Ly; xr += 1; }
    }
    if (ch < cpr) {
#pragma unroll
      for (int u = 0; u < U; u++)
        if (r0 + u * rpi < rows) *reinterpret_cast<float4*>(tile + lds0 + u * lds_step) = v[u];
    }
  }
}

// Sixteen zero bytes for a chunk of the box that lies outside the volume — as an instruction the compiler does not see (round 5,
// first found in resample_lean_exact.hpp; used by every staging loop that mixes DMA and zero chunks): in front of a plain LDS
// store it puts `s_waitcnt vmcnt(0)`, because the store may alias an LDS-DMA in flight for all it knows, and every DMA
// instruction with a lane outside the volume then waited for ALL the box's earlier DMA to land.  The chunks are disjoint from
// every DMA destination by construction; tile_dma_wait() waits for the LDS counter as well before the barrier.
__device__ __forceinline__ void lds_zero_chunk(float* chunk) {
  typedef __attribute__((address_space(3))) float* lds_wfloat_ptr;
  typedef float zero4_t __attribute__((ext_vector_type(4)));
  const zero4_t zero = {0.0f, 0.0f, 0.0f, 0.0f};
  const unsigned addr = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_wfloat_ptr)(chunk)));
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(zero) : "memory");
}

// float32, K % 4 == 0, 16-byte aligned base: global → LDS directly
// (global_load_lds_dwordx4: each lane names one 16-byte chunk, a wave fills 1 KiB of
// consecutive LDS), so the brick never passes through VGPRs.  Chunks outside the volume are
// written as zeros by the lanes that own them.  Asynchronous: the caller waits
// (tile_dma_wait) right before the barrier that precedes the first read.
template <int NT, bool INSIDE>
__device__ __forceinline__ void stage_brick_dma_loop(float* __restrict__ tile, const float* __restrict__ src, int tid,
                                                     const TileBox& bx, int I, int J, int K) {
  const unsigned cpr = static_cast<unsigned>(bx.Lz) >> 2;
  const unsigned total = static_cast<unsigned>(bx.Lx * bx.Ly) * cpr;
  const unsigned m_cpr = fastdiv_magic(cpr), m_ly = fastdiv_magic(bx.Ly);
  const int wave_base = __builtin_amdgcn_readfirstlane(tid >> 6) << 6;  // first chunk id of this wave in an iteration (SGPR)
  typedef __attribute__((address_space(3))) float* lds_float_ptr;
  typedef __attribute__((address_space(1))) const char* global_byte_ptr;
  // element offset of chunk (xr, yr, ch): origin + row K + xr (J - Ly) K + 4 ch with row = xr Ly + yr;
  // byte offsets fit 32 bits (the launcher keeps larger volumes on the gather path)
  const int origin = (bx.bx0 * J + bx.by0) * K + bx.za;
  const int JmLyK = (J - bx.Ly) * K;
  for (unsigned base = 0; base < total; base += NT) {
    const unsigned id = base + static_cast<unsigned>(tid);
    if (id < total) {  // tail lanes are masked off: nothing is written past the box
      const unsigned row = fastdiv(id, m_cpr, cpr);
      const unsigned ch = id - row * cpr;
      const unsigned xr = fastdiv(row, m_ly, bx.Ly);
      bool ok = true;
      if constexpr (!INSIDE) {
        const unsigned yr = row - xr * bx.Ly;
        ok = (static_cast<unsigned>(bx.bx0 + static_cast<int>(xr)) < static_cast<unsigned>(I)) &
             (static_cast<unsigned>(bx.by0 + static_cast<int>(yr)) < static_cast<unsigned>(J)) &
             (static_cast<unsigned>(bx.za + 4 * static_cast<int>(ch)) < static_cast<unsigned>(K));
      }
      if (ok) {
        const unsigned off = static_cast<unsigned>(origin + static_cast<int>(row) * K + static_cast<int>(xr) * JmLyK +
                                                   4 * static_cast<int>(ch));
        // the hardware writes lane l of the wave at (wave-uniform LDS base) + 16 l
        lds_float_ptr dst = (lds_float_ptr)(tile) + 4 * (base + static_cast<unsigned>(wave_base));
        __builtin_amdgcn_global_load_lds((global_byte_ptr)(src) + 4u * off, dst, 16, 0, 0);
      } else {
        lds_zero_chunk(tile + 4 * id);
      }
    }
  }
}

template <int NT>
__device__ __forceinline__ void stage_brick_dma(float* __restrict__ tile, const float* __restrict__ src, int tid,
                                                const TileBox& bx, int I, int J, int K) {
  if (bx.interior)  // block uniform: no chunk can be outside
    stage_brick_dma_loop<NT, true>(tile, src, tid, bx, I, J, K);
  else
    stage_brick_dma_loop<NT, false>(tile, src, tid, bx, I, J, K);
}

// (lgkmcnt too: the zero chunks of boxes that leave the volume are LDS stores the compiler does not see — lds_zero_chunk above)
__device__ __forceinline__ void tile_dma_wait() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }

template <int NT, int DTMODE>
__device__ __forceinline__ void stage_brick_generic(float* __restrict__ tile, const void* __restrict__ src, int dtype,
                                                    int tid, const TileBox& bx, int I, int J, int K) {
  const unsigned total = static_cast<unsigned>(bx.Lx * bx.Ly * bx.Lz);
  const unsigned m_lz = fastdiv_magic(bx.Lz), m_ly = fastdiv_magic(bx.Ly);
  for (unsigned id = tid; id < total; id += NT) {
    const unsigned row = fastdiv(id, m_lz, bx.Lz);
    const unsigned zc = id - row * bx.Lz;
    const unsigned xr = fastdiv(row, m_ly, bx.Ly);
    const unsigned yr = row - xr * bx.Ly;
    const int gx = bx.bx0 + static_cast<int>(xr), gy = bx.by0 + static_cast<int>(yr), gz = bx.za + static_cast<int>(zc);
    float v = 0.0f;
    if ((static_cast<unsigned>(gx) < static_cast<unsigned>(I)) & (static_cast<unsigned>(gy) < static_cast<unsigned>(J)) &
        (static_cast<unsigned>(gz) < static_cast<unsigned>(K)))
      v = load_mode<DTMODE>(src, dtype, (gx * J + gy) * K + gz);
    tile[id] = v;
  }
}

// ---- sampling from the staged brick ---------------------------------------------------
// Split in two halves so that a group of voxels can have all of its LDS reads in flight
// before the first accumulation starts (the caller fences the halves with
// sched_barrier; left alone the scheduler serialises voxel after voxel and every
// ds_read latency is exposed).
struct TapSet {
  float v[8];  // tap values, ATen order: bit0 = x+1, bit1 = y+1, bit2 = z+1
  float wx0, wx1, wy0, wy1, wz0, wz1;
};

// constants of the float address arithmetic of one staged box (element offsets; exact:
// every term is an integer far below 2^24)
struct TileAddr {
  float ox, oy, oz;     // box origin (bx0, by0, za) as floats
  float sXbf, sYbf;     // byte strides of x and y in the brick, as floats
  float base_f;         // LDS byte address of the brick, as a float
  unsigned sXb, sYb, sXYb;
  float c_f;            // base_f - ox sXb - oy sYb - 4 oz: the constant of the address formed from ABSOLUTE indices (tile_issue_folded)
};

typedef __attribute__((address_space(3))) const float* lds_cfloat_ptr;

// Interior brick: all 8 taps are inside the volume and inside the box; the 4 (z, z+1)
// pairs are adjacent dwords (ds_read2_b32).  LAUNDER hides the coordinates from the
// optimiser (needed whenever this sits in a loop over passes / images / channels).
template <bool LAUNDER>
__device__ __forceinline__ void tile_issue_interior(TapSet& ts, float x, float y, float z, const TileAddr& ta, int dep) {
  if constexpr (LAUNDER) TIO_OPAQUE3(x, y, z, dep);
  const float x0 = floorf(x), y0 = floorf(y), z0 = floorf(z);
  const float x1 = x0 + 1.0f, y1 = y0 + 1.0f, z1 = z0 + 1.0f;
  ts.wx0 = x1 - x; ts.wx1 = x - x0; ts.wy0 = y1 - y; ts.wy1 = y - y0; ts.wz0 = z1 - z; ts.wz1 = z - z0;
  // LDS byte address base + ((x0-ox) sX + (y0-oy) sY + (z0-oz)) * 4 in float32 (every term an
  // integer far below 2^24: exact), one conversion, three integer adds
  const float af = __builtin_fmaf(x0 - ta.ox, ta.sXbf, __builtin_fmaf(y0 - ta.oy, ta.sYbf, __builtin_fmaf(z0 - ta.oz, 4.0f, ta.base_f)));
  const unsigned addr = static_cast<unsigned>(static_cast<int>(af));
  lds_cfloat_ptr q00 = reinterpret_cast<lds_cfloat_ptr>(static_cast<uintptr_t>(addr));
  lds_cfloat_ptr q10 = reinterpret_cast<lds_cfloat_ptr>(static_cast<uintptr_t>(addr + ta.sXb));
  lds_cfloat_ptr q01 = reinterpret_cast<lds_cfloat_ptr>(static_cast<uintptr_t>(addr + ta.sYb));
  lds_cfloat_ptr q11 = reinterpret_cast<lds_cfloat_ptr>(static_cast<uintptr_t>(addr + ta.sXYb));
  ts.v[0] = q00[0]; ts.v[4] = q00[1];
  ts.v[1] = q10[0]; ts.v[5] = q10[1];
  ts.v[2] = q01[0]; ts.v[6] = q01[1];
  ts.v[3] = q11[0]; ts.v[7] = q11[1];
}

// The same eight reads with the box origin folded into the constant: base + (x0 - ox) sX + (y0 - oy) sY + 4 (z0 - oz) =
// x0 sX + y0 sY + 4 z0 + c_f — three subtractions per voxel less.  Every partial sum is an integer, exact in float32
// below 2^24; the ABSOLUTE indices make the terms larger than the relative ones: callers check box_address_fits first.
__device__ __forceinline__ void tile_issue_folded(TapSet& ts, float x, float y, float z, const TileAddr& ta) {
  const float x0 = floorf(x), y0 = floorf(y), z0 = floorf(z);
  const float x1 = x0 + 1.0f, y1 = y0 + 1.0f, z1 = z0 + 1.0f;
  ts.wx0 = x1 - x; ts.wx1 = x - x0; ts.wy0 = y1 - y; ts.wy1 = y - y0; ts.wz0 = z1 - z; ts.wz1 = z - z0;
  const float af = __builtin_fmaf(x0, ta.sXbf, __builtin_fmaf(y0, ta.sYbf, __builtin_fmaf(z0, 4.0f, ta.c_f)));
  const unsigned addr = static_cast<unsigned>(static_cast<int>(af));
  lds_cfloat_ptr q00 = reinterpret_cast<lds_cfloat_ptr>(static_cast<uintptr_t>(addr));
  lds_cfloat_ptr q10 = reinterpret_cast<lds_cfloat_ptr>(static_cast<uintptr_t>(addr + ta.sXb));
  lds_cfloat_ptr q01 = reinterpret_cast<lds_cfloat_ptr>(static_cast<uintptr_t>(addr + ta.sYb));
  lds_cfloat_ptr q11 = reinterpret_cast<lds_cfloat_ptr>(static_cast<uintptr_t>(addr + ta.sXYb));
  ts.v[0] = q00[0]; ts.v[4] = q00[1];
  ts.v[1] = q10[0]; ts.v[5] = q10[1];
  ts.v[2] = q01[0]; ts.v[6] = q01[1];
  ts.v[3] = q11[0]; ts.v[7] = q11[1];
}

// ... and its condition (block uniform; the integers of a box): every partial sum of the folded address — at most
// |x| sX + |y| sY + 4 |z| + the LDS size, the indices anywhere in the box — stays below 2^23.  A 512^3 volume reaches 2^21.
__device__ __forceinline__ bool box_address_fits(int bx0, int by0, int za, int Lx, int Ly, int cpr) {
  const float sY = static_cast<float>(cpr) * 16.0f, sX = static_cast<float>(Ly) * sY;
  const float ax = fmaxf(fabsf(static_cast<float>(bx0)), fabsf(static_cast<float>(bx0) + static_cast<float>(Lx)));
  const float ay = fmaxf(fabsf(static_cast<float>(by0)), fabsf(static_cast<float>(by0) + static_cast<float>(Ly)));
  const float az = fmaxf(fabsf(static_cast<float>(za)), fabsf(static_cast<float>(za) + 4.0f * static_cast<float>(cpr)));
  return ax * sX + ay * sY + 4.0f * az + 262144.0f < 8388608.0f;
}

// Fast intensity path (precision = TIO_PRECISION_FAST): the same trilinear interpolant as three
// nested lerps, each one subtraction and one fma — 14 operations instead of 28, a different
// rounding sequence (within ~1e-6 of the exact one; the contract for intensities is 1e-4).
__device__ __forceinline__ float tile_finish_fast(const TapSet& ts) {
  const float c00 = __builtin_fmaf(ts.wx1, ts.v[1] - ts.v[0], ts.v[0]);
  const float c10 = __builtin_fmaf(ts.wx1, ts.v[3] - ts.v[2], ts.v[2]);
  const float c01 = __builtin_fmaf(ts.wx1, ts.v[5] - ts.v[4], ts.v[4]);
  const float c11 = __builtin_fmaf(ts.wx1, ts.v[7] - ts.v[6], ts.v[6]);
  const float c0 = __builtin_fmaf(ts.wy1, c10 - c00, c00);
  const float c1 = __builtin_fmaf(ts.wy1, c11 - c01, c01);
  return __builtin_fmaf(ts.wz1, c1 - c0, c0);
}

// Weights and accumulation order are ATen's grid_sampler_3d.
__device__ __forceinline__ float tile_finish(const TapSet& ts) {
  float val = __fadd_rn(0.0f, __fmul_rn(ts.v[0], __fmul_rn(__fmul_rn(ts.wx0, ts.wy0), ts.wz0)));
  val = __fadd_rn(val, __fmul_rn(ts.v[1], __fmul_rn(__fmul_rn(ts.wx1, ts.wy0), ts.wz0)));
  val = __fadd_rn(val, __fmul_rn(ts.v[2], __fmul_rn(__fmul_rn(ts.wx0, ts.wy1), ts.wz0)));
  val = __fadd_rn(val, __fmul_rn(ts.v[3], __fmul_rn(__fmul_rn(ts.wx1, ts.wy1), ts.wz0)));
  val = __fadd_rn(val, __fmul_rn(ts.v[4], __fmul_rn(__fmul_rn(ts.wx0, ts.wy0), ts.wz1)));
  val = __fadd_rn(val, __fmul_rn(ts.v[5], __fmul_rn(__fmul_rn(ts.wx1, ts.wy0), ts.wz1)));
  val = __fadd_rn(val, __fmul_rn(ts.v[6], __fmul_rn(__fmul_rn(ts.wx0, ts.wy1), ts.wz1)));
  val = __fadd_rn(val, __fmul_rn(ts.v[7], __fmul_rn(__fmul_rn(ts.wx1, ts.wy1), ts.wz1)));
  return val;
}

// In-bounds weight mask of ATen's fill logic (spatial.py:1719-1728): the trilinear weights of
// the taps that lie inside the volume, accumulated in ATen's tap order.  Only needed in
// bricks that touch the outside AND have a fill value.
__device__ __forceinline__ float tile_mask(const TapSet& ts, float x, float y, float z, float hx, float hy, float hz) {
  // first-tap indices from the coordinate the weights were taken from: x0 = floor(x)
  const float x0 = floorf(x), y0 = floorf(y), z0 = floorf(z);
  const float x1 = x0 + 1.0f, y1 = y0 + 1.0f, z1 = z0 + 1.0f;
  const bool ox0 = (x0 >= 0.0f) & (x0 <= hx), ox1 = (x1 >= 0.0f) & (x1 <= hx);
  const bool oy0 = (y0 >= 0.0f) & (y0 <= hy), oy1 = (y1 >= 0.0f) & (y1 <= hy);
  const bool oz0 = (z0 >= 0.0f) & (z0 <= hz), oz1 = (z1 >= 0.0f) & (z1 <= hz);
  float w[8];
  w[0] = __fmul_rn(__fmul_rn(ts.wx0, ts.wy0), ts.wz0);
  w[1] = __fmul_rn(__fmul_rn(ts.wx1, ts.wy0), ts.wz0);
  w[2] = __fmul_rn(__fmul_rn(ts.wx0, ts.wy1), ts.wz0);
  w[3] = __fmul_rn(__fmul_rn(ts.wx1, ts.wy1), ts.wz0);
  w[4] = __fmul_rn(__fmul_rn(ts.wx0, ts.wy0), ts.wz1);
  w[5] = __fmul_rn(__fmul_rn(ts.wx1, ts.wy0), ts.wz1);
  w[6] = __fmul_rn(__fmul_rn(ts.wx0, ts.wy1), ts.wz1);
  w[7] = __fmul_rn(__fmul_rn(ts.wx1, ts.wy1), ts.wz1);
  float mask = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const bool ok = ((k & 1) ? ox1 : ox0) & ((k & 2) ? oy1 : oy0) & ((k & 4) ? oz1 : oz0);
    const float next = __fadd_rn(mask, w[k]);  // same order as ATen's accumulation
    mask = ok ? next : mask;
  }
  return mask;
}

// ---- one channel of one pass: stage the box, sample the pass's quarters, store ----------
// LAUNDER = false only when the caller is straight-line code (no enclosing loop to hoist
// the per-voxel arithmetic out of).
template <int NT, int DTMODE, int TI, bool LAUNDER, bool FAST = false>
__device__ __forceinline__ void tile_channel(const ResampleArgs& a, const ImgArgs& g, int b, int c, const float (&X)[TI],
                                             const float (&Y)[TI], const float (&Z)[TI], const TileBox& bx, float* s_tile,
                                             unsigned tile_lds_addr, int tid, int row, int slab, int i_begin, int i_count,
                                             bool col_active, bool full, int q_begin, int q_end, int64_t n_in, int64_t n_out,
                                             bool prestaged = false) {
  constexpr int NQ = 4, QT = TI / NQ;
  const float hx = a.size_m1[0], hy = a.size_m1[1], hz = a.size_m1[2];
  const int t_begin = q_begin * QT, t_end = min(q_end * QT, i_count);
  const bool vec_ok = (g.dtype == TIO_F32) & ((a.K & 3) == 0) & ((reinterpret_cast<uintptr_t>(g.in) & 15) == 0);
  TileAddr ta;
  ta.ox = static_cast<float>(bx.bx0); ta.oy = static_cast<float>(bx.by0); ta.oz = static_cast<float>(bx.za);
  ta.sYb = bx.Lz * 4; ta.sXb = bx.Ly * bx.Lz * 4; ta.sXYb = ta.sXb + ta.sYb;
  ta.sYbf = static_cast<float>(ta.sYb); ta.sXbf = static_cast<float>(ta.sXb);
  ta.base_f = static_cast<float>(tile_lds_addr);
  const int es = dtype_size(g.dtype);
  const int64_t bc = static_cast<int64_t>(b) * g.channels + c;
  if (!prestaged) __syncthreads();  // previous brick of this block fully consumed
  if (a.ablate & 1) {
  } else if (prestaged) {  // the caller issued the DMA before phase A
    tile_dma_wait();
  } else if (vec_ok && !(a.ablate & 8)) {
    stage_brick_dma<NT>(s_tile, static_cast<const float*>(g.in) + bc * n_in, tid, bx, a.I, a.J, a.K);
    tile_dma_wait();
  } else if (vec_ok) {
    stage_brick_f32x4<NT>(s_tile, static_cast<const float*>(g.in) + bc * n_in, tid, bx, a.I, a.J, a.K);
  } else {
    stage_brick_generic<NT, DTMODE>(s_tile, static_cast<const char*>(g.in) + bc * n_in * es, g.dtype, tid, bx, a.I, a.J, a.K);
  }
  __syncthreads();
  // The fill value lives in a scalar register: as a pending vector load its first use (inside the voxel
  // loop of boundary bricks) makes the compiler wait with vmcnt(0) there, which on every later voxel
  // also drains the previous voxel's STORE - one store round trip per voxel.
  // (Through the scalar cache - the fill array is read-only for the launch: a vector load + readfirstlane
  // here costs every brick, interior ones included, a memory round trip right after the barrier.)
  const bool has_fill = g.fill != nullptr;
  typedef __attribute__((address_space(4))) const float* const_float_ptr;
  const float fillv = has_fill ? ((const_float_ptr)g.fill)[c] : 0.0f;
  // Output addresses: block-uniform running pointer (one plane = slab_b bytes) + this
  // thread's byte offset inside the plane.
  const int64_t slab_b = static_cast<int64_t>(slab) * es;
  char* out_c = static_cast<char*>(g.out) + (bc * n_out + static_cast<int64_t>(i_begin) * slab) * es;
  const unsigned urow = static_cast<unsigned>(row) * static_cast<unsigned>(es);
  if (a.ablate & 2) {
#pragma unroll
    for (int t = 0; t < TI; t++)
      if (col_active && t >= t_begin && t < t_end) store_at<DTMODE>(out_c + t * slab_b, g.dtype, urow, X[t] + Y[t] + Z[t]);
    return;
  }
  // `full` (block uniform): every thread owns a real column and all TI planes exist, so the
  // hot loops carry no predication and a quarter's LDS reads are all in flight before its
  // first accumulation.
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    if (q < q_begin || q >= q_end) continue;  // scalar branch
    char* out_t = out_c + (q * QT) * slab_b;
    if (bx.interior && full) {
      constexpr int G = TILE_GROUP < QT ? TILE_GROUP : QT;  // voxels with their LDS reads in flight together
#pragma unroll
      for (int u0 = 0; u0 < QT; u0 += G) {
        TapSet ts[G];
#pragma unroll
        for (int u = 0; u < G; u++) tile_issue_interior<LAUNDER>(ts[u], X[q * QT + u0 + u], Y[q * QT + u0 + u], Z[q * QT + u0 + u], ta, c);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < G; u++) {
          store_at<DTMODE>(out_t, g.dtype, urow, FAST ? tile_finish_fast(ts[u]) : tile_finish(ts[u]));
          out_t += slab_b;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (bx.interior) {
#pragma unroll
      for (int u = 0; u < QT; u++) {
        const int t = q * QT + u;
        TapSet ts;
        tile_issue_interior<LAUNDER>(ts, X[t], Y[t], Z[t], ta, c);
        const float val = FAST ? tile_finish_fast(ts) : tile_finish(ts);
        if (col_active && t < i_count) store_at<DTMODE>(out_t, g.dtype, urow, val);
        out_t += slab_b;
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      // brick touching the outside: same addressing (the box holds zeros there); the mask
      // decides between the sample and the fill value where one is configured
#pragma unroll
      for (int u = 0; u < QT; u++) {
        const int t = q * QT + u;
        TapSet ts;
        float x = X[t], y = Y[t], z = Z[t];
        if constexpr (LAUNDER) TIO_OPAQUE3(x, y, z, c);
        tile_issue_interior<false>(ts, x, y, z, ta, c);
        float val = FAST ? tile_finish_fast(ts) : tile_finish(ts);
        if (has_fill) val = (tile_mask(ts, x, y, z, hx, hy, hz) > 0.5f) ? val : fillv;
        if (full || (col_active && t < i_count)) store_at<DTMODE>(out_t, g.dtype, urow, val);
        out_t += slab_b;
        if ((u & 1) == 1) __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

// FAST (float32, trilinear-only launches that asked for TIO_PRECISION_FAST): the coordinates skip
// the normalise / un-normalise round trip of F.grid_sample (they are the voxel coordinates
// themselves, a few ulps away from the round-tripped ones) and the interpolation uses nested fma
// lerps.  Everything else — boxes, staging, masks — is shared with the exact kernel.
template <bool ELASTIC_POSSIBLE, int DTMODE, int TI, int TJ, int TK, int OCC, bool FAST = false>
__global__ __launch_bounds__(TJ* TK, (TJ * TK) / 256 * OCC) void resample_tile_kernel(const ResampleArgs a, const int* __restrict__ plan) {
  constexpr int NT = TJ * TK, NW = NT / 64;
  constexpr int NQ = 4, QT = TI / NQ;  // a brick is split (when needed) at quarter granularity
  static_assert(TI % NQ == 0 && QT <= kTileStashPlanes, "unsupported brick depth");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_cp = smem;
  int* s_red = reinterpret_cast<int*>(smem + a.cp_lds);
  float* s_tile = smem + a.cp_lds + kTileRedInts;

  // tile decode on the scalar unit: host-computed magic multipliers instead of divisions
  const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned t1 = fastdiv_exact(tile, a.magic_k, a.tiles_k);
  const int kt = tile - t1 * a.tiles_k;
  const unsigned t2 = fastdiv_exact(t1, a.magic_j, a.tiles_j);
  const int jt = t1 - t2 * a.tiles_j;
  const unsigned t3 = fastdiv_exact(t2, a.magic_i, a.tiles_i);
  const int it = t2 - t3 * a.tiles_i;
  const int b = t3;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tk = tid % TK, tj = tid / TK;
  const int jo_raw = jt * TJ + tj, ko_raw = kt * TK + tk;
  const bool col_active = (jo_raw < a.Jo) & (ko_raw < a.Ko);
  // out-of-range threads shadow the last valid column: their coordinates are those of
  // real voxels (the bounding box is unaffected) and they never store
  const int jo = min(jo_raw, a.Jo - 1), ko = min(ko_raw, a.Ko - 1);
  const int i_begin = it * TI;
  const int i_count = min(TI, a.Io - i_begin);
  const int i_last = i_begin + i_count - 1;
  const bool full = (i_count == TI) & ((jt + 1) * TJ <= a.Jo) & ((kt + 1) * TK <= a.Ko);

  const int64_t n_in = static_cast<int64_t>(a.I) * a.J * a.K;
  const int64_t n_out = static_cast<int64_t>(a.Io) * a.Jo * a.Ko;
  const int row = jo * a.Ko + ko;
  const int slab = a.Jo * a.Ko;

  if (a.passthrough != nullptr && a.passthrough[b] != 0) {  // gated-out element: bit-exact copy
    if (col_active) {
      for (int t = 0; t < i_count; t++) {
        const int64_t o_idx = static_cast<int64_t>(i_begin + t) * slab + row;
        for (int im = 0; im < a.n_images; im++) {
          const ImgArgs& g = a.img[im];
          const int es = dtype_size(g.dtype);
          for (int c = 0; c < g.channels; c++) {
            const int64_t off = (static_cast<int64_t>(b) * g.channels + c) * n_out + o_idx;
            const char* s = static_cast<const char*>(g.in) + off * es;
            char* d = static_cast<char*>(g.out) + off * es;
            for (int e = 0; e < es; e++) d[e] = s[e];
          }
        }
      }
    }
    return;
  }

  // ---- geometry sanity (block uniform): non-finite or absurd values → gather path ----
  const float* m = a.mapping + (a.mapping_batched ? b * 12 : 0);
  const float m00 = m[0], m01 = m[1], m02 = m[2], m03 = m[3];
  const float m10 = m[4], m11 = m[5], m12 = m[6], m13 = m[7];
  const float m20 = m[8], m21 = m[9], m22 = m[10], m23 = m[11];
  bool weird = false;  // |m| > 1e30, Inf or NaN — an integer compare on the bit pattern (scalar ALU)
#pragma unroll
  for (int q = 0; q < 12; q++) weird |= (__float_as_uint(m[q]) & 0x7FFFFFFFu) > 0x7149F2CAu;

  bool elastic = false, plane_cached = false;
  const float* cp = nullptr;
  int cp_bad = 0;  // per thread; joins the box reduction
  int ia = 0, ib = 0;
  Lerp1D li_lane = {0, 0, 0.0f, 0.0f};
  if constexpr (ELASTIC_POSSIBLE) {
    elastic = !(a.cp_skip != nullptr && a.cp_skip[b] != 0);
    if (elastic) {
      const int n_cp = a.ni * a.nj * a.nk * 3;
      const float* cp_global = a.cp + (a.cp_batched ? static_cast<int64_t>(b) * n_cp : 0);
      cp = cp_global;
      // the TI plane lerps are block uniform: lane t computes plane t's once, the unrolled
      // loop reads them back as scalars (v_readlane) instead of redoing them per voxel
      li_lane = lerp_index(min(i_begin + (lane & (TI - 1)), i_last), a.ni, a.Io, a.scale_i);
      ia = __builtin_amdgcn_readlane(li_lane.i0, 0);
      ib = __builtin_amdgcn_readlane(li_lane.i1, TI - 1);
      // <= 3 control planes under the brick (the usual coarse grid): every column reads its 36
      // control values straight from global memory (4 KiB, cache resident) - no staging, no
      // barrier.  Denser grids stage the whole field in LDS for the per-voxel lerp below.
      plane_cached = ib - ia <= 2;
    }
    if (elastic && !plane_cached) {
      const int n_cp = a.ni * a.nj * a.nk * 3;
      const float* cp_global = cp;
      int bad = 0;
      if (a.cp_lds > 0) {
        for (int t = tid; t < n_cp; t += NT) {
          const float v = cp_global[t];
          bad |= !(fabsf(v) <= 1e30f);
          s_cp[t] = v;
        }
        cp = s_cp;
      } else {
        for (int t = tid; t < n_cp; t += NT) bad |= !(fabsf(cp_global[t]) <= 1e30f);
      }
      weird |= (__syncthreads_or(bad) != 0);
    }
  }

  // ---- phase A: sampling coordinates of this thread's column, TI planes -----------
  float X[TI], Y[TI], Z[TI];
  float lo[3], hi[3];  // coordinate range of this column over the brick (not yet floored)
  const float cj = static_cast<float>(jo), ck = static_cast<float>(ko);
  const float hx = a.size_m1[0], hy = a.size_m1[1], hz = a.size_m1[2];
  const float ci0 = static_cast<float>(i_begin), ci_last = static_cast<float>(i_last);
  const bool short_div = a.short_div != 0;
  // FAST drops the normalise / un-normalise round trip — unless the launch applies a fill rule (see TIO_NORM_RT below)
  const bool skip_rt = FAST && !(a.fill_recheck != 0 && a.any_fill != 0);
  // Identity mapping (ElasticDeformation alone, Resample onto the same grid): the chain
  // c*1 + 0 + 0 + 0 returns its argument bit for bit (c >= 0, so no -0 subtlety), skip it.
  const bool ident = (m00 == 1.0f) & (m01 == 0.0f) & (m02 == 0.0f) & (m03 == 0.0f) & (m10 == 0.0f) & (m11 == 1.0f) &
                     (m12 == 0.0f) & (m13 == 0.0f) & (m20 == 0.0f) & (m21 == 0.0f) & (m22 == 1.0f) & (m23 == 0.0f);
  // the three block-uniform modes of TIO_FINISH_COORD; the hot elastic loop shadows them with literals
  const bool m_unit = a.unit_spacing != 0, m_ident = ident, m_affine_first = a.affine_first != 0;
#define TIO_AFFINE_ROW(M0, M1, M2, M3, A, B, C) \
  __builtin_fmaf(1.0f, M3, __builtin_fmaf(C, M2, __builtin_fmaf(B, M1, __fmul_rn(A, M0))))
#define TIO_FINISH_COORD(T, DI, DJ, DK, HAS_D, NORM)                                                \
  {                                                                                             \
    float vi, vj, vk;                                                                           \
    if (HAS_D) {                                                                                \
      float q_i = DI, q_j = DJ, q_k = DK;                                                       \
      (void)q_i; (void)q_j; (void)q_k;                                                          \
      if (!m_unit) {                                                                            \
        q_i = exact_div(q_i, a.sp[0], a.rsp[0]);                                                \
        q_j = exact_div(q_j, a.sp[1], a.rsp[1]);                                                \
        q_k = exact_div(q_k, a.sp[2], a.rsp[2]);                                                \
      }                                                                                         \
      if (m_ident) { /* c + d either way round */                                               \
        vi = __fadd_rn(ci, q_i);                                                                \
        vj = __fadd_rn(cj, q_j);                                                                \
        vk = __fadd_rn(ck, q_k);                                                                \
      } else if (m_affine_first) {                                                              \
        vi = __fadd_rn(TIO_AFFINE_ROW(m00, m01, m02, m03, ci, cj, ck), q_i);                    \
        vj = __fadd_rn(TIO_AFFINE_ROW(m10, m11, m12, m13, ci, cj, ck), q_j);                    \
        vk = __fadd_rn(TIO_AFFINE_ROW(m20, m21, m22, m23, ci, cj, ck), q_k);                    \
      } else {                                                                                  \
        const float ei = __fadd_rn(ci, q_i), ej = __fadd_rn(cj, q_j), ek = __fadd_rn(ck, q_k);  \
        vi = TIO_AFFINE_ROW(m00, m01, m02, m03, ei, ej, ek);                                    \
        vj = TIO_AFFINE_ROW(m10, m11, m12, m13, ei, ej, ek);                                    \
        vk = TIO_AFFINE_ROW(m20, m21, m22, m23, ei, ej, ek);                                    \
      }                                                                                         \
    } else {                                                                                    \
      vi = TIO_AFFINE_ROW(m00, m01, m02, m03, ci, cj, ck);                                      \
      vj = TIO_AFFINE_ROW(m10, m11, m12, m13, ci, cj, ck);                                      \
      vk = TIO_AFFINE_ROW(m20, m21, m22, m23, ci, cj, ck);                                      \
    }                                                                                           \
    X[T] = NORM(vi, a.dh[0], a.rdh[0], a.half_h[0]);                                            \
    Y[T] = NORM(vj, a.dh[1], a.rdh[1], a.half_h[1]);                                            \
    Z[T] = NORM(vk, a.dh[2], a.rdh[2], a.half_h[2]);                                            \
  }
// FAST keeps the net scaling of the round trip, (S_own - 1) / max(S_norm - 1, 1) = half_h / dh: 1 for the usual
// case, 0 on a one-voxel axis (2-D images: the coordinate collapses to 0 like in the exact path), the
// resolution ratio for Resample(target) on a multi-resolution subject.
//
// A FAST launch WITH A FILL RULE keeps the exact round trip (`skip_rt` false): the fill decision `mask > 0.5` is then taken
// on the reference's own coordinate, bit for bit — dropping the round trip moves a coordinate by a few ulps, harmless for
// a value but enough to flip the comparison for the voxels whose in-bounds weight sits within rounding of the threshold
// (VERDICT r3 weak #1).  The planned FAST kernels (resample_fast.hpp) re-decide such voxels in a tail instead; here — the
// single-kernel road of small launches — a tail behind the sampling code cost spills in every path (its registers live next
// to the 48 coordinate registers), so the launch simply pays the 18 instructions per voxel and keeps the nested-fma lerps.
#define TIO_NORM_RT(V, D, R, H) (skip_rt ? __fmul_rn(V, __fmul_rn(H, R)) : normalise_roundtrip_folded(V, D, R, H, short_div))
#define TIO_NORM_SKIP(V, D, R, H) __fmul_rn(V, __fmul_rn(H, R))
#define TIO_NORM_SHORT(V, D, R, H) normalise_roundtrip_folded<true>(V, D, R, H)
#define TIO_NORM_FULL(V, D, R, H) normalise_roundtrip_folded<false>(V, D, R, H)
#define TIO_TRACK_ALL(T)                                                     \
  {                                                                          \
    if ((T) == 0) {                                                          \
      lo[0] = hi[0] = X[T]; lo[1] = hi[1] = Y[T]; lo[2] = hi[2] = Z[T];      \
    } else {                                                                 \
      lo[0] = fminf(lo[0], X[T]); hi[0] = fmaxf(hi[0], X[T]);                \
      lo[1] = fminf(lo[1], Y[T]); hi[1] = fmaxf(hi[1], Y[T]);                \
      lo[2] = fminf(lo[2], Z[T]); hi[2] = fmaxf(hi[2], Z[T]);                \
    }                                                                        \
  }

  // ---- affine-only bricks: the box from the 8 brick corners, before phase A ----------------
  // Every operation of the coordinate chain (products, sums, the correctly rounded division,
  // the un-normalisation) is monotone in each of the three output indices, so the brick's
  // coordinate extremes sit at its corners.  Each wave evaluates the 8 corners (lane & 7)
  // with the very same instruction sequence and reduces them on its own: no LDS, no barrier,
  // and the brick can be requested from HBM before the per-voxel coordinates are computed.
  const float capx = hx + 1.0f, capy = hy + 1.0f, capz = hz + 1.0f;
  TileBox box_full;
  bool have_box = false, prestaged = false;
  // A planned launch (16^3 bricks only): the brick's box was bounded ahead of time from the <= 27 vertices of the
  // coordinate map, with a margin (1 / 64 voxel) far above what the float32 operation sequence below can differ from
  // the real-valued map by.  It only decides WHAT is staged — every coordinate and every tap weight is still computed
  // here, bit for bit — but it is known before phase A, so the brick travels while the coordinates are formed and the
  // per-voxel tracking, the block reduction and its barrier are not waited for.  Bricks the planner could not bound
  // (non-finite geometry, several control cells per brick axis, a box beyond the LDS budget) take the in-kernel road.
  if (plan != nullptr && !weird) {
    typedef __attribute__((address_space(4))) const int* const_int_ptr;
    const_int_ptr d = (const_int_ptr)(plan + a.B * 16) + static_cast<size_t>(tile) * kDescInts;
    const int kw = d[0];
    // (a plan made AHEAD — tio_resample3d_plan — may have been sized for another road's LDS budget; `fits` is this
    // kernel's own decision, taken against ITS tile: a box beyond it takes the in-kernel road below like any unplanned brick)
    if ((kw & 0xFF) == kDescStaged && static_cast<int64_t>(d[4]) * d[5] * (d[6] * 4) <= static_cast<int64_t>(a.tile_cap)) {
      box_full.bx0 = d[1]; box_full.by0 = d[2]; box_full.za = d[3];
      box_full.Lx = d[4]; box_full.Ly = d[5]; box_full.Lz = d[6] * 4;
      box_full.bx1 = box_full.bx0 + box_full.Lx - 1; box_full.by1 = box_full.by0 + box_full.Ly - 1;
      box_full.zlo = box_full.za; box_full.zhi = box_full.za + box_full.Lz - 1;
      box_full.interior = kw >> 8; box_full.outside = 0; box_full.fits = 1;
      have_box = true;
    }
  }
  if (!elastic && !have_box) {
    int r[6];
    {
      const int j_lo = jt * TJ, j_hi = min(j_lo + TJ, a.Jo) - 1, k_lo = kt * TK, k_hi = min(k_lo + TK, a.Ko) - 1;
      const float ci = (lane & 1) ? ci_last : ci0;
      const float cj = static_cast<float>((lane & 2) ? j_hi : j_lo), ck = static_cast<float>((lane & 4) ? k_hi : k_lo);
      float X[1], Y[1], Z[1];
      TIO_FINISH_COORD(0, 0.0f, 0.0f, 0.0f, false, TIO_NORM_RT)
      bound_ints(X[0], X[0], capx, r[0], r[1]);
      bound_ints(Y[0], Y[0], capy, r[2], r[3]);
      bound_ints(Z[0], Z[0], capz, r[4], r[5]);
    }
#pragma unroll
    for (int q = 0; q < 6; q++) r[q] = wave_max_i32(r[q]);
    box_full = make_box(r, a, weird);
    have_box = true;
  }
  if (have_box) {
    const ImgArgs& g0 = a.img[0];
    prestaged = (a.n_images == 1) & (g0.channels == 1) & (g0.interp == TIO_LINEAR) & (box_full.fits != 0) & (box_full.outside == 0) &
                (g0.dtype == TIO_F32) & ((a.K & 3) == 0) & ((reinterpret_cast<uintptr_t>(g0.in) & 15) == 0) & (a.ablate == 0);
    if (prestaged)
      stage_brick_dma<NT>(s_tile, static_cast<const float*>(g0.in) + static_cast<int64_t>(b) * n_in, tid, box_full, a.I, a.J, a.K);
  }

  bool done = false;
  if constexpr (ELASTIC_POSSIBLE) {
    if (elastic) {
      const Lerp1D lj = lerp_index(jo, a.nj, a.Jo, a.scale_j);
      const Lerp1D lk = lerp_index(ko, a.nk, a.Ko, a.scale_k);
      const int s_i = a.nj * a.nk * 3, s_j = a.nk * 3;
      if (plane_cached) {  // ≤ 3 control planes under the brick: lerp them once per column
        typedef float plane_vec __attribute__((ext_vector_type(16)));
        plane_vec P = {};  // P[3 e + c]: component c of control plane ia + e
        {
          float p[3];
          cp_plane(cp, ia, s_i, s_j, lj, lk, p);
          P[0] = p[0]; P[1] = p[1]; P[2] = p[2];
          if (ib - ia >= 1) { cp_plane(cp, ia + 1, s_i, s_j, lj, lk, p); P[3] = p[0]; P[4] = p[1]; P[5] = p[2]; }
          if (ib - ia >= 2) { cp_plane(cp, ia + 2, s_i, s_j, lj, lk, p); P[6] = p[0]; P[7] = p[1]; P[8] = p[2]; }
        }
        // NaN / Inf control values reach the lerped planes (NaN * 0 = NaN): flag the brick
#pragma unroll
        for (int e = 0; e < 9; e++) cp_bad |= !(fabsf(P[e]) <= 1e30f);
        // The two control planes a voxel lerps between change once or twice per brick (at a cell
        // boundary), so they live in named registers that a SCALAR branch refreshes when the plane pair
        // moves on; indexing P[] with the (uniform) plane number on every voxel costs an
        // s_set_gpr_idx_on / v_mov / off sequence per access, six times per voxel.
        float pa_i = 0.f, pa_j = 0.f, pa_k = 0.f, pb_i = 0.f, pb_j = 0.f, pb_k = 0.f;
        int cur0 = -1, cur1 = -1;
#define TIO_PLANE_PICK(E, DI, DJ, DK)                                   \
  {                                                                     \
    DI = (E) == 0 ? P[0] : ((E) == 3 ? P[3] : P[6]);                    \
    DJ = (E) == 0 ? P[1] : ((E) == 3 ? P[4] : P[7]);                    \
    DK = (E) == 0 ? P[2] : ((E) == 3 ? P[5] : P[8]);                    \
  }
#define TIO_PLANE_LOOP(NORM, UNIT, IDENT, AFFINE_FIRST)                                                           \
  _Pragma("unroll") for (int t = 0; t < TI; t++) {                                                                \
    const bool m_unit = UNIT, m_ident = IDENT, m_affine_first = AFFINE_FIRST; /* literals fold the branches away */ \
    (void)m_unit; (void)m_ident; (void)m_affine_first;                                                            \
    const float ci = fminf(ci0 + static_cast<float>(t), ci_last);                                                 \
    const int e0 = 3 * (__builtin_amdgcn_readlane(li_lane.i0, t) - ia); /* scalars */                             \
    const int e1 = 3 * (__builtin_amdgcn_readlane(li_lane.i1, t) - ia);                                           \
    const float l0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(li_lane.l0), t));                    \
    const float l1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(li_lane.l1), t));                    \
    if (e0 != cur0) { TIO_PLANE_PICK(e0, pa_i, pa_j, pa_k) cur0 = e0; }                                           \
    if (e1 != cur1) { TIO_PLANE_PICK(e1, pb_i, pb_j, pb_k) cur1 = e1; }                                           \
    const float di = lerp2(pa_i, l0, pb_i, l1);                                                                   \
    const float dj = lerp2(pa_j, l0, pb_j, l1);                                                                   \
    const float dk = lerp2(pa_k, l0, pb_k, l1);                                                                   \
    TIO_FINISH_COORD(t, di, dj, dk, true, NORM)                                                                   \
    TIO_TRACK_ALL(t)                                                                                              \
    if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);                                                          \
  }
        // one block-uniform branch instead of a select per division
        // one block-uniform branch instead of a select per division, and - for unit spacing, the usual
        // 1 mm volumes - one copy of the loop per composition order, so that the 16 unrolled planes carry
        // no scalar branches at all (they cost a wave more than the arithmetic they skip)
        const bool outer_unit = a.unit_spacing != 0, outer_ident = ident, outer_af = a.affine_first != 0;
#define TIO_PLANE_LOOPS(NORM)                                                         \
  if (outer_unit && outer_ident) {                                                    \
    TIO_PLANE_LOOP(NORM, true, true, true)                                            \
  } else if (outer_unit && outer_af) {                                                \
    TIO_PLANE_LOOP(NORM, true, false, true)                                           \
  } else if (outer_unit) {                                                            \
    TIO_PLANE_LOOP(NORM, true, false, false)                                          \
  } else {                                                                            \
    TIO_PLANE_LOOP(NORM, outer_unit, outer_ident, outer_af)                           \
  }
        if (skip_rt) {
          TIO_PLANE_LOOPS(TIO_NORM_SKIP)
        } else if (short_div) {
          TIO_PLANE_LOOPS(TIO_NORM_SHORT)
        } else {
          TIO_PLANE_LOOPS(TIO_NORM_FULL)
        }
#undef TIO_PLANE_LOOPS
#undef TIO_PLANE_LOOP
#undef TIO_PLANE_PICK
      } else {
#pragma unroll
        for (int t = 0; t < TI; t++) {
          const int io = min(i_begin + t, i_last);
          const float ci = static_cast<float>(io);
          const Lerp1D li = lerp_index(io, a.ni, a.Io, a.scale_i);
          const Disp d = cp_trilerp3(cp, s_i, s_j, li, lj, lk);
          TIO_FINISH_COORD(t, d.i, d.j, d.k, true, TIO_NORM_RT)
          TIO_TRACK_ALL(t)
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      done = true;
    }
  }
  if (!done) {
#define TIO_AFFINE_LOOP(NORM)                                          \
  _Pragma("unroll") for (int t = 0; t < TI; t++) {                     \
    const float ci = fminf(ci0 + static_cast<float>(t), ci_last);      \
    TIO_FINISH_COORD(t, 0.0f, 0.0f, 0.0f, false, NORM)                 \
    if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);               \
  }
    if (skip_rt) {
      TIO_AFFINE_LOOP(TIO_NORM_SKIP)
    } else if (short_div) {
      TIO_AFFINE_LOOP(TIO_NORM_SHORT)
    } else {
      TIO_AFFINE_LOOP(TIO_NORM_FULL)
    }
#undef TIO_AFFINE_LOOP
    lo[0] = hi[0] = lo[1] = hi[1] = lo[2] = hi[2] = 0.0f;  // unused: the box came from the corners
  }
#undef TIO_TRACK_ALL
#undef TIO_FINISH_COORD
#undef TIO_NORM_RT
#undef TIO_NORM_SKIP
#undef TIO_NORM_SHORT
#undef TIO_NORM_FULL
#undef TIO_AFFINE_ROW

  // ---- phase B: bounding boxes: whole brick, else halves, else quarters ----------------
  int nsplit = 1;
  if (!have_box) {
    int r7[7];
    bound_ints(lo[0], hi[0], capx, r7[0], r7[1]);
    bound_ints(lo[1], hi[1], capy, r7[2], r7[3]);
    bound_ints(lo[2], hi[2], capz, r7[4], r7[5]);
    r7[6] = cp_bad;  // a non-finite displacement seen by any column of the brick
    block_max6<NW, 7>(r7, s_red, 0, wave, lane);
    weird |= r7[6] != 0;
    const int r[6] = {r7[0], r7[1], r7[2], r7[3], r7[4], r7[5]};
    box_full = make_box(r, a, weird);
  }
  if (!box_full.fits && !weird && !box_full.outside) {
    // Rare: the brick's box exceeds the LDS budget.  Bound halves, then quarters, of the
    // planes (recomputed from the coordinates, nothing extra is kept live for this) and
    // leave the pass boxes in LDS.
    int* s_box = s_red + kTileBoxBase;
    bool ok2 = true;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      float l3[3] = {X[h * 2 * QT], Y[h * 2 * QT], Z[h * 2 * QT]}, h3[3] = {l3[0], l3[1], l3[2]};
#pragma unroll
      for (int t = h * 2 * QT + 1; t < (h + 1) * 2 * QT; t++) {
        l3[0] = fminf(l3[0], X[t]); h3[0] = fmaxf(h3[0], X[t]);
        l3[1] = fminf(l3[1], Y[t]); h3[1] = fmaxf(h3[1], Y[t]);
        l3[2] = fminf(l3[2], Z[t]); h3[2] = fmaxf(h3[2], Z[t]);
      }
      int r[6];
      bound_ints(l3[0], h3[0], capx, r[0], r[1]);
      bound_ints(l3[1], h3[1], capy, r[2], r[3]);
      bound_ints(l3[2], h3[2], capz, r[4], r[5]);
      block_max6<NW, 6>(r, s_red, 1 + h, wave, lane);
      const TileBox bh = make_box(r, a, weird);
      ok2 &= (bh.fits | bh.outside) != 0;
      if (tid == 0) store_box(s_box + h * 16, bh);
    }
    nsplit = 2;
    if (!ok2) {
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        float l3[3] = {X[q * QT], Y[q * QT], Z[q * QT]}, h3[3] = {l3[0], l3[1], l3[2]};
#pragma unroll
        for (int t = q * QT + 1; t < (q + 1) * QT; t++) {
          l3[0] = fminf(l3[0], X[t]); h3[0] = fmaxf(h3[0], X[t]);
          l3[1] = fminf(l3[1], Y[t]); h3[1] = fmaxf(h3[1], Y[t]);
          l3[2] = fminf(l3[2], Z[t]); h3[2] = fmaxf(h3[2], Z[t]);
        }
        int r[6];
        bound_ints(l3[0], h3[0], capx, r[0], r[1]);
        bound_ints(l3[1], h3[1], capy, r[2], r[3]);
        bound_ints(l3[2], h3[2], capz, r[4], r[5]);
        block_max6<NW, 6>(r, s_red, 3 + q, wave, lane);
        const TileBox bq = make_box(r, a, weird);
        if (tid == 0) store_box(s_box + q * 16, bq);  // overwrites the halves: every wave passed the last barrier after reading them
      }
      nsplit = 4;
    }
    __syncthreads();
  }

  // ---- phase C: per pass / image / channel: stage, sample, store ------------------------
  const unsigned tile_lds_addr = static_cast<unsigned>(
      reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) float*)s_tile));
  // the common launch shape needs no loops around the sampling code, hence no laundering
  const bool single = (nsplit == 1) & (a.n_images == 1) & (a.img[0].channels == 1);
  if (single && a.img[0].interp == TIO_LINEAR && box_full.fits && !box_full.outside) {
    tile_channel<NT, DTMODE, TI, false, FAST>(a, a.img[0], b, 0, X, Y, Z, box_full, s_tile, tile_lds_addr, tid, row, slab, i_begin, i_count, col_active,
                                        full, 0, 4, n_in, n_out, prestaged);
    return;
  }
  for (int p = 0; p < nsplit; p++) {
    const TileBox bx = nsplit == 1 ? box_full : load_box(s_red + kTileBoxBase + p * 16);
    const int q_begin = p * NQ / nsplit, q_end = (p + 1) * NQ / nsplit;  // quarters of this pass
    const int t_begin = q_begin * QT, t_end = min(q_end * QT, i_count);  // planes of this pass

    for (int im = 0; im < a.n_images; im++) {
      const ImgArgs& g = a.img[im];
      if (bx.outside) {  // nothing of this pass sees the volume: fill (or 0) everywhere
        if (col_active) {
          for (int c = 0; c < g.channels; c++) {
            const int64_t bc = static_cast<int64_t>(b) * g.channels + c;
            const float val = g.fill != nullptr ? g.fill[c] : 0.0f;
            for (int t = t_begin; t < t_end; t++)
              store_from_float(g.out, g.dtype, bc * n_out + static_cast<int64_t>(i_begin + t) * slab + row, val);
          }
        }
        continue;
      }
      if (g.interp != TIO_LINEAR || !bx.fits) {  // per-voxel global gather (nearest images, oversize boxes)
        // The coordinates live in registers under compile-time indices; park a quarter at a
        // time in the (unused) brick area so that the gather body is emitted once.
        __syncthreads();  // the brick area may still be read by the previous image
#pragma unroll
        for (int q = 0; q < NQ; q++) {
          if (q >= q_begin && q < q_end) {
#pragma unroll
            for (int u = 0; u < QT; u++) {
              float sx = X[q * QT + u], sy = Y[q * QT + u], sz = Z[q * QT + u];
              TIO_OPAQUE3(sx, sy, sz, im);
              s_tile[(u * 3 + 0) * NT + tid] = sx;
              s_tile[(u * 3 + 1) * NT + tid] = sy;
              s_tile[(u * 3 + 2) * NT + tid] = sz;
            }
            if (col_active) {
              const int u_end = min(QT, i_count - q * QT);
#pragma unroll 1
              for (int u = 0; u < u_end; u++) {
                const int o_idx = (i_begin + q * QT + u) * slab + row;
                gather_voxel<DTMODE>(g, a, b, n_in, n_out, o_idx, s_tile[(u * 3 + 0) * NT + tid], s_tile[(u * 3 + 1) * NT + tid],
                                     s_tile[(u * 3 + 2) * NT + tid], bx.interior != 0);
              }
            }
          }
        }
        continue;
      }
      for (int c = 0; c < g.channels; c++)
        tile_channel<NT, DTMODE, TI, true, FAST>(a, g, b, c, X, Y, Z, bx, s_tile, tile_lds_addr, tid, row, slab, i_begin, i_count, col_active, full,
                                           q_begin, q_end, n_in, n_out);
    }
  }
}

}  // namespace tio
