// Nearest-neighbour images without a fill rule (label maps, masks): their own kernel, bit-identical to the exact
// coordinate chain at a fraction of its cost.  Included by resample.hip after ResampleArgs, the exact-chain helpers
// (exact_div, normalise_roundtrip, cp_trilerp3) and resample_fast.hpp.
//
// Reference: `_sample_batch_grid_sample(..., mode="nearest")` -> F.grid_sample(padding_mode="zeros", align_corners=True)
// (transforms/spatial/spatial.py:1695-1731): the value of the input voxel at nearbyint(coordinate) per axis, zero
// outside.  The result depends on the coordinate only through three roundings, so the float32 op sequence of the
// reference's chain (affine_grid-style matmul, field upsampling, normalise / un-normalise: ~115 vector instructions per
// voxel with control points) matters only where a coordinate lies within rounding error of a half-integer.  The kernel
//   1. evaluates the FAST coordinate line of the column (resample_fast.hpp: one fma per axis and plane),
//   2. takes the voxel as decided when every axis is at least eps = nn_eps * (S + |x|) away from the next
//      half-integer (the in-bounds test flips at -0.5 and S - 0.5: half-integers too) — the FAST and the exact
//      coordinate differ by far less (bound and measurement: DESIGN.md section 4.1b),
//   3. re-evaluates the few undecided voxels of the column with the exact chain (exact_voxel_coords below: the
//      operation sequence of resample_kernel, which tests/ pin against the oracle and the reference's golden vectors).
// One load per voxel straight from global memory (no LDS: a nearest image has no taps to share), element size 1 / 2 / 4 / 8
// bytes copied as bits, every image and channel of the launch that has that size.
#pragma once

namespace tio {

// The exact coordinate chain of ONE voxel — the sequence of resample_kernel (resample.hip), operation for operation.
template <bool ELASTIC_POSSIBLE>
__device__ __forceinline__ void exact_voxel_coords(const ResampleArgs& a, const float (&m)[12], bool elastic, const float* __restrict__ cp,
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

template <int ES> struct NearestBits;
template <> struct NearestBits<1> { typedef uint8_t type; };
template <> struct NearestBits<2> { typedef uint16_t type; };
template <> struct NearestBits<4> { typedef uint32_t type; };
template <> struct NearestBits<8> { typedef uint64_t type; };

// nearbyint per axis (round half to even: v_rndne_f32), zero padding: offset of the source voxel, or -1
__device__ __forceinline__ int nearest_offset(float x, float y, float z, float hx, float hy, float hz, int J, int K) {
  const float xn = rintf(x), yn = rintf(y), zn = rintf(z);
  const bool ok = (xn >= 0.0f) & (xn <= hx) & (yn >= 0.0f) & (yn <= hy) & (zn >= 0.0f) & (zn <= hz);  // NaN fails
  return ok ? (static_cast<int>(xn) * J + static_cast<int>(yn)) * K + static_cast<int>(zn) : -1;
}

// |frac(x) - 1/2| >= eps (S + |x|): the rounding of x, and its in-bounds test, cannot differ for a coordinate within
// that distance of x.  NaN and coordinates beyond float32's integer range come out undecided / trivially decided.
__device__ __forceinline__ bool nearest_decided(float x, float size, float eps) {
  const float d = fabsf((x - floorf(x)) - 0.5f);
  return d >= eps * (size + fabsf(x));
}

// One block per 16 x 16 x 16 brick of the output, one column of 16 planes per thread.
template <bool ELASTIC_POSSIBLE, int ES>
__global__ __launch_bounds__(256) void resample_nearest_kernel(const ResampleArgs a) {
  typedef typename NearestBits<ES>::type bits_t;
  constexpr int TI = 16, TJ = 16, TK = 16;
  const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned t1 = fastdiv(tile, a.magic_k, a.tiles_k);
  const int kt = tile - t1 * a.tiles_k;
  const unsigned t2 = fastdiv(t1, a.magic_j, a.tiles_j);
  const int jt = t1 - t2 * a.tiles_j;
  const unsigned t3 = fastdiv(t2, a.magic_i, a.tiles_i);
  const int it = t2 - t3 * a.tiles_i;
  const int b = t3;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int tk = tid & (TK - 1), tj = tid / TK;
  const int i_begin = it * TI, j_lo = jt * TJ, k_lo = kt * TK;
  const int i_count = min(TI, a.Io - i_begin), nv = min(TJ, a.Jo - j_lo), nw = min(TK, a.Ko - k_lo);
  const bool col_active = (tj < nv) & (tk < nw);
  const int jv = min(tj, nv - 1), kw = min(tk, nw - 1);  // idle threads shadow the last column (they never store)
  const int jo = j_lo + jv, ko = k_lo + kw;
  const int64_t n_in = static_cast<int64_t>(a.I) * a.J * a.K;
  const int64_t n_out = static_cast<int64_t>(a.Io) * a.Jo * a.Ko;
  const int slab = a.Jo * a.Ko;
  const int64_t col = static_cast<int64_t>(i_begin) * slab + static_cast<int64_t>(jo) * a.Ko + ko;  // first voxel of the column

  if (a.passthrough != nullptr && a.passthrough[b] != 0) {  // gated-out element: bit-exact copy (spatial.py:1101-1106)
    if (col_active) {
      for (int im = 0; im < a.n_images; im++) {
        const ImgArgs& g = a.img[im];
        if (dtype_size(g.dtype) != ES) continue;
        for (int c = 0; c < g.channels; c++) {
          const int64_t bc = static_cast<int64_t>(b) * g.channels + c;
          const bits_t* src = static_cast<const bits_t*>(g.in) + bc * n_out + col;
          bits_t* dst = static_cast<bits_t*>(g.out) + bc * n_out + col;
          for (int t = 0; t < i_count; t++) dst[static_cast<int64_t>(t) * slab] = src[static_cast<int64_t>(t) * slab];
        }
      }
    }
    return;
  }

  // the element's mapping: raw for the exact chain, rows scaled by the axis ratios for the FAST line
  float m[12];
  {
    const float* mp = a.mapping + (a.mapping_batched ? b * 12 : 0);
#pragma unroll
    for (int q = 0; q < 12; q++) m[q] = mp[q];
  }
  FastFrameG f;
  bool weird = false;
#pragma unroll
  for (int q = 0; q < 12; q++) {
    weird |= (__float_as_uint(m[q]) & 0x7FFFFFFFu) > 0x7149F2CAu;  // |m| > 1e30, Inf or NaN: every voxel goes the exact way
    f.m[q] = m[q] * (a.half_h[q >> 2] / a.dh[q >> 2]);
  }
  f.affine_first = a.affine_first != 0;
  f.ni = a.ni; f.nj = a.nj; f.nk = a.nk; f.sci = a.scale_i; f.scj = a.scale_j; f.sck = a.scale_k;
  f.elastic = false; f.cp = nullptr;
#pragma unroll
  for (int e = 0; e < 3; e++) f.dsc[e] = a.rsp[e] * (f.affine_first ? a.half_h[e] / a.dh[e] : 1.0f);
  if constexpr (ELASTIC_POSSIBLE) {
    f.elastic = !(a.cp_skip != nullptr && a.cp_skip[b] != 0);
    f.cp = f.elastic ? a.cp + (a.cp_batched ? static_cast<int64_t>(b) * (a.ni * a.nj * a.nk * 3) : 0) : nullptr;
  }
  pipe_brick_frame(f, j_lo, k_lo);
  float C3[3], col3[3];
  const float fv = static_cast<float>(jv), fw = static_cast<float>(kw);
#pragma unroll
  for (int r = 0; r < 3; r++) {
    C3[r] = static_cast<float>(static_cast<double>(f.m[4 * r]) * i_begin + f.c[r]);
    col3[r] = __builtin_fmaf(f.m[4 * r + 1], fv, f.m[4 * r + 2] * fw);
  }
  Lerp1D lj{0, 0, 1.0f, 0.0f}, lk{0, 0, 1.0f, 0.0f};
  if constexpr (ELASTIC_POSSIBLE) {
    if (f.elastic) {
      lj = lerp_index(jo, a.nj, a.Jo, a.scale_j);
      lk = lerp_index(ko, a.nk, a.Ko, a.scale_k);
    }
  }
  const float hx = a.size_m1[0], hy = a.size_m1[1], hz = a.size_m1[2];
  const float eps = weird ? 1.0f : a.nn_eps;  // (1: nothing is ever decided by the FAST line)

  // ---- 1 + 2: the FAST line, plane by plane; undecided planes are remembered ---------------------------------------
  int offs[TI];
  unsigned undecided = 0u;
  {
    ColumnPlanes planes;
    planes.cell = -2;
#pragma unroll
    for (int e = 0; e < 3; e++) { planes.P0[e] = 0.0f; planes.P1[e] = 0.0f; }
    float A3[3], B3[3];
    const int u1 = i_begin + i_count;
    int run0 = i_begin;
    int run1 = fast_column_line(f, lj, lk, planes, run0, u1, i_begin, C3, col3, lane, A3, B3);
#pragma unroll
    for (int t = 0; t < TI; t++) {
      offs[t] = -1;
      if (t < i_count) {  // (block uniform)
        if (i_begin + t >= run1) {  // next control cell (wave uniform by construction of the runs)
          run0 = run1;
          run1 = fast_column_line(f, lj, lk, planes, run0, u1, i_begin, C3, col3, lane, A3, B3);
        }
        const float s = static_cast<float>(i_begin + t - run0);
        const float x = __builtin_fmaf(s, B3[0], A3[0]), y = __builtin_fmaf(s, B3[1], A3[1]), z = __builtin_fmaf(s, B3[2], A3[2]);
        const bool decided = nearest_decided(x, hx + 1.0f, eps) & nearest_decided(y, hy + 1.0f, eps) & nearest_decided(z, hz + 1.0f, eps);
        offs[t] = nearest_offset(x, y, z, hx, hy, hz, a.J, a.K);
        undecided |= decided ? 0u : (1u << t);
      }
    }
  }
  if (!col_active) undecided = 0u;

  // ---- the decided voxels: every image and channel of this element size -------------------------------------------------
  for (int im = 0; im < a.n_images; im++) {
    const ImgArgs& g = a.img[im];
    if (dtype_size(g.dtype) != ES) continue;
    for (int c = 0; c < g.channels; c++) {
      const int64_t bc = static_cast<int64_t>(b) * g.channels + c;
      const bits_t* __restrict__ src = static_cast<const bits_t*>(g.in) + bc * n_in;
      bits_t* __restrict__ dst = static_cast<bits_t*>(g.out) + bc * n_out + col;
      bits_t v[TI];
#pragma unroll
      for (int t = 0; t < TI; t++) v[t] = (col_active && offs[t] >= 0) ? src[offs[t]] : static_cast<bits_t>(0);
#pragma unroll
      for (int t = 0; t < TI; t++)
        if (col_active && t < i_count && !((undecided >> t) & 1u)) dst[static_cast<int64_t>(t) * slab] = v[t];
    }
  }

  // ---- 3: the undecided ones through the exact chain ---------------------------------------------------------------------
  const float cj = static_cast<float>(jo), ck = static_cast<float>(ko);
  while (__builtin_amdgcn_ballot_w64(undecided != 0u) != 0ull) {
    if (undecided != 0u) {
      const int t = __builtin_ctz(undecided);
      undecided &= undecided - 1u;
      float x, y, z;
      exact_voxel_coords<ELASTIC_POSSIBLE>(a, m, f.elastic, f.cp, lj, lk, i_begin + t, cj, ck, x, y, z);
      const int off = nearest_offset(x, y, z, hx, hy, hz, a.J, a.K);
      for (int im = 0; im < a.n_images; im++) {
        const ImgArgs& g = a.img[im];
        if (dtype_size(g.dtype) != ES) continue;
        for (int c = 0; c < g.channels; c++) {
          const int64_t bc = static_cast<int64_t>(b) * g.channels + c;
          const bits_t* src = static_cast<const bits_t*>(g.in) + bc * n_in;
          bits_t* dst = static_cast<bits_t*>(g.out) + bc * n_out + col;
          dst[static_cast<int64_t>(t) * slab] = off >= 0 ? src[off] : static_cast<bits_t>(0);
        }
      }
    }
  }
}

}  // namespace tio
