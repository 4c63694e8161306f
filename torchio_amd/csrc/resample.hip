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
// once); no MFMA: this is a gather/stencil op.
#include "common.hpp"

namespace tio {

struct ImgArgs {
  const void* in;
  void* out;
  const float* fill;  // nullptr → no mask step
  int channels;
  int dtype;
  int interp;
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
  float sp0, sp1, sp2;           // spacing that converts mm → voxels
  float scale_i, scale_j, scale_k;  // ATen lerp scales of the control grid
  int n_images;
  ImgArgs img[TIO_MAX_IMAGES];
  // tiling
  int tiles_k, tiles_j;          // tiles per row / per slab
};

constexpr int kRowsPerBlock = 4;   // one wave per output row (jo), 4 rows per block
constexpr int kLanes = 64;         // contiguous ko per wave → coalesced stores
constexpr int kMaxCpLds = 6144;    // floats of control points staged in LDS (24 KiB)

// [c,1] @ M^T for one row of M: the rounding sequence of MKL sgemm (K = 4),
// pinned against the reference in tests/golden.
__device__ __forceinline__ float affine_row(const float* __restrict__ m, float a, float b, float c) {
  float t = a * m[0];
  t = __builtin_fmaf(b, m[1], t);
  t = __builtin_fmaf(c, m[2], t);
  t = __builtin_fmaf(1.0f, m[3], t);
  return t;
}

// g = 2 v / max(S-1,1) - 1 (spatial.py:1638-1646) followed by ATen's
// grid_sampler_unnormalize(align_corners=True): ((g + 1) / 2) * (S - 1).
__device__ __forceinline__ float normalise_roundtrip(float v, int size) {
  const float denom = static_cast<float>(max(size - 1, 1));
  const float g = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, v), denom), 1.0f);
  return __fmul_rn(__fmul_rn(__fadd_rn(g, 1.0f), 0.5f), static_cast<float>(size - 1));
}

__device__ __forceinline__ bool in_bounds(float f, int n) {
  return f >= 0.0f && f <= static_cast<float>(n - 1);
}

// trilinear lookup of one displacement component from the (ni,nj,nk,3) field
__device__ __forceinline__ float cp_trilerp(const float* v, int s_i, int s_j, const Lerp1D& li,
                                            const Lerp1D& lj, const Lerp1D& lk) {
  const float* p00 = v + li.i0 * s_i + lj.i0 * s_j;
  const float* p01 = v + li.i0 * s_i + lj.i1 * s_j;
  const float* p10 = v + li.i1 * s_i + lj.i0 * s_j;
  const float* p11 = v + li.i1 * s_i + lj.i1 * s_j;
  const float a00 = lerp2(p00[lk.i0 * 3], lk.l0, p00[lk.i1 * 3], lk.l1);
  const float a01 = lerp2(p01[lk.i0 * 3], lk.l0, p01[lk.i1 * 3], lk.l1);
  const float a10 = lerp2(p10[lk.i0 * 3], lk.l0, p10[lk.i1 * 3], lk.l1);
  const float a11 = lerp2(p11[lk.i0 * 3], lk.l0, p11[lk.i1 * 3], lk.l1);
  const float b0 = lerp2(a00, lj.l0, a01, lj.l1);
  const float b1 = lerp2(a10, lj.l0, a11, lj.l1);
  return lerp2(b0, li.l0, b1, li.l1);
}

__global__ __launch_bounds__(kRowsPerBlock* kLanes) void resample_kernel(const ResampleArgs a) {
  extern __shared__ __attribute__((aligned(16))) float s_cp[];

  // tile decode: XCD-contiguous chunks of (b, io, jt, kt), kt fastest
  const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
  const int kt = tile % a.tiles_k;
  const unsigned t1 = tile / a.tiles_k;
  const int jt = t1 % a.tiles_j;
  const unsigned t2 = t1 / a.tiles_j;
  const int io = t2 % a.Io;
  const int b = t2 / a.Io;

  const int lane = threadIdx.x & (kLanes - 1);
  const int wave = threadIdx.x / kLanes;
  const int jo = jt * kRowsPerBlock + wave;
  const int ko = kt * kLanes + lane;

  const bool elastic = a.cp != nullptr && !(a.cp_skip != nullptr && a.cp_skip[b] != 0);
  const bool pass = a.passthrough != nullptr && a.passthrough[b] != 0;
  const int n_cp = a.ni * a.nj * a.nk * 3;
  const float* cp = nullptr;
  if (elastic && !pass) {
    const float* src = a.cp + (a.cp_batched ? static_cast<int64_t>(b) * n_cp : 0);
    if (n_cp <= kMaxCpLds) {
      for (int t = threadIdx.x; t < n_cp; t += blockDim.x) s_cp[t] = src[t];
      __syncthreads();
      cp = s_cp;
    } else {
      cp = src;  // oversized control grid: read through the cache
    }
  }
  if (jo >= a.Jo || ko >= a.Ko) return;

  const int64_t n_in = static_cast<int64_t>(a.I) * a.J * a.K;
  const int64_t n_out = static_cast<int64_t>(a.Io) * a.Jo * a.Ko;
  const int64_t o_idx = (static_cast<int64_t>(io) * a.Jo + jo) * a.Ko + ko;

  if (pass) {  // gated-out element: bit-exact copy (spatial.py:1101-1106)
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
    return;
  }

  const float* m = a.mapping + (a.mapping_batched ? b * 12 : 0);
  const float ci = static_cast<float>(io), cj = static_cast<float>(jo), ck = static_cast<float>(ko);
  float vi, vj, vk;
  if (elastic) {
    const Lerp1D li = lerp_index(io, a.ni, a.Io, a.scale_i);
    const Lerp1D lj = lerp_index(jo, a.nj, a.Jo, a.scale_j);
    const Lerp1D lk = lerp_index(ko, a.nk, a.Ko, a.scale_k);
    const int s_i = a.nj * a.nk * 3, s_j = a.nk * 3;
    const float di = cp_trilerp(cp + 0, s_i, s_j, li, lj, lk);
    const float dj = cp_trilerp(cp + 1, s_i, s_j, li, lj, lk);
    const float dk = cp_trilerp(cp + 2, s_i, s_j, li, lj, lk);
    if (a.affine_first) {  // spatial.py:1570-1573
      vi = __fadd_rn(affine_row(m + 0, ci, cj, ck), __fdiv_rn(di, a.sp0));
      vj = __fadd_rn(affine_row(m + 4, ci, cj, ck), __fdiv_rn(dj, a.sp1));
      vk = __fadd_rn(affine_row(m + 8, ci, cj, ck), __fdiv_rn(dk, a.sp2));
    } else {  // spatial.py:1574-1577
      const float ei = __fadd_rn(ci, __fdiv_rn(di, a.sp0));
      const float ej = __fadd_rn(cj, __fdiv_rn(dj, a.sp1));
      const float ek = __fadd_rn(ck, __fdiv_rn(dk, a.sp2));
      vi = affine_row(m + 0, ei, ej, ek);
      vj = affine_row(m + 4, ei, ej, ek);
      vk = affine_row(m + 8, ei, ej, ek);
    }
  } else {  // spatial.py:1542-1543
    vi = affine_row(m + 0, ci, cj, ck);
    vj = affine_row(m + 4, ci, cj, ck);
    vk = affine_row(m + 8, ci, cj, ck);
  }
  // torchio axis i ≡ grid x ≡ ATen W ; j ≡ y ≡ H ; k ≡ z ≡ D
  const float x = normalise_roundtrip(vi, a.I);
  const float y = normalise_roundtrip(vj, a.J);
  const float z = normalise_roundtrip(vk, a.K);

  // ATen grid_sampler_3d corner weights, order tnw,tne,tsw,tse,bnw,bne,bsw,bse
  const float x0 = floorf(x), y0 = floorf(y), z0 = floorf(z);
  const float x1 = x0 + 1.0f, y1 = y0 + 1.0f, z1 = z0 + 1.0f;
  const float wx0 = x1 - x, wx1 = x - x0;
  const float wy0 = y1 - y, wy1 = y - y0;
  const float wz0 = z1 - z, wz1 = z - z0;
  float w[8];
  w[0] = __fmul_rn(__fmul_rn(wx0, wy0), wz0);
  w[1] = __fmul_rn(__fmul_rn(wx1, wy0), wz0);
  w[2] = __fmul_rn(__fmul_rn(wx0, wy1), wz0);
  w[3] = __fmul_rn(__fmul_rn(wx1, wy1), wz0);
  w[4] = __fmul_rn(__fmul_rn(wx0, wy0), wz1);
  w[5] = __fmul_rn(__fmul_rn(wx1, wy0), wz1);
  w[6] = __fmul_rn(__fmul_rn(wx0, wy1), wz1);
  w[7] = __fmul_rn(__fmul_rn(wx1, wy1), wz1);

  const bool bx0 = in_bounds(x0, a.I), bx1 = in_bounds(x1, a.I);
  const bool by0 = in_bounds(y0, a.J), by1 = in_bounds(y1, a.J);
  const bool bz0 = in_bounds(z0, a.K), bz1 = in_bounds(z1, a.K);
  bool ok[8];
  int off[8];
#pragma unroll
  for (int t = 0; t < 8; t++) {
    const bool bx = (t & 1) ? bx1 : bx0, by = (t & 2) ? by1 : by0, bz = (t & 4) ? bz1 : bz0;
    ok[t] = bx && by && bz;
    const float fx = (t & 1) ? x1 : x0, fy = (t & 2) ? y1 : y0, fz = (t & 4) ? z1 : z0;
    off[t] = ok[t] ? (static_cast<int>(fx) * a.J + static_cast<int>(fy)) * a.K + static_cast<int>(fz) : 0;
  }
  float mask = 0.0f;  // == F.grid_sample(ones) (spatial.py:1721-1727)
#pragma unroll
  for (int t = 0; t < 8; t++) mask = ok[t] ? __fadd_rn(mask, w[t]) : mask;

  // nearest: nearbyint = round half to even (v_rndne_f32)
  const float xn = rintf(x), yn = rintf(y), zn = rintf(z);
  const bool okn = in_bounds(xn, a.I) && in_bounds(yn, a.J) && in_bounds(zn, a.K);
  const int offn = okn ? (static_cast<int>(xn) * a.J + static_cast<int>(yn)) * a.K + static_cast<int>(zn) : 0;

  for (int im = 0; im < a.n_images; im++) {
    const ImgArgs& g = a.img[im];
    for (int c = 0; c < g.channels; c++) {
      const int64_t bc = static_cast<int64_t>(b) * g.channels + c;
      const int64_t base_in = bc * n_in;
      float val;
      if (g.interp == TIO_LINEAR) {
        val = 0.0f;
        if (g.dtype == TIO_F32) {
          const float* p = static_cast<const float*>(g.in) + base_in;
#pragma unroll
          for (int t = 0; t < 8; t++) {
            const float v = p[off[t]];
            val = ok[t] ? __fadd_rn(val, __fmul_rn(v, w[t])) : val;
          }
        } else {
#pragma unroll
          for (int t = 0; t < 8; t++) {
            const float v = load_as_float(g.in, g.dtype, base_in + off[t]);
            val = ok[t] ? __fadd_rn(val, __fmul_rn(v, w[t])) : val;
          }
        }
      } else {
        const float v = load_as_float(g.in, g.dtype, base_in + offn);
        val = okn ? v : 0.0f;
      }
      if (g.fill != nullptr) val = (mask > 0.5f) ? val : g.fill[c];
      store_from_float(g.out, g.dtype, bc * n_out + o_idx, val);
    }
  }
}

}  // namespace tio

extern "C" int tio_resample3d(const tio_resample_geom* geom, int32_t n_images,
                              const tio_resample_image* images, void* stream) {
  using namespace tio;
  if (geom == nullptr || images == nullptr) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_resample3d: null argument");
  if (n_images < 1 || n_images > TIO_MAX_IMAGES)
    return fail(TIO_ERR_INVALID_ARGUMENT, "tio_resample3d: n_images=%d not in [1, %d]", n_images, TIO_MAX_IMAGES);
  if (geom->mapping_dev == nullptr) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_resample3d: mapping_dev is null");
  if (geom->batch < 0) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_resample3d: negative batch");
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
  a.sp0 = sp[0]; a.sp1 = sp[1]; a.sp2 = sp[2];
  if (a.cp != nullptr) {
    a.scale_i = lerp_scale(a.ni, a.Io);
    a.scale_j = lerp_scale(a.nj, a.Jo);
    a.scale_k = lerp_scale(a.nk, a.Ko);
  }
  a.n_images = n_images;
  for (int i = 0; i < n_images; i++) {
    const tio_resample_image& s = images[i];
    if (s.in == nullptr || s.out == nullptr || s.channels < 1)
      return fail(TIO_ERR_INVALID_ARGUMENT, "tio_resample3d: image %d has null data or no channels", i);
    if (dtype_size(s.dtype) == 0) return fail(TIO_ERR_UNSUPPORTED_DTYPE, "tio_resample3d: image %d dtype %d", i, s.dtype);
    if (s.interp != TIO_NEAREST && s.interp != TIO_LINEAR)
      return fail(TIO_ERR_INVALID_ARGUMENT, "tio_resample3d: image %d interp %d", i, s.interp);
    a.img[i] = ImgArgs{s.in, s.out, s.fill_dev, s.channels, s.dtype, s.interp};
  }
  if (a.B == 0) return TIO_OK;

  a.tiles_k = (a.Ko + kLanes - 1) / kLanes;
  a.tiles_j = (a.Jo + kRowsPerBlock - 1) / kRowsPerBlock;
  const int64_t blocks = static_cast<int64_t>(a.B) * a.Io * a.tiles_j * a.tiles_k;
  if (blocks >= (1LL << 31)) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_resample3d: grid too large");
  size_t lds = 0;
  if (a.cp != nullptr) {
    const int n_cp = a.ni * a.nj * a.nk * 3;
    if (n_cp <= kMaxCpLds) lds = static_cast<size_t>(n_cp) * sizeof(float);
  }
  hipLaunchKernelGGL(resample_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kRowsPerBlock * kLanes), lds,
                     static_cast<hipStream_t>(stream), a);
  return check_launch("tio_resample3d");
}
