// tio_patch_accumulate — dense-inference patch aggregation on the device.
//
// Reference: PatchAggregator.add_batch / _add_crop / _add_average / _add_hann
// (src/torchio/data/aggregator.py:76-232), which moves every model output to the host
// (`tensor.cpu()`, :95) and adds it with one Python slice assignment per patch.
//
// Design: the kernel is a GATHER over volume voxels, not a scatter over patch voxels.
// One thread owns one (channel, i, j, k) of the bounding box of the call's placements and
// applies the patches that cover it in patch order, so overlapping contributions are added
// in exactly the reference's order (bit-exact float sums, last writer wins for 'crop')
// without atomics.  Lanes run along K: patch reads and volume read-modify-writes are
// coalesced dword / 8-byte streams.  HBM-bound: per call it reads the patches once and
// read-modify-writes the touched part of the accumulators.
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace tio {

struct AggregateArgs {
  void* out;
  void* weight_sum;
  const void* patches;
  const float* window[3];
  int channels;
  int vol[3];
  int patch[3];
  int box_ini[3], box_ext[3];  // bounding box of the placements inside the volume
  int n_patches;
  int mode;
  tio_patch_placement place[TIO_MAX_PATCHES];
};

// storage type <-> compute type: float64 computes in double, the rest in float32
template <int DT>
struct Compute {
  using type = float;
  static __device__ __forceinline__ float load(const void* p, int64_t i) { return Elem<DT>::load(p, i); }
  static __device__ __forceinline__ void store(void* p, int64_t i, float v) { Elem<DT>::store(p, i, v); }
};
template <>
struct Compute<TIO_F64> {
  using type = double;
  static __device__ __forceinline__ double load(const void* p, int64_t i) { return static_cast<const double*>(p)[i]; }
  static __device__ __forceinline__ void store(void* p, int64_t i, double v) { static_cast<double*>(p)[i] = v; }
};

// the value `x` has after a round trip through the storage dtype (what an in-place op on a
// float16 / bfloat16 tensor leaves behind)
template <int DT>
__device__ __forceinline__ typename Compute<DT>::type round_to_storage(typename Compute<DT>::type x) {
  if constexpr (DT == TIO_F16) {
    return static_cast<float>(static_cast<_Float16>(x));
  } else if constexpr (DT == TIO_BF16) {
    return bf16_bits_to_float(float_to_bf16_bits(x));
  } else {
    return x;
  }
}

template <int DT, int MODE>
__global__ __launch_bounds__(256) void accumulate_kernel(const AggregateArgs a) {
  using T = typename Compute<DT>::type;
  const int64_t box_n = static_cast<int64_t>(a.box_ext[0]) * a.box_ext[1] * a.box_ext[2];
  const int64_t total = box_n * a.channels;
  const int64_t vol_n = static_cast<int64_t>(a.vol[0]) * a.vol[1] * a.vol[2];
  const int64_t patch_n = static_cast<int64_t>(a.patch[0]) * a.patch[1] * a.patch[2];
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total;
       t += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int c = static_cast<int>(t / box_n);
    int64_t r = t - c * box_n;
    const int k = a.box_ini[2] + static_cast<int>(r % a.box_ext[2]);
    r /= a.box_ext[2];
    const int j = a.box_ini[1] + static_cast<int>(r % a.box_ext[1]);
    const int i = a.box_ini[0] + static_cast<int>(r / a.box_ext[1]);
    const int64_t v_idx = c * vol_n + (static_cast<int64_t>(i) * a.vol[1] + j) * a.vol[2] + k;

    T acc = T(0), weight = T(0);
    bool touched = false, loaded = false;
    for (int p = 0; p < a.n_patches; p++) {
      const tio_patch_placement& q = a.place[p];
      const int di = i - q.dst_ini[0], dj = j - q.dst_ini[1], dk = k - q.dst_ini[2];
      const bool inside = (static_cast<unsigned>(di) < static_cast<unsigned>(q.extent[0])) &
                          (static_cast<unsigned>(dj) < static_cast<unsigned>(q.extent[1])) &
                          (static_cast<unsigned>(dk) < static_cast<unsigned>(q.extent[2]));
      if (!inside) continue;
      const int si = q.src_ini[0] + di, sj = q.src_ini[1] + dj, sk = q.src_ini[2] + dk;
      const int64_t p_idx = (static_cast<int64_t>(p) * a.channels + c) * patch_n +
                            (static_cast<int64_t>(si) * a.patch[1] + sj) * a.patch[2] + sk;
      if constexpr (MODE == TIO_OVERLAP_CROP) {
        // element copy, any dtype: the last covering patch wins
        const int es = dtype_size(DT);
        const char* s = static_cast<const char*>(a.patches) + p_idx * es;
        char* d = static_cast<char*>(a.out) + v_idx * es;
        for (int e = 0; e < es; e++) d[e] = s[e];
      } else {
        if (!loaded) {
          acc = Compute<DT>::load(a.out, v_idx);
          weight = Compute<DT>::load(a.weight_sum, v_idx);
          loaded = true;
        }
        const T value = Compute<DT>::load(a.patches, p_idx);
        if constexpr (MODE == TIO_OVERLAP_AVERAGE) {
          acc = round_to_storage<DT>(acc + value);
          weight = round_to_storage<DT>(weight + T(1));
        } else {
          // (wi * wj) * wk in float32, then promoted to the compute type (patch * window)
          const float w32 = __fmul_rn(__fmul_rn(a.window[0][si], a.window[1][sj]), a.window[2][sk]);
          const T w = static_cast<T>(w32);
          acc = round_to_storage<DT>(acc + value * w);
          weight = round_to_storage<DT>(weight + w);
        }
        touched = true;
      }
    }
    if constexpr (MODE != TIO_OVERLAP_CROP) {
      if (touched) {
        Compute<DT>::store(a.out, v_idx, acc);
        Compute<DT>::store(a.weight_sum, v_idx, weight);
      }
    }
  }
}

template <int DT>
int launch_accumulate(const AggregateArgs& a, hipStream_t s) {
  const int64_t total = static_cast<int64_t>(a.box_ext[0]) * a.box_ext[1] * a.box_ext[2] * a.channels;
  if (total == 0) return TIO_OK;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;  // grid-stride beyond 64 blocks per CU
  const dim3 grid(static_cast<unsigned>(blocks)), block(256);
  switch (a.mode) {
    case TIO_OVERLAP_CROP: hipLaunchKernelGGL((accumulate_kernel<DT, TIO_OVERLAP_CROP>), grid, block, 0, s, a); break;
    case TIO_OVERLAP_AVERAGE:
      if constexpr (DT == TIO_F32 || DT == TIO_F64 || DT == TIO_F16 || DT == TIO_BF16)
        hipLaunchKernelGGL((accumulate_kernel<DT, TIO_OVERLAP_AVERAGE>), grid, block, 0, s, a);
      break;
    default:
      if constexpr (DT == TIO_F32 || DT == TIO_F64 || DT == TIO_F16 || DT == TIO_BF16)
        hipLaunchKernelGGL((accumulate_kernel<DT, TIO_OVERLAP_HANN>), grid, block, 0, s, a);
      break;
  }
  return check_launch("tio_patch_accumulate");
}

}  // namespace tio

extern "C" int tio_patch_accumulate(void* out, void* weight_sum, int32_t dtype, int32_t channels,
                                    const int32_t vol_shape[3], const void* patches, int32_t n_patches,
                                    const int32_t patch_shape[3], const tio_patch_placement* placements_host,
                                    int32_t mode, const float* window_i_dev, const float* window_j_dev,
                                    const float* window_k_dev, void* stream) {
  using namespace tio;
  if (out == nullptr || patches == nullptr || vol_shape == nullptr || patch_shape == nullptr || placements_host == nullptr)
    return fail(TIO_ERR_INVALID_ARGUMENT, "tio_patch_accumulate: null argument");
  if (mode != TIO_OVERLAP_CROP && mode != TIO_OVERLAP_AVERAGE && mode != TIO_OVERLAP_HANN)
    return fail(TIO_ERR_INVALID_ARGUMENT, "tio_patch_accumulate: mode %d", mode);
  if (channels < 1 || n_patches < 0 || n_patches > TIO_MAX_PATCHES)
    return fail(TIO_ERR_INVALID_ARGUMENT, "tio_patch_accumulate: channels=%d n_patches=%d (max %d)", channels, n_patches,
                TIO_MAX_PATCHES);
  if (dtype_size(dtype) == 0) return fail(TIO_ERR_UNSUPPORTED_DTYPE, "tio_patch_accumulate: dtype %d", dtype);
  if (mode != TIO_OVERLAP_CROP) {
    if (!is_float_dtype(dtype))
      return fail(TIO_ERR_UNSUPPORTED_DTYPE, "tio_patch_accumulate: 'average' / 'hann' need a floating dtype, got %d", dtype);
    if (weight_sum == nullptr) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_patch_accumulate: weight_sum is null");
  }
  if (mode == TIO_OVERLAP_HANN && (window_i_dev == nullptr || window_j_dev == nullptr || window_k_dev == nullptr))
    return fail(TIO_ERR_INVALID_ARGUMENT, "tio_patch_accumulate: 'hann' needs the three windows");
  AggregateArgs a{};
  a.out = out; a.weight_sum = weight_sum; a.patches = patches;
  a.window[0] = window_i_dev; a.window[1] = window_j_dev; a.window[2] = window_k_dev;
  a.channels = channels; a.n_patches = n_patches; a.mode = mode;
  int lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {0, 0, 0};
  for (int d = 0; d < 3; d++) {
    if (vol_shape[d] < 1 || patch_shape[d] < 1) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_patch_accumulate: shapes must be >= 1");
    a.vol[d] = vol_shape[d];
    a.patch[d] = patch_shape[d];
  }
  for (int p = 0; p < n_patches; p++) {
    const tio_patch_placement& q = placements_host[p];
    for (int d = 0; d < 3; d++) {
      if (q.extent[d] < 0 || q.src_ini[d] < 0 || q.src_ini[d] + q.extent[d] > patch_shape[d] || q.dst_ini[d] < 0 ||
          q.dst_ini[d] + q.extent[d] > vol_shape[d])
        return fail(TIO_ERR_INVALID_ARGUMENT, "tio_patch_accumulate: placement %d leaves the patch or the volume on axis %d", p, d);
    }
    a.place[p] = q;
    if (q.extent[0] == 0 || q.extent[1] == 0 || q.extent[2] == 0) continue;
    for (int d = 0; d < 3; d++) {
      lo[d] = q.dst_ini[d] < lo[d] ? q.dst_ini[d] : lo[d];
      hi[d] = q.dst_ini[d] + q.extent[d] > hi[d] ? q.dst_ini[d] + q.extent[d] : hi[d];
    }
  }
  if (n_patches == 0 || lo[0] == INT32_MAX) return TIO_OK;
  for (int d = 0; d < 3; d++) {
    a.box_ini[d] = lo[d];
    a.box_ext[d] = hi[d] - lo[d];
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case TIO_F32: return launch_accumulate<TIO_F32>(a, s);
    case TIO_F64: return launch_accumulate<TIO_F64>(a, s);
    case TIO_F16: return launch_accumulate<TIO_F16>(a, s);
    case TIO_BF16: return launch_accumulate<TIO_BF16>(a, s);
    case TIO_U8: return launch_accumulate<TIO_U8>(a, s);
    case TIO_I8: return launch_accumulate<TIO_I8>(a, s);
    case TIO_I16: return launch_accumulate<TIO_I16>(a, s);
    case TIO_I32: return launch_accumulate<TIO_I32>(a, s);
    default: return launch_accumulate<TIO_I64>(a, s);
  }
}
