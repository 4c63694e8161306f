// labels.hip — the sorted set of values of a label map, `torch.unique(data)`, for the partial-volume
// label mode (transforms/spatial/spatial.py:1360: it sizes the one-hot encoding).
//
// torch.unique sorts the whole tensor (a 512^3 int16 label map: 1.7 ms on this GPU, more than half of
// the fused label resampling it feeds).  Label maps are 8- or 16-bit integers in practice, so the set
// fits a presence bitmap of at most 65536 bits: one streaming pass marks bits (each block in its own
// LDS copy; a plain read first, the atomic only for a bit that is not set yet — after the first few
// hundred elements every lane finds its bit set and the pass is a pure load stream), one small block
// compacts the bitmap into the ascending table.  Wider or floating dtypes are refused (the host keeps
// torch.unique for those).
#include "common.hpp"

namespace tio {
namespace {

constexpr int kBitmapWords = 2048;  // 65536 bits

template <typename T>
__device__ __forceinline__ unsigned ordered_index(T v);  // ascending index == ascending value
template <>
__device__ __forceinline__ unsigned ordered_index<uint8_t>(uint8_t v) { return v; }
template <>
__device__ __forceinline__ unsigned ordered_index<int8_t>(int8_t v) { return static_cast<unsigned>(static_cast<int>(v) + 128); }
template <>
__device__ __forceinline__ unsigned ordered_index<int16_t>(int16_t v) { return static_cast<unsigned>(static_cast<int>(v) + 32768); }

template <typename T>
__global__ __launch_bounds__(256) void mark_labels_kernel(const T* __restrict__ x, int64_t n, unsigned* __restrict__ bitmap) {
  constexpr int PER = 16 / sizeof(T);  // elements per 16-byte load
  __shared__ unsigned seen[kBitmapWords];
  for (int w = threadIdx.x; w < kBitmapWords; w += 256) seen[w] = 0u;
  __syncthreads();
  auto mark = [&](T v) {
    const unsigned idx = ordered_index<T>(v), bit = 1u << (idx & 31u);
    if ((seen[idx >> 5] & bit) == 0u) atomicOr(&seen[idx >> 5], bit);
  };
  const int64_t vectors = n / PER;
  const uint4* xv = reinterpret_cast<const uint4*>(x);  // the entry point checks the 16-byte alignment
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < vectors; i += static_cast<int64_t>(gridDim.x) * 256) {
    const uint4 raw = xv[i];
    T values[PER];
    __builtin_memcpy(values, &raw, 16);
#pragma unroll
    for (int e = 0; e < PER; e++) mark(values[e]);
  }
  if (blockIdx.x == 0)
    for (int64_t i = vectors * PER + threadIdx.x; i < n; i += 256) mark(x[i]);
  __syncthreads();
  for (int w = threadIdx.x; w < kBitmapWords; w += 256)
    if (seen[w] != 0u) atomicOr(&bitmap[w], seen[w]);
}

// One block of 1024 threads, two words each: popcount prefix sum, then every thread writes its labels.
__global__ __launch_bounds__(1024) void compact_labels_kernel(const unsigned* __restrict__ bitmap, int bias, double* __restrict__ table,
                                                              int32_t* __restrict__ count) {
  __shared__ int sums[1024];
  const int t = threadIdx.x;
  const unsigned w0 = bitmap[2 * t], w1 = bitmap[2 * t + 1];
  const int mine = __popc(w0) + __popc(w1);
  sums[t] = mine;
  __syncthreads();
  for (int step = 1; step < 1024; step <<= 1) {  // inclusive Hillis-Steele scan
    const int add = t >= step ? sums[t - step] : 0;
    __syncthreads();
    sums[t] += add;
    __syncthreads();
  }
  int at = sums[t] - mine;
  for (int half = 0; half < 2; half++) {
    unsigned w = half == 0 ? w0 : w1;
    while (w != 0u) {
      const int bit = __ffs(static_cast<int>(w)) - 1;
      w &= w - 1u;
      table[at++] = static_cast<double>((2 * t + half) * 32 + bit - bias);
    }
  }
  if (t == 1023) *count = sums[t];
}

}  // namespace
}  // namespace tio

extern "C" int tio_unique_labels(const void* x, int32_t dtype, int64_t n, double* table_dev, int32_t* count_dev,
                                 void* workspace_dev, void* stream) {
  using namespace tio;
  if (dtype != TIO_U8 && dtype != TIO_I8 && dtype != TIO_I16)
    return fail(TIO_ERR_UNSUPPORTED_DTYPE, "tio_unique_labels: 8- and 16-bit integer label maps only (dtype %d)", dtype);
  if (n < 0) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_unique_labels: negative size");
  if (table_dev == nullptr || count_dev == nullptr || workspace_dev == nullptr || (n > 0 && x == nullptr))
    return fail(TIO_ERR_INVALID_ARGUMENT, "tio_unique_labels: null argument");
  if (reinterpret_cast<uintptr_t>(x) % 16 != 0) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_unique_labels: data must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  unsigned* bitmap = static_cast<unsigned*>(workspace_dev);
  if (hipMemsetAsync(bitmap, 0, kBitmapWords * sizeof(unsigned), s) != hipSuccess) return fail(TIO_ERR_LAUNCH, "tio_unique_labels: memset failed");
  if (n > 0) {
    const int es = dtype_size(dtype);
    int64_t blocks = (n / (16 / es) + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    const dim3 grid(static_cast<unsigned>(blocks));
    if (dtype == TIO_U8) hipLaunchKernelGGL(mark_labels_kernel<uint8_t>, grid, dim3(256), 0, s, static_cast<const uint8_t*>(x), n, bitmap);
    else if (dtype == TIO_I8) hipLaunchKernelGGL(mark_labels_kernel<int8_t>, grid, dim3(256), 0, s, static_cast<const int8_t*>(x), n, bitmap);
    else hipLaunchKernelGGL(mark_labels_kernel<int16_t>, grid, dim3(256), 0, s, static_cast<const int16_t*>(x), n, bitmap);
    if (const int rc = check_launch("tio_unique_labels (mark)")) return rc;
  }
  const int bias = dtype == TIO_U8 ? 0 : (dtype == TIO_I8 ? 128 : 32768);
  hipLaunchKernelGGL(compact_labels_kernel, dim3(1), dim3(1024), 0, s, bitmap, bias, table_dev, count_dev);
  return check_launch("tio_unique_labels (compact)");
}
