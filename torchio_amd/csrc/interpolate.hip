// tio_interpolate3d / tio_axis_gather_lerp — F.interpolate users on the augmentation path
// (SURVEY §8f rank 3): Resize (transforms/spatial/resize.py:57-82) and Anisotropy
// (transforms/spatial/anisotropy.py:128-392).
//
// Both are pure streaming kernels (HBM bound: read the input once, write the output once),
// one thread per output voxel with lanes along K.  The arithmetic is ATen's:
//   nearest (legacy "nearest"): src = min(floor(dst * float(in) / float(out)), in - 1)
//   trilinear, align_corners=True: lerp_index() per axis and the K-, J-, I-nested lerp2()
//   (common.hpp; pinned bit for bit against F.interpolate by the oracle's golden vectors).
#include <hip/hip_runtime.h>

#include <string.h>

#include <type_traits>

#include "common.hpp"

namespace tio {

static inline uint16_t float_to_bf16_bits_host(float f) {  // round to nearest even, like torch
  uint32_t x;
  memcpy(&x, &f, 4);
  if ((x & 0x7FFFFFFFu) > 0x7F800000u) return 0x7FC0u;
  x += 0x7FFFu + ((x >> 16) & 1u);
  return static_cast<uint16_t>(x >> 16);
}

struct InterpArgs {
  const void* x;
  void* y;
  int64_t n_bc;
  int in[3], out[3];
  float scale[3];  // nearest: float(in) / float(out); linear: (in - 1) / (out - 1)
  int mode;
};

template <int DT>
__global__ __launch_bounds__(256) void interpolate_kernel(const InterpArgs a) {
  const int64_t n_out = static_cast<int64_t>(a.out[0]) * a.out[1] * a.out[2];
  const int64_t n_in = static_cast<int64_t>(a.in[0]) * a.in[1] * a.in[2];
  const int64_t total = n_out * a.n_bc;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total;
       t += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t bc = t / n_out;
    int64_t r = t - bc * n_out;
    const int k = static_cast<int>(r % a.out[2]);
    r /= a.out[2];
    const int j = static_cast<int>(r % a.out[1]);
    const int i = static_cast<int>(r / a.out[1]);
    const int64_t base = bc * n_in;
    float value;
    if (a.mode == TIO_NEAREST) {
      const int si = min(static_cast<int>(floorf(__fmul_rn(static_cast<float>(i), a.scale[0]))), a.in[0] - 1);
      const int sj = min(static_cast<int>(floorf(__fmul_rn(static_cast<float>(j), a.scale[1]))), a.in[1] - 1);
      const int sk = min(static_cast<int>(floorf(__fmul_rn(static_cast<float>(k), a.scale[2]))), a.in[2] - 1);
      // a pure element move: copy the bits (no float round trip for 64-bit types)
      const int es = dtype_size(DT);
      const char* s = static_cast<const char*>(a.x) + (base + (static_cast<int64_t>(si) * a.in[1] + sj) * a.in[2] + sk) * es;
      char* d = static_cast<char*>(a.y) + t * es;
      for (int e = 0; e < es; e++) d[e] = s[e];
      continue;
    }
    const Lerp1D li = lerp_index(i, a.in[0], a.out[0], a.scale[0]);
    const Lerp1D lj = lerp_index(j, a.in[1], a.out[1], a.scale[1]);
    const Lerp1D lk = lerp_index(k, a.in[2], a.out[2], a.scale[2]);
    const int64_t r00 = base + (static_cast<int64_t>(li.i0) * a.in[1] + lj.i0) * a.in[2];
    const int64_t r01 = base + (static_cast<int64_t>(li.i0) * a.in[1] + lj.i1) * a.in[2];
    const int64_t r10 = base + (static_cast<int64_t>(li.i1) * a.in[1] + lj.i0) * a.in[2];
    const int64_t r11 = base + (static_cast<int64_t>(li.i1) * a.in[1] + lj.i1) * a.in[2];
    const float a00 = lerp2(Elem<DT>::load(a.x, r00 + lk.i0), lk.l0, Elem<DT>::load(a.x, r00 + lk.i1), lk.l1);
    const float a01 = lerp2(Elem<DT>::load(a.x, r01 + lk.i0), lk.l0, Elem<DT>::load(a.x, r01 + lk.i1), lk.l1);
    const float a10 = lerp2(Elem<DT>::load(a.x, r10 + lk.i0), lk.l0, Elem<DT>::load(a.x, r10 + lk.i1), lk.l1);
    const float a11 = lerp2(Elem<DT>::load(a.x, r11 + lk.i0), lk.l0, Elem<DT>::load(a.x, r11 + lk.i1), lk.l1);
    value = lerp2(lerp2(a00, lj.l0, a01, lj.l1), li.l0, lerp2(a10, lj.l0, a11, lj.l1), li.l1);
    Elem<DT>::store(a.y, t, value);
  }
}

struct AxisArgs {
  const void* x;
  void* y;
  const int32_t* lower;   // (B, length) source index along the axis
  const int32_t* upper;   // (B, length) or nullptr (nearest: a gather)
  const float* weight;    // (B, length) weight of `upper`
  const uint8_t* active;  // (B) 0 = copy the element unchanged; nullptr = all active
  int batch, channels;
  int shape[3];
  int axis;
};

template <int DT>
__global__ __launch_bounds__(256) void axis_gather_lerp_kernel(const AxisArgs a) {
  const int64_t n = static_cast<int64_t>(a.shape[0]) * a.shape[1] * a.shape[2];
  const int64_t total = n * a.batch * a.channels;
  const int length = a.shape[a.axis];
  const int64_t stride = a.axis == 0 ? static_cast<int64_t>(a.shape[1]) * a.shape[2] : (a.axis == 1 ? a.shape[2] : 1);
  const int es = dtype_size(DT);
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total;
       t += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t bc = t / n;
    const int b = static_cast<int>(bc / a.channels);
    const int64_t r = t - bc * n;
    const int k = static_cast<int>(r % a.shape[2]);
    const int j = static_cast<int>((r / a.shape[2]) % a.shape[1]);
    const int i = static_cast<int>(r / (static_cast<int64_t>(a.shape[1]) * a.shape[2]));
    const int p = a.axis == 0 ? i : (a.axis == 1 ? j : k);
    if (a.active != nullptr && a.active[b] == 0) {  // elements with factor <= 1: bit-exact copy
      const char* s = static_cast<const char*>(a.x) + t * es;
      char* d = static_cast<char*>(a.y) + t * es;
      for (int e = 0; e < es; e++) d[e] = s[e];
      continue;
    }
    const int64_t line = t - static_cast<int64_t>(p) * stride;  // position 0 of this voxel's line along the axis
    const int lo = a.lower[static_cast<int64_t>(b) * length + p];
    if (a.upper == nullptr) {  // nearest: gather(data.float()).to(dtype) == an element move
      const char* s = static_cast<const char*>(a.x) + (line + lo * stride) * es;
      char* d = static_cast<char*>(a.y) + t * es;
      for (int e = 0; e < es; e++) d[e] = s[e];
      continue;
    }
    const int hi = a.upper[static_cast<int64_t>(b) * length + p];
    const float w = a.weight[static_cast<int64_t>(b) * length + p];
    // lower * (1.0 - w) + upper * w: three tensor ops in the reference, every one rounds (anisotropy.py:207)
    const float lower_term = __fmul_rn(Elem<DT>::load(a.x, line + lo * stride), __fsub_rn(1.0f, w));
    const float upper_term = __fmul_rn(Elem<DT>::load(a.x, line + hi * stride), w);
    Elem<DT>::store(a.y, t, __fadd_rn(lower_term, upper_term));
  }
}

struct FlipArgs {
  const void* x;
  void* y;
  const uint8_t* flags;  // (B, 3) per-element flags or nullptr (then `mask` applies to every element)
  int batch, channels, shape[3];
  int mask;              // bit a: flip spatial axis a
};

template <int ES>
__global__ __launch_bounds__(256) void flip_kernel(const FlipArgs a) {
  using RAW = typename std::conditional<ES == 1, uint8_t, typename std::conditional<ES == 2, uint16_t,
              typename std::conditional<ES == 4, uint32_t, uint64_t>::type>::type>::type;
  const int64_t n = static_cast<int64_t>(a.shape[0]) * a.shape[1] * a.shape[2];
  const int64_t total = n * a.batch * a.channels;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total;
       t += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t bc = t / n;
    const int b = static_cast<int>(bc / a.channels);
    const int64_t r = t - bc * n;
    int k = static_cast<int>(r % a.shape[2]);
    int j = static_cast<int>((r / a.shape[2]) % a.shape[1]);
    int i = static_cast<int>(r / (static_cast<int64_t>(a.shape[1]) * a.shape[2]));
    const int mask = a.flags != nullptr ? ((a.flags[b * 3] ? 1 : 0) | (a.flags[b * 3 + 1] ? 2 : 0) | (a.flags[b * 3 + 2] ? 4 : 0)) : a.mask;
    if (mask & 1) i = a.shape[0] - 1 - i;
    if (mask & 2) j = a.shape[1] - 1 - j;
    if (mask & 4) k = a.shape[2] - 1 - k;
    static_cast<RAW*>(a.y)[t] = static_cast<const RAW*>(a.x)[bc * n + (static_cast<int64_t>(i) * a.shape[1] + j) * a.shape[2] + k];
  }
}

struct PadArgs {
  const void* x;
  void* y;
  const void* fill_per_element;  // (B) values of the image dtype, or nullptr (then `fill_bits` for every element)
  uint64_t fill_bits;            // the constant, already cast to the image dtype (its low bytes)
  int batch, channels, in[3], out[3], before[3];
  int mode;
};

// source index of output position p (already shifted by the leading pad) for the non-constant modes
__device__ __forceinline__ int pad_source(int q, int n, int mode) {
  if (mode == TIO_PAD_REPLICATE) return min(max(q, 0), n - 1);
  if (mode == TIO_PAD_CIRCULAR) {
    q %= n;
    return q < 0 ? q + n : q;
  }
  // reflect (no edge repeat): ... 2 1 | 0 1 2 ... n-1 | n-2 n-3 ...
  if (n == 1) return 0;
  const int period = 2 * (n - 1);
  q %= period;
  if (q < 0) q += period;
  return q < n ? q : period - q;
}

template <int ES>
__global__ __launch_bounds__(256) void pad_kernel(const PadArgs a) {
  using RAW = typename std::conditional<ES == 1, uint8_t, typename std::conditional<ES == 2, uint16_t,
              typename std::conditional<ES == 4, uint32_t, uint64_t>::type>::type>::type;
  const int64_t n_out = static_cast<int64_t>(a.out[0]) * a.out[1] * a.out[2];
  const int64_t n_in = static_cast<int64_t>(a.in[0]) * a.in[1] * a.in[2];
  const int64_t total = n_out * a.batch * a.channels;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total;
       t += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t bc = t / n_out;
    const int64_t r = t - bc * n_out;
    int k = static_cast<int>(r % a.out[2]) - a.before[2];
    int j = static_cast<int>((r / a.out[2]) % a.out[1]) - a.before[1];
    int i = static_cast<int>(r / (static_cast<int64_t>(a.out[1]) * a.out[2])) - a.before[0];
    const bool inside = (static_cast<unsigned>(i) < static_cast<unsigned>(a.in[0])) & (static_cast<unsigned>(j) < static_cast<unsigned>(a.in[1])) &
                        (static_cast<unsigned>(k) < static_cast<unsigned>(a.in[2]));
    RAW value;
    if (!inside && a.mode == TIO_PAD_CONSTANT) {
      value = a.fill_per_element != nullptr ? static_cast<const RAW*>(a.fill_per_element)[bc / a.channels] : static_cast<RAW>(a.fill_bits);
    } else {
      if (!inside) {
        i = pad_source(i, a.in[0], a.mode);
        j = pad_source(j, a.in[1], a.mode);
        k = pad_source(k, a.in[2], a.mode);
      }
      value = static_cast<const RAW*>(a.x)[bc * n_in + (static_cast<int64_t>(i) * a.in[1] + j) * a.in[2] + k];
    }
    static_cast<RAW*>(a.y)[t] = value;
  }
}

template <typename Kernel, typename Args>
static int launch_stream(Kernel kernel, const Args& a, int64_t total, hipStream_t s, const char* what) {
  if (total == 0) return TIO_OK;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, a);
  return check_launch(what);
}

}  // namespace tio

#define TIO_DISPATCH_DTYPE(DTYPE, CALL) \
  switch (DTYPE) {                      \
    case TIO_F32: CALL(TIO_F32); break; \
    case TIO_F64: CALL(TIO_F64); break; \
    case TIO_F16: CALL(TIO_F16); break; \
    case TIO_BF16: CALL(TIO_BF16); break; \
    case TIO_U8: CALL(TIO_U8); break;   \
    case TIO_I8: CALL(TIO_I8); break;   \
    case TIO_I16: CALL(TIO_I16); break; \
    case TIO_I32: CALL(TIO_I32); break; \
    default: CALL(TIO_I64); break;      \
  }

extern "C" int tio_interpolate3d(const void* x, void* y, int32_t dtype, int64_t n_batch_channels, const int32_t in_shape[3],
                                 const int32_t out_shape[3], int32_t mode, void* stream) {
  using namespace tio;
  if (in_shape == nullptr || out_shape == nullptr) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_interpolate3d: null shape");
  if (mode != TIO_NEAREST && mode != TIO_LINEAR) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_interpolate3d: mode %d", mode);
  if (dtype_size(dtype) == 0) return fail(TIO_ERR_UNSUPPORTED_DTYPE, "tio_interpolate3d: dtype %d", dtype);
  if (n_batch_channels < 0) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_interpolate3d: negative batch");
  InterpArgs a{};
  a.x = x; a.y = y; a.n_bc = n_batch_channels; a.mode = mode;
  for (int d = 0; d < 3; d++) {
    if (in_shape[d] < 1 || out_shape[d] < 1) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_interpolate3d: shapes must be >= 1");
    a.in[d] = in_shape[d];
    a.out[d] = out_shape[d];
    a.scale[d] = mode == TIO_NEAREST ? static_cast<float>(in_shape[d]) / static_cast<float>(out_shape[d])
                                     : lerp_scale(in_shape[d], out_shape[d]);
  }
  if (n_batch_channels == 0) return TIO_OK;
  if (x == nullptr || y == nullptr) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_interpolate3d: null data");
  const int64_t total = static_cast<int64_t>(out_shape[0]) * out_shape[1] * out_shape[2] * n_batch_channels;
  hipStream_t s = static_cast<hipStream_t>(stream);
#define TIO_CALL(DT) return launch_stream(interpolate_kernel<DT>, a, total, s, "tio_interpolate3d")
  TIO_DISPATCH_DTYPE(dtype, TIO_CALL)
#undef TIO_CALL
  return TIO_OK;
}

extern "C" int tio_axis_gather_lerp(const void* x, void* y, int32_t dtype, int32_t batch, int32_t channels, const int32_t shape[3],
                                    int32_t axis, const int32_t* lower_dev, const int32_t* upper_dev, const float* weight_dev,
                                    const uint8_t* active_dev, void* stream) {
  using namespace tio;
  if (shape == nullptr) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_axis_gather_lerp: null shape");
  if (axis < 0 || axis > 2) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_axis_gather_lerp: axis %d", axis);
  if (dtype_size(dtype) == 0) return fail(TIO_ERR_UNSUPPORTED_DTYPE, "tio_axis_gather_lerp: dtype %d", dtype);
  if (batch < 0 || channels < 1) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_axis_gather_lerp: bad batch / channels");
  if (batch == 0) return TIO_OK;
  if (x == nullptr || y == nullptr || lower_dev == nullptr) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_axis_gather_lerp: null argument");
  if (upper_dev != nullptr && weight_dev == nullptr) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_axis_gather_lerp: upper without weights");
  AxisArgs a{};
  a.x = x; a.y = y; a.lower = lower_dev; a.upper = upper_dev; a.weight = weight_dev; a.active = active_dev;
  a.batch = batch; a.channels = channels; a.axis = axis;
  for (int d = 0; d < 3; d++) {
    if (shape[d] < 1) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_axis_gather_lerp: shapes must be >= 1");
    a.shape[d] = shape[d];
  }
  const int64_t total = static_cast<int64_t>(shape[0]) * shape[1] * shape[2] * batch * channels;
  hipStream_t s = static_cast<hipStream_t>(stream);
#define TIO_CALL(DT) return launch_stream(axis_gather_lerp_kernel<DT>, a, total, s, "tio_axis_gather_lerp")
  TIO_DISPATCH_DTYPE(dtype, TIO_CALL)
#undef TIO_CALL
  return TIO_OK;
}

extern "C" int tio_flip3d(const void* x, void* y, int32_t dtype, int32_t batch, int32_t channels, const int32_t shape[3],
                          int32_t axes_mask, const uint8_t* flags_dev, void* stream) {
  using namespace tio;
  if (shape == nullptr) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_flip3d: null shape");
  const int es = dtype_size(dtype);
  if (es == 0) return fail(TIO_ERR_UNSUPPORTED_DTYPE, "tio_flip3d: dtype %d", dtype);
  if (batch < 0 || channels < 1 || axes_mask < 0 || axes_mask > 7) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_flip3d: bad argument");
  if (batch == 0) return TIO_OK;
  if (x == nullptr || y == nullptr) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_flip3d: null data");
  FlipArgs a{};
  a.x = x; a.y = y; a.flags = flags_dev; a.batch = batch; a.channels = channels; a.mask = axes_mask;
  for (int d = 0; d < 3; d++) {
    if (shape[d] < 1) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_flip3d: shapes must be >= 1");
    a.shape[d] = shape[d];
  }
  const int64_t total = static_cast<int64_t>(shape[0]) * shape[1] * shape[2] * batch * channels;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (es) {
    case 1: return launch_stream(flip_kernel<1>, a, total, s, "tio_flip3d");
    case 2: return launch_stream(flip_kernel<2>, a, total, s, "tio_flip3d");
    case 4: return launch_stream(flip_kernel<4>, a, total, s, "tio_flip3d");
    default: return launch_stream(flip_kernel<8>, a, total, s, "tio_flip3d");
  }
}

extern "C" int tio_pad3d(const void* x, void* y, int32_t dtype, int32_t batch, int32_t channels, const int32_t in_shape[3],
                         const int32_t padding[6], int32_t mode, double fill, const void* fill_per_element_dev, void* stream) {
  using namespace tio;
  if (in_shape == nullptr || padding == nullptr) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_pad3d: null argument");
  const int es = dtype_size(dtype);
  if (es == 0) return fail(TIO_ERR_UNSUPPORTED_DTYPE, "tio_pad3d: dtype %d", dtype);
  if (mode < TIO_PAD_CONSTANT || mode > TIO_PAD_CIRCULAR) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_pad3d: mode %d", mode);
  if (batch < 0 || channels < 1) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_pad3d: bad batch / channels");
  PadArgs a{};
  a.x = x; a.y = y; a.fill_per_element = fill_per_element_dev; a.batch = batch; a.channels = channels; a.mode = mode;
  for (int d = 0; d < 3; d++) {
    if (in_shape[d] < 1 || padding[2 * d] < 0 || padding[2 * d + 1] < 0)
      return fail(TIO_ERR_INVALID_ARGUMENT, "tio_pad3d: shapes must be >= 1 and paddings >= 0");
    // F.pad's own limits: reflect needs pad < size, circular pad <= size
    if (mode == TIO_PAD_REFLECT && (padding[2 * d] >= in_shape[d] || padding[2 * d + 1] >= in_shape[d]))
      return fail(TIO_ERR_INVALID_ARGUMENT, "tio_pad3d: reflect padding must be smaller than the axis (axis %d)", d);
    if (mode == TIO_PAD_CIRCULAR && (padding[2 * d] > in_shape[d] || padding[2 * d + 1] > in_shape[d]))
      return fail(TIO_ERR_INVALID_ARGUMENT, "tio_pad3d: circular padding must not exceed the axis (axis %d)", d);
    a.in[d] = in_shape[d];
    a.before[d] = padding[2 * d];
    a.out[d] = in_shape[d] + padding[2 * d] + padding[2 * d + 1];
  }
  if (batch == 0) return TIO_OK;
  if (x == nullptr || y == nullptr) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_pad3d: null data");
  // the constant in the image dtype, like F.pad(value=fill) casts it
  switch (dtype) {
    case TIO_F32: { float v = static_cast<float>(fill); uint32_t b; memcpy(&b, &v, 4); a.fill_bits = b; break; }
    case TIO_F64: { uint64_t b; memcpy(&b, &fill, 8); a.fill_bits = b; break; }
    case TIO_F16: { _Float16 v = static_cast<_Float16>(static_cast<float>(fill)); uint16_t b; memcpy(&b, &v, 2); a.fill_bits = b; break; }
    case TIO_BF16: { a.fill_bits = float_to_bf16_bits_host(static_cast<float>(fill)); break; }
    case TIO_U8: a.fill_bits = static_cast<uint8_t>(static_cast<int64_t>(fill)); break;
    case TIO_I8: a.fill_bits = static_cast<uint8_t>(static_cast<int8_t>(static_cast<int64_t>(fill))); break;
    case TIO_I16: a.fill_bits = static_cast<uint16_t>(static_cast<int16_t>(static_cast<int64_t>(fill))); break;
    case TIO_I32: a.fill_bits = static_cast<uint32_t>(static_cast<int32_t>(static_cast<int64_t>(fill))); break;
    default: a.fill_bits = static_cast<uint64_t>(static_cast<int64_t>(fill)); break;
  }
  const int64_t total = static_cast<int64_t>(a.out[0]) * a.out[1] * a.out[2] * batch * channels;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (es) {
    case 1: return launch_stream(pad_kernel<1>, a, total, s, "tio_pad3d");
    case 2: return launch_stream(pad_kernel<2>, a, total, s, "tio_pad3d");
    case 4: return launch_stream(pad_kernel<4>, a, total, s, "tio_pad3d");
    default: return launch_stream(pad_kernel<8>, a, total, s, "tio_pad3d");
  }
}

// =====================================================================================================================
// tio_bspline_prefilter: B-spline coefficients of orders 2 / 3, half-sample-symmetric boundary (include/tio_hip.h).
// One thread per line of the axis being filtered, the line in place in global memory, the same operations in the same
// order as oracle/tio_oracle.c (spline_filter_line).  Lines along I and J are coalesced across a wave (neighbouring
// threads = neighbouring k); lines along K are not — this is the straightforward version of a path the headline
// pipeline does not use.
// =====================================================================================================================
namespace tio {

template <int DT>
__global__ __launch_bounds__(256) void bspline_load_kernel(const void* __restrict__ x, float* __restrict__ y, int64_t n) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) y[i] = Elem<DT>::load(x, i);
}

__global__ __launch_bounds__(256) void bspline_axis_kernel(float* __restrict__ c, int64_t n_lines, int I, int J, int K, int axis, int order) {
  const int64_t line = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (line >= n_lines) return;
  int64_t base, stride;
  int n;
  if (axis == 0) {  // lines (v, j, k)
    const int64_t jk = static_cast<int64_t>(J) * K;
    const int64_t v = line / jk, r = line - v * jk;
    base = v * I * jk + r; stride = jk; n = I;
  } else if (axis == 1) {  // lines (v, i, k)
    const int64_t vi = line / K, k = line - vi * K;
    base = vi * J * K + k; stride = K; n = J;
  } else {  // lines (v, i, j)
    base = line * K; stride = 1; n = K;
  }
  if (n < 2) return;
  float* p = c + base;
  // the poles of the order (oracle/tio_oracle.c: spline_poles), one after the other
  float poles[3] = {0.0f, 0.0f, 0.0f};
  int n_poles = 1;
  switch (order) {
    case 2: poles[0] = -0.17157287525380990f; break;
    case 3: poles[0] = -0.26794919243112270f; break;
    case 4: poles[0] = -0.36134122590022033f; poles[1] = -0.013725429297339118f; n_poles = 2; break;
    case 5: poles[0] = -0.43057534709997358f; poles[1] = -0.043096288203264665f; n_poles = 2; break;
    case 6: poles[0] = -0.48829458930304598f; poles[1] = -0.081679271076237445f; poles[2] = -0.0014141518083258169f; n_poles = 3; break;
    default: poles[0] = -0.53528043079643883f; poles[1] = -0.12255461519232658f; poles[2] = -0.0091486948096082803f; n_poles = 3; break;
  }
  for (int pole = 0; pole < n_poles; pole++) {
  const float z = pole == 0 ? poles[0] : (pole == 1 ? poles[1] : poles[2]);
  const float gain = __fmul_rn(__fsub_rn(1.0f, z), __fsub_rn(1.0f, __fdiv_rn(1.0f, z)));
  for (int i = 0; i < n; i++) p[i * stride] = __fmul_rn(p[i * stride], gain);
  float z_n = 1.0f;
  for (int i = 0; i < n; i++) z_n = __fmul_rn(z_n, z);
  float z_i = z;
  const float c0 = p[0];
  float acc = __fadd_rn(p[0], __fmul_rn(z_n, p[(n - 1) * stride]));
  for (int i = 1; i < n; i++) {
    acc = __fadd_rn(acc, __fmul_rn(z_i, __fadd_rn(p[i * stride], __fmul_rn(z_n, p[(n - 1 - i) * stride]))));
    z_i = __fmul_rn(z_i, z);
  }
  acc = __fmul_rn(acc, __fdiv_rn(z, __fsub_rn(1.0f, __fmul_rn(z_n, z_n))));
  float prev = __fadd_rn(acc, c0);
  p[0] = prev;
  for (int i = 1; i < n; i++) {
    prev = __fadd_rn(p[i * stride], __fmul_rn(z, prev));
    p[i * stride] = prev;
  }
  prev = __fmul_rn(prev, __fdiv_rn(z, __fsub_rn(z, 1.0f)));
  p[(n - 1) * stride] = prev;
  for (int i = n - 2; i >= 0; i--) {
    prev = __fmul_rn(z, __fsub_rn(prev, p[i * stride]));
    p[i * stride] = prev;
  }
  }  // poles
}

}  // namespace tio

extern "C" int tio_bspline_prefilter(const void* x, float* y, int32_t dtype, int64_t n_bc, const int32_t shape[3], int32_t order,
                                     void* stream) {
  using namespace tio;
  if (x == nullptr || y == nullptr || shape == nullptr) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_bspline_prefilter: null argument");
  if (order < 2 || order > 7) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_bspline_prefilter: order %d (2 ... 7 are implemented)", order);
  if (dtype_size(dtype) == 0) return fail(TIO_ERR_UNSUPPORTED_DTYPE, "tio_bspline_prefilter: dtype %d", dtype);
  if (n_bc < 0 || shape[0] < 1 || shape[1] < 1 || shape[2] < 1) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_bspline_prefilter: bad shape");
  if (n_bc == 0) return TIO_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int I = shape[0], J = shape[1], K = shape[2];
  const int64_t n = n_bc * I * J * K;
  const unsigned blocks = static_cast<unsigned>((n + 255) / 256);
  switch (dtype) {
#define TIO_CASE(DT) case DT: hipLaunchKernelGGL((bspline_load_kernel<DT>), dim3(blocks), dim3(256), 0, s, x, y, n); break;
    TIO_CASE(TIO_F32) TIO_CASE(TIO_F64) TIO_CASE(TIO_F16) TIO_CASE(TIO_BF16) TIO_CASE(TIO_U8) TIO_CASE(TIO_I8) TIO_CASE(TIO_I16)
    TIO_CASE(TIO_I32) TIO_CASE(TIO_I64)
#undef TIO_CASE
    default: return fail(TIO_ERR_UNSUPPORTED_DTYPE, "tio_bspline_prefilter: dtype %d", dtype);
  }
  for (int axis = 0; axis < 3; axis++) {
    const int64_t lines = n / shape[axis];
    hipLaunchKernelGGL(bspline_axis_kernel, dim3(static_cast<unsigned>((lines + 255) / 256)), dim3(256), 0, s, y, lines, I, J, K, axis, order);
  }
  return check_launch("tio_bspline_prefilter");
}
