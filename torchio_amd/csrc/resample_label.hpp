// "label" partial-volume resampling of ONE output voxel, fused (TIO_LABEL_PV).
//
// Reference: _resample_label_partial_volume, transforms/spatial/spatial.py:1275-1389, the
// C == 1 / antialias=False / one_hot_label_interpolation="linear" pipeline:
//   labels  = torch.unique(data)                        (sorted ascending)
//   one_hot = (data == labels[l]).float()               (B, L, I, J, K)
//   sampled = grid_sample(one_hot, linear, zeros)       fill 0.0 -> no mask step (spatial.py:2075-2076)
//   winner  = labels[sampled.argmax(dim=1)]             first maximum
//   out     = sampled.sum(dim=1) > 0.5 ? winner : default_pad_label
// Here the (B, L, I, J, K) tensor never exists: channel l of a voxel is the sum, in ATen's
// tap order and starting from 0, of the weights of the in-bounds taps whose label is l
// (adding the 0 * w products of the other taps never changes a partial sum), so the 8 taps
// carry everything.  The channel sum reproduces ATen's cascade_sum (multi_row_sum,
// aten/src/ATen/native/cpu/SumKernel.cpp, torch 2.10): accumulator 0 takes the channels in
// order and is dumped into accumulator 1 after every `step` = 16 channels (and so on for
// two more levels); only positions of non-zero channels matter, hence the label table.
#pragma once

#include "common.hpp"

namespace tio {

__device__ __forceinline__ double label_key(const void* p, int dtype, int64_t i) {
  switch (dtype) {
    case TIO_F32: return static_cast<double>(static_cast<const float*>(p)[i]);
    case TIO_F64: return static_cast<const double*>(p)[i];
    case TIO_F16: return static_cast<double>(static_cast<float>(static_cast<const _Float16*>(p)[i]));
    case TIO_BF16: return static_cast<double>(bf16_bits_to_float(static_cast<const uint16_t*>(p)[i]));
    case TIO_U8: return static_cast<double>(static_cast<const uint8_t*>(p)[i]);
    case TIO_I8: return static_cast<double>(static_cast<const int8_t*>(p)[i]);
    case TIO_I16: return static_cast<double>(static_cast<const int16_t*>(p)[i]);
    case TIO_I32: return static_cast<double>(static_cast<const int32_t*>(p)[i]);
    default: return static_cast<double>(static_cast<const int64_t*>(p)[i]);
  }
}

// torch.full_like(<tensor of the image dtype>, default_pad_label): scalar → dtype cast
__device__ __forceinline__ void store_pad_label(void* p, int dtype, int64_t i, double v) {
  switch (dtype) {
    case TIO_F32: static_cast<float*>(p)[i] = static_cast<float>(v); break;
    case TIO_F64: static_cast<double*>(p)[i] = v; break;
    case TIO_F16: static_cast<_Float16*>(p)[i] = static_cast<_Float16>(static_cast<float>(v)); break;
    case TIO_BF16: static_cast<uint16_t*>(p)[i] = float_to_bf16_bits(static_cast<float>(v)); break;
    case TIO_U8: static_cast<uint8_t*>(p)[i] = static_cast<uint8_t>(static_cast<int64_t>(v)); break;
    case TIO_I8: static_cast<int8_t*>(p)[i] = static_cast<int8_t>(static_cast<int64_t>(v)); break;
    case TIO_I16: static_cast<int16_t*>(p)[i] = static_cast<int16_t>(static_cast<int64_t>(v)); break;
    case TIO_I32: static_cast<int32_t*>(p)[i] = static_cast<int32_t>(static_cast<int64_t>(v)); break;
    default: static_cast<int64_t*>(p)[i] = static_cast<int64_t>(v); break;
  }
}

// position of `key` in the ascending table (lower bound)
__device__ __forceinline__ int64_t label_rank(const double* __restrict__ table, int n, double key) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (table[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// ATen's four-level running sum, fed only with the non-zero channels in ascending position.
struct CascadeSum {
  float acc[4];
  int64_t step, step2, step3, full;  // dump periods; `full`: channels covered by whole steps
  int64_t prev;                      // position of the last channel added (-1: none yet)

  __device__ __forceinline__ void init(int64_t n_channels) {
    acc[0] = acc[1] = acc[2] = acc[3] = 0.0f;
    // level_power = max(4, CeilLog2(size) / 4); CeilLog2(x) = x <= 2 ? 1 : 64 - clz(x - 1)
    const int ceil_log2 = n_channels <= 2 ? 1 : 64 - __clzll(static_cast<long long>(n_channels - 1));
    const int power = ceil_log2 / 4 > 4 ? ceil_log2 / 4 : 4;
    step = int64_t{1} << power;
    step2 = step << power;
    step3 = step2 << power;
    full = (n_channels >> power) << power;
    prev = -1;
  }
  // the dumps that happen after channel `prev` and before channel `upto` (counts e, prev < e <= upto)
  __device__ __forceinline__ void dumps_until(int64_t upto) {
    if (prev < 0) return;  // nothing accumulated yet: dumping zeros changes nothing
    const int64_t last = upto < full ? upto : full;
    if (last / step <= prev / step) return;
    acc[1] = __fadd_rn(acc[1], acc[0]); acc[0] = 0.0f;
    if (last / step2 <= prev / step2) return;
    acc[2] = __fadd_rn(acc[2], acc[1]); acc[1] = 0.0f;
    if (last / step3 <= prev / step3) return;
    acc[3] = __fadd_rn(acc[3], acc[2]); acc[2] = 0.0f;
  }
  __device__ __forceinline__ void add(int64_t position, float v) {
    dumps_until(position);
    acc[0] = __fadd_rn(acc[0], v);
    prev = position;
  }
  __device__ __forceinline__ float total() {
    dumps_until(full);
    float t = __fadd_rn(acc[0], acc[1]);
    t = __fadd_rn(t, acc[2]);
    return __fadd_rn(t, acc[3]);
  }
};

// w / off / okbits: the 8 trilinear taps exactly as the boundary path of the intensity
// sampler computes them (weights, clamped element offsets, in-bounds bits).
__device__ __noinline__ void label_pv_voxel(const void* in, void* out, int dtype, const double* labels, int n_labels,
                                            double pad_label, int64_t in_base, int64_t out_index, const float* w8,
                                            const int* off8, unsigned okbits) {
  double key[8];
  float value[8];
#pragma unroll
  for (int k = 0; k < 8; k++) key[k] = ((okbits >> k) & 1u) ? label_key(in, dtype, in_base + off8[k]) : 0.0;
  // channel value of tap k's label: in-bounds weights of equal-label taps, tap order, from 0
#pragma unroll
  for (int k = 0; k < 8; k++) {
    float v = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const bool same = ((okbits >> j) & 1u) && key[j] == key[k];
      v = same ? __fadd_rn(v, w8[j]) : v;
    }
    value[k] = v;
  }
  // argmax over the channels: first maximum = smallest label among the maximal ones
  int best = -1;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if (!((okbits >> k) & 1u)) continue;
    if (best < 0 || value[k] > value[best] || (value[k] == value[best] && key[k] < key[best])) best = k;
  }
  // channel sum over the distinct labels in ascending order
  CascadeSum sum;
  sum.init(n_labels > 0 ? n_labels : 1);
  double last = 0.0;
  bool have_last = false;
  for (int round = 0; round < 8; round++) {
    int next = -1;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (!((okbits >> k) & 1u)) continue;
      if (have_last && !(key[k] > last)) continue;
      if (next < 0 || key[k] < key[next]) next = k;
    }
    if (next < 0) break;
    const int64_t position = (labels != nullptr && n_labels > 0) ? label_rank(labels, n_labels, key[next]) : round;
    sum.add(position, value[next]);
    last = key[next];
    have_last = true;
  }
  const bool in_bounds = best >= 0 && sum.total() > 0.5f;
  const int es = dtype_size(dtype);
  if (in_bounds) {  // the winner's element, bit for bit
    const char* s = static_cast<const char*>(in) + (in_base + off8[best]) * es;
    char* d = static_cast<char*>(out) + out_index * es;
    for (int e = 0; e < es; e++) d[e] = s[e];
  } else {
    store_pad_label(out, dtype, out_index, pad_label);
  }
}

}  // namespace tio
