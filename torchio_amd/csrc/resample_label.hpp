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

__device__ __forceinline__ uint64_t label_bits(const void* p, int es, int64_t i) {
  switch (es) {
    case 1: return static_cast<const uint8_t*>(p)[i];
    case 2: return static_cast<const uint16_t*>(p)[i];
    case 4: return static_cast<const uint32_t*>(p)[i];
    default: return static_cast<const uint64_t*>(p)[i];
  }
}

// Everything a voxel needs besides its coordinates (plain scalars: nothing lives in memory,
// so the call to the general path below does not force the caller's arrays into scratch).
struct LabelSite {
  const void* in;
  void* out;
  const double* labels;
  double pad_label;
  int64_t in_base, out_index;
  int dtype, n_labels;
  int J, K;
  float hx, hy, hz;  // input size - 1 per axis
};

// the 8 trilinear taps of ATen's grid_sampler_3d at (x, y, z): weights, clamped element offsets,
// in-bounds bits — the boundary path of the intensity sampler (resample.hip)
__device__ __forceinline__ unsigned label_taps(const LabelSite& s, float x, float y, float z, float (&w)[8], int (&off)[8]) {
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
  const bool bx0 = (x0 >= 0.0f) & (x0 <= s.hx), bx1 = (x1 >= 0.0f) & (x1 <= s.hx);
  const bool by0 = (y0 >= 0.0f) & (y0 <= s.hy), by1 = (y1 >= 0.0f) & (y1 <= s.hy);
  const bool bz0 = (z0 >= 0.0f) & (z0 <= s.hz), bz1 = (z1 >= 0.0f) & (z1 <= s.hz);
  const int ix0 = static_cast<int>(fminf(fmaxf(x0, 0.0f), s.hx)), ix1 = static_cast<int>(fminf(fmaxf(x1, 0.0f), s.hx));
  const int iy0 = static_cast<int>(fminf(fmaxf(y0, 0.0f), s.hy)), iy1 = static_cast<int>(fminf(fmaxf(y1, 0.0f), s.hy));
  const int iz0 = static_cast<int>(fminf(fmaxf(z0, 0.0f), s.hz)), iz1 = static_cast<int>(fminf(fmaxf(z1, 0.0f), s.hz));
  unsigned okbits = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const bool ok = ((k & 1) ? bx1 : bx0) & ((k & 2) ? by1 : by0) & ((k & 4) ? bz1 : bz0);
    off[k] = (((k & 1) ? ix1 : ix0) * s.J + ((k & 2) ? iy1 : iy0)) * s.K + ((k & 4) ? iz1 : iz0);
    okbits |= ok ? (1u << k) : 0u;
  }
  return okbits;
}

__device__ __forceinline__ void label_pv_voxel_general(LabelSite s, float x, float y, float z);

// Fast path: every in-bounds tap carries the same element (the interior of a label region,
// i.e. almost every voxel).  Then exactly one one-hot channel is non-zero, its value is the
// tap-order sum of the in-bounds weights, it wins the argmax, and the channel sum IS that
// value (a single non-zero term survives every level of the cascade unchanged).
template <typename RAW>
__device__ __forceinline__ void label_pv_voxel_sized(const LabelSite& s, float x, float y, float z) {
  float w[8];
  int off[8];
  const unsigned okbits = label_taps(s, x, y, z, w, off);
  const RAW* p = static_cast<const RAW*>(s.in) + s.in_base;
  RAW bits[8];
#pragma unroll
  for (int k = 0; k < 8; k++) bits[k] = p[off[k]];  // offsets are clamped into the volume: all 8 loads issue together
  RAW first = 0;
  bool have = false, uniform = true;
  float value = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const bool ok = (okbits >> k) & 1u;
    first = (ok && !have) ? bits[k] : first;
    have |= ok;
    uniform &= !ok || bits[k] == first;  // (+0.0 / -0.0 float labels differ in bits: they take the general path)
    value = ok ? __fadd_rn(value, w[k]) : value;
  }
  // A wave whose lanes are all uniform leaves through a SCALAR branch: a per-lane `if` around the
  // two-label arithmetic below gets if-converted, and every voxel of a uniform region then pays for it
  // (measured: 0.97 -> 1.26 ms on a uniform 512^3 map).
  if (__builtin_amdgcn_ballot_w64(!uniform) == 0) {
    if (have && value > 0.5f) {
      static_cast<RAW*>(s.out)[s.out_index] = first;
    } else {
      store_pad_label(s.out, s.dtype, s.out_index, s.pad_label);
    }
    return;
  }
  if constexpr (sizeof(RAW) > 2) {  // wide label types keep the register budget of the kernel: general path only
    if (!uniform) {
      label_pv_voxel_general(s, x, y, z);
      return;
    }
  }
  if (!uniform) {
    // Second fast path: exactly TWO labels among the in-bounds taps (the face between two regions, i.e.
    // nearly every non-uniform voxel).  Two channels are non-zero: their values are the tap-order sums
    // of the weights, the argmax is one comparison (ties go to the smaller label = the earlier channel)
    // and the channel sum is the cascade fed with two terms.  Three labels or more, and float labels
    // whose bits differ but compare equal (+0.0 / -0.0), take the general path.
    RAW second = first;
    bool have_second = false;
    int off_first = 0, off_second = 0;
#pragma unroll
    for (int k = 7; k >= 0; k--) {  // descending: the LAST assignment wins, so these end up as the first taps in order
      const bool ok = (okbits >> k) & 1u;
      const bool other = ok && bits[k] != first;
      second = other ? bits[k] : second;
      off_second = other ? off[k] : off_second;
      off_first = (ok && bits[k] == first) ? off[k] : off_first;
      have_second |= other;
    }
    bool two = have_second;
    float value_first = 0.0f, value_second = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const bool ok = (okbits >> k) & 1u;
      const bool is_first = ok && bits[k] == first, is_second = ok && bits[k] == second;
      two &= !ok || is_first || is_second;
      value_first = is_first ? __fadd_rn(value_first, w[k]) : value_first;
      value_second = is_second ? __fadd_rn(value_second, w[k]) : value_second;
    }
    const double key_first = label_key(s.in, s.dtype, s.in_base + off_first);
    const double key_second = label_key(s.in, s.dtype, s.in_base + off_second);
    const bool first_is_lower = key_first < key_second;
    if (!two || !(first_is_lower || key_second < key_first)) {
      label_pv_voxel_general(s, x, y, z);
      return;
    }
    const float value_low = first_is_lower ? value_first : value_second, value_high = first_is_lower ? value_second : value_first;
    const RAW bits_low = first_is_lower ? first : second, bits_high = first_is_lower ? second : first;
    CascadeSum sum;
    sum.init(s.n_labels > 0 ? s.n_labels : 1);
    const bool ranked = s.labels != nullptr && s.n_labels >= 16;
    sum.add(ranked ? label_rank(s.labels, s.n_labels, first_is_lower ? key_first : key_second) : 0, value_low);
    sum.add(ranked ? label_rank(s.labels, s.n_labels, first_is_lower ? key_second : key_first) : 1, value_high);
    if (sum.total() > 0.5f) {
      static_cast<RAW*>(s.out)[s.out_index] = value_high > value_low ? bits_high : bits_low;  // first maximum
    } else {
      store_pad_label(s.out, s.dtype, s.out_index, s.pad_label);
    }
    return;
  }
  if (have && value > 0.5f) {
    static_cast<RAW*>(s.out)[s.out_index] = first;
  } else {
    store_pad_label(s.out, s.dtype, s.out_index, s.pad_label);
  }
}

__device__ __forceinline__ void label_pv_voxel(const LabelSite& s, float x, float y, float z) {
  switch (dtype_size(s.dtype)) {  // launch uniform: one scalar branch, then straight-line loads
    case 1: label_pv_voxel_sized<uint8_t>(s, x, y, z); break;
    case 2: label_pv_voxel_sized<uint16_t>(s, x, y, z); break;
    case 4: label_pv_voxel_sized<uint32_t>(s, x, y, z); break;
    default: label_pv_voxel_sized<uint64_t>(s, x, y, z); break;
  }
}

// General path (mixed neighbourhood).  Everything is indexed statically — the 8 keys, values and
// offsets stay in registers — and the call is inlined into the dedicated label kernel.
__device__ __forceinline__ void label_pv_voxel_general(LabelSite s, float x, float y, float z) {
  float w8[8];
  int off8[8];
  const unsigned okbits = label_taps(s, x, y, z, w8, off8);
  double key[8];
  float value[8];
#pragma unroll
  for (int k = 0; k < 8; k++) key[k] = ((okbits >> k) & 1u) ? label_key(s.in, s.dtype, s.in_base + off8[k]) : 0.0;
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
  bool have_best = false;
  float best_value = 0.0f;
  double best_key = 0.0;
  int best_off = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const bool ok = (okbits >> k) & 1u;
    const bool better = ok && (!have_best || value[k] > best_value || (value[k] == best_value && key[k] < best_key));
    best_value = better ? value[k] : best_value;
    best_key = better ? key[k] : best_key;
    best_off = better ? off8[k] : best_off;
    have_best |= ok;
  }
  // channel sum over the distinct labels in ascending order; positions only matter once the
  // cascade dumps (16 channels or more)
  CascadeSum sum;
  sum.init(s.n_labels > 0 ? s.n_labels : 1);
  const bool ranked = s.labels != nullptr && s.n_labels >= 16;
  double last = 0.0;
  bool have_last = false;
#pragma unroll 1
  for (int round = 0; round < 8; round++) {
    bool found = false;
    double next_key = 0.0;
    float next_value = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const bool candidate = ((okbits >> k) & 1u) && (!have_last || key[k] > last) && (!found || key[k] < next_key);
      next_key = candidate ? key[k] : next_key;
      next_value = candidate ? value[k] : next_value;
      found |= candidate;
    }
    if (!found) break;
    sum.add(ranked ? label_rank(s.labels, s.n_labels, next_key) : round, next_value);
    last = next_key;
    have_last = true;
  }
  const int es = dtype_size(s.dtype);
  if (have_best && sum.total() > 0.5f) {  // the winner's element, bit for bit
    const char* src = static_cast<const char*>(s.in) + (s.in_base + best_off) * es;
    char* dst = static_cast<char*>(s.out) + s.out_index * es;
    for (int e = 0; e < es; e++) dst[e] = src[e];
  } else {
    store_pad_label(s.out, s.dtype, s.out_index, s.pad_label);
  }
}

}  // namespace tio
