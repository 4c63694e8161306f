// kspace.hip — Motion's k-space compositing as ONE real GEMM along the first spatial axis, on MFMA.
//
// Reference (transforms/intensity/motion.py:334-372): `spectrum = fftn(x_0)`; for every motion
// segment s >= 1 the planes [bounds[s], bounds[s+1]) of `spectrum` along the FIRST spatial axis are
// replaced by the same planes of `fftn(x_s)` (x_s = the rigidly moved image); the result is
// `ifftn(spectrum).real`.  Only the first axis is ever masked, so the J and K transforms cancel
// and the whole composite is linear and real:
//
//     out[i, n] = sum_s sum_i' W_s[i', i] * x_s[i', n],
//     W_s[i', i] = (1/I) * sum_{f in slab s} cos(2 pi f (i - i') / I)          (n = j*K + k)
//
// i.e. a (I x S*I) by (S*I x J*K) float32 GEMM per (batch, channel) with circulant band-pass
// blocks — no FFT, no complex temporaries (the FFT route moves (S+1) * 128 MiB complex volumes
// through HBM several times per 256^3 image).  This is the one GEMM-shaped piece of the path, so
// it runs on the matrix cores: `v_mfma_f32_32x32x2_f32` is exact float32 (a k-ordered fmaf chain)
// at 157 TFLOP/s; x_s streams through LDS once, the 768 KiB table stays in L2.
//
// The slabs tile k-space, so sum_s W_s = 1 and the still image's block never has to be multiplied:
//
//     out = x_0 + sum_{s >= 1} W_s (x_s - x_0)
//
// which drops 1 / (S + 1) of the GEMM (a third at the default two motion events), makes zero motion
// return the image exactly, and costs one extra read of x_0 per event (the difference is formed on the
// way into LDS) plus one as the accumulators' initial value.
//
// Block = 4 waves stacked along the rows: (128 * RM) x 128 output tile, wave = (32 * RM) x 128 =
// RM x 4 accumulators of 32 x 32; K runs in chunks of 16 through double-buffered LDS with the next
// chunk's global loads in flight during the MFMAs (one barrier per chunk).  LDS is k-major for
// both operands, so the MFMA fragment reads (lane l: k = l >> 5, row/col = l & 31) are
// conflict-free 128-byte rows.
#include <math.h>
#include <string.h>

#include "common.hpp"

namespace tio {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct MixArgs {
  const float* seg[TIO_MAX_SEGMENTS];
  const float* mix;       // (n_seg, I, I): mix[s][i'][i]
  void* out;              // (B*C, I, N) in `dtype`
  const uint8_t* active;  // (B) or null
  int n_seg, I, N, dtype, channels;
};

constexpr int kBN = 128;  // output columns per block
constexpr int kKC = 16;   // k per LDS chunk

// Epilogue: the accumulators (which started from the still image) converted to the image dtype.  C/D map of
// the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).  Every store is a
// block-uniform pointer (scalar registers) plus ONE per-lane 32-bit offset that never changes, so no vector
// register is rewritten between stores and they all stay in flight (with per-store address registers the
// compiler reuses them and waits for the previous store each time: 128 serialised round trips per lane).
template <int RM, int DT, bool ALIGNED>
__device__ __forceinline__ void store_tile(const f32x16 (&acc)[RM][4], void* out, int64_t tile_base, int row_base,
                                           int col_base, int lane, int I, int N) {
  using T = typename Elem<DT>::type;
  const int lane_row = 4 * (lane >> 5), lane_col = lane & 31;
  const unsigned lane_off = static_cast<unsigned>(lane_row) * static_cast<unsigned>(N) + static_cast<unsigned>(lane_col);
#pragma unroll
  for (int rm = 0; rm < RM; rm++)
#pragma unroll
    for (int cn = 0; cn < 4; cn++) {
      const bool col_ok = col_base + cn * 32 + lane_col < N;
#pragma unroll
      for (int g = 0; g < 4; g++) {  // groups of four consecutive rows: registers 4g .. 4g+3
        const int row_u = row_base + rm * 32 + 8 * g;  // uniform part of the rows
        T* base = static_cast<T*>(out) + tile_base + static_cast<int64_t>(rm * 32 + 8 * g) * N + cn * 32;
        if constexpr (ALIGNED) {  // I % 4 == 0: a group of four rows is inside or outside as a whole
          if (col_ok && row_u + lane_row < I) {
#pragma unroll
            for (int q = 0; q < 4; q++) Elem<DT>::store(base + static_cast<int64_t>(q) * N, lane_off, acc[rm][cn][4 * g + q]);
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4; q++)
            if (col_ok && row_u + lane_row + q < I) Elem<DT>::store(base + static_cast<int64_t>(q) * N, lane_off, acc[rm][cn][4 * g + q]);
        }
      }
    }
}

template <int RM, bool ALIGNED>
__global__ __launch_bounds__(256, 2) void segment_mix_kernel(MixArgs a) {
  constexpr int BM = 128 * RM;
  constexpr int A_V4 = kKC * BM / 4 / 256;  // float4 per thread of the W chunk (RM)
  __shared__ float As[2][kKC][BM];
  __shared__ float Bs[2][kKC][kBN];

  __shared__ const float* seg_table[TIO_MAX_SEGMENTS];  // dynamic indexing of the by-value argument would go through scratch

  const int bc = blockIdx.z;
  if (a.active != nullptr && a.active[bc / a.channels] == 0) return;  // block-uniform
  if (threadIdx.x < TIO_MAX_SEGMENTS) {
    const float* p = nullptr;
#pragma unroll
    for (int s = 0; s < TIO_MAX_SEGMENTS; s++)
      if (static_cast<int>(threadIdx.x) == s) p = a.seg[s];
    seg_table[threadIdx.x] = p;
  }
  __syncthreads();
  const float* const still = a.seg[0];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = blockIdx.x * kBN, m0 = blockIdx.y * BM;
  const int I = a.I, N = a.N;
  const int chunks_per_seg = (I + kKC - 1) / kKC, total = (a.n_seg - 1) * chunks_per_seg;  // segments 1 .. n_seg-1

  // this thread's slots in the two chunk loads
  const int b_k = tid >> 5, b_c = (tid & 31) * 4;
  constexpr int B_V4 = kKC / 8;  // image float4 per thread: 8 rows of 32 float4 per pass
  float4 ra[A_V4], rb[B_V4], rq[B_V4];  // W chunk, moved rows, still rows (subtracted on the way into LDS, after the MFMAs)

  // ALIGNED (I and N multiples of 4): every float4 is either fully inside or fully outside, so the loads are
  // unconditional from a clamped address and the zeroing happens in stash(), after the MFMAs - a branch
  // around a load makes the compiler drain vmcnt at the join, which serialises the prefetch with the math.
  unsigned ok = 0;  // bit r: W float4 r is real; bit 8 + r: image float4 r is real
  auto fetch = [&](int chunk) {
    const int s = 1 + chunk / chunks_per_seg, kbase = (chunk - (s - 1) * chunks_per_seg) * kKC;
    const float* w = a.mix + static_cast<int64_t>(s) * I * I;
    ok = 0;
#pragma unroll
    for (int r = 0; r < A_V4; r++) {
      const int f = tid + r * 256, kk = f / (BM / 4), col = (f % (BM / 4)) * 4;
      const int ip = kbase + kk, i = m0 + col;
      if constexpr (ALIGNED) {
        ok |= (ip < I && i < I) ? (1u << r) : 0u;
        ra[r] = *reinterpret_cast<const float4*>(w + static_cast<int64_t>(min(ip, I - 1)) * I + min(i, I - 4));
      } else {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ip < I) {
          const float* p = w + static_cast<int64_t>(ip) * I + i;
          if (i < I) v.x = p[0];
          if (i + 1 < I) v.y = p[1];
          if (i + 2 < I) v.z = p[2];
          if (i + 3 < I) v.w = p[3];
        }
        ra[r] = v;
        ok |= 1u << r;
      }
    }
#pragma unroll
    for (int r = 0; r < B_V4; r++) {
      const int ip = kbase + b_k + 8 * r, n = n0 + b_c;
      if constexpr (ALIGNED) {
        ok |= (ip < I && n < N) ? (256u << r) : 0u;
        const int64_t at = (static_cast<int64_t>(bc) * I + min(ip, I - 1)) * N + min(n, N - 4);
        rb[r] = *reinterpret_cast<const float4*>(seg_table[s] + at);
        rq[r] = *reinterpret_cast<const float4*>(still + at);
      } else {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f), x0 = v;
        if (ip < I) {
          const int64_t at = (static_cast<int64_t>(bc) * I + ip) * N + n;
          const float *p = seg_table[s] + at, *q = still + at;
          if (n < N) { v.x = p[0]; x0.x = q[0]; }
          if (n + 1 < N) { v.y = p[1]; x0.y = q[1]; }
          if (n + 2 < N) { v.z = p[2]; x0.z = q[2]; }
          if (n + 3 < N) { v.w = p[3]; x0.w = q[3]; }
        }
        rb[r] = v;
        rq[r] = x0;
        ok |= 256u << r;
      }
    }
  };
  auto stash = [&](int buf) {
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < A_V4; r++) {
      const int f = tid + r * 256, kk = f / (BM / 4), col = (f % (BM / 4)) * 4;
      float4 v = ra[r];
      if (((ok >> r) & 1u) == 0) v = zero;
      *reinterpret_cast<float4*>(&As[buf][kk][col]) = v;
    }
#pragma unroll
    for (int r = 0; r < B_V4; r++) {
      float4 diff = make_float4(rb[r].x - rq[r].x, rb[r].y - rq[r].y, rb[r].z - rq[r].z, rb[r].w - rq[r].w);  // moved - still
      if ((ok & (256u << r)) == 0) diff = zero;
      *reinterpret_cast<float4*>(&Bs[buf][b_k + 8 * r][b_c]) = diff;
    }
  };

  // the accumulators start from the still image (out = x_0 + ...): 128 unconditional loads from clamped
  // addresses, in flight together with the first chunk's prefetch; rows / columns outside are never stored
  const int row0 = m0 + wave * RM * 32 + 4 * (lane >> 5), col0 = n0 + (lane & 31);
  const int64_t out_base = static_cast<int64_t>(bc) * I * N;
  f32x16 acc[RM][4];
#pragma unroll
  for (int rm = 0; rm < RM; rm++)
#pragma unroll
    for (int cn = 0; cn < 4; cn++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = row0 + rm * 32 + (r & 3) + 8 * (r >> 2), col = col0 + cn * 32;
        acc[rm][cn][r] = still[out_base + static_cast<int64_t>(min(row, I - 1)) * N + min(col, N - 1)];
      }

  if (total > 0) {
    fetch(0);
    stash(0);
  }
  __syncthreads();
  // pin the initial values here: with the loads still pending at the loop header the compiler's wait for
  // them lands INSIDE the loop as vmcnt(0), which also drains every iteration's prefetch before the MFMAs
#pragma unroll
  for (int rm = 0; rm < RM; rm++)
#pragma unroll
    for (int cn = 0; cn < 4; cn++) asm volatile("" : "+v"(acc[rm][cn]));
  const int frag_k = lane >> 5, frag_x = lane & 31;
  for (int chunk = 0; chunk < total; chunk++) {
    const int buf = chunk & 1;
    if (chunk + 1 < total) fetch(chunk + 1);  // in flight during the MFMAs below
    // fragments one k-step ahead of the MFMAs that use them: the LDS latency hides under 8 MFMAs
    float af[RM], bf[4];
#pragma unroll
    for (int rm = 0; rm < RM; rm++) af[rm] = As[buf][frag_k][(wave * RM + rm) * 32 + frag_x];
#pragma unroll
    for (int cn = 0; cn < 4; cn++) bf[cn] = Bs[buf][frag_k][cn * 32 + frag_x];
#pragma unroll
    for (int k2 = 0; k2 < kKC; k2 += 2) {
      float an[RM], bn[4];
      if (k2 + 2 < kKC) {
#pragma unroll
        for (int rm = 0; rm < RM; rm++) an[rm] = As[buf][k2 + 2 + frag_k][(wave * RM + rm) * 32 + frag_x];
#pragma unroll
        for (int cn = 0; cn < 4; cn++) bn[cn] = Bs[buf][k2 + 2 + frag_k][cn * 32 + frag_x];
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the reads above the MFMAs
#pragma unroll
      for (int rm = 0; rm < RM; rm++)
#pragma unroll
        for (int cn = 0; cn < 4; cn++)
          acc[rm][cn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[rm], bf[cn], acc[rm][cn], 0, 0, 0);
      if (k2 + 2 < kKC) {
#pragma unroll
        for (int rm = 0; rm < RM; rm++) af[rm] = an[rm];
#pragma unroll
        for (int cn = 0; cn < 4; cn++) bf[cn] = bn[cn];
      }
    }
    if (chunk + 1 < total) stash(buf ^ 1);  // the other buffer: its readers passed the previous barrier
    __syncthreads();
  }

  const int row_base = m0 + wave * RM * 32;  // uniform
  const int64_t tile_base = out_base + static_cast<int64_t>(row_base) * N + n0;
  switch (a.dtype) {  // uniform: one specialised copy of the epilogue runs
    case TIO_F32: store_tile<RM, TIO_F32, ALIGNED>(acc, a.out, tile_base, row_base, n0, lane, I, N); break;
    case TIO_F64: store_tile<RM, TIO_F64, ALIGNED>(acc, a.out, tile_base, row_base, n0, lane, I, N); break;
    case TIO_F16: store_tile<RM, TIO_F16, ALIGNED>(acc, a.out, tile_base, row_base, n0, lane, I, N); break;
    case TIO_BF16: store_tile<RM, TIO_BF16, ALIGNED>(acc, a.out, tile_base, row_base, n0, lane, I, N); break;
    case TIO_U8: store_tile<RM, TIO_U8, ALIGNED>(acc, a.out, tile_base, row_base, n0, lane, I, N); break;
    case TIO_I8: store_tile<RM, TIO_I8, ALIGNED>(acc, a.out, tile_base, row_base, n0, lane, I, N); break;
    case TIO_I16: store_tile<RM, TIO_I16, ALIGNED>(acc, a.out, tile_base, row_base, n0, lane, I, N); break;
    case TIO_I32: store_tile<RM, TIO_I32, ALIGNED>(acc, a.out, tile_base, row_base, n0, lane, I, N); break;
    default: store_tile<RM, TIO_I64, ALIGNED>(acc, a.out, tile_base, row_base, n0, lane, I, N); break;
  }
}

}  // namespace
}  // namespace tio

// W_s[i'][i] in float64, stored as float32: (1/I) * sum_{f = bounds[s]}^{bounds[s+1]-1} cos(2 pi f (i - i') / I).
// The angle is reduced as an exact integer (f * d mod I) before the cosine.
extern "C" int tio_kspace_mix_table(int32_t length, int32_t n_segments, const int32_t* bounds, float* table_host) {
  using namespace tio;
  if (bounds == nullptr || table_host == nullptr) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_kspace_mix_table: null argument");
  if (length < 1 || n_segments < 1) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_kspace_mix_table: bad sizes");
  if (bounds[0] != 0 || bounds[n_segments] != length)
    return fail(TIO_ERR_INVALID_ARGUMENT, "tio_kspace_mix_table: bounds must run from 0 to the axis length");
  for (int s = 0; s < n_segments; s++)
    if (bounds[s + 1] < bounds[s]) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_kspace_mix_table: bounds must not decrease");
  const double step = 2.0 * M_PI / static_cast<double>(length);
  for (int s = 0; s < n_segments; s++) {
    float* block = table_host + static_cast<int64_t>(s) * length * length;
    for (int d = 0; d < length; d++) {  // the circulant's generator: value at (i - i') mod I == d
      double sum = 0.0;
      for (int f = bounds[s]; f < bounds[s + 1]; f++)
        sum += cos(step * static_cast<double>((static_cast<int64_t>(f) * d) % length));
      const float value = static_cast<float>(sum / static_cast<double>(length));
      for (int ip = 0; ip < length; ip++) block[static_cast<int64_t>(ip) * length + (ip + d) % length] = value;
    }
  }
  return TIO_OK;
}

extern "C" int tio_kspace_segment_mix(const void* const* segments, int32_t n_segments, const int32_t* bounds,
                                      const float* mix_dev, void* out, int32_t dtype, int32_t batch, int32_t channels,
                                      const int32_t shape[3], const uint8_t* active_dev, void* stream) {
  using namespace tio;
  if (segments == nullptr || bounds == nullptr || shape == nullptr) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_kspace_segment_mix: null argument");
  if (n_segments < 1 || n_segments > TIO_MAX_SEGMENTS)
    return fail(TIO_ERR_INVALID_ARGUMENT, "tio_kspace_segment_mix: %d segments (1..%d)", n_segments, TIO_MAX_SEGMENTS);
  if (dtype_size(dtype) == 0) return fail(TIO_ERR_UNSUPPORTED_DTYPE, "tio_kspace_segment_mix: dtype %d", dtype);
  if (batch < 0 || channels < 1 || shape[0] < 1 || shape[1] < 1 || shape[2] < 1)
    return fail(TIO_ERR_INVALID_ARGUMENT, "tio_kspace_segment_mix: bad shape");
  if (bounds[0] != 0 || bounds[n_segments] != shape[0])
    return fail(TIO_ERR_INVALID_ARGUMENT, "tio_kspace_segment_mix: bounds must run from 0 to shape[0]");
  for (int s = 0; s < n_segments; s++)
    if (bounds[s + 1] < bounds[s]) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_kspace_segment_mix: bounds must not decrease");
  if (batch == 0) return TIO_OK;
  if (mix_dev == nullptr || out == nullptr) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_kspace_segment_mix: null data");
  const int64_t n = static_cast<int64_t>(shape[1]) * shape[2], bc = static_cast<int64_t>(batch) * channels;
  if (n >= (1ll << 31) || bc > 65535) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_kspace_segment_mix: plane or batch*channels too large");
  MixArgs a{};
  for (int s = 0; s < n_segments; s++) {
    if (segments[s] == nullptr) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_kspace_segment_mix: null segment %d", s);
    if (segments[s] == out) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_kspace_segment_mix: out must not alias a segment");
    a.seg[s] = static_cast<const float*>(segments[s]);
  }
  a.mix = mix_dev; a.out = out; a.active = active_dev;
  a.n_seg = n_segments; a.I = shape[0]; a.N = static_cast<int>(n); a.dtype = dtype; a.channels = channels;
  const bool aligned = (shape[0] % 4 == 0) && (n % 4 == 0);
  const int rm = shape[0] > 128 ? 2 : 1, bm = 128 * rm;
  dim3 grid(static_cast<unsigned>((n + kBN - 1) / kBN), static_cast<unsigned>((shape[0] + bm - 1) / bm), static_cast<unsigned>(bc));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (rm == 2) {
    if (aligned) hipLaunchKernelGGL((segment_mix_kernel<2, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((segment_mix_kernel<2, false>), grid, dim3(256), 0, s, a);
  } else {
    if (aligned) hipLaunchKernelGGL((segment_mix_kernel<1, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((segment_mix_kernel<1, false>), grid, dim3(256), 0, s, a);
  }
  return check_launch("tio_kspace_segment_mix");
}
