// mt19937.hip — torch's CPU `randn` stream, drawn on the device from a host-made plan (tio_mt19937_randn_device).
//
// Reference: `torch.randn(data.shape, generator=cpu_gen)` of Noise (transforms/intensity/noise.py:108-116, 166-178).  The
// stream is at::mt19937 -> 24-bit uniforms -> normal_fill_16_AVX2 (ATen/native/cpu/DistributionKernels.cpp) with
// avx_mathfun.h's log256_ps / sincos256_ps.  host_rng.cpp restates it for the host cores and is pinned against
// torch.randn; this file is the same float32 operation sequence for the device — every multiply-add below is fused exactly
// where host_rng.cpp (i.e. the compiler of the torch build) fuses it, and nothing else is (-ffp-contract=off) — so the
// two produce the same bits (tests/test_gpu_device_rng.py compares them, and torch.randn itself).
//
// The only sequential part of the stream, the chain of state twists, stays on ONE host core (tio_host_mt19937_plan):
// it leaves a snapshot of the 624-word state every 128 blocks.  Here one workgroup per snapshot replays its 128 twists
// with the state in LDS — a twist is three data-parallel segments (new[i] needs old[i], old[i + 1] and the word 227
// places back, which is new from the second segment on) — and turns each block into 624 normals where it lands in the
// output.  HBM traffic: 4 bytes written per draw; the plan is 2.5 KB per 79 872 draws.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tio_hip.h"
#include "common.hpp"

namespace tio {
namespace {

constexpr int kN = 624, kM = 397;
constexpr int64_t kPlanUnitBlocks = 128;
constexpr int64_t kPlanHeader = 16, kPlanTail = kPlanHeader + kN, kPlanSnapshots = kPlanTail + 16;
constexpr uint32_t kPlanMagic = 0x4D54504Cu;
constexpr int kThreads = 256;  // (320 — one thread per pair of a state block — was measured: five waves per block no longer fit
                               //  seven blocks on a CU, the 1 681 blocks of a bench batch need a second round: 0.41 -> 0.57 ms)

__device__ __forceinline__ uint32_t twist_word(uint32_t a, uint32_t b, uint32_t c) {
  const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
  return c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

__device__ __forceinline__ uint32_t temper(uint32_t y) {
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

// avx_mathfun.h log256_ps as host_rng.cpp's log_ps states it
__device__ __forceinline__ float log_ps(float x) {
  x = x < __uint_as_float(0x00800000u) ? __uint_as_float(0x00800000u) : x;
  int32_t imm0 = static_cast<int32_t>(__float_as_uint(x) >> 23);
  x = __uint_as_float((__float_as_uint(x) & ~0x7f800000u) | __float_as_uint(0.5f));
  imm0 -= 0x7f;
  float e = __fadd_rn(static_cast<float>(imm0), 1.0f);
  const bool below = x < 0.707106781186547524f;
  const float tmp = below ? x : 0.0f;
  x = __fsub_rn(x, 1.0f);
  e = __fsub_rn(e, below ? 1.0f : 0.0f);
  x = __fadd_rn(x, tmp);
  const float z = __fmul_rn(x, x);
  float y = 7.0376836292E-2f;
  y = __builtin_fmaf(y, x, -1.1514610310E-1f);
  y = __builtin_fmaf(y, x, 1.1676998740E-1f);
  y = __builtin_fmaf(y, x, -1.2420140846E-1f);
  y = __builtin_fmaf(y, x, 1.4249322787E-1f);
  y = __builtin_fmaf(y, x, -1.6668057665E-1f);
  y = __builtin_fmaf(y, x, 2.0000714765E-1f);
  y = __builtin_fmaf(y, x, -2.4999993993E-1f);
  y = __builtin_fmaf(y, x, 3.3333331174E-1f);
  y = __fmul_rn(y, x);
  y = __builtin_fmaf(y, z, __fmul_rn(e, -2.12194440e-4f));
  y = __builtin_fmaf(-z, 0.5f, y);
  x = __fadd_rn(x, y);
  return __builtin_fmaf(e, 0.693359375f, x);
}

// avx_mathfun.h sincos256_ps as host_rng.cpp's sincos_ps states it
__device__ __forceinline__ void sincos_ps(float x, float& s, float& c) {
  uint32_t sign_sin = __float_as_uint(x) & 0x80000000u;
  x = __uint_as_float(__float_as_uint(x) & 0x7fffffffu);
  float y = __fmul_rn(x, 1.27323954473516f);
  int32_t j = static_cast<int32_t>(y);  // (truncation, like _mm256_cvttps_epi32; y >= 0 and far below 2^31)
  j = (j + 1) & ~1;
  y = static_cast<float>(j);
  const uint32_t swap_sin = static_cast<uint32_t>(j & 4) << 29;
  const bool poly = (j & 2) == 0;
  x = __builtin_fmaf(y, -0.78515625f, x);
  x = __builtin_fmaf(y, -2.4187564849853515625e-4f, x);
  x = __builtin_fmaf(y, -3.77489497744594108e-8f, x);
  const uint32_t sign_cos = static_cast<uint32_t>(~(j - 2) & 4) << 29;
  sign_sin ^= swap_sin;
  const float z = __fmul_rn(x, x);
  float yc = 2.443315711809948E-005f;
  yc = __builtin_fmaf(yc, z, -1.388731625493765E-003f);
  yc = __builtin_fmaf(yc, z, 4.166664568298827E-002f);
  yc = __fmul_rn(yc, z);
  yc = __builtin_fmaf(yc, z, -__fmul_rn(z, 0.5f));
  yc = __fadd_rn(yc, 1.0f);
  float ys = -1.9515295891E-4f;
  ys = __builtin_fmaf(ys, z, 8.3321608736E-3f);
  ys = __builtin_fmaf(ys, z, -1.6666654611E-1f);
  ys = __fmul_rn(ys, z);
  ys = __builtin_fmaf(ys, x, x);
  const float ysin2 = poly ? ys : 0.0f, ysin1 = poly ? 0.0f : yc;
  ys = __fsub_rn(ys, ysin2);
  yc = __fsub_rn(yc, ysin1);
  s = __uint_as_float(__float_as_uint(__fadd_rn(ysin1, ysin2)) ^ sign_sin);
  c = __uint_as_float(__float_as_uint(__fadd_rn(yc, ys)) ^ sign_cos);
}

// lanes i and i + 8 of one group of 16 raw words -> the two normals normal_fill_16_AVX2 leaves there
__device__ __forceinline__ void normal_pair(uint32_t w1, uint32_t w2, float& at_i, float& at_i8) {
  const float u1 = __fsub_rn(1.0f, __fmul_rn(static_cast<float>(temper(w1) & 0xFFFFFFu), 1.0f / 16777216.0f));
  const float u2 = __fmul_rn(static_cast<float>(temper(w2) & 0xFFFFFFu), 1.0f / 16777216.0f);
  const float radius = __builtin_sqrtf(__fmul_rn(-2.0f, log_ps(u1)));  // (correctly rounded: __fsqrt_rn is the NATIVE square root in this toolchain)
  const float theta = __fmul_rn(6.283185307179586476925286766559f, u2);
  float s, c;
  sincos_ps(theta, s, c);
  at_i = __builtin_fmaf(__fmul_rn(radius, c), 1.0f, 0.0f);  // fmadd(n, std = 1, mean = 0): n, except that -0 becomes +0
  at_i8 = __builtin_fmaf(__fmul_rn(radius, s), 1.0f, 0.0f);
}

// What the draws are added to when the kernel is Noise itself (tio_mt19937_add_noise_device): out = x + (mean + std z), the
// three roundings of tio_add_noise / noise.py:178, :119; mean and std per batch element of n_per_element values, or scalars
struct NoiseTarget {
  const float* x;
  const float* mean_b;
  const float* std_b;
  float mean, std;
  int64_t n_per_element;
};

// mean / std of the (at most two: n_per_element >= 624 is the host's condition) batch elements a run of <= 624 values
// starting at `first` falls in: one division per run, not per value
struct NoiseRun {
  float mean[2], std[2];
  int64_t boundary;  // first index of the second element
};

__device__ __forceinline__ NoiseRun noise_run(const NoiseTarget& t, int64_t first, int64_t n) {
  NoiseRun r;
  r.mean[0] = r.mean[1] = t.mean; r.std[0] = r.std[1] = t.std;
  r.boundary = INT64_MAX;
  if (t.mean_b != nullptr || t.std_b != nullptr) {
    const int64_t b = first / t.n_per_element;
    const int64_t b1 = min(b + 1, (n - 1) / t.n_per_element);
    r.boundary = (b + 1) * t.n_per_element;
    if (t.mean_b != nullptr) { r.mean[0] = t.mean_b[b]; r.mean[1] = t.mean_b[b1]; }
    if (t.std_b != nullptr) { r.std[0] = t.std_b[b]; r.std[1] = t.std_b[b1]; }
  }
  return r;
}

__device__ __forceinline__ float noisy(const NoiseTarget& t, const NoiseRun& r, int64_t index, float z) {
  const bool second = index >= r.boundary;
  return __fadd_rn(t.x[index], __fadd_rn(second ? r.mean[1] : r.mean[0], __fmul_rn(second ? r.std[1] : r.std[0], z)));
}

// `groups` complete groups of 16 raw words (LDS or global) -> normals at out[first .. first + 16 groups)
template <bool ADD, typename Words>
__device__ __forceinline__ void emit_groups(Words words, int groups, float* __restrict__ out, int64_t first, const NoiseTarget& target,
                                            const NoiseRun& run, int tid) {
  for (int p = tid; p < groups * 8; p += kThreads) {
    const int at = (p >> 3) * 16 + (p & 7);
    float a, b;
    normal_pair(words[at], words[at + 8], a, b);
    if constexpr (ADD) {
      a = noisy(target, run, first + at, a);
      b = noisy(target, run, first + at + 8, b);
    }
    out[first + at] = a;
    out[first + at + 8] = b;
  }
}

// Block 0: the rest of the state block the stream stood in.  Block u + 1: unit u — kPlanUnitBlocks twists from its snapshot.
// (seven waves per SIMD = seven blocks per CU: the 1 681 blocks of a bench batch then run in ONE round on 256 CUs — at 73
// registers, one more than that allows, the launch took a second round for its last 145 blocks: 0.41 -> 0.47 ms)
template <bool ADD>
__global__ __launch_bounds__(kThreads, 7) void mt19937_randn_kernel(const uint32_t* __restrict__ plan, float* __restrict__ out, const NoiseTarget target) {
  constexpr int kRing = 5, kChunk = 4;  // state buffers; state blocks turned into draws together (see the loop)
  __shared__ uint32_t s_state[kRing][kN];
  const int tid = threadIdx.x;
  const int64_t head = plan[1];
  const int64_t total_blocks = static_cast<int64_t>(plan[2]) | (static_cast<int64_t>(plan[3]) << 32);
  const int64_t n = static_cast<int64_t>(plan[6]) | (static_cast<int64_t>(plan[7]) << 32);
  const int64_t n_full = n & ~static_cast<int64_t>(15);  // normal_fill transforms i < size - 15; the tail is the caller's copy
  if (blockIdx.x == 0) {
    NoiseRun run{};
    if constexpr (ADD) run = noise_run(target, 0, n);
    emit_groups<ADD>(plan + kPlanHeader, static_cast<int>(head / 16), out, 0, target, run, tid);
    return;
  }
  const int64_t unit = static_cast<int64_t>(blockIdx.x) - 1;
  const uint32_t* snapshot = plan + kPlanSnapshots + unit * kN;
  for (int i = tid; i < kN; i += kThreads) s_state[0][i] = snapshot[i];
  __syncthreads();
  const int64_t b_end = min((unit + 1) * kPlanUnitBlocks, total_blocks);
  // the parameters of the (at most two) batch elements this unit's 79 872 values fall in, fetched ONCE when elements are
  // at least that long (inside the loop the fetch is a memory round trip per state block)
  const bool run_per_unit = target.n_per_element >= kPlanUnitBlocks * kN;
  NoiseRun run{};
  if constexpr (ADD) {
    if (run_per_unit) run = noise_run(target, head + unit * kPlanUnitBlocks * kN, n);
  }
  // One twist, ONE barrier (round 6; three until then): the third operand of new[j] is new[j - 227] — lane i makes new[i],
  // new[i + 227], new[i + 454] one after the other and has it in a register; everything else it reads is OLD.  The one word
  // that wraps, new[623] = f(old[623], NEW[0], new[396]), is lane 169's third: it makes new[0] again for itself.
  auto twist = [&](const uint32_t* o, uint32_t* w) {
    if (tid < kN - kM) {
      constexpr int kS = kN - kM;  // 227
      const uint32_t n0 = twist_word(o[tid], o[tid + 1], o[tid + kM]);
      const uint32_t n1 = twist_word(o[tid + kS], o[tid + kS + 1], n0);
      w[tid] = n0; w[tid + kS] = n1;
      if (tid < kN - 1 - 2 * kS) w[tid + 2 * kS] = twist_word(o[tid + 2 * kS], o[tid + 2 * kS + 1], n1);
      else if (tid == kN - 1 - 2 * kS) w[kN - 1] = twist_word(o[kN - 1], twist_word(o[0], o[1], o[kM]), n1);
    }
    __syncthreads();
  };
  int cur = 0;
  int64_t b = unit * kPlanUnitBlocks;
  // FOUR state blocks at a time (round 6): a block is 312 pairs of draws — 1.22 rounds of the 256 threads, i.e. two rounds the
  // second of which keeps 56 lanes busy (61 % of the lanes over the kernel's arithmetic, which is all it does: ~135 vector
  // instructions per pair); four blocks are 1 248 pairs = 4.875 rounds, five rounds at 97.5 %.  The twists run ahead through a ring
  // of five state buffers (the next chunk's first twist writes the one buffer this chunk's draws do not read; its barrier then
  // orders every later write behind them).  Whole blocks of whole groups only, and parameters fetched per unit: the rest — a
  // ragged last block, elements shorter than a unit — goes block by block as before.
  while (b + kChunk <= b_end && head + (b + kChunk) * kN <= n_full && (!ADD || run_per_unit)) {
#pragma unroll
    for (int q = 0; q < kChunk; q++) {
      const int from = cur;
      cur = cur + 1 == kRing ? 0 : cur + 1;
      twist(s_state[from], s_state[cur]);
    }
    // (cur is the LAST of the four new blocks: block q of the chunk sits (kChunk - 1 - q) buffers back)
    for (int p = tid; p < kChunk * (kN / 2); p += kThreads) {
      const int q = (p >= kN / 2) + (p >= kN) + (p >= 3 * (kN / 2));
      const int pp = p - q * (kN / 2);
      const int at = (pp >> 3) * 16 + (pp & 7);
      int buf = cur - (kChunk - 1 - q);
      buf = buf < 0 ? buf + kRing : buf;
      const uint32_t* words = s_state[buf];
      const int64_t first = head + (b + q) * kN;
      float x, y;
      normal_pair(words[at], words[at + 8], x, y);
      if constexpr (ADD) {
        x = noisy(target, run, first + at, x);
        y = noisy(target, run, first + at + 8, y);
      }
      out[first + at] = x;
      out[first + at + 8] = y;
    }
    b += kChunk;
  }
  for (; b < b_end; b++) {
    const int from = cur;
    cur = cur + 1 == kRing ? 0 : cur + 1;
    twist(s_state[from], s_state[cur]);
    const uint32_t* w = s_state[cur];
    const int64_t at = head + b * kN;                 // first output index of this block
    const int64_t count = min(static_cast<int64_t>(kN), n - at);
    const int whole = static_cast<int>((min(at + count, n_full) - at) / 16);
    if constexpr (ADD) {
      if (!run_per_unit) run = noise_run(target, at, n);
    }
    if (whole > 0) emit_groups<ADD>(w, whole, out, at, target, run, tid);
    // (no barrier here: the next twist only READS the buffer these groups read, and writes another one)
  }
}

// torch's tail rule under ADD: the last 16 values of the stream (made by the plan) through the same sum
__global__ __launch_bounds__(64) void mt19937_tail_kernel(const uint32_t* __restrict__ plan, float* __restrict__ out, const NoiseTarget target, int64_t n) {
  const int t = threadIdx.x;
  if (t < 16) out[n - 16 + t] = noisy(target, noise_run(target, n - 16, n), n - 16 + t, __uint_as_float(plan[kPlanTail + t]));
}

// ---- the plan's snapshots, made on the device (round 6) -------------------------------------------------------------------
// Snapshot u is the state kPlanUnitBlocks u twists after snapshot 0.  One WORD step of mt19937 — x[k + 624] = x[k + 397] ^
// A(x[k], x[k + 1]) — is a linear map f of the state over GF(2), the state J word steps on is g(f) s with g = x^J mod the
// characteristic polynomial (host_rng_jump.cpp; degree < 19937), and f^k(s) is nothing but the WINDOW of the generator's own word
// sequence at offset k.  So a jump is a correlation of the polynomial's bits with that sequence:
//     out[m] = XOR over the set coefficients k of g of x[k + m],      m = 0 .. 623,
// with x[0 .. 19937 + 624) made once per workgroup from snapshot 0 (33 twists) and kept in LDS (82 KB).  One workgroup per
// SEGMENT of the chain: it jumps to its segment's first state (the host caches the segment polynomials per segment length and
// uploads them once: 2.5 KB each), then twists through its segment and leaves the snapshots.  ~0.1 ms for the 215 k twists of a
// bench batch on ~210 CUs, one workgroup each — beside whatever else the device runs — instead of 0.6 ms on 32 host threads
// (5 ms on the 15 threads a rank gets when eight ranks share a host).
// (The jumped state's first word carries 31 low bits that are not part of the generator's state: the twist never reads
// them, and a snapshot's own words are never turned into draws.)
constexpr int kJumpWords = 19937 + kN + 31;  // words of the sequence a jump can read (rounded up below)
constexpr int kSeqBlocks = (kJumpWords + kN - 1) / kN;  // 33 state blocks
constexpr int kSnapThreads = 960;  // fifteen waves share a jump polynomial's words; the chain behind the jump keeps the first four

__global__ __launch_bounds__(kSnapThreads, 1) void mt19937_snapshots_kernel(uint32_t* __restrict__ plan, const uint32_t* __restrict__ polys,
                                                                            int64_t total_blocks, int64_t segment_blocks) {
  extern __shared__ __attribute__((aligned(16))) uint32_t s_seq[];  // the word sequence from snapshot 0: kSeqBlocks x 624 words (+ 16 of slack), then two state buffers
  uint32_t* s_state = s_seq + kSeqBlocks * kN + 16;
  const int tid = threadIdx.x;
  const int64_t segment = blockIdx.x;
  uint32_t* snapshots = plan + kPlanSnapshots;
  const int64_t b_begin = segment * segment_blocks, b_end = min(b_begin + segment_blocks, total_blocks);
  if (b_begin >= b_end) return;
  if (segment == 0) {
    for (int i = tid; i < kN; i += kSnapThreads) s_state[i] = snapshots[i];
  } else {
    for (int i = tid; i < kN; i += kSnapThreads) s_seq[i] = snapshots[i];
    if (tid < 16) s_seq[kSeqBlocks * kN + tid] = 0u;  // (slack the last window's 16-byte reads run into; multiplied by unset bits only)
    __syncthreads();
    // x[i + 624] = step(x[i], x[i + 1], x[i + 397]): 227 words at a time read only words that exist
    for (int base = 0; base < (kSeqBlocks - 1) * kN; base += kN - kM) {
      const int i = base + tid;
      if (tid < kN - kM && i < (kSeqBlocks - 1) * kN) s_seq[i + kN] = twist_word(s_seq[i], s_seq[i + 1], s_seq[i + kM]);
      __syncthreads();
    }
    // The correlation.  A lane owns FOUR consecutive words m = 4 l .. 4 l + 3 of the jumped state (156 lanes); for the 32
    // coefficients of one polynomial word it needs x[32 w + m .. 32 w + m + 34]: nine aligned 16-byte reads into registers,
    // then one xor per (set bit, word) — the bits are uniform per wave, the 32 tests are scalar branches.  Five groups of lanes
    // share the polynomial's words (w mod 5) and meet through LDS.
    // The correlation.  A lane owns TWELVE consecutive words m = 12 l .. 12 l + 11 of the jumped state: 52 lanes of ONE wave
    // cover all 624, so a coefficient is tested once per polynomial word, not once per wave that shares the outputs — the
    // tests are scalar instructions and branches, and the first builds of this kernel, with four words per lane and three
    // waves per polynomial word, spent their time on the CU's scalar unit (0.2 ms per jump; this form: see profiles/r06_noise_plan.md).
    // For the 32 coefficients of one polynomial word the lane needs x[32 w + m .. 32 w + m + 42]: eleven aligned 16-byte reads
    // into registers, then one xor per (set bit, word).  The waves of the block share the polynomial's words (w mod waves) and
    // meet through LDS.
    constexpr int kWaves = kSnapThreads / 64, kOwn = 12;
    const int wave = tid >> 6, l = tid & 63;
    const bool busy = l < kN / kOwn;
    uint32_t* s_poly = s_state;  // (the state buffers are free until the jump is done: the polynomial, fetched once, coalesced)
    for (int i = tid; i < kN; i += kSnapThreads) s_poly[i] = polys[(segment - 1) * kN + i];
    __syncthreads();
    uint32_t acc[kOwn];
#pragma unroll
    for (int e = 0; e < kOwn; e++) acc[e] = 0u;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    if (busy) {
      uint32_t next_bits = __builtin_amdgcn_readfirstlane(s_poly[wave]);
      for (int w = wave; w < kN; w += kWaves) {
        const uint32_t bits = next_bits;
        if (w + kWaves < kN) next_bits = __builtin_amdgcn_readfirstlane(s_poly[w + kWaves]);
        if (bits == 0u) continue;
        uint32_t r[44];
        const u32x4* src = reinterpret_cast<const u32x4*>(s_seq + 32 * w + kOwn * l);
#pragma unroll
        for (int q = 0; q < 11; q++) {
          const u32x4 v = src[q];
          r[4 * q] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int j = 0; j < 32; j++) {
          if ((bits >> j) & 1u) {
#pragma unroll
            for (int e = 0; e < kOwn; e++) acc[e] ^= r[j + e];
          }
        }
      }
    }
    __syncthreads();  // (everybody is done with the sequence: its first words become the exchange buffer)
    if (busy) {
#pragma unroll
      for (int e = 0; e < kOwn; e++) s_seq[wave * kN + kOwn * l + e] = acc[e];
    }
    __syncthreads();
    for (int i = tid; i < kN; i += kSnapThreads) {
      uint32_t v = 0u;
#pragma unroll
      for (int g = 0; g < kWaves; g++) v ^= s_seq[g * kN + i];
      s_state[i] = v;
    }
  }
  __syncthreads();
  if (tid >= 256) return;  // (the chain is 227 lanes' work: a barrier of four waves, not fifteen — finished waves do not count)
  __syncthreads();
  int cur = 0;
  for (int64_t b = b_begin; b < b_end; b++) {
    const uint32_t* o = s_state + cur * kN;
    uint32_t* w = s_state + (cur ^ 1) * kN;
    if (b % kPlanUnitBlocks == 0 && b != 0) {  // (snapshot 0 is the host's)
      uint32_t* dst = snapshots + (b / kPlanUnitBlocks) * kN;
      for (int i = tid; i < kN; i += 256) dst[i] = o[i];
    }
    // One twist, ONE barrier: the third operand of new[j] is new[j - 227] — lane i makes new[i], new[i + 227], new[i + 454] one
    // after the other and has it in a register; everything else it reads is OLD (new[i] needs old[i + 1], old[i + 397]; the
    // in-place form reads old[j + 1] too).  The one word that wraps, new[623] = f(old[623], NEW[0], new[396]), is lane 169's
    // third (169 + 454 = 623, 169 + 227 = 396): it makes new[0] again for itself from three old words.
    if (tid < kN - kM) {
      constexpr int kS = kN - kM;  // 227
      const uint32_t n0 = twist_word(o[tid], o[tid + 1], o[tid + kM]);
      const uint32_t n1 = twist_word(o[tid + kS], o[tid + kS + 1], n0);
      w[tid] = n0; w[tid + kS] = n1;
      if (tid < kN - 1 - 2 * kS) w[tid + 2 * kS] = twist_word(o[tid + 2 * kS], o[tid + 2 * kS + 1], n1);
      else if (tid == kN - 1 - 2 * kS) w[kN - 1] = twist_word(o[kN - 1], twist_word(o[0], o[1], o[kM]), n1);
    }
    __syncthreads();
    cur ^= 1;
  }
}

}  // namespace
}  // namespace tio

// plan_dev: a plan whose prefix (header, head words, snapshot 0: tio_host_mt19937_plan_prefix) has been uploaded; the other
// snapshots are written here.  segment_blocks: a multiple of 128; polys_dev: the polynomials of segments 1 .. n_segments - 1
// (tio_host_mt19937_segment_polynomials), n_segments = ceil(total_blocks / segment_blocks).
extern "C" int tio_mt19937_device_snapshots(uint32_t* plan_dev, int64_t total_blocks, int64_t segment_blocks, const uint32_t* polys_dev, void* stream) {
  using namespace tio;
  if (plan_dev == nullptr || total_blocks < 1 || segment_blocks < kPlanUnitBlocks || segment_blocks % kPlanUnitBlocks != 0)
    return fail(TIO_ERR_INVALID_ARGUMENT, "tio_mt19937_device_snapshots: bad argument");
  const int64_t n_segments = (total_blocks + segment_blocks - 1) / segment_blocks;
  if (n_segments > 1 && polys_dev == nullptr) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_mt19937_device_snapshots: no segment polynomials");
  if (n_segments > (1 << 20)) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_mt19937_device_snapshots: too many segments");
  const size_t lds = ((static_cast<size_t>(kSeqBlocks) + 2) * kN + 16) * sizeof(uint32_t);
  static_assert((kSeqBlocks - 1) * kN + kN >= 19937 + kN, "the sequence must cover every window a jump reads");
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(mt19937_snapshots_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess)
    return fail(TIO_ERR_LAUNCH, "tio_mt19937_device_snapshots: cannot reserve %zu bytes of LDS", lds);
  hipLaunchKernelGGL(mt19937_snapshots_kernel, dim3(static_cast<unsigned>(n_segments)), dim3(kSnapThreads), lds, static_cast<hipStream_t>(stream), plan_dev, polys_dev,
                     total_blocks, segment_blocks);
  return check_launch("tio_mt19937_device_snapshots");
}

extern "C" int tio_mt19937_add_noise_device(const uint32_t* plan_host, const uint32_t* plan_dev, const float* x_dev, float* out_dev,
                                            int64_t n_per_element, float mean, float std, const float* mean_dev, const float* std_dev,
                                            void* stream) {
  using namespace tio;
  if (plan_host == nullptr || plan_dev == nullptr || x_dev == nullptr || out_dev == nullptr || n_per_element < 1)
    return fail(TIO_ERR_INVALID_ARGUMENT, "tio_mt19937_add_noise_device: bad argument");
  if (plan_host[0] != kPlanMagic) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_mt19937_add_noise_device: not a plan of tio_host_mt19937_plan");
  const int64_t n_units = plan_host[4];
  const int64_t n = static_cast<int64_t>(plan_host[6]) | (static_cast<int64_t>(plan_host[7]) << 32);
  if ((mean_dev != nullptr || std_dev != nullptr) && n_per_element < kN)
    return fail(TIO_ERR_UNSUPPORTED_CONFIG, "tio_mt19937_add_noise_device: per-element parameters need elements of at least 624 values");
  if (x_dev == out_dev && plan_host[5] != 0u)  // (the tail rule re-reads x for the last 16 values after the main kernel has written them)
    return fail(TIO_ERR_UNSUPPORTED_CONFIG, "tio_mt19937_add_noise_device: in place only for draw counts that are multiples of 16");
  if (n % n_per_element != 0) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_mt19937_add_noise_device: the plan's %lld draws are not whole elements of %lld", static_cast<long long>(n), static_cast<long long>(n_per_element));
  const NoiseTarget target{x_dev, mean_dev, std_dev, mean, std, n_per_element};
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(mt19937_randn_kernel<true>, dim3(static_cast<unsigned>(n_units + 1)), dim3(kThreads), 0, s, plan_dev, out_dev, target);
  if (plan_host[5] != 0u) hipLaunchKernelGGL(mt19937_tail_kernel, dim3(1), dim3(64), 0, s, plan_dev, out_dev, target, n);
  return check_launch("tio_mt19937_add_noise_device");
}

extern "C" int tio_mt19937_randn_device(const uint32_t* plan_host, const uint32_t* plan_dev, float* out_dev, void* stream) {
  using namespace tio;
  if (plan_host == nullptr || plan_dev == nullptr || out_dev == nullptr) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_mt19937_randn_device: null pointer");
  if (plan_host[0] != kPlanMagic) return fail(TIO_ERR_INVALID_ARGUMENT, "tio_mt19937_randn_device: not a plan of tio_host_mt19937_plan");
  const int64_t n_units = plan_host[4];
  const int64_t n = static_cast<int64_t>(plan_host[6]) | (static_cast<int64_t>(plan_host[7]) << 32);
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(mt19937_randn_kernel<false>, dim3(static_cast<unsigned>(n_units + 1)), dim3(kThreads), 0, s, plan_dev, out_dev, NoiseTarget{});
  if (plan_host[5] != 0u) {  // torch's tail rule: the last 16 values come from 16 fresh draws (made by the plan)
    if (hipMemcpyAsync(out_dev + n - 16, plan_dev + kPlanTail, 16 * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess)
      return fail(TIO_ERR_LAUNCH, "tio_mt19937_randn_device: cannot place the tail draws");
  }
  return check_launch("tio_mt19937_randn_device");
}
