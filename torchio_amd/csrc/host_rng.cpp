// host_rng.cpp — torch's CPU `randn` stream, reproduced on the host cores (tio_host_mt19937_*).
//
// The reference's Noise draws `torch.randn(data.shape, generator=cpu_gen)` from ONE seeded CPU generator
// (reference transforms/intensity/noise.py:108-116, 166-178).  That stream is what "reference-identical noise" means, and
// torch produces it on one thread: 134 M draws for the bench batch = 0.35 s per step (18 volumes/s, VERDICT r2 missing #2).
// The stream itself is public arithmetic — restated here, bit for bit, and split so that only the part that MUST be
// sequential is:
//   * mt19937 (ATen/core/MT19937RNGEngine.h): the state twist is a chain, but one twist of 624 words is data parallel
//     inside (new[i] needs old[i], old[i+1] and a word 227 places back): ONE thread runs the chain with 8 / 16-lane
//     integer vectors and drops every new state block, untempered, straight into the output buffer;
//   * tempering, the 24-bit uniform (ATen/core/TransformationHelper.h uniform_real: (x & (2^24 - 1)) * 2^-24) and the
//     Box-Muller step on groups of 16 (ATen/native/cpu/DistributionKernels.cpp normal_fill_16_AVX2: u1 = 1 - data[0:8],
//     u2 = data[8:16], radius = sqrt(-2 log u1), theta = 2 pi u2, data[0:8] = radius cos theta, data[8:16] = radius sin
//     theta; log256_ps / sincos256_ps of avx_mathfun.h, with the multiply-adds fused exactly where the compiler of the
//     torch build fuses them — pinned against torch.randn on 10^8 draws, tests/test_host_rng.py) are independent per
//     group: worker threads follow the chain through the buffer and turn the raw words into normals IN PLACE.
// A state block is 624 = 39 x 16 words, so groups never straddle blocks.
//
// Host code only (no device code, no HIP call); part of libtio_hip.so so that the binding loads one library.
#include <immintrin.h>
#include <stdint.h>
#include <string.h>

#include <limits.h>
#include <linux/futex.h>
#include <pthread.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/tio_hip.h"
#include "host_rng_jump.hpp"

namespace {

constexpr int kN = 624, kM = 397;

struct MtState {
  uint32_t s[kN + 16];  // + mirror of the first 16 new words (vector loads across the wrap)
  int32_t pos;          // next unread word of the current block (kN = block exhausted)
  int32_t seeded;
  int64_t owed_blocks;  // twists `s` is still behind (round 6: tio_host_mt19937_plan_prefix — the DEVICE ran the chain; settled by settle() when the words are next needed)
};
static_assert(sizeof(MtState) <= TIO_HOST_MT_STATE_BYTES, "tio_host_mt_state is too small");

void mt_seed(MtState* st, uint32_t seed) {
  st->s[0] = seed;
  for (int j = 1; j < kN; j++) st->s[j] = 1812433253u * (st->s[j - 1] ^ (st->s[j - 1] >> 30)) + static_cast<uint32_t>(j);
  st->pos = kN;
  st->seeded = 1;
  st->owed_blocks = 0;
}

inline uint32_t twist_word(uint32_t a, uint32_t b, uint32_t c) {
  const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
  return c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

void twist_scalar(uint32_t* s) {
  for (int i = 0; i < kN - kM; i++) s[i] = twist_word(s[i], s[i + 1], s[i + kM]);
  for (int i = kN - kM; i < kN - 1; i++) s[i] = twist_word(s[i], s[i + 1], s[i - (kN - kM)]);
  s[kN - 1] = twist_word(s[kN - 1], s[0], s[kM - 1]);
}

__attribute__((target("avx2"))) inline void twist_step_avx2(uint32_t* s, int i, const uint32_t* third) {
  const __m256i upper = _mm256_set1_epi32(static_cast<int>(0x80000000u)), lower = _mm256_set1_epi32(0x7fffffff);
  const __m256i matrix = _mm256_set1_epi32(static_cast<int>(0x9908b0dfu)), one = _mm256_set1_epi32(1);
  const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s + i));
  const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s + i + 1));
  const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(third));
  const __m256i y = _mm256_or_si256(_mm256_and_si256(a, upper), _mm256_and_si256(b, lower));
  const __m256i mag = _mm256_and_si256(_mm256_sub_epi32(_mm256_setzero_si256(), _mm256_and_si256(y, one)), matrix);
  _mm256_storeu_si256(reinterpret_cast<__m256i*>(s + i), _mm256_xor_si256(_mm256_xor_si256(c, _mm256_srli_epi32(y, 1)), mag));
}

__attribute__((target("avx2"))) void twist_avx2(uint32_t* s) {
  twist_step_avx2(s, 0, s + kM);
  memcpy(s + kN, s, 16 * sizeof(uint32_t));  // (only the first 8 are new yet: the next vector refreshes the rest)
  twist_step_avx2(s, 8, s + 8 + kM);
  memcpy(s + kN, s, 16 * sizeof(uint32_t));
  // words 16 .. 231 take their third operand at i + 397 (old words up to 623, then the mirror of the new words 0 .. 12)
  for (int i = 16; i < 232; i += 8) twist_step_avx2(s, i, s + i + kM);
  for (int i = 232; i < kN; i += 8) twist_step_avx2(s, i, s + i - (kN - kM));  // (i = 616 reads s[617 .. 624]: the mirror of new word 0)
}

// (the chain is what bounds the device-side stream — 215 k twists per 134 M draws on ONE core — so the step is written
// for the fewest operations: a bit select and a three-way xor are one vpternlogd each, the conditional xor with the
// matrix constant is a masked xor on the low bit of y)
__attribute__((target("avx512f"))) inline void twist_step_avx512(uint32_t* s, int i, const uint32_t* third) {
  const __m512i lower = _mm512_set1_epi32(0x7fffffff), matrix = _mm512_set1_epi32(static_cast<int>(0x9908b0dfu)), one = _mm512_set1_epi32(1);
  const __m512i a = _mm512_loadu_si512(s + i), b = _mm512_loadu_si512(s + i + 1), c = _mm512_loadu_si512(third);
  const __m512i y = _mm512_ternarylogic_epi32(lower, b, a, 0xCA);  // lower ? b : a, bit by bit
  const __m512i r = _mm512_xor_si512(c, _mm512_srli_epi32(y, 1));
  _mm512_storeu_si512(s + i, _mm512_mask_xor_epi32(r, _mm512_test_epi32_mask(y, one), r, matrix));
}

__attribute__((target("avx512f"))) void twist_avx512(uint32_t* s) {
  twist_step_avx512(s, 0, s + kM);
  memcpy(s + kN, s, 16 * sizeof(uint32_t));
  for (int i = 16; i < 240; i += 16) twist_step_avx512(s, i, s + i + kM);  // (i = 224: old words 621 .. 623, then the mirror of new 0 .. 12)
  for (int i = 240; i < kN; i += 16) twist_step_avx512(s, i, s + i - (kN - kM));
}

// The same twist for a 64-byte aligned state, without a single split load: old[i + 1 ..] and the third operand (397 places
// on = 384 + 13) are assembled from aligned vectors with valignd.  The third operands form ONE sliding sequence — old
// words 384 .. 623, then the NEW words 0, 16, ... — so every step loads two aligned vectors (the next `a`, the next
// third) and keeps the previous ones in registers.
__attribute__((target("avx512f"))) void twist_avx512_aligned(uint32_t* s) {
  const __m512i lower = _mm512_set1_epi32(0x7fffffff), matrix = _mm512_set1_epi32(static_cast<int>(0x9908b0dfu)), one = _mm512_set1_epi32(1);
  __m512i a = _mm512_load_si512(s);            // old words i .. i + 15
  __m512i c_lo = _mm512_load_si512(s + 384);   // third-operand window: its low vector
  __m512i new0 = _mm512_setzero_si512();       // new words 0 .. 15 (stand in for "old" words 624 ..)
  for (int i = 0; i < kN; i += 16) {
    const __m512i a_next = i + 16 < kN ? _mm512_load_si512(s + i + 16) : new0;  // (i = 608: word 624 is new word 0)
    __m512i c_hi;
    if (i < 224) c_hi = _mm512_load_si512(s + i + 400);           // old words
    else if (i == 224) c_hi = new0;                               // old 608 .. 623 | new 0 .. 15
    else c_hi = _mm512_load_si512(s + i - 224);                   // new words (stored by an earlier step)
    const __m512i b = _mm512_alignr_epi32(a_next, a, 1);
    const __m512i c = _mm512_alignr_epi32(c_hi, c_lo, 13);
    const __m512i y = _mm512_ternarylogic_epi32(lower, b, a, 0xCA);
    const __m512i r = _mm512_xor_si512(c, _mm512_srli_epi32(y, 1));
    const __m512i w = _mm512_mask_xor_epi32(r, _mm512_test_epi32_mask(y, one), r, matrix);
    _mm512_store_si512(s + i, w);
    if (i == 0) new0 = w;
    a = a_next;
    c_lo = c_hi;
  }
  _mm512_store_si512(s + kN, new0);  // (the mirror the unaligned forms rely on: kept consistent)
}

void twist(uint32_t* s) {
  static const int level = __builtin_cpu_supports("avx512f") ? 2 : (__builtin_cpu_supports("avx2") ? 1 : 0);
  if (level == 2) twist_avx512(s);
  else if (level == 1) twist_avx2(s);
  else twist_scalar(s);
}

// A few long-lived worker threads for the parallel parts (creating and joining 15 threads per call is ~0.3 ms of a 1.2-ms
// plan).  One job at a time (callers are serialised by the mutex); a forked child (DataLoader workers) starts its own.
// (Linux futexes: one system call wakes every sleeping worker, and nobody queues up behind a mutex on the way out of it.
// Measured on the GPU box's host, the plan of 134 M draws on 27 segments: 0.82 ms with one condition variable for the pool
// — the woken threads take its mutex one after the other —, 0.64 ms with one per worker — the caller's 26 notifications, ~5 us
// each, are then the start of the critical path.)
inline void futex_wait(std::atomic<uint32_t>* word, uint32_t expected) {
  syscall(SYS_futex, reinterpret_cast<uint32_t*>(word), FUTEX_WAIT_PRIVATE, expected, nullptr, nullptr, 0);
}
inline void futex_wake_all(std::atomic<uint32_t>* word) {
  syscall(SYS_futex, reinterpret_cast<uint32_t*>(word), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
}
static_assert(sizeof(std::atomic<uint32_t>) == sizeof(uint32_t), "a futex word");

class Workers {
 public:
  // runs job(t) for t = 0 .. count - 1: t = 0 on the calling thread, the rest on the pool; returns when all are done
  void run(int count, const std::function<void(int)>& job) {
    if (count <= 1) { if (count == 1) job(0); return; }
    std::unique_lock<std::mutex> call(call_mutex_);  // one call at a time
    while (static_cast<int>(threads_.size()) < count - 1 && threads_.size() < 63) {
      const int index = static_cast<int>(threads_.size());
      const uint32_t born = generation_.load();  // (the calls it has nothing to do with)
      threads_.emplace_back(new std::thread([this, index, born] { loop(index, born); }));
    }
    // (more jobs than workers: worker w takes jobs w + 1, w + 1 + W, ...)
    job_ = &job; count_ = count; stride_ = static_cast<int>(threads_.size());
    // EVERY worker answers every call, the ones it has no job for at once: none can be a call behind when the next one starts
    pending_.store(static_cast<uint32_t>(threads_.size()));
    generation_.fetch_add(1);
    futex_wake_all(&generation_);
    job(0);
    for (uint32_t left = pending_.load(); left != 0; left = pending_.load()) futex_wait(&pending_, left);
    job_ = nullptr;
  }

 private:
  void loop(int index, uint32_t seen) {
    for (;;) {
      uint32_t now;
      while ((now = generation_.load()) == seen) futex_wait(&generation_, seen);
      seen = now;  // (job_, count_, stride_ were written before the generation changed)
      for (int t = index + 1; t < count_; t += stride_) (*job_)(t);
      if (pending_.fetch_sub(1) == 1) futex_wake_all(&pending_);  // (the caller sleeps on the count; only zero is worth waking it)
    }
  }
  std::mutex call_mutex_;
  std::vector<std::thread*> threads_;  // (never joined: they live as long as the process)
  const std::function<void(int)>* job_ = nullptr;
  int count_ = 0, stride_ = 1;
  std::atomic<uint32_t> generation_{0}, pending_{0};
};

// One pool per PROCESS: a forked child (DataLoader workers) inherits the parent's object with its mutexes and condition
// variables in whatever state the parent's threads left them — and none of those threads — so it starts a fresh one
// (the old object is leaked; measured: reusing it hangs in the first notify).
Workers* g_workers = nullptr;
std::mutex g_workers_mutex;

Workers& workers() {
  static const int registered = pthread_atfork(nullptr, nullptr, [] {
    g_workers = nullptr;
    new (&g_workers_mutex) std::mutex();  // (the parent may have held it at the fork)
  });
  (void)registered;
  std::lock_guard<std::mutex> lock(g_workers_mutex);
  if (g_workers == nullptr) g_workers = new Workers();  // (leaked on purpose: its threads outlive static destruction)
  return *g_workers;
}

struct alignas(64) AlignedState { uint32_t s[kN + 16]; };

void twist_aligned(uint32_t* s) {  // s: 64-byte aligned, kN + 16 words
  static const bool wide = __builtin_cpu_supports("avx512f");
  if (wide) twist_avx512_aligned(s);
  else twist(s);
}

inline uint32_t temper(uint32_t y) {
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

// ---- one group of 16 raw words -> 16 normals, scalar (the reference for the vector form, and the fallback) ----------------
inline float as_f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline uint32_t as_u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

inline float log_ps(float x) {  // avx_mathfun.h log256_ps, fused as in the torch build
  x = x < as_f(0x00800000u) ? as_f(0x00800000u) : x;
  int32_t imm0 = static_cast<int32_t>(as_u(x) >> 23);
  x = as_f((as_u(x) & ~0x7f800000u) | as_u(0.5f));
  imm0 -= 0x7f;
  float e = static_cast<float>(imm0) + 1.0f;
  const bool below = x < 0.707106781186547524f;
  const float tmp = below ? x : 0.0f;
  x = x - 1.0f;
  e = e - (below ? 1.0f : 0.0f);
  x = x + tmp;
  const float z = x * x;
  float y = 7.0376836292E-2f;
  y = __builtin_fmaf(y, x, -1.1514610310E-1f);
  y = __builtin_fmaf(y, x, 1.1676998740E-1f);
  y = __builtin_fmaf(y, x, -1.2420140846E-1f);
  y = __builtin_fmaf(y, x, 1.4249322787E-1f);
  y = __builtin_fmaf(y, x, -1.6668057665E-1f);
  y = __builtin_fmaf(y, x, 2.0000714765E-1f);
  y = __builtin_fmaf(y, x, -2.4999993993E-1f);
  y = __builtin_fmaf(y, x, 3.3333331174E-1f);
  y = y * x;
  y = __builtin_fmaf(y, z, e * -2.12194440e-4f);
  y = __builtin_fmaf(-z, 0.5f, y);
  x = x + y;
  return __builtin_fmaf(e, 0.693359375f, x);
}

inline void sincos_ps(float x, float* s, float* c) {  // avx_mathfun.h sincos256_ps, fused as in the torch build
  uint32_t sign_sin = as_u(x) & 0x80000000u;
  x = as_f(as_u(x) & 0x7fffffffu);
  float y = x * 1.27323954473516f;
  int32_t j = static_cast<int32_t>(y);
  j = (j + 1) & ~1;
  y = static_cast<float>(j);
  const uint32_t swap_sin = static_cast<uint32_t>(j & 4) << 29;
  const bool poly = (j & 2) == 0;
  x = __builtin_fmaf(y, -0.78515625f, x);
  x = __builtin_fmaf(y, -2.4187564849853515625e-4f, x);
  x = __builtin_fmaf(y, -3.77489497744594108e-8f, x);
  const uint32_t sign_cos = static_cast<uint32_t>(~(j - 2) & 4) << 29;
  sign_sin ^= swap_sin;
  const float z = x * x;
  float yc = 2.443315711809948E-005f;
  yc = __builtin_fmaf(yc, z, -1.388731625493765E-003f);
  yc = __builtin_fmaf(yc, z, 4.166664568298827E-002f);
  yc = yc * z;
  yc = __builtin_fmaf(yc, z, -(z * 0.5f));
  yc = yc + 1.0f;
  float ys = -1.9515295891E-4f;
  ys = __builtin_fmaf(ys, z, 8.3321608736E-3f);
  ys = __builtin_fmaf(ys, z, -1.6666654611E-1f);
  ys = ys * z;
  ys = __builtin_fmaf(ys, x, x);
  const float ysin2 = poly ? ys : 0.0f, ysin1 = poly ? 0.0f : yc;
  ys = ys - ysin2;
  yc = yc - ysin1;
  *s = as_f(as_u(ysin1 + ysin2) ^ sign_sin);
  *c = as_f(as_u(yc + ys) ^ sign_cos);
}

void group16_scalar(uint32_t* words) {  // in place: 16 raw state words -> 16 float32 normals
  float u[16];
  for (int i = 0; i < 16; i++) u[i] = static_cast<float>(temper(words[i]) & 0xFFFFFFu) * (1.0f / 16777216.0f);
  float* out = reinterpret_cast<float*>(words);
  for (int i = 0; i < 8; i++) {
    const float radius = __builtin_sqrtf(-2.0f * log_ps(1.0f - u[i]));
    const float theta = 6.283185307179586476925286766559f * u[i + 8];
    float s, c;
    sincos_ps(theta, &s, &c);
    out[i] = __builtin_fmaf(radius * c, 1.0f, 0.0f);  // fmadd(n, std = 1, mean = 0): n, except that -0 becomes +0 (u1 == 1)
    out[i + 8] = __builtin_fmaf(radius * s, 1.0f, 0.0f);
  }
}

// ---- the same, 8 lanes at a time -------------------------------------------------------------------------------------
__attribute__((target("avx2,fma"))) inline __m256 uniform8(const uint32_t* words) {
  __m256i y = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(words));
  y = _mm256_xor_si256(y, _mm256_srli_epi32(y, 11));
  y = _mm256_xor_si256(y, _mm256_and_si256(_mm256_slli_epi32(y, 7), _mm256_set1_epi32(static_cast<int>(0x9d2c5680u))));
  y = _mm256_xor_si256(y, _mm256_and_si256(_mm256_slli_epi32(y, 15), _mm256_set1_epi32(static_cast<int>(0xefc60000u))));
  y = _mm256_xor_si256(y, _mm256_srli_epi32(y, 18));
  y = _mm256_and_si256(y, _mm256_set1_epi32(0xFFFFFF));
  return _mm256_mul_ps(_mm256_cvtepi32_ps(y), _mm256_set1_ps(1.0f / 16777216.0f));
}

__attribute__((target("avx2,fma"))) inline __m256 log256(__m256 x) {
  const __m256 one = _mm256_set1_ps(1.0f);
  x = _mm256_max_ps(x, _mm256_castsi256_ps(_mm256_set1_epi32(0x00800000)));
  __m256i imm0 = _mm256_srli_epi32(_mm256_castps_si256(x), 23);
  x = _mm256_and_ps(x, _mm256_castsi256_ps(_mm256_set1_epi32(~0x7f800000)));
  x = _mm256_or_ps(x, _mm256_set1_ps(0.5f));
  imm0 = _mm256_sub_epi32(imm0, _mm256_set1_epi32(0x7f));
  __m256 e = _mm256_add_ps(_mm256_cvtepi32_ps(imm0), one);
  const __m256 mask = _mm256_cmp_ps(x, _mm256_set1_ps(0.707106781186547524f), _CMP_LT_OS);
  const __m256 tmp = _mm256_and_ps(x, mask);
  x = _mm256_sub_ps(x, one);
  e = _mm256_sub_ps(e, _mm256_and_ps(one, mask));
  x = _mm256_add_ps(x, tmp);
  const __m256 z = _mm256_mul_ps(x, x);
  __m256 y = _mm256_set1_ps(7.0376836292E-2f);
  y = _mm256_fmadd_ps(y, x, _mm256_set1_ps(-1.1514610310E-1f));
  y = _mm256_fmadd_ps(y, x, _mm256_set1_ps(1.1676998740E-1f));
  y = _mm256_fmadd_ps(y, x, _mm256_set1_ps(-1.2420140846E-1f));
  y = _mm256_fmadd_ps(y, x, _mm256_set1_ps(1.4249322787E-1f));
  y = _mm256_fmadd_ps(y, x, _mm256_set1_ps(-1.6668057665E-1f));
  y = _mm256_fmadd_ps(y, x, _mm256_set1_ps(2.0000714765E-1f));
  y = _mm256_fmadd_ps(y, x, _mm256_set1_ps(-2.4999993993E-1f));
  y = _mm256_fmadd_ps(y, x, _mm256_set1_ps(3.3333331174E-1f));
  y = _mm256_mul_ps(y, x);
  y = _mm256_fmadd_ps(y, z, _mm256_mul_ps(e, _mm256_set1_ps(-2.12194440e-4f)));
  y = _mm256_fnmadd_ps(z, _mm256_set1_ps(0.5f), y);
  x = _mm256_add_ps(x, y);
  return _mm256_fmadd_ps(e, _mm256_set1_ps(0.693359375f), x);
}

__attribute__((target("avx2,fma"))) inline void sincos256(__m256 x, __m256* s, __m256* c) {
  const __m256 sign_mask = _mm256_castsi256_ps(_mm256_set1_epi32(static_cast<int>(0x80000000u)));
  __m256 sign_sin = _mm256_and_ps(x, sign_mask);
  x = _mm256_andnot_ps(sign_mask, x);
  __m256 y = _mm256_mul_ps(x, _mm256_set1_ps(1.27323954473516f));
  __m256i j = _mm256_cvttps_epi32(y);
  j = _mm256_and_si256(_mm256_add_epi32(j, _mm256_set1_epi32(1)), _mm256_set1_epi32(~1));
  y = _mm256_cvtepi32_ps(j);
  const __m256 swap_sin = _mm256_castsi256_ps(_mm256_slli_epi32(_mm256_and_si256(j, _mm256_set1_epi32(4)), 29));
  const __m256 poly = _mm256_castsi256_ps(_mm256_cmpeq_epi32(_mm256_and_si256(j, _mm256_set1_epi32(2)), _mm256_setzero_si256()));
  x = _mm256_fmadd_ps(y, _mm256_set1_ps(-0.78515625f), x);
  x = _mm256_fmadd_ps(y, _mm256_set1_ps(-2.4187564849853515625e-4f), x);
  x = _mm256_fmadd_ps(y, _mm256_set1_ps(-3.77489497744594108e-8f), x);
  const __m256i j2 = _mm256_sub_epi32(j, _mm256_set1_epi32(2));
  const __m256 sign_cos = _mm256_castsi256_ps(_mm256_slli_epi32(_mm256_andnot_si256(j2, _mm256_set1_epi32(4)), 29));
  sign_sin = _mm256_xor_ps(sign_sin, swap_sin);
  const __m256 z = _mm256_mul_ps(x, x);
  __m256 yc = _mm256_set1_ps(2.443315711809948E-005f);
  yc = _mm256_fmadd_ps(yc, z, _mm256_set1_ps(-1.388731625493765E-003f));
  yc = _mm256_fmadd_ps(yc, z, _mm256_set1_ps(4.166664568298827E-002f));
  yc = _mm256_mul_ps(yc, z);
  yc = _mm256_fmsub_ps(yc, z, _mm256_mul_ps(z, _mm256_set1_ps(0.5f)));
  yc = _mm256_add_ps(yc, _mm256_set1_ps(1.0f));
  __m256 ys = _mm256_set1_ps(-1.9515295891E-4f);
  ys = _mm256_fmadd_ps(ys, z, _mm256_set1_ps(8.3321608736E-3f));
  ys = _mm256_fmadd_ps(ys, z, _mm256_set1_ps(-1.6666654611E-1f));
  ys = _mm256_mul_ps(ys, z);
  ys = _mm256_fmadd_ps(ys, x, x);
  const __m256 ysin2 = _mm256_and_ps(poly, ys), ysin1 = _mm256_andnot_ps(poly, yc);
  ys = _mm256_sub_ps(ys, ysin2);
  yc = _mm256_sub_ps(yc, ysin1);
  *s = _mm256_xor_ps(_mm256_add_ps(ysin1, ysin2), sign_sin);
  *c = _mm256_xor_ps(_mm256_add_ps(yc, ys), sign_cos);
}

__attribute__((target("avx2,fma"))) void groups_avx2(uint32_t* words, int64_t n_groups) {
  for (int64_t g = 0; g < n_groups; g++, words += 16) {
    const __m256 u1 = _mm256_sub_ps(_mm256_set1_ps(1.0f), uniform8(words));
    const __m256 u2 = uniform8(words + 8);
    const __m256 radius = _mm256_sqrt_ps(_mm256_mul_ps(_mm256_set1_ps(-2.0f), log256(u1)));
    const __m256 theta = _mm256_mul_ps(_mm256_set1_ps(6.283185307179586476925286766559f), u2);
    __m256 s, c;
    sincos256(theta, &s, &c);
    const __m256 one = _mm256_set1_ps(1.0f), zero = _mm256_setzero_ps();  // (std = 1, mean = 0: turns a -0 product into +0)
    _mm256_storeu_ps(reinterpret_cast<float*>(words), _mm256_fmadd_ps(_mm256_mul_ps(radius, c), one, zero));
    _mm256_storeu_ps(reinterpret_cast<float*>(words) + 8, _mm256_fmadd_ps(_mm256_mul_ps(radius, s), one, zero));
  }
}

void groups(uint32_t* words, int64_t n_groups) {
  static const bool vector = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
  if (vector) { groups_avx2(words, n_groups); return; }
  for (int64_t g = 0; g < n_groups; g++) group16_scalar(words + 16 * g);
}

// The twists a device-made plan left owing (tio_host_mt19937_plan_prefix): `s` is brought to the block the stream stands in —
// by jump-ahead to ONE twist before it and a real twist (the jumped window's first word carries 31 bits that are not part of
// the generator's state: host_rng_jump.cpp; after a twist every bit is the chained one's), or by plain chaining when the
// characteristic polynomial is not available.  Only a generator that is used AGAIN pays this (a Noise call with several images).
void settle(MtState* st) {
  int64_t owed = st->owed_blocks;
  if (owed <= 0) return;
  st->owed_blocks = 0;
  if (owed > 64 && tio_host_rng::jump_available()) {
    const std::vector<tio_host_rng::JumpPolynomial> g = tio_host_rng::jump_polynomials(owed - 1, 1);
    if (g.size() == 1) {
      uint32_t jumped[kN];
      tio_host_rng::jump_state(st->s, *g[0], jumped);
      memcpy(st->s, jumped, sizeof(jumped));
      memcpy(st->s + kN, st->s, 16 * sizeof(uint32_t));
      owed = 1;
    }
  }
  for (int64_t b = 0; b < owed; b++) twist(st->s);
}

inline uint32_t next_word(MtState* st) {
  settle(st);
  if (st->pos >= kN) { twist(st->s); st->pos = 0; }
  return st->s[st->pos++];
}

}  // namespace

extern "C" int tio_host_mt19937_seed(tio_host_mt_state* state, uint64_t seed) {
  if (state == nullptr) return TIO_ERR_INVALID_ARGUMENT;
  mt_seed(reinterpret_cast<MtState*>(state), static_cast<uint32_t>(seed));  // (at::mt19937 keeps the low 32 bits of the seed)
  return TIO_OK;
}

// ---- the plan of a device-side draw (tio_mt19937_randn_device, mt19937.hip) ----------------------------------------------
// words: [0] magic  [1] head  [2,3] total_blocks  [4] n_units  [5] has_tail  [6,7] n  [8..15] 0
//        [16 .. 16 + 624) the rest of the current state block (raw words; `head` of them count)
//        [640 .. 656) torch's tail rule: the last 16 values, final (float bits)
//        [656 ..) one state snapshot per unit of kPlanUnitBlocks blocks
constexpr int64_t kPlanUnitBlocks = 128;
constexpr int64_t kPlanHeader = 16, kPlanTail = kPlanHeader + kN, kPlanSnapshots = kPlanTail + 16;
constexpr uint32_t kPlanMagic = 0x4D54504Cu;

extern "C" int64_t tio_host_mt19937_plan_words(int64_t n) {
  if (n < 0) return 0;
  const int64_t blocks = (n + kN - 1) / kN + 1;
  return kPlanSnapshots + ((blocks + kPlanUnitBlocks - 1) / kPlanUnitBlocks) * kN;
}

extern "C" int tio_host_mt19937_plan(tio_host_mt_state* state, int64_t n, uint32_t* plan, int64_t capacity_words, int64_t* used_words,
                                     int32_t n_threads) {
  MtState* st = reinterpret_cast<MtState*>(state);
  if (st == nullptr || plan == nullptr || used_words == nullptr || n < 0 || st->seeded != 1) return TIO_ERR_INVALID_ARGUMENT;
  if (n < 16) return TIO_ERR_UNSUPPORTED_CONFIG;
  settle(st);
  const int64_t head = std::min<int64_t>(n, kN - st->pos);
  if (head % 16 != 0) return TIO_ERR_UNSUPPORTED_CONFIG;  // groups would straddle state blocks: the host road (state untouched)
  const int64_t body_words = n - head;
  const int64_t total_blocks = (body_words + kN - 1) / kN;
  const int64_t n_units = (total_blocks + kPlanUnitBlocks - 1) / kPlanUnitBlocks;
  const int64_t used = kPlanSnapshots + n_units * kN;
  if (capacity_words < used) return TIO_ERR_INVALID_ARGUMENT;
  memset(plan, 0, static_cast<size_t>(kPlanSnapshots) * sizeof(uint32_t));
  plan[0] = kPlanMagic;
  plan[1] = static_cast<uint32_t>(head);
  plan[2] = static_cast<uint32_t>(total_blocks); plan[3] = static_cast<uint32_t>(static_cast<uint64_t>(total_blocks) >> 32);
  plan[4] = static_cast<uint32_t>(n_units);
  plan[5] = (n % 16) != 0 ? 1u : 0u;
  plan[6] = static_cast<uint32_t>(n); plan[7] = static_cast<uint32_t>(static_cast<uint64_t>(n) >> 32);
  memcpy(plan + kPlanHeader, st->s + st->pos, static_cast<size_t>(head) * sizeof(uint32_t));
  st->pos += static_cast<int32_t>(head);
  // The chain (on aligned copies of the state: no split vector loads).  Long chains are cut into segments of a power of
  // two of blocks: thread t JUMPS to the start of segment t (host_rng_jump.cpp: ~0.3 ms whatever the distance) and chains
  // from there; the last segment leaves the generator's new state.
  constexpr int64_t kMinSegmentBlocks = 4096;  // 2.5 M draws: below, a jump costs more than the chain it saves
  int64_t segment = total_blocks;
  int n_segments = 1;
  if (n_threads > 64) n_threads = 64;  // (the worker pool's size)
  if (n_threads > 1 && total_blocks >= 2 * kMinSegmentBlocks && tio_host_rng::jump_available()) {
    const int64_t wanted = std::max<int64_t>((total_blocks + n_threads - 1) / n_threads, kMinSegmentBlocks);
    segment = kMinSegmentBlocks;
    while (segment < wanted) segment *= 2;  // (few distinct lengths: their polynomials are cached)
    n_segments = static_cast<int>((total_blocks + segment - 1) / segment);
  }
  std::vector<tio_host_rng::JumpPolynomial> ahead;
  if (n_segments > 1) ahead = tio_host_rng::jump_polynomials(segment, n_segments - 1);
  if (static_cast<int>(ahead.size()) != n_segments - 1) { segment = total_blocks; n_segments = 1; }
  auto chain = [&](int t) {
    AlignedState work;
    if (t == 0) memcpy(work.s, st->s, kN * sizeof(uint32_t));
    else tio_host_rng::jump_state(st->s, *ahead[static_cast<size_t>(t) - 1], work.s);
    memcpy(work.s + kN, work.s, 16 * sizeof(uint32_t));
    const int64_t b_end = std::min<int64_t>((t + 1) * segment, total_blocks);
    for (int64_t b = t * segment; b < b_end; b++) {
      if (b % kPlanUnitBlocks == 0) memcpy(plan + kPlanSnapshots + (b / kPlanUnitBlocks) * kN, work.s, kN * sizeof(uint32_t));
      twist_aligned(work.s);
    }
    return work;
  };
  if (total_blocks > 0) {
    AlignedState first, last;
    workers().run(n_segments, [&](int t) {
      const AlignedState done = chain(t);
      if (t == 0) first = done;
      if (t == n_segments - 1) last = done;
    });
    memcpy(st->s, n_segments == 1 ? first.s : last.s, sizeof(first.s));  // (st->s is read by the jumps: written last)
  }
  if (total_blocks > 0) st->pos = static_cast<int32_t>(body_words - (total_blocks - 1) * kN);
  if (n % 16 != 0) {  // normal_fill: "recompute the last 16 values" from 16 FRESH draws
    uint32_t last[16];
    for (int i = 0; i < 16; i++) last[i] = next_word(st);
    groups(last, 1);
    memcpy(plan + kPlanTail, last, sizeof(last));
  }
  *used_words = used;
  return TIO_OK;
}

// ---- the plan whose snapshots the DEVICE makes (round 6; VERDICT r5 missing #3) ---------------------------------------------
// Eight ranks on one host each ran the state chain of 134 M draws on 15 worker threads per step: 4.8 - 5.7 ms per step and rank
// under contention against 1.8 ms of GPU time (profiles/r05_host_stress_gpu_box.json) — the reference-identical noise mode was
// host bound on an 8-GPU node.  mt19937 is linear over GF(2), and every snapshot is an independent polynomial evaluation: the
// device makes them (mt19937.hip: tio_mt19937_device_snapshots).  What is left for the host is this PREFIX — the header, the
// rest of the current block, the state the chain starts from (it IS snapshot 0) — and bookkeeping: the state is left
// `owed_blocks` twists behind and settled when somebody reads it again (a second image of the same Noise call).
//   returns TIO_ERR_UNSUPPORTED_CONFIG (state untouched) where tio_host_mt19937_plan would, and when n is not a multiple of 16
//   (torch's tail rule redraws the last 16 values from 16 FRESH draws: they need the final state now) or there is no body
extern "C" int tio_host_mt19937_plan_prefix(tio_host_mt_state* state, int64_t n, uint32_t* plan, int64_t capacity_words, int64_t* prefix_words,
                                            int64_t* used_words, int64_t* total_blocks_out) {
  MtState* st = reinterpret_cast<MtState*>(state);
  if (st == nullptr || plan == nullptr || prefix_words == nullptr || used_words == nullptr || total_blocks_out == nullptr || n < 0 || st->seeded != 1)
    return TIO_ERR_INVALID_ARGUMENT;
  if (n < 16 || (n % 16) != 0) return TIO_ERR_UNSUPPORTED_CONFIG;
  settle(st);
  const int64_t head = std::min<int64_t>(n, kN - st->pos);
  if (head % 16 != 0) return TIO_ERR_UNSUPPORTED_CONFIG;
  const int64_t body_words = n - head;
  const int64_t total_blocks = (body_words + kN - 1) / kN;
  if (total_blocks < 1) return TIO_ERR_UNSUPPORTED_CONFIG;
  const int64_t n_units = (total_blocks + kPlanUnitBlocks - 1) / kPlanUnitBlocks;
  const int64_t used = kPlanSnapshots + n_units * kN;
  if (capacity_words < kPlanSnapshots + kN) return TIO_ERR_INVALID_ARGUMENT;
  memset(plan, 0, static_cast<size_t>(kPlanSnapshots) * sizeof(uint32_t));
  plan[0] = kPlanMagic;
  plan[1] = static_cast<uint32_t>(head);
  plan[2] = static_cast<uint32_t>(total_blocks); plan[3] = static_cast<uint32_t>(static_cast<uint64_t>(total_blocks) >> 32);
  plan[4] = static_cast<uint32_t>(n_units);
  plan[5] = 0u;
  plan[6] = static_cast<uint32_t>(n); plan[7] = static_cast<uint32_t>(static_cast<uint64_t>(n) >> 32);
  memcpy(plan + kPlanHeader, st->s + st->pos, static_cast<size_t>(head) * sizeof(uint32_t));
  memcpy(plan + kPlanSnapshots, st->s, kN * sizeof(uint32_t));  // snapshot 0: the state the first twist starts from
  st->owed_blocks = total_blocks;
  st->pos = static_cast<int32_t>(body_words - (total_blocks - 1) * kN);
  *prefix_words = kPlanSnapshots + kN;
  *used_words = used;
  *total_blocks_out = total_blocks;
  return TIO_OK;
}

// the jump polynomials of the device's segment starts: count polynomials of kN words (bit k of word k / 32 = the coefficient of
// x^k; degree < 19937), the t-th carrying a state t * segment_blocks twists ahead, t = 1 .. count.  Cached per segment length
// (the first call for a length costs `count` polynomial products: tens of milliseconds).
extern "C" int tio_host_mt19937_segment_polynomials(int64_t segment_blocks, int32_t count, uint32_t* out) {
  if (segment_blocks < 1 || count < 1 || out == nullptr) return TIO_ERR_INVALID_ARGUMENT;
  if (!tio_host_rng::jump_available()) return TIO_ERR_UNSUPPORTED_CONFIG;
  const std::vector<tio_host_rng::JumpPolynomial> list = tio_host_rng::jump_polynomials(segment_blocks, count);
  if (static_cast<int>(list.size()) != count) return TIO_ERR_UNSUPPORTED_CONFIG;
  for (int t = 0; t < count; t++) {
    uint32_t* row = out + static_cast<size_t>(t) * kN;
    memset(row, 0, kN * sizeof(uint32_t));
    const std::vector<uint64_t>& g = *list[static_cast<size_t>(t)];
    for (size_t w = 0; w < g.size() && 2 * w + 1 < static_cast<size_t>(kN); w++) {
      row[2 * w] = static_cast<uint32_t>(g[w]);
      row[2 * w + 1] = static_cast<uint32_t>(g[w] >> 32);
    }
  }
  return TIO_OK;
}

// ---- the plan started AHEAD of its launch (round 4) -------------------------------------------------------------------
// tio_host_mt19937_plan_begin hands the call above to a native thread and returns at once; tio_host_mt19937_plan_end waits
// for it.  Between the two the caller (a Python thread that knows the seed one millisecond of enqueue work before it
// launches the noise kernel: Compose's draw-ahead road) does something else — no interpreter lock is involved, the job
// never touches Python.  One job per state at a time; the state belongs to the job until _end has returned.
namespace {
struct PlanJob {
  std::thread worker;
  int status = TIO_OK;
  int64_t used = 0;
};
std::mutex g_jobs_mu;
std::vector<std::pair<int64_t, PlanJob*>> g_jobs;
int64_t g_next_job = 1;
}  // namespace

extern "C" int64_t tio_host_mt19937_plan_begin(tio_host_mt_state* state, int64_t n, uint32_t* plan, int64_t capacity_words, int32_t n_threads) {
  if (state == nullptr || plan == nullptr) return 0;
  PlanJob* job = new PlanJob();
  job->worker = std::thread([=] { job->status = tio_host_mt19937_plan(state, n, plan, capacity_words, &job->used, n_threads); });
  std::lock_guard<std::mutex> lock(g_jobs_mu);
  const int64_t handle = g_next_job++;
  g_jobs.emplace_back(handle, job);
  return handle;
}

extern "C" int tio_host_mt19937_plan_end(int64_t handle, int64_t* used_words) {
  PlanJob* job = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_jobs_mu);
    for (size_t i = 0; i < g_jobs.size(); i++)
      if (g_jobs[i].first == handle) { job = g_jobs[i].second; g_jobs.erase(g_jobs.begin() + static_cast<long>(i)); break; }
  }
  if (job == nullptr) return TIO_ERR_INVALID_ARGUMENT;
  job->worker.join();
  const int status = job->status;
  if (used_words != nullptr) *used_words = job->used;
  delete job;
  return status;
}

extern "C" int tio_host_mt19937_randn(tio_host_mt_state* state, float* out, int64_t n, int32_t n_threads) {
  MtState* st = reinterpret_cast<MtState*>(state);
  if (st == nullptr || out == nullptr || n < 0 || st->seeded != 1) return TIO_ERR_INVALID_ARGUMENT;
  if (n < 16) return TIO_ERR_UNSUPPORTED_CONFIG;  // torch takes its scalar normal_distribution path below 16 values: not restated
  settle(st);  // (twists a device-made plan left owing)
  uint32_t* words = reinterpret_cast<uint32_t*>(out);
  const int64_t n_full = n & ~static_cast<int64_t>(15);  // values transformed by the in-order groups (normal_fill: i < size - 15)
  constexpr int64_t kUnitBlocks = 128;                  // state blocks per unit of parallel work (80 k values)

  // head: the rest of the current state block (a previous call may have stopped inside it)
  const int64_t head = std::min<int64_t>(n, kN - st->pos);
  const bool aligned = (head % 16) == 0;  // groups of 16 then never straddle state blocks (624 = 39 x 16)
  const int64_t body_words = n - head;
  const int64_t total_blocks = (body_words + kN - 1) / kN;  // state blocks the body touches (the last one maybe partly)
  if (!aligned || n_threads <= 1 || total_blocks < 2 * kUnitBlocks) {
    // the plain road, on this thread: raw words in order, then the groups (also what a misaligned continuation takes)
    for (int64_t i = 0; i < n; i++) words[i] = next_word(st);
    groups(words, n_full / 16);
  } else {
    for (int64_t i = 0; i < head; i++) words[i] = st->s[st->pos++];
    // Phase A, the only sequential part: run the chain through the body, keeping a snapshot of the state every
    // kUnitBlocks blocks (2.5 KB each; everything stays in this core's cache — nothing of the output is touched here)
    const int64_t n_units = (total_blocks + kUnitBlocks - 1) / kUnitBlocks;
    std::vector<uint32_t> snapshots(static_cast<size_t>(n_units) * kN);
    {
      AlignedState work;
      memcpy(work.s, st->s, sizeof(work.s));
      for (int64_t b = 0; b < total_blocks; b++) {
        if (b % kUnitBlocks == 0) memcpy(&snapshots[static_cast<size_t>(b / kUnitBlocks) * kN], work.s, kN * sizeof(uint32_t));
        twist_aligned(work.s);
      }
      memcpy(st->s, work.s, sizeof(work.s));
    }
    const int64_t tail_words = body_words - (total_blocks - 1) * kN;  // words of the last block that belong to this call (1 .. 624)
    st->pos = static_cast<int32_t>(tail_words);
    // Phase B, parallel: every unit replays its kUnitBlocks twists from its snapshot and turns each block into normals
    // where it lands in the output (complete groups only: a trailing partial group is torch's tail rule, below)
    int threads = n_threads > 64 ? 64 : n_threads;
    if (threads > n_units) threads = static_cast<int>(n_units);
    auto run = [&](int t) {
      MtState local;
      for (int64_t u = t; u < n_units; u += threads) {
        memcpy(local.s, &snapshots[static_cast<size_t>(u) * kN], kN * sizeof(uint32_t));
        const int64_t b_end = std::min<int64_t>((u + 1) * kUnitBlocks, total_blocks);
        for (int64_t b = u * kUnitBlocks; b < b_end; b++) {
          twist(local.s);
          const int64_t at = head + b * kN;                                   // first output index of this block
          const int64_t count = std::min<int64_t>(kN, n - at);                // words of it inside this call
          memcpy(words + at, local.s, static_cast<size_t>(count) * sizeof(uint32_t));
          const int64_t whole = (std::min<int64_t>(at + count, n_full) - at) / 16;  // complete groups below n_full
          if (whole > 0) groups(words + at, whole);
        }
      }
    };
    workers().run(threads, [&](int t) {
      if (t == 0) groups(words, head / 16);  // (the head's groups)
      run(t);
    });
  }
  if (n != n_full) {  // normal_fill: "recompute the last 16 values" from 16 FRESH draws
    uint32_t last[16];
    for (int i = 0; i < 16; i++) last[i] = next_word(st);
    groups(last, 1);
    memcpy(words + n - 16, last, sizeof(last));
  }
  return TIO_OK;
}
