// host_rng_jump.cpp — jump-ahead for the mt19937 state chain (host code; used by tio_host_mt19937_plan).
//
// The chain of state twists is the one sequential part of torch's CPU randn stream (host_rng.cpp): 215 k twists for the
// 134 M draws of a bench batch, 5 - 6.5 ms of one core, and since the draws themselves moved to the device it is what the
// reference-identical Noise waits for.  mt19937 is linear over GF(2): one WORD step is x[k + 624] = x[k + 397] ^ A(x[k] upper
// bit | x[k + 1] lower bits), so the state after J word steps is g_J(f) applied to the state now, with g_J = x^J mod phi
// and phi the characteristic polynomial of the step (degree 19937) — the method of Haramoto, Matsumoto, Nishimura, Panneton
// and L'Ecuyer, "Efficient jump ahead for F2-linear random number generators" (2008), in its plain Horner form:
//   h = 0;  for k = deg g .. 0:  h = f(h);  if g_k: h ^= s
// on a LINEAR buffer of 624 + 19937 words (a word step appends one word; no wrap), ~0.2 - 0.4 ms per jump whatever J.
// With it the plan splits the chain into segments: thread t jumps to the start of segment t and chains from there.
//
// Nothing here is taken on trust: phi is COMPUTED (Berlekamp-Massey on the low bit of the first 2 x 19937 + 64 output
// words) and checked (degree 19937; phi(f) annihilates a second seed's stream), and tests/test_host_rng.py compares jumped
// states with chained ones.  The 31 low bits of the window's first word are not part of the generator's state (the
// recurrence never reads them): a jumped window may differ from the chained one there, and nowhere else; one twist
// later every bit is the chained one's.
#include <stdint.h>
#include <string.h>

#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "host_rng_jump.hpp"

namespace tio_host_rng {
namespace {

constexpr int kN = 624, kM = 397;
constexpr int kDegree = 19937;
constexpr int kPolyWords = (kDegree + 63) / 64 + 1;  // 313 words of 64 bits: coefficients 0 .. 19937 (+ slack)

inline uint32_t step_word(uint32_t a, uint32_t b, uint32_t c) {
  const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
  return c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

void seed_words(uint32_t* s, uint32_t seed) {
  s[0] = seed;
  for (int j = 1; j < kN; j++) s[j] = 1812433253u * (s[j - 1] ^ (s[j - 1] >> 30)) + static_cast<uint32_t>(j);
}

typedef std::vector<uint64_t> Poly;  // bit k of word k / 64 = coefficient of x^k

inline bool bit(const Poly& p, int k) { return (p[k >> 6] >> (k & 63)) & 1u; }
inline void flip(Poly& p, int k) { p[k >> 6] ^= 1ull << (k & 63); }

int degree(const Poly& p) {
  for (int w = static_cast<int>(p.size()) - 1; w >= 0; w--)
    if (p[w] != 0) return w * 64 + 63 - __builtin_clzll(p[w]);
  return -1;
}

void shift_left_one(Poly& p) {
  uint64_t carry = 0;
  for (size_t w = 0; w < p.size(); w++) {
    const uint64_t next = p[w] >> 63;
    p[w] = (p[w] << 1) | carry;
    carry = next;
  }
}

// Berlekamp-Massey over GF(2): the shortest connection polynomial C (c_0 = 1) with s_n = sum_{i >= 1} c_i s_{n - i}
Poly berlekamp_massey(const std::vector<uint8_t>& s, int* length) {
  const size_t words = s.size() / 64 + 2;
  Poly c(words, 0), b(words, 0), window(words, 0), t;
  c[0] = 1; b[0] = 1;
  int l = 0;
  for (size_t n = 0; n < s.size(); n++) {
    shift_left_one(window);
    window[0] |= s[n];  // bit j of window = s_{n - j}
    uint64_t acc = 0;
    const size_t used = static_cast<size_t>(l) / 64 + 1;
    for (size_t w = 0; w < used && w < words; w++) acc ^= c[w] & window[w];
    shift_left_one(b);  // b(x) <- x b(x)
    if (__builtin_parityll(acc)) {
      if (2 * l <= static_cast<int>(n)) {
        t = c;
        for (size_t w = 0; w < words; w++) c[w] ^= b[w];
        l = static_cast<int>(n) + 1 - l;
        b = t;
      } else {
        for (size_t w = 0; w < words; w++) c[w] ^= b[w];
      }
    }
  }
  *length = l;
  return c;
}

// ---- arithmetic modulo phi ----------------------------------------------------------------------------------------------
struct Field {
  Poly phi;                              // the characteristic polynomial, degree 19937
  std::vector<Poly> phi_shifted;         // phi << r for r = 0 .. 63 (word-aligned xors in the reduction)
  bool ok = false;
};

void xor_at(Poly& into, const Poly& what, size_t word_offset) {
  for (size_t w = 0; w < what.size() && w + word_offset < into.size(); w++) into[w + word_offset] ^= what[w];
}

// (a * b) mod phi; a, b of degree < 19937
Poly mulmod(const Field& f, const Poly& a, const Poly& b) {
  std::vector<Poly> shifted(64, Poly(kPolyWords + 1, 0));
  for (int r = 0; r < 64; r++) {
    for (int w = 0; w < kPolyWords; w++) {
      shifted[r][w] |= r ? (b[w] << r) : b[w];
      if (r) shifted[r][w + 1] |= b[w] >> (64 - r);
    }
  }
  Poly product(2 * kPolyWords + 2, 0);
  for (int w = 0; w < kPolyWords; w++) {
    uint64_t bits = a[w];
    while (bits) {
      const int r = __builtin_ctzll(bits);
      bits &= bits - 1;
      xor_at(product, shifted[r], static_cast<size_t>(w));
    }
  }
  for (int k = 2 * kDegree; k >= kDegree; k--) {  // reduce: clear the coefficients from the top
    if (!((product[k >> 6] >> (k & 63)) & 1u)) continue;
    const int shift = k - kDegree;
    xor_at(product, f.phi_shifted[shift & 63], static_cast<size_t>(shift >> 6));
  }
  product.resize(kPolyWords);
  return product;
}

Field build_field() {
  Field f;
  // the low bit of the first 2 x 19937 + 64 output words of some seed: a linear functional of the state per step
  const int count = 2 * kDegree + 64;
  std::vector<uint32_t> x(static_cast<size_t>(kN) + count);
  seed_words(x.data(), 5489u);
  for (int k = 0; k < count; k++) x[k + kN] = step_word(x[k], x[k + 1], x[k + kM]);
  std::vector<uint8_t> sequence(count);
  for (int k = 0; k < count; k++) sequence[k] = static_cast<uint8_t>(x[k + kN] & 1u);
  int length = 0;
  const Poly c = berlekamp_massey(sequence, &length);
  if (length != kDegree) return f;
  f.phi.assign(kPolyWords, 0);
  for (int k = 0; k <= kDegree; k++)  // phi(x) = x^L C(1 / x)
    if (bit(c, kDegree - k)) flip(f.phi, k);
  if (degree(f.phi) != kDegree || !bit(f.phi, 0)) return f;
  f.phi_shifted.assign(64, Poly(kPolyWords + 1, 0));
  for (int r = 0; r < 64; r++)
    for (int w = 0; w < kPolyWords; w++) {
      f.phi_shifted[r][w] |= r ? (f.phi[w] << r) : f.phi[w];
      if (r) f.phi_shifted[r][w + 1] |= f.phi[w] >> (64 - r);
    }
  // check on another seed: sum_k phi_k x[n + k] = 0 for every bit of the words (beyond the first: its low bits are free)
  std::vector<uint32_t> y(static_cast<size_t>(kN) + kDegree + 8);
  seed_words(y.data(), 20240229u);
  for (int k = 0; k + kN < static_cast<int>(y.size()); k++) y[k + kN] = step_word(y[k], y[k + 1], y[k + kM]);
  for (int n = 1; n < 6; n++) {
    uint32_t acc = 0;
    for (int k = 0; k <= kDegree; k++)
      if (bit(f.phi, k)) acc ^= y[n + k];
    if (acc != 0) return f;
  }
  f.ok = true;
  return f;
}

const Field& field() {
  static const Field f = build_field();
  return f;
}

// x^e mod phi for e = word steps
Poly power_of_x(const Field& f, uint64_t e) {
  Poly result(kPolyWords, 0), base(kPolyWords, 0);
  result[0] = 1;  // 1
  base[0] = 2;    // x
  while (e) {
    if (e & 1u) result = mulmod(f, result, base);
    e >>= 1;
    if (e) base = mulmod(f, base, base);
  }
  return result;
}

std::mutex g_cache_mutex;
std::map<int64_t, std::vector<JumpPolynomial>> g_cache;  // segment length in blocks -> g for 1, 2, ... segments

}  // namespace

bool jump_available() { return field().ok; }

// the polynomials that carry a state `t * segment_blocks` twists ahead, t = 1 .. count (cached per segment length)
std::vector<JumpPolynomial> jump_polynomials(int64_t segment_blocks, int count) {
  const Field& f = field();
  if (!f.ok || segment_blocks <= 0 || count <= 0) return {};
  std::lock_guard<std::mutex> lock(g_cache_mutex);
  std::vector<JumpPolynomial>& list = g_cache[segment_blocks];
  if (list.empty()) list.push_back(std::make_shared<const Poly>(power_of_x(f, static_cast<uint64_t>(segment_blocks) * kN)));
  while (static_cast<int>(list.size()) < count) list.push_back(std::make_shared<const Poly>(mulmod(f, *list.back(), *list[0])));
  return std::vector<JumpPolynomial>(list.begin(), list.begin() + count);
}

// out[0 .. 624) = the window `g(f)` carries `in` to: Horner over word steps on a linear buffer, kWindow coefficients at
// a time —  h <- f^w(h) ^ sum_i g_{k - i} f^{w - 1 - i}(s)  — with the 2^w possible sums tabulated first (they are windows
// of s's own sequence, shifted by 0 .. w - 1 words): a quarter (w = 8: 2 500 + the 255 of the table) of the 624-word xors of the plain form; measured here 0.55 -> 0.28 (w = 4) -> 0.20 ms (w = 8).
constexpr int kWindow = 8;

__attribute__((target_clones("avx512f", "avx2", "default")))  // (the 624-word xor is the whole cost: as wide as the host allows)
void jump_state(const uint32_t* in, const std::vector<uint64_t>& g, uint32_t* out) {
  const int deg = degree(g);
  if (deg < 0) { memset(out, 0, kN * sizeof(uint32_t)); return; }
  constexpr int kSums = 1 << kWindow;
  // s's sequence, kWindow - 1 words on
  uint32_t xs[kN + kWindow];
  memcpy(xs, in, kN * sizeof(uint32_t));
  for (int i = 0; i + 1 < kWindow; i++) xs[kN + i] = step_word(xs[i], xs[i + 1], xs[i + kM]);
  // table[j] = xor over the set bits i of j of f^i(s) = the window of xs at offset i
  // (scratch kept per thread: 640 KB + 100 KB allocated afresh by every jump are an mmap / munmap pair and a few hundred page
  // faults each, taken under the process's address-space lock by all planning threads at once)
  static thread_local std::vector<uint32_t> table, line;
  table.resize(static_cast<size_t>(kSums) * kN);
  memset(table.data(), 0, kN * sizeof(uint32_t));  // (row 0: the empty sum; every other row is written below)
  for (int j = 1; j < kSums; j++) {
    const int low = __builtin_ctz(j);
    const uint32_t* base = &table[static_cast<size_t>(j & (j - 1)) * kN];
    uint32_t* row = &table[static_cast<size_t>(j) * kN];
    for (int m = 0; m < kN; m++) row[m] = base[m] ^ xs[low + m];
  }
  const int groups = deg / kWindow + 1;  // coefficient groups, the top one padded with zeros
  line.resize(static_cast<size_t>(kN) + static_cast<size_t>(groups) * kWindow + 16);
  uint32_t* h = line.data();  // the window: h[0 .. 624), all zero (what lies beyond is written before it is read)
  memset(h, 0, kN * sizeof(uint32_t));
  for (int q = groups - 1; q >= 0; q--) {
    for (int i = 0; i < kWindow; i++) h[kN + i] = step_word(h[i], h[i + 1], h[i + kM]);  // kWindow word steps (independent: 397 + w < 624)
    h += kWindow;
    // the group's coefficients g_{w q + w - 1} .. g_{w q}: bit i of j pairs g_{w q + i} with f^i(s)
    unsigned j = 0;
    for (int i = 0; i < kWindow; i++) {
      const int k = q * kWindow + i;
      if (k <= deg && ((g[k >> 6] >> (k & 63)) & 1u)) j |= 1u << i;
    }
    if (j != 0) {
      const uint32_t* row = &table[static_cast<size_t>(j) * kN];
      for (int m = 0; m < kN; m++) h[m] ^= row[m];
    }
  }
  memcpy(out, h, kN * sizeof(uint32_t));
}

}  // namespace tio_host_rng
