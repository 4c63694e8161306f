// lds_rates.cpp — micro-benchmark behind the resampler's tap reads (tests/native/build.sh builds it,
// scripts run it on the GPU box): what does a (z, z+1) tap pair cost as ds_read2_b32, as a
// ds_read_b64 on a 4-byte-aligned address (legal on ROCm: the kernel driver runs gfx9 compute queues
// in unaligned access mode), and how do row pitch / wave layout change the bank conflicts?
//
//   lds_rates            prints cycles per wave instruction for each pattern (one wave per SIMD and 4 per SIMD)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define HIP_CHECK(x)                                                                    \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) {                                                             \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                          \
    }                                                                                   \
  } while (0)

constexpr int kIters = 512;
constexpr int kLdsFloats = 16384;

__global__ void lds_read2(const int* __restrict__ lane_addr, float* __restrict__ out, long long* __restrict__ cycles) {
  __shared__ float lds[kLdsFloats];
  for (int t = threadIdx.x; t < kLdsFloats; t += blockDim.x) lds[t] = static_cast<float>(t);
  __syncthreads();
  const unsigned base = static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) float*)lds));
  const unsigned addr = base + 4u * static_cast<unsigned>(lane_addr[threadIdx.x & 63]);
  float s0 = 0.0f, s1 = 0.0f;
  const long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < kIters; it += 8) {
    float2 v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(v[u]) : "v"(addr));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 8; u++) { s0 += v[u].x; s1 += v[u].y; }
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  out[(blockIdx.x * blockDim.x + threadIdx.x) * 2] = s0;
  out[(blockIdx.x * blockDim.x + threadIdx.x) * 2 + 1] = s1;
}

__global__ void lds_read64(const int* __restrict__ lane_addr, float* __restrict__ out, long long* __restrict__ cycles) {
  __shared__ float lds[kLdsFloats];
  for (int t = threadIdx.x; t < kLdsFloats; t += blockDim.x) lds[t] = static_cast<float>(t);
  __syncthreads();
  const unsigned base = static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) float*)lds));
  const unsigned addr = base + 4u * static_cast<unsigned>(lane_addr[threadIdx.x & 63]);
  float s0 = 0.0f, s1 = 0.0f;
  const long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < kIters; it += 8) {
    float2 v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) asm volatile("ds_read_b64 %0, %1" : "=v"(v[u]) : "v"(addr));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 8; u++) { s0 += v[u].x; s1 += v[u].y; }
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  out[(blockIdx.x * blockDim.x + threadIdx.x) * 2] = s0;
  out[(blockIdx.x * blockDim.x + threadIdx.x) * 2 + 1] = s1;
}

struct Pattern {
  const char* name;
  int addr[64];
};

static void run(const Pattern& p, int threads) {
  int* d_addr;
  float* d_out;
  long long* d_cyc;
  const int blocks = 256;
  HIP_CHECK(hipMalloc(&d_addr, 64 * sizeof(int)));
  HIP_CHECK(hipMalloc(&d_out, static_cast<size_t>(blocks) * threads * 2 * sizeof(float)));
  HIP_CHECK(hipMalloc(&d_cyc, blocks * sizeof(long long)));
  HIP_CHECK(hipMemcpy(d_addr, p.addr, 64 * sizeof(int), hipMemcpyHostToDevice));
  double res[2];
  bool ok[2] = {true, true};
  for (int mode = 0; mode < 2; mode++) {
    for (int rep = 0; rep < 2; rep++) {
      if (mode == 0) lds_read2<<<blocks, threads>>>(d_addr, d_out, d_cyc);
      else lds_read64<<<blocks, threads>>>(d_addr, d_out, d_cyc);
      HIP_CHECK(hipDeviceSynchronize());
    }
    long long cyc[256];
    HIP_CHECK(hipMemcpy(cyc, d_cyc, sizeof(cyc), hipMemcpyDeviceToHost));
    double mean = 0;
    for (int b = 0; b < blocks; b++) mean += static_cast<double>(cyc[b]);
    mean /= blocks;
    // cycles per wave instruction as seen by one wave; LDS-pipe cycles per instruction ~ mean / kIters / (waves per CU sharing)
    res[mode] = mean / kIters;
    float out[128];
    HIP_CHECK(hipMemcpy(out, d_out, sizeof(out), hipMemcpyDeviceToHost));
    for (int l = 0; l < 64; l++) {
      const float e0 = static_cast<float>(kIters) * p.addr[l], e1 = static_cast<float>(kIters) * (p.addr[l] + 1);
      if (out[2 * l] != e0 || out[2 * l + 1] != e1) ok[mode] = false;
    }
  }
  const int waves = threads / 64;
  printf("%-44s %2d waves/CU  read2_b32: %6.2f cyc/instr/wave (%5.2f LDS cyc) %s   read_b64: %6.2f (%5.2f) %s\n", p.name, waves, res[0],
         res[0] / waves, ok[0] ? "ok" : "WRONG", res[1], res[1] / waves, ok[1] ? "ok" : "WRONG");
  hipFree(d_addr); hipFree(d_out); hipFree(d_cyc);
}

int main() {
  Pattern pats[16];
  int np = 0;
  auto add = [&](const char* name, auto fn) {
    pats[np].name = name;
    for (int l = 0; l < 64; l++) pats[np].addr[l] = fn(l);
    np++;
  };
  add("consecutive even (8B aligned)", [](int l) { return 2 * l; });
  add("consecutive pairs overlap (z=l)", [](int l) { return l; });
  add("consecutive odd start (z=l+1)", [](int l) { return l + 1; });
  add("4 rows x16, pitch 24 (16x16 tile)", [](int l) { return (l >> 4) * 24 + (l & 15); });
  add("4 rows x16, pitch 28", [](int l) { return (l >> 4) * 28 + (l & 15); });
  add("4 rows x16, pitch 16/48 (ideal)", [](int l) { return (l >> 4) * 48 + (l & 15); });
  add("rows 0,2,1,3 x16, pitch 24", [](int l) { const int r[4] = {0, 2, 1, 3}; return r[l >> 4] * 24 + (l & 15); });
  add("2 rows x32, pitch 40 (8x32 tile)", [](int l) { return (l >> 5) * 40 + (l & 31); });
  add("2 rows x32, pitch 44", [](int l) { return (l >> 5) * 44 + (l & 31); });
  add("2 rows x32 scale 1.1, pitch 40", [](int l) { return (l >> 5) * 40 + ((l & 31) * 11) / 10; });
  add("2 rows x32 scale 0.9, pitch 40", [](int l) { return (l >> 5) * 40 + ((l & 31) * 9) / 10; });
  add("2 rows x32, row change mid (pitch 40)", [](int l) { return (l >> 5) * 40 + (l & 31) + ((l & 31) >= 13 ? 40 : 0); });
  add("4 rows x16 scale 1.1, pitch 24", [](int l) { return (l >> 4) * 24 + ((l & 15) * 11) / 10; });
  for (int threads : {256, 1024}) {
    for (int i = 0; i < np; i++) run(pats[i], threads);
    printf("\n");
  }
  return 0;
}
