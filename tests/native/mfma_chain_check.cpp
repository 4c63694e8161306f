// mfma_chain_check.cpp — can the matrix pipe evaluate the reference's affine row chain bit for bit?
//
// The reference forms a sampling coordinate as MKL sgemm does: fma(1, m3, fma(k, m2, fma(j, m1, i * m0))) per row
// (spatial.py:1604-1624).  v_mfma_f32_4x4x1f32 computes, for 16 independent 4x4 blocks, D[r][c] = A[r] * B[c] + C[r][c] —
// ONE multiply-add per element and instruction (K = 1).  With B = the lane's own coordinate (column c = the voxel), A = one
// column of the mapping (row r = the output axis) and C = the running sum, four such instructions are the chain — IF the
// unit rounds like one fused multiply-add (a single rounding, no flush of denormals).  This program answers that on the
// hardware: part 1 compares the two forms bit for bit over random operands (integers as the kernel sees them, and arbitrary
// floats, and denormal products); part 2 times a coordinate phase shaped like resample_lean_exact_kernel's with the chain on
// the vector ALU against the same with the chain on the matrix pipe (three waves per SIMD, as LDS allows the real kernel).
// Test infrastructure (measurement); nothing links it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float chain_valu(const float* m, int r, float ci, float cj, float ck) {
  return __builtin_fmaf(1.0f, m[4 * r + 3], __builtin_fmaf(ck, m[4 * r + 2], __builtin_fmaf(cj, m[4 * r + 1], __fmul_rn(ci, m[4 * r]))));
}

// mapping column `term` as the A operand: lane l feeds row l & 3 of its block (row 3: zero)
__device__ __forceinline__ float a_operand(const float* m, int term, int lane) {
  const int r = lane & 3;
  return r < 3 ? m[4 * r + term] : 0.0f;
}

__global__ void check_kernel(const float* __restrict__ mats, const float* __restrict__ coords, uint32_t* __restrict__ out, int n_cases) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x;
  if (c >= n_cases) return;
  const float* m = mats + c * 12;
  const float ci = coords[(c * 64 + lane) * 3], cj = coords[(c * 64 + lane) * 3 + 1], ck = coords[(c * 64 + lane) * 3 + 2];
  v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a_operand(m, 0, lane), ci, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a_operand(m, 1, lane), cj, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a_operand(m, 2, lane), ck, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a_operand(m, 3, lane), 1.0f, acc, 0, 0, 0);
  uint32_t* o = out + (c * 64 + lane) * 6;
  o[0] = __float_as_uint(acc[0]); o[1] = __float_as_uint(acc[1]); o[2] = __float_as_uint(acc[2]);
  o[3] = __float_as_uint(chain_valu(m, 0, ci, cj, ck));
  o[4] = __float_as_uint(chain_valu(m, 1, ci, cj, ck));
  o[5] = __float_as_uint(chain_valu(m, 2, ci, cj, ck));
}

// ---- part 2: a coordinate phase, chain on the vector ALU or on the matrix pipe ---------------------------------------------
__device__ __forceinline__ float roundtrip(float v, float dh, float rdh, float half_h) {
  float q = __fmul_rn(v, rdh);
  const float e = __builtin_fmaf(-dh, q, v);
  q = __builtin_fmaf(e, rdh, q);
  const float g = __fsub_rn(q, 1.0f);
  return __fmul_rn(__fadd_rn(g, 1.0f), half_h);
}

template <int FORM>  // 0: VALU chain + round trip, 1: MFMA chain + round trip, 2: round trip only (chain replaced by one add),
                     // 3: MFMA chains of four planes interleaved (no dependent instruction back to back)
__global__ __launch_bounds__(256, 3) void phase_kernel(const float* __restrict__ mats, float* __restrict__ out, int iters) {
  extern __shared__ float smem[];
  const int lane = threadIdx.x & 63;
  float m[12];
#pragma unroll
  for (int q = 0; q < 12; q++) m[q] = mats[q];
  const float cj = static_cast<float>(threadIdx.x >> 4), ck = static_cast<float>(threadIdx.x & 15);
  const float dh = 127.5f, rdh = 1.0f / 127.5f, half_h = 127.5f;
  const float a0 = a_operand(m, 0, lane), a1 = a_operand(m, 1, lane), a2 = a_operand(m, 2, lane), a3 = a_operand(m, 3, lane);
  float s = 0.0f;
  for (int it = 0; it < iters; it++) {
    float X[16], Y[16], Z[16];
    if constexpr (FORM == 3) {
#pragma unroll
      for (int g = 0; g < 4; g++) {
        v4f acc[4];
#pragma unroll
        for (int u = 0; u < 4; u++) acc[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, static_cast<float>(it + 4 * g + u), v4f{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; u++) acc[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1, cj, acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; u++) acc[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(a2, ck, acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; u++) acc[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(a3, 1.0f, acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; u++) {
          X[4 * g + u] = roundtrip(acc[u][0], dh, rdh, half_h); Y[4 * g + u] = roundtrip(acc[u][1], dh, rdh, half_h);
          Z[4 * g + u] = roundtrip(acc[u][2], dh, rdh, half_h);
        }
      }
    } else {
#pragma unroll
    for (int t = 0; t < 16; t++) {
      const float ci = static_cast<float>(it + t);
      float vi, vj, vk;
      if constexpr (FORM == 0) {
        vi = chain_valu(m, 0, ci, cj, ck); vj = chain_valu(m, 1, ci, cj, ck); vk = chain_valu(m, 2, ci, cj, ck);
      } else if constexpr (FORM == 1) {
        v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, ci, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a1, cj, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a2, ck, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a3, 1.0f, acc, 0, 0, 0);
        vi = acc[0]; vj = acc[1]; vk = acc[2];
      } else {
        vi = ci + cj; vj = ci + ck; vk = cj + ci;
      }
      X[t] = roundtrip(vi, dh, rdh, half_h); Y[t] = roundtrip(vj, dh, rdh, half_h); Z[t] = roundtrip(vk, dh, rdh, half_h);
    }
    }
#pragma unroll
    for (int t = 0; t < 16; t++) s += X[t] + Y[t] * Z[t];
  }
  if (s == 12345.678f) smem[threadIdx.x] = s;  // (keeps the LDS allocation alive)
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

static uint32_t rng_state = 12345u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state; }
static float urand(float lo, float hi) { return lo + (hi - lo) * ((rnd() >> 8) * (1.0f / 16777216.0f)); }

template <int FORM>
static int time_phase(const char* name, const float* d_m, float* d_out) {
  const int blocks = 256 * 3 * 8, iters = 64;
  hipEvent_t e0, e1;
  HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
  const size_t lds = 53 * 1024;  // three blocks per CU, as the real kernel's tile
  HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&phase_kernel<FORM>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
  phase_kernel<FORM><<<blocks, 256, lds>>>(d_m, d_out, iters);
  HIP_CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; rep++) {
    HIP_CHECK(hipEventRecord(e0));
    phase_kernel<FORM><<<blocks, 256, lds>>>(d_m, d_out, iters);
    HIP_CHECK(hipEventRecord(e1));
    HIP_CHECK(hipEventSynchronize(e1));
    float ms; HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double planes = static_cast<double>(blocks) * 256 * iters * 16;
  printf("%-44s %8.3f ms  %7.2f ps per voxel-plane  (%.1f G voxel-planes/s)\n", name, best, best * 1e9 / planes, planes / best * 1e-6);
  return 0;
}

int main() {
  const int n_cases = 4096;
  float* h_m = static_cast<float*>(malloc(n_cases * 12 * sizeof(float)));
  float* h_c = static_cast<float*>(malloc(n_cases * 64 * 3 * sizeof(float)));
  for (int c = 0; c < n_cases; c++) {
    const int family = c & 3;
    for (int q = 0; q < 12; q++) {
      float v;
      if (family == 0) {  // rotation-like rows, translations of tens of voxels
        v = (q & 3) == 3 ? urand(-40.f, 40.f) : urand(-1.2f, 1.2f);
      } else if (family == 1) {  // anything
        v = urand(-1.f, 1.f) * powf(2.0f, static_cast<float>(static_cast<int>(rnd() % 40) - 20));
      } else if (family == 2) {  // tiny entries: denormal products and sums
        v = urand(-1.f, 1.f) * 1e-38f;
      } else {  // exact cancellations / signed zeros: small integers
        v = static_cast<float>(static_cast<int>(rnd() % 5) - 2);
      }
      h_m[c * 12 + q] = v;
    }
    for (int l = 0; l < 64; l++)
      for (int e = 0; e < 3; e++) {
        float v;
        if (family == 0 || family == 3) v = static_cast<float>(rnd() % 512);              // voxel indices
        else if (family == 1) v = urand(-600.f, 600.f);                                       // displaced (non-integer) positions
        else v = urand(0.f, 4.f);
        h_c[(c * 64 + l) * 3 + e] = v;
      }
  }
  float *d_m, *d_c; uint32_t* d_o;
  HIP_CHECK(hipMalloc(&d_m, n_cases * 12 * sizeof(float)));
  HIP_CHECK(hipMalloc(&d_c, n_cases * 64 * 3 * sizeof(float)));
  HIP_CHECK(hipMalloc(&d_o, n_cases * 64 * 6 * sizeof(uint32_t)));
  HIP_CHECK(hipMemcpy(d_m, h_m, n_cases * 12 * sizeof(float), hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(d_c, h_c, n_cases * 64 * 3 * sizeof(float), hipMemcpyHostToDevice));
  check_kernel<<<n_cases, 64>>>(d_m, d_c, d_o, n_cases);
  HIP_CHECK(hipDeviceSynchronize());
  uint32_t* h_o = static_cast<uint32_t*>(malloc(n_cases * 64 * 6 * sizeof(uint32_t)));
  HIP_CHECK(hipMemcpy(h_o, d_o, n_cases * 64 * 6 * sizeof(uint32_t), hipMemcpyDeviceToHost));
  long mism[4] = {0, 0, 0, 0}, total[4] = {0, 0, 0, 0};
  int shown = 0;
  for (int c = 0; c < n_cases; c++)
    for (int l = 0; l < 64; l++)
      for (int r = 0; r < 3; r++) {
        const uint32_t a = h_o[(c * 64 + l) * 6 + r], b = h_o[(c * 64 + l) * 6 + 3 + r];
        total[c & 3]++;
        if (a != b) {
          mism[c & 3]++;
          if (shown < 12) {
            float fa, fb; memcpy(&fa, &a, 4); memcpy(&fb, &b, 4);
            printf("  mismatch family %d case %d lane %d row %d: mfma %.9g (%08x)  valu %.9g (%08x)\n", c & 3, c, l, r, fa, a, fb, b);
            shown++;
          }
        }
      }
  const char* names[4] = {"rotation rows x voxel indices", "arbitrary floats", "denormal range", "small integers (cancellation, zeros)"};
  for (int f = 0; f < 4; f++) printf("mfma chain vs fma chain, %-40s mismatches %ld of %ld\n", names[f], mism[f], total[f]);

  float* d_out;
  HIP_CHECK(hipMalloc(&d_out, 256 * 3 * 8 * 256 * sizeof(float)));
  if (time_phase<0>("chain on the vector ALU + round trip (31 / plane)", d_m, d_out)) return 2;
  if (time_phase<1>("chain on the matrix pipe + round trip (19 + 4 mfma)", d_m, d_out)) return 2;
  if (time_phase<3>("... four planes' chains interleaved", d_m, d_out)) return 2;
  if (time_phase<2>("round trip only (3 adds for the chain)", d_m, d_out)) return 2;
  return (mism[0] == 0 && mism[3] == 0) ? 0 : 1;
}
