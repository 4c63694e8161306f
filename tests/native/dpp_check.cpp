// dpp_check.cpp — validates the DPP wave-reduction used by the tile kernel and times a few
// select forms.  Test infrastructure.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__device__ __forceinline__ int wave_max_dpp(int v) {
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false));  // row_half_mirror
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false));  // row_mirror
  const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
  const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  return max(max(a, b), max(c, d));
}

__global__ void check_kernel(const int* in, int* out) {
  const int v = in[blockIdx.x * 64 + threadIdx.x];
  const int r = wave_max_dpp(v);
  if (threadIdx.x == 0) out[blockIdx.x] = r;
}

constexpr int ITERS = 4096;
template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(float* out, float seed, int sel) {
  float a[8];
  typedef float v16 __attribute__((ext_vector_type(16)));
  v16 P;
#pragma unroll
  for (int i = 0; i < 16; i++) P[i] = seed * i + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; i++) a[i] = seed + i + threadIdx.x;
  const float b = seed * 0.5f;
  for (int it = 0; it < ITERS; it++) {
    const int e = (it + sel) & 3;  // uniform, loop varying
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if constexpr (OP == 0) a[i] = (e == 0) ? a[i] * b : a[i] + b;         // compiler's choice for a uniform select
      if constexpr (OP == 1) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(b) : "vcc");
      if constexpr (OP == 2) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b));
      if constexpr (OP == 3) a[i] += P[(e * 3 + i) & 15];                    // dynamic uniform register index
      if constexpr (OP == 4) a[i] = __shfl_xor(a[i], 1 << (i % 6), 64);
      if constexpr (OP == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
    }
  }
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; i++) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
static int run(const char* name, float* d_out) {
  const int w = 4, blocks = 256 * w;
  hipEvent_t e0, e1;
  HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
  rate_kernel<OP><<<blocks, 256>>>(d_out, 1.0f, 1);
  HIP_CHECK(hipDeviceSynchronize());
  HIP_CHECK(hipEventRecord(e0));
  rate_kernel<OP><<<blocks, 256>>>(d_out, 1.0f, 1);
  HIP_CHECK(hipEventRecord(e1));
  HIP_CHECK(hipEventSynchronize(e1));
  float ms; HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-28s %7.3f ms  %.2f nominal cycles per statement per SIMD (4 waves/SIMD)\n", name, ms, ms * 1e-3 * 2.4e9 / (double(ITERS) * 8 * w));
  return 0;
}

int main() {
  const int nb = 64;
  int h_in[nb * 64], h_out[nb], *d_in, *d_out;
  srand(7);
  for (int i = 0; i < nb * 64; i++) h_in[i] = (rand() % 2000001) - 1000000;
  HIP_CHECK(hipMalloc(&d_in, sizeof(h_in))); HIP_CHECK(hipMalloc(&d_out, sizeof(h_out)));
  HIP_CHECK(hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice));
  check_kernel<<<nb, 64>>>(d_in, d_out);
  HIP_CHECK(hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost));
  int bad = 0;
  for (int b = 0; b < nb; b++) {
    int m = h_in[b * 64];
    for (int l = 1; l < 64; l++) m = h_in[b * 64 + l] > m ? h_in[b * 64 + l] : m;
    if (m != h_out[b]) bad++;
  }
  printf("dpp wave max: %d / %d waves wrong\n", bad, nb);
  float* d_f;
  HIP_CHECK(hipMalloc(&d_f, 1024 * 256 * sizeof(float)));
  run<0>("uniform select (compiler)", d_f); run<1>("v_cmp + v_cndmask vcc", d_f); run<2>("v_cndmask_e64 sgpr mask", d_f);
  run<3>("dynamic uniform reg index", d_f); run<4>("__shfl_xor", d_f); run<5>("v_cndmask vcc (no cmp)", d_f);
  return bad != 0;
}
