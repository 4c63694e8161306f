// valu_rates.cpp — instruction-rate micro-benchmark for gfx950 (wave64): how many cycles per
// wave-instruction per SIMD the ops used by the resampler cost.  Test infrastructure.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float float4_t __attribute__((ext_vector_type(4)));
constexpr int ITERS = 4096;

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(float* out, float seed, int iseed) {
  float a[8];
  float2_t p[8];
  int n[8];
  unsigned long long q[8];
  float4_t v4[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { q[i] = i; v4[i] = float4_t{0, 0, 0, 0}; a[i] = seed + i + threadIdx.x; p[i] = float2_t{a[i], a[i] + 1.0f}; n[i] = iseed + i * 3 + threadIdx.x; }
  const float b = seed * 0.5f, c = seed * 0.25f;
  const float2_t pb = {b, b}, pc = {c, c};
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if constexpr (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if constexpr (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
      if constexpr (OP == 2) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(n[i]) : "v"(iseed));
      if constexpr (OP == 3) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(n[i]) : "v"(iseed));
      if constexpr (OP == 4) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(n[i]) : "v"(iseed));
      if constexpr (OP == 5) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
      if constexpr (OP == 6) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(n[i]) : "v"(a[i]));
      if constexpr (OP == 7) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if constexpr (OP == 8) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
      if constexpr (OP == 9) asm volatile("v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(n[i]));
      if constexpr (OP == 10) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(n[i]) : "v"(iseed));
      if constexpr (OP == 11) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if constexpr (OP == 12) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if constexpr (OP == 13) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(n[i]) : "v"(iseed));
      if constexpr (OP == 15) asm volatile("v_add_u32 %0, %0, %1" : "+v"(n[i]) : "v"(iseed));
      if constexpr (OP == 16) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if constexpr (OP == 17) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if constexpr (OP == 18) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if constexpr (OP == 19) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
      if constexpr (OP == 20) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
      if constexpr (OP == 21) asm volatile("v_lshlrev_b32 %0, 2, %0" : "+v"(n[i]));
      if constexpr (OP == 22) asm volatile("v_and_b32 %0, %0, %1" : "+v"(n[i]) : "v"(iseed));
      if constexpr (OP == 23) asm volatile("v_mov_b32 %0, %1" : "=v"(n[i]) : "v"(iseed));
      if constexpr (OP == 24) asm volatile("v_rndne_f32 %0, %0" : "+v"(a[i]));
      if constexpr (OP == 25) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a[i]) : "v"(n[i]));
      if constexpr (OP == 26) asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]));
      if constexpr (OP == 27) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(n[i]) : "v"(iseed));
      if constexpr (OP == 28) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if constexpr (OP == 29) asm volatile("v_readlane_b32 s20, %0, 5" : : "v"(n[i]) : "s20");
      if constexpr (OP == 30) asm volatile("v_mul_f32 %0, %0, %1\n s_nop 0" : "+v"(a[i]) : "v"(b));
      if constexpr (OP == 31) asm volatile("v_mul_f32 %0, %0, %1\n s_add_u32 s20, s20, 1" : "+v"(a[i]) : "v"(b) : "s20");
      if constexpr (OP == 32) asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1" : "=v"(p[i]) : "v"((n[i] & 1023) * 4));
      if constexpr (OP == 33) asm volatile("ds_read_b32 %0, %1" : "=v"(a[i]) : "v"((n[i] & 1023) * 4));
      if constexpr (OP == 34) asm volatile("v_mul_f32 %0, %0, %1\n v_add_u32 %2, %2, %3" : "+v"(a[i]), "+v"(n[i]) : "v"(b), "v"(iseed));
      if constexpr (OP == 35) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(n[i]) : "v"(a[i]));
      if constexpr (OP == 36) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(n[i]) : "v"(iseed));
      if constexpr (OP == 37) asm volatile("v_med3_i32 %0, %0, %1, %1" : "+v"(n[i]) : "v"(iseed));
      if constexpr (OP == 38) asm volatile("v_max_i32 %0, %0, %1" : "+v"(n[i]) : "v"(iseed));
      if constexpr (OP == 14) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(b), "v"(c));
      if constexpr (OP == 39) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(n[i]), "v"(iseed) : "vcc");
      if constexpr (OP == 40) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(n[i]) : "v"(iseed));
      if constexpr (OP == 41) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
      if constexpr (OP == 42) asm volatile("v_sin_f32 %0, %0" : "+v"(a[i]));
      if constexpr (OP == 43) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
      if constexpr (OP == 44) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
      if constexpr (OP == 45) asm volatile("ds_read_b128 %0, %1" : "=v"(v4[i]) : "v"((n[i] & 255) * 16));
      if constexpr (OP == 46) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      // selects: the VCC form, the SGPR-pair form, a compare + select pair, and the mask forms that avoid v_cndmask
      if constexpr (OP == 47) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b) : "s20", "s21");
      if constexpr (OP == 48) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(n[i]) : "v"(iseed), "v"(n[(i + 1) & 7]));
      if constexpr (OP == 49) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
      if constexpr (OP == 50) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b) : "s20", "s21");
      if constexpr (OP == 51) asm volatile("v_sub_u32 %1, %0, %2\n v_ashrrev_i32 %1, 31, %1\n v_and_b32 %0, %0, %1" : "+v"(n[i]), "+v"(n[(i + 1) & 7]) : "v"(iseed));
      if constexpr (OP == 52) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if constexpr (OP == 53) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(b), "v"(c) : );
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y + static_cast<float>(n[i]) + static_cast<float>(q[i]) + v4[i].x + v4[i].w;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
static int run(const char* name, float* d_out, int waves_per_simd) {
  hipDeviceProp_t prop;
  HIP_CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD
  hipEvent_t e0, e1;
  HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
  rate_kernel<OP><<<blocks, 256>>>(d_out, 1.0f, 3);
  HIP_CHECK(hipDeviceSynchronize());
  HIP_CHECK(hipEventRecord(e0));
  rate_kernel<OP><<<blocks, 256>>>(d_out, 1.0f, 3);
  HIP_CHECK(hipEventRecord(e1));
  HIP_CHECK(hipEventSynchronize(e1));
  float ms; HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double instr_per_simd = double(ITERS) * 8 * waves_per_simd;
  const double cycles = ms * 1e-3 * prop.clockRate * 1e3;  // clockRate in kHz
  printf("%-22s waves/SIMD %d: %7.3f ms  %.2f cycles per wave-instruction per SIMD (at %.0f MHz nominal)\n", name, waves_per_simd, ms,
         cycles / instr_per_simd, prop.clockRate / 1e3);
  return 0;
}

int main() {
  float* d_out;
  HIP_CHECK(hipMalloc(&d_out, 256 * 256 * 16 * sizeof(float)));
  for (int w : {3, 6}) {
    run<0>("v_fma_f32", d_out, w); run<28>("v_fmac_f32", d_out, w); run<16>("v_sub_f32", d_out, w);
    run<15>("v_add_u32", d_out, w); run<36>("v_sub_u32", d_out, w); run<17>("v_min_f32", d_out, w); run<18>("v_max_f32", d_out, w);
    run<19>("v_cndmask_b32", d_out, w); run<20>("v_cmp_lt_f32", d_out, w); run<21>("v_lshlrev_b32", d_out, w);
    run<22>("v_and_b32", d_out, w); run<23>("v_mov_b32", d_out, w); run<24>("v_rndne_f32", d_out, w);
    run<25>("v_cvt_f32_i32", d_out, w); run<35>("v_cvt_u32_f32", d_out, w); run<26>("v_fract_f32", d_out, w);
    run<27>("v_lshl_add_u32", d_out, w); run<29>("v_readlane_b32", d_out, w); run<37>("v_med3_i32", d_out, w); run<38>("v_max_i32", d_out, w);
    run<30>("v_mul_f32 + s_nop", d_out, w); run<31>("v_mul_f32 + s_add", d_out, w); run<34>("v_mul_f32 + v_add_u32", d_out, w);
    run<32>("ds_read2_b32", d_out, w); run<33>("ds_read_b32", d_out, w);
    run<39>("v_mad_u64_u32", d_out, w); run<4>("v_mul_hi_u32", d_out, w); run<2>("v_mul_lo_u32", d_out, w); run<40>("v_xor_b32", d_out, w);
    run<41>("v_log_f32", d_out, w); run<42>("v_sin_f32", d_out, w); run<43>("v_sqrt_f32", d_out, w); run<46>("v_exp_f32", d_out, w);
    run<44>("v_pk_add_f32", d_out, w);
    run<47>("v_cndmask_b32_e64 sgpr", d_out, w); run<53>("v_cndmask (no dep)", d_out, w); run<48>("v_bfi_b32", d_out, w);
    run<49>("v_cmp+v_cndmask vcc", d_out, w); run<50>("v_cmp+v_cndmask sgpr", d_out, w); run<51>("sub+ashr+and (mask)", d_out, w);
  }
  return 0;
}
