// brute force: is q1 = fma(fma(-d,q0,n), r, q0), q0 = RN(n*r), r = RN(1/d) always RN(n/d)
// for d = k/2, k = 1..8192 and all float n in [1,2) and [2,4) (scale invariance)?
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <omp.h>
int main() {
  long total_bad1 = 0, total_bad2 = 0;
  int first_bad_k = -1;
#pragma omp parallel for schedule(dynamic, 8) reduction(+ : total_bad1, total_bad2)
  for (int k = 1; k <= 8192; k++) {
    const float d = 0.5f * (float)k;
    const float r = 1.0f / d;
    long bad1 = 0, bad2 = 0;
    for (int binade = 0; binade < 2; binade++) {
      for (uint32_t m = 0; m < (1u << 23); m += 8) {
        float nn[8];
        for (int l = 0; l < 8; l++) { uint32_t bits = ((127u + binade) << 23) | (m + l); memcpy(&nn[l], &bits, 4); }
        __m256 n = _mm256_loadu_ps(nn), vd = _mm256_set1_ps(d), vr = _mm256_set1_ps(r);
        __m256 ref = _mm256_div_ps(n, vd);
        __m256 q0 = _mm256_mul_ps(n, vr);
        __m256 e = _mm256_fnmadd_ps(vd, q0, n);
        __m256 q1 = _mm256_fmadd_ps(e, vr, q0);
        __m256 e2 = _mm256_fnmadd_ps(vd, q1, n);
        __m256 q2 = _mm256_fmadd_ps(e2, vr, q1);
        int m1 = _mm256_movemask_ps(_mm256_cmp_ps(q1, ref, _CMP_NEQ_UQ));
        int m2 = _mm256_movemask_ps(_mm256_cmp_ps(q2, ref, _CMP_NEQ_UQ));
        bad1 += __builtin_popcount(m1); bad2 += __builtin_popcount(m2);
      }
    }
    if (bad1) {
#pragma omp critical
      { if (first_bad_k < 0 || k < first_bad_k) first_bad_k = k; }
    }
    total_bad1 += bad1; total_bad2 += bad2;
  }
  printf("one refinement: %ld mismatches (first bad k=%d), two refinements: %ld mismatches\n", total_bad1, first_bad_k, total_bad2);
  return 0;
}
