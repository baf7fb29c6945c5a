// div255_check.cpp — development proof (not product): x / 255.0f == fmaf(fmaf(-q, 255, x), y, q) with q = x * y, y = RN(1/255), for EVERY
// float x in [0, 256] (1 132 462 081 values, denormals included; 36 s on 8 threads) — the identity rt_core.h::rt_div255f relies on.
//   g++ -O2 -fopenmp -ffp-contract=off tools/analysis/div255_check.cpp -o /tmp/div255 && /tmp/div255
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
int main() {
  const float y = 1.0f / 255.0f;
  printf("y = %.9g (%a)\n", y, y);
  unsigned long long bad1 = 0, bad2 = 0, n = 0;
  uint32_t hi; float f256 = 256.0f; memcpy(&hi, &f256, 4);
#pragma omp parallel for reduction(+ : bad1, bad2, n)
  for (uint32_t b = 0; b <= hi; ++b) {
    float x; memcpy(&x, &b, 4);
    const float want = x / 255.0f;
    float q = x * y;
    float q1 = __builtin_fmaf(__builtin_fmaf(-q, 255.0f, x), y, q);
    float q2 = __builtin_fmaf(__builtin_fmaf(-q1, 255.0f, x), y, q1);
    bad1 += (q1 != want); bad2 += (q2 != want); n++;
  }
  printf("all %llu floats in [0, 256]: one correction wrong %llu, two corrections wrong %llu\n", n, bad1, bad2);
  // the values the kernel actually divides: bytes, and 0.7f * byte
  int w1 = 0, w2 = 0;
  for (int p = 0; p < 256; ++p) for (int k = 0; k < 2; ++k) {
    float x = k ? 0.7f * (float)p : (float)p;
    float want = x / 255.0f, q = x * y;
    float q1 = __builtin_fmaf(__builtin_fmaf(-q, 255.0f, x), y, q);
    w1 += q1 != want;
  }
  printf("bytes and 0.7f*bytes: one correction wrong %d\n", w1);
}
