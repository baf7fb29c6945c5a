// valu_rates.hip — development microbenchmark: issue cost of the vector instructions the megakernel leans on, as cycles
// per instruction per wave with 4 waves per SIMD resident (16 waves per CU, like the megakernel).  Each kernel runs a
// long chain of ONE instruction kind, 8 independent chains per lane.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define N_ITER 4096
template <int KIND>
__global__ __launch_bounds__(1024) void k(uint64_t* out, uint32_t seed) {
  uint32_t a[8], b = seed | 1u; uint64_t c[8]; double d[8], e = 1.0000001 + seed * 1e-9;
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 2654435761u + i + seed; c[i] = a[i]; d[i] = 1.0 + a[i] * 1e-10; }
  const long long t0 = clock64();
  for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (KIND == 0) c[i] = (uint64_t)a[i] * b + c[i], a[i] = (uint32_t)(c[i] >> 32) ^ (uint32_t)c[i];   // v_mad_u64_u32 (+ xor)
      if (KIND == 1) d[i] = __builtin_fma(d[i], e, 1e-9);                                                     // v_fma_f64
      if (KIND == 2) a[i] = a[i] * b + 12345u;                                                                // v_mul_lo_u32 (+ add) / v_mad_u32
      if (KIND == 3) a[i] = __umulhi(a[i], b) ^ a[i];                                                         // v_mul_hi_u32 (+ xor)
      if (KIND == 4) a[i] = (a[i] ^ b) + 0x9E3779B9u;                                                         // xor + add (2 simple ops)
      if (KIND == 5) d[i] = d[i] * e;                                                                         // v_mul_f64
      if (KIND == 6) d[i] = d[i] + e;                                                                         // v_add_f64
      if (KIND == 7) { float f = __uint_as_float(a[i]); f = __builtin_fmaf(f, 1.0000001f, 1e-9f); a[i] = __float_as_uint(f); }  // v_fma_f32
    }
  }
  const long long t1 = clock64();
  uint64_t acc = 0;
  for (int i = 0; i < 8; ++i) acc += a[i] + c[i] + (uint64_t)d[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + (uint64_t)(t1 - t0);
  if (threadIdx.x == 0) out[gridDim.x * blockDim.x + blockIdx.x] = (uint64_t)(t1 - t0);
}
int main() {
  const int blocks = 256, threads = 1024;
  uint64_t* d; hipMalloc(&d, (blocks * threads + blocks) * 8);
  const char* names[8] = {"v_mad_u64_u32 + xor", "v_fma_f64", "u32 mul + add", "v_mul_hi_u32 + xor", "xor + add", "v_mul_f64", "v_add_f64", "v_fma_f32"};
  for (int kind = 0; kind < 8; ++kind) {
    for (int rep = 0; rep < 2; ++rep) {
      switch (kind) {
        case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, d, 7u); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, d, 7u); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(threads), 0, 0, d, 7u); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(threads), 0, 0, d, 7u); break;
        case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(threads), 0, 0, d, 7u); break;
        case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(threads), 0, 0, d, 7u); break;
        case 6: hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(threads), 0, 0, d, 7u); break;
        case 7: hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(threads), 0, 0, d, 7u); break;
      }
      hipDeviceSynchronize();
    }
    std::vector<uint64_t> h(blocks);
    hipMemcpy(h.data(), d + (size_t)blocks * threads, blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
    // clock64 = s_memtime, 100 MHz; 4 waves per SIMD share the issue port: per wave and instruction group
    const double us = mean / 100.0;
    const double groups = (double)N_ITER * 8;               // per wave
    std::printf("%-22s %8.1f us per wave  -> %.2f ns per group per wave; x 2.4 GHz / 4 waves = %.2f cycles per group\n", names[kind], us, us * 1e3 / groups,
                us * 1e3 / groups * 2.4 / 4.0);
  }
  return 0;
}
