// ubench.hip — instruction-throughput probes for gfx950 (design input for the megakernel; not product code).
// Each probe runs REP x 16 independent instructions per wave; cycles from s_memtime.
// Output: cycles per wave-instruction per SIMD at 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>
#include <algorithm>

#define REP 512
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint64_t now() { return __builtin_readcyclecounter(); }

#define PROBE_ASM(NAME, DECL, BODY, SINK)                                                    \
  __global__ void NAME(uint64_t* out, double seed) {                                         \
    DECL;                                                                                    \
    uint64_t t0 = now();                                                                     \
    for (int i = 0; i < REP; ++i) { BODY; }                                                  \
    uint64_t t1 = now();                                                                     \
    SINK;                                                                                    \
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0; \
  }

// 16 independent chains each
#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

// ---- f32 fma
#define D_F32(i) float a##i = (float)seed + i;
#define B_FMA32(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a##i) : "v"(c));
#define S_F32(i) s += a##i;
PROBE_ASM(k_fma_f32, R16(D_F32) float c = (float)seed * 0.5f; float s = 0, R16(B_FMA32), R16(S_F32) if (s == 1.2345f) out[0] = 1)
// ---- pk fma f32
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define D_PK(i) f32x2 p##i; p##i.x = (float)seed + i; p##i.y = (float)seed - i;
#define B_PK(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p##i) : "v"(pc));
#define S_PK(i) s += p##i.x + p##i.y;
PROBE_ASM(k_pk_fma_f32, R16(D_PK) f32x2 pc; pc.x = (float)seed; pc.y = 0.5f; float s = 0, R16(B_PK), R16(S_PK) if (s == 1.2345f) out[0] = 1)
// ---- f64
#define D_F64(i) double d##i = seed + i;
#define S_F64(i) sd += d##i;
#define B_FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d##i) : "v"(dc));
#define B_ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d##i) : "v"(dc));
#define B_MUL64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d##i) : "v"(dc));
#define B_RCP64(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(d##i));
#define B_RSQ64(i) asm volatile("v_rsq_f64 %0, %0" : "+v"(d##i));
#define B_SQRT64(i) asm volatile("v_sqrt_f64 %0, %0" : "+v"(d##i));
#define F64_PROBE(NAME, B) PROBE_ASM(NAME, R16(D_F64) double dc = seed * 0.5; double sd = 0, R16(B), R16(S_F64) if (sd == 1.2345) out[0] = 1)
F64_PROBE(k_fma_f64, B_FMA64)
F64_PROBE(k_add_f64, B_ADD64)
F64_PROBE(k_mul_f64, B_MUL64)
F64_PROBE(k_rcp_f64, B_RCP64)
F64_PROBE(k_rsq_f64, B_RSQ64)
F64_PROBE(k_sqrt_f64, B_SQRT64)
// compiler sequences for IEEE division and sqrt
#define B_DIV64(i) d##i = dc / d##i;
#define B_CSQRT64(i) d##i = sqrt(d##i + dc);
F64_PROBE(k_div_f64_ieee, B_DIV64)
F64_PROBE(k_sqrt_f64_ieee, B_CSQRT64)
// ---- integer
#define D_U32(i) uint32_t u##i = (uint32_t)seed + i;
#define S_U32(i) su += u##i;
#define B_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u##i) : "v"(uc));
#define B_MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u##i) : "v"(uc));
#define B_XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u##i) : "v"(uc));
#define B_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(u##i) : "v"(uc));
#define U32_PROBE(NAME, B) PROBE_ASM(NAME, R16(D_U32) uint32_t uc = (uint32_t)seed * 3u + 1u; uint32_t su = 0, R16(B), R16(S_U32) if (su == 12345u) out[0] = 1)
U32_PROBE(k_mul_lo_u32, B_MULLO)
U32_PROBE(k_mul_hi_u32, B_MULHI)
U32_PROBE(k_xor_b32, B_XOR)
U32_PROBE(k_add3_u32, B_ADD3)
#define D_U64(i) uint64_t w##i = (uint64_t)seed + i;
#define S_U64(i) sw += w##i;
#define B_MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w##i) : "v"(uc), "v"(ud) : "vcc");
PROBE_ASM(k_mad_u64_u32, R16(D_U64) uint32_t uc = (uint32_t)seed * 3u + 1u; uint32_t ud = uc + 7u; uint64_t sw = 0, R16(B_MAD64), R16(S_U64) if (sw == 12345u) out[0] = 1)
// conversions / transcendental f32
#define B_RCP32(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a##i));
#define B_RSQ32(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(a##i));
PROBE_ASM(k_rcp_f32, R16(D_F32) float c = 0; float s = c, R16(B_RCP32), R16(S_F32) if (s == 1.2345f) out[0] = 1)
PROBE_ASM(k_rsq_f32, R16(D_F32) float c = 0; float s = c, R16(B_RSQ32), R16(S_F32) if (s == 1.2345f) out[0] = 1)
#define B_CVT(i) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d##i) : "v"(u##i));
PROBE_ASM(k_cvt_f64_u32, R16(D_F64) R16(D_U32) double sd = 0, R16(B_CVT), R16(S_F64) if (sd == 1.2345) out[0] = 1)

// ---- LDS gathers: per-lane random 32 B (2 x ds_read_b128) / 16 B / 8 B / 4 B / 2 B reads out of a 32 KB table
template <int BYTES>
__global__ void k_lds_gather(uint64_t* out, double seed) {
  __shared__ __attribute__((aligned(16))) unsigned char tab[32768];
  for (int i = threadIdx.x; i < 32768 / 4; i += blockDim.x) ((uint32_t*)tab)[i] = i * 2654435761u;
  __syncthreads();
  uint32_t idx = (threadIdx.x * 2654435761u + (uint32_t)seed) >> 8;
  double acc = 0; uint32_t acci = 0;
  uint64_t t0 = now();
  for (int i = 0; i < REP; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      uint32_t off = (idx % (32768 / BYTES)) * BYTES;
      if constexpr (BYTES == 32) { double4 v = *(const double4*)(tab + off); acc += v.x + v.w; acci += (uint32_t)__double_as_longlong(v.y); }
      else if constexpr (BYTES == 16) { uint4 v = *(const uint4*)(tab + off); acci += v.x + v.w; }
      else if constexpr (BYTES == 8) { uint2 v = *(const uint2*)(tab + off); acci += v.x + v.y; }
      else if constexpr (BYTES == 4) { acci += *(const uint32_t*)(tab + off); }
      else { acci += *(const uint16_t*)(tab + off); }
      idx = idx * 1664525u + 1013904223u + acci;   // dependent address: latency-bound chain per wave, throughput across waves
    }
  }
  uint64_t t1 = now();
  if (acc == 1.2345 || acci == 12345u) out[0] = 1;
  if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
// independent (throughput) version: 8 independent address streams
template <int BYTES>
__global__ void k_lds_gather_tp(uint64_t* out, double seed) {
  __shared__ __attribute__((aligned(16))) unsigned char tab[32768];
  for (int i = threadIdx.x; i < 32768 / 4; i += blockDim.x) ((uint32_t*)tab)[i] = i * 2654435761u;
  __syncthreads();
  uint32_t idx = (threadIdx.x * 2654435761u + (uint32_t)seed) >> 8;
  uint32_t acci = 0;
  uint64_t t0 = now();
  for (int i = 0; i < REP; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      uint32_t off = ((idx + j * 977u) % (32768 / BYTES)) * BYTES;
      if constexpr (BYTES == 32) { uint4 v = *(const uint4*)(tab + off); uint4 w = *(const uint4*)(tab + off + 16); acci += v.x + w.w; }
      else if constexpr (BYTES == 16) { uint4 v = *(const uint4*)(tab + off); acci += v.x + v.w; }
      else if constexpr (BYTES == 8) { uint2 v = *(const uint2*)(tab + off); acci += v.x + v.y; }
      else if constexpr (BYTES == 4) { acci += *(const uint32_t*)(tab + off); }
      else { acci += *(const uint16_t*)(tab + off); }
    }
    idx = idx * 1664525u + 1013904223u;
  }
  uint64_t t1 = now();
  if (acci == 12345u) out[0] = 1;
  if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
// scalar-cache broadcast loads: wave-uniform pseudo-random 32 B record out of a 16 KB / 320 KB table
__global__ void k_sload(uint64_t* out, double seed, const float* __restrict__ tab, uint32_t n_rec) {
  typedef const float __attribute__((address_space(4)))* KP;
  KP t = (KP)(uintptr_t)tab;
  uint32_t idx = __builtin_amdgcn_readfirstlane((uint32_t)seed + blockIdx.x * 7919u);
  float acc = 0;
  uint64_t t0 = now();
  for (int i = 0; i < REP; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      uint32_t r = (idx + j * 977u) % n_rec;
      KP p = t + (size_t)r * 8;
      acc += p[0] + p[7];
    }
    idx = idx * 1664525u + 1013904223u;
  }
  uint64_t t1 = now();
  if (acc == 1.2345f) out[0] = 1;
  if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
// per-lane global gathers (L1/L2 resident table): 32 B records
__global__ void k_global_gather(uint64_t* out, double seed, const uint4* __restrict__ tab, uint32_t n_rec) {
  uint32_t idx = ((blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + (uint32_t)seed) >> 8;
  uint32_t acci = 0;
  uint64_t t0 = now();
  for (int i = 0; i < REP / 4; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      uint32_t r = (idx + j * 977u) % n_rec;
      uint4 v = tab[2 * r]; uint4 w = tab[2 * r + 1];
      acci += v.x + w.w;
    }
    idx = idx * 1664525u + 1013904223u;
  }
  uint64_t t1 = now();
  if (acci == 12345u) out[0] = 1;
  if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <typename F>
int run(const char* name, F launch, uint64_t* d_out, int insts_per_wave) {
  for (int wps : {1, 2, 4}) {  // waves per SIMD: one block of 256*wps threads per CU
    int threads = 256 * wps, blocks = 256;
    if (threads > 1024) { threads = 1024; }
    int n_waves = blocks * threads / 64;
    float best_ms = 1e30f;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
      hipEventRecord(e0);
      launch(blocks, threads);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); best_ms = std::min(best_ms, ms);
    }
    if (hipGetLastError() != hipSuccess) { printf("%s: launch error\n", name); return 1; }
    std::vector<uint64_t> h(n_waves);
    hipMemcpy(h.data(), d_out, n_waves * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    double med = (double)h[n_waves / 2];
    // s_memtime ticks at 100 MHz on some parts: report both tick-based and wall-based numbers
    double per_inst_ticks = med / insts_per_wave * wps;  // ticks per wave-instruction per SIMD
    double wall_cyc = best_ms * 1e-3 * 2.4e9 / insts_per_wave * wps;  // assuming 2.4 GHz and full chip
    printf("{\"probe\": \"%s\", \"waves_per_simd\": %d, \"ticks_per_inst_per_simd\": %.3f, \"wall_cyc_per_inst_per_simd_at_2.4GHz\": %.3f, \"kernel_ms\": %.4f}\n",
           name, wps, per_inst_ticks, wall_cyc, best_ms);
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
  return 0;
}

int main() {
  uint64_t* d_out; CHK(hipMalloc(&d_out, 1 << 20));
  const int N = REP * 16;
#define RUN(K) run(#K, [&](int b, int t) { hipLaunchKernelGGL(K, dim3(b), dim3(t), 0, 0, d_out, 1.5); }, d_out, N)
  RUN(k_fma_f32); RUN(k_pk_fma_f32); RUN(k_fma_f64); RUN(k_add_f64); RUN(k_mul_f64);
  RUN(k_rcp_f64); RUN(k_rsq_f64); RUN(k_sqrt_f64); RUN(k_div_f64_ieee); RUN(k_sqrt_f64_ieee);
  RUN(k_mul_lo_u32); RUN(k_mul_hi_u32); RUN(k_mad_u64_u32); RUN(k_xor_b32); RUN(k_add3_u32);
  RUN(k_rcp_f32); RUN(k_rsq_f32); RUN(k_cvt_f64_u32);
  RUN(k_lds_gather<32>); RUN(k_lds_gather<16>); RUN(k_lds_gather<4>); RUN(k_lds_gather<2>);
  RUN(k_lds_gather_tp<32>); RUN(k_lds_gather_tp<16>); RUN(k_lds_gather_tp<8>); RUN(k_lds_gather_tp<4>); RUN(k_lds_gather_tp<2>);
  float* d_tab; CHK(hipMalloc((void**)&d_tab, 1 << 20)); CHK(hipMemset(d_tab, 0, 1 << 20));
  for (uint32_t n_rec : {512u, 10240u}) {
    char nm[64]; snprintf(nm, sizeof nm, "k_sload_%u_rec32B", n_rec);
    run(nm, [&](int b, int t) { hipLaunchKernelGGL(k_sload, dim3(b), dim3(t), 0, 0, d_out, 1.5, d_tab, n_rec); }, d_out, N);
    snprintf(nm, sizeof nm, "k_global_gather_%u_rec32B", n_rec);
    run(nm, [&](int b, int t) { hipLaunchKernelGGL(k_global_gather, dim3(b), dim3(t), 0, 0, d_out, 1.5, (const uint4*)d_tab, n_rec); }, d_out, N / 4);
  }
  return 0;
}
