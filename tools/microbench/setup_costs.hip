// setup_costs.hip — what the HIP runtime's first-use costs are in a fresh process, one at a time and side by side
// (round 6: where the 36 ms of a one-shot frame's set-up go).  hipcc --offload-arch=gfx950 -O2 setup_costs.hip -o setup_costs -lpthread
//   ./setup_costs seq | par
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>
using clk = std::chrono::steady_clock;
static double ms(clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); }
__global__ void k_touch(int* p) { if (p) p[threadIdx.x] = 1; }
int main(int argc, char** argv) {
  const bool par = argc > 1 && !strcmp(argv[1], "par");
  auto t0 = clk::now();
  int n = 0; hipGetDeviceCount(&n);
  printf("hipGetDeviceCount %.2f ms (n=%d)\n", ms(t0), n);
  auto t1 = clk::now();
  hipSetDevice(0); void* p = nullptr; hipMalloc(&p, 256);
  printf("first hipMalloc (context) %.2f ms\n", ms(t1));
  hipStream_t s[4] = {};
  void* pinned[2] = {};
  std::vector<char> host(1 << 20, 1);
  void* d1 = nullptr; hipMalloc(&d1, 32 << 20);
  struct Job { const char* name; std::function<void()> fn; double ms; };
  std::vector<Job> jobs = {
    {"first kernel launch (code object)", [&] { hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, nullptr, (int*)nullptr); hipDeviceSynchronize(); }, 0},
    {"hipStreamCreateWithFlags #1", [&] { hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking); }, 0},
    {"hipStreamCreateWithFlags #2", [&] { hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking); }, 0},
    {"hipHostMalloc 2.88 MB #1", [&] { hipHostMalloc(&pinned[0], 2880000, hipHostMallocDefault); }, 0},
    {"hipHostMalloc 2.88 MB #2", [&] { hipHostMalloc(&pinned[1], 2880000, hipHostMallocDefault); }, 0},
    {"first pageable H2D copy 64 KB", [&] { hipMemcpy(d1, host.data(), 65536, hipMemcpyHostToDevice); }, 0},
  };
  auto tp = clk::now();
  if (par) {
    std::vector<std::thread> th;
    for (auto& j : jobs) th.emplace_back([&j] { hipSetDevice(0); auto t = clk::now(); j.fn(); j.ms = ms(t); });
    for (auto& t : th) t.join();
  } else for (auto& j : jobs) { auto t = clk::now(); j.fn(); j.ms = ms(t); }
  printf("%s: all jobs %.2f ms\n", par ? "PARALLEL" : "SEQUENTIAL", ms(tp));
  for (auto& j : jobs) printf("  %-40s %.2f ms\n", j.name, j.ms);
  // second-use costs
  auto t = clk::now(); hipMemcpy(d1, host.data(), 65536, hipMemcpyHostToDevice); printf("second pageable H2D 64 KB %.3f ms\n", ms(t));
  t = clk::now(); hipMemcpy(d1, host.data(), 1 << 20, hipMemcpyHostToDevice); printf("pageable H2D 1 MB %.3f ms\n", ms(t));
  std::vector<char> big(29 << 20, 2);
  t = clk::now(); hipMemcpy(d1, big.data(), big.size(), hipMemcpyHostToDevice); printf("pageable H2D 29 MB %.3f ms\n", ms(t));
  t = clk::now(); hipMemcpyAsync(pinned[0], d1, 2880000, hipMemcpyDeviceToHost, s[0]); hipStreamSynchronize(s[0]); printf("first D2H pinned 2.88 MB on stream %.3f ms\n", ms(t));
  t = clk::now(); hipMemcpyAsync(pinned[0], d1, 2880000, hipMemcpyDeviceToHost, s[0]); hipStreamSynchronize(s[0]); printf("second D2H pinned 2.88 MB %.3f ms\n", ms(t));
  t = clk::now(); hipMemcpy(host.data(), d1, 1 << 20, hipMemcpyDeviceToHost); printf("first D2H pageable 1 MB %.3f ms\n", ms(t));
  if (argc > 2 && !strcmp(argv[2], "after_copy")) {  // (round 6) what the FIRST kernel after a 29 MB pageable upload waits for
    const size_t n = 29u << 20;
    char* src = (char*)malloc(n);
    { std::vector<std::thread> th; for (int k = 0; k < 8; ++k) th.emplace_back([=] { memset(src + (n / 8) * k, k + 1, n / 8); }); for (auto& x : th) x.join(); }
    void* d2 = nullptr; hipMalloc(&d2, n);
    t = clk::now(); hipMemcpy(d2, src, n, hipMemcpyHostToDevice); printf("fresh pageable H2D 29 MB returns %.3f ms\n", ms(t));
    t = clk::now(); hipDeviceSynchronize(); printf("  hipDeviceSynchronize %.3f ms\n", ms(t));
    t = clk::now(); hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, nullptr, (int*)nullptr); double a = ms(t); hipDeviceSynchronize(); printf("  first kernel after it: enqueue %.3f ms, done %.3f ms\n", a, ms(t));
    t = clk::now(); hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, nullptr, (int*)nullptr); hipDeviceSynchronize(); printf("  second kernel %.3f ms\n", ms(t));
    t = clk::now(); free(src); printf("  free(src) %.3f ms\n", ms(t));
    return 0;
  }
  // (round 6) a 29 MB texel upload, to completion: pageable source vs pinned staging
  { std::vector<char> tex(29 << 20, 3); void* d2 = nullptr; hipMalloc(&d2, 32 << 20);
    t = clk::now(); hipMemcpy(d2, tex.data(), tex.size(), hipMemcpyHostToDevice); double a = ms(t); hipDeviceSynchronize(); printf("pageable H2D 29 MB: returns %.3f ms, device done %.3f ms\n", a, ms(t));
    t = clk::now(); hipMemcpy(d2, tex.data(), tex.size(), hipMemcpyHostToDevice); a = ms(t); hipDeviceSynchronize(); printf("pageable H2D 29 MB again: returns %.3f ms, device done %.3f ms\n", a, ms(t));
    t = clk::now(); void* pin = nullptr; hipHostMalloc(&pin, 29 << 20, hipHostMallocDefault); printf("hipHostMalloc 29 MB %.3f ms\n", ms(t));
    t = clk::now(); memcpy(pin, tex.data(), tex.size()); printf("memcpy into pinned 29 MB %.3f ms\n", ms(t));
    t = clk::now(); hipMemcpyAsync(d2, pin, 29 << 20, hipMemcpyHostToDevice, nullptr); a = ms(t); hipDeviceSynchronize(); printf("pinned H2D 29 MB: returns %.3f ms, device done %.3f ms\n", a, ms(t));
    t = clk::now(); hipHostFree(pin); printf("hipHostFree 29 MB %.3f ms\n", ms(t));
    t = clk::now(); hipHostRegister(tex.data(), tex.size(), hipHostRegisterDefault); printf("hipHostRegister 29 MB %.3f ms\n", ms(t));
    t = clk::now(); hipMemcpyAsync(d2, tex.data(), 29 << 20, hipMemcpyHostToDevice, nullptr); a = ms(t); hipDeviceSynchronize(); printf("registered H2D 29 MB: returns %.3f ms, device done %.3f ms\n", a, ms(t));
    t = clk::now(); hipHostUnregister(tex.data()); printf("hipHostUnregister 29 MB %.3f ms\n", ms(t));
  }
  t = clk::now(); hipStreamCreateWithFlags(&s[2], hipStreamNonBlocking); printf("third stream %.3f ms\n", ms(t));
  t = clk::now(); hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, s[2], (int*)nullptr); hipStreamSynchronize(s[2]); printf("first launch on third stream %.3f ms\n", ms(t));
  t = clk::now(); hipHostRegister(big.data(), 2880000, hipHostRegisterDefault); printf("hipHostRegister 2.88 MB %.3f ms\n", ms(t));
  t = clk::now(); hipEvent_t e; hipEventCreate(&e); printf("hipEventCreate %.3f ms\n", ms(t));
  t = clk::now(); void* q; hipMalloc(&q, 147 << 20); printf("hipMalloc 147 MB %.3f ms\n", ms(t));
  return 0;
}
