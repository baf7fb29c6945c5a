// fuzz_tables.cpp — development tool: adversarial sphere sets through build_tables / build_grid and the grid walk
// (the CPU build of rt_core.h that tests/hostsim uses), under ASan/UBSan.  Every frame is also rendered with the
// grid switched off (brute force) and the two must agree bit for bit.
//   g++ -O1 -g -std=c++17 -ffp-contract=off -fsanitize=address,undefined -fopenmp tools/fuzz/fuzz_tables.cpp tests/hostsim/hostsim.cpp \
//       -Iinclude -o /tmp/fuzz_tables && /tmp/fuzz_tables <seed> <iterations>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <random>
#include <vector>
#include <pthread.h>
#include <signal.h>
#include <thread>
#include <chrono>
#include <atomic>
#include "../../include/rt_abi.h"
extern "C" int hostsim_render(const RtScene* scene, const RtRowTiles* tiles, uint8_t* rgb8, float* linear, RtStats* stats, int use_cull);

static std::atomic<int> g_iter{-1};
int main(int argc, char** argv) {
  // watchdog (FUZZ_WATCHDOG=seconds, run with OMP_NUM_THREADS=1): an iteration that does not finish gets SIGSEGV on the
  // main thread, so that ASan prints where it was stuck
  if (const char* wd = std::getenv("FUZZ_WATCHDOG")) {
    const int limit = atoi(wd);
    const pthread_t main_thread = pthread_self();
    std::thread([limit, main_thread]() {
      int last = -2, age = 0;
      for (;;) {
        std::this_thread::sleep_for(std::chrono::seconds(1));
        const int cur = g_iter.load();
        if (cur == last) { if (++age >= limit) { std::fprintf(stderr, "watchdog: iteration %d stuck\n", cur); pthread_kill(main_thread, SIGSEGV); return; } }
        else { last = cur; age = 0; }
      }
    }).detach();
  }
  std::mt19937_64 rng(argc > 1 ? atoll(argv[1]) : 1);
  const int iters = argc > 2 ? atoi(argv[2]) : 100;
  auto U = [&](double a, double b) { return a + (b - a) * (double)(rng() >> 11) * 0x1p-53; };
  int n_diff = 0, n_rejected = 0;
  for (int it = 0; it < iters; ++it) {
    g_iter = it;
    const uint32_t n = (rng() % 4 == 0) ? (uint32_t)(rng() % 30) : (uint32_t)(24 + rng() % 1500);
    std::vector<RtSphere> sp(n);
    uint32_t n_lights = 0;
    const int flavour = (int)(rng() % 8);
    const double scale = std::pow(10.0, U(-3, flavour == 1 ? 12 : 3));
    for (uint32_t i = 0; i < n; ++i) {
      RtSphere& s = sp[i];
      std::memset(&s, 0, sizeof s);
      for (int k = 0; k < 3; ++k) s.center[k] = U(-1, 1) * scale * (flavour == 2 && k == 1 ? 1e-9 : 1.0);
      if (flavour == 3) s.center[1] = 0.0;                 // a plane of spheres: one-cell-high grid
      if (flavour == 4 && i) { for (int k = 0; k < 3; ++k) s.center[k] = sp[0].center[k]; }  // all coincident
      s.radius = U(0.001, 0.2) * scale * (rng() % 16 == 0 ? -1.0 : 1.0);
      int odd = (int)(rng() % 200);
      if (const char* m = std::getenv("FUZZ_ODD_MASK")) { if (odd < 9 && !((atoi(m) >> odd) & 1)) odd = 100; }
      if (odd < 9 && std::getenv("FUZZ_VERBOSE")) std::printf("  sphere %u odd %d\n", i, odd);
      if (odd == 0) s.radius = 0.0;
      if (odd == 1) s.radius = std::numeric_limits<double>::quiet_NaN();
      if (odd == 2) s.radius = std::numeric_limits<double>::infinity();
      if (odd == 3) s.center[rng() % 3] = std::numeric_limits<double>::quiet_NaN();
      if (odd == 4) s.center[rng() % 3] = -std::numeric_limits<double>::infinity();
      if (odd == 5) s.radius = 1e-310;
      if (odd == 6) s.radius = 1e300;
      if (odd == 7) s.center[rng() % 3] = 1e308;
      if (odd == 8) s.radius = scale * 50;
      // (at most two lights: ray_color's light recursion is a branching process with 0.1 n_lights^2 x occlusion offspring
      //  per level — the reference overflows its stack on such scenes, here the nesting cap makes them merely very slow)
      s.kind = (uint32_t)(rng() % 10 < 6 ? RT_MAT_LAMBERTIAN : (rng() % 3 == 0 ? RT_MAT_GLASS : (n_lights < 2 && rng() % 20 == 0 ? RT_MAT_LIGHT : RT_MAT_METAL)));
      if (s.kind == RT_MAT_LIGHT) n_lights++;
      s.albedo[0] = (float)U(0, 1); s.albedo[1] = (float)U(0, 1); s.albedo[2] = (float)U(0, 1.2);
      s.fuzz_or_ior = s.kind == RT_MAT_GLASS ? 1.5 : U(0, 0.5);
    }
    RtScene sc;
    std::memset(&sc, 0, sizeof sc);
    sc.abi_version = RT_ABI_VERSION; sc.width = 24; sc.height = 16; sc.samples_per_pixel = 2; sc.max_depth = 6;
    sc.sky_mode = RT_SKY_GRADIENT;
    const double d = scale * 3;
    sc.cam_origin[0] = d; sc.cam_origin[1] = d * 0.3; sc.cam_origin[2] = d * 0.2;
    if (rng() % 8 == 0) { sc.cam_origin[0] = sp.empty() ? 0 : sp[0].center[0]; sc.cam_origin[1] = sp.empty() ? 0 : sp[0].center[1]; sc.cam_origin[2] = sp.empty() ? 0 : sp[0].center[2]; }
    sc.cam_lower_left[0] = -d; sc.cam_lower_left[1] = -d * 0.6 ; sc.cam_lower_left[2] = -d * 0.9;
    sc.cam_horizontal[2] = d * 1.8; sc.cam_vertical[1] = d * 1.2;
    sc.spheres = sp.data(); sc.n_spheres = n; sc.seed = rng();
    std::vector<uint8_t> a(24 * 16 * 3), b(24 * 16 * 3);
    std::vector<float> la(24 * 16 * 3), lb(24 * 16 * 3);
    RtStats sa, sb;
    if (std::getenv("FUZZ_VERBOSE")) { std::printf("iter %d flavour %d n %u scale %g\n", it, flavour, n, scale); std::fflush(stdout); }
    const int ra = hostsim_render(&sc, nullptr, a.data(), la.data(), &sa, 3);   // the product's hit_world: large list + grid walk
    const int rb = hostsim_render(&sc, nullptr, b.data(), lb.data(), &sb, 0);   // every sphere, object order
    if (ra != rb) { std::printf("iter %d: rc %d vs %d\n", it, ra, rb); return 1; }
    if (ra != 0) { n_rejected++; continue; }
    if (std::memcmp(a.data(), b.data(), a.size()) || std::memcmp(la.data(), lb.data(), la.size() * 4) || sa.segments != sb.segments) {
      std::printf("iter %d (seed %s, flavour %d, n %u, scale %g): grid walk != brute force\n", it, argv[1], flavour, n, scale);
      n_diff++;
    }
  }
  std::printf("done: %d iterations, %d rejected scenes, %d mismatches\n", iters, n_rejected, n_diff);
  return n_diff ? 1 : 0;
}
