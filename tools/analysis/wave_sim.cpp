// wave_sim.cpp — development analysis (not product): replay the megakernel's lockstep
// schedule on the CPU (64 lanes = one 8x8 tile, one ray segment per iteration) to estimate
// SIMD utilisation: live-lane fraction, per-iteration max of cull survivors (the confirm loop
// runs max-over-lanes iterations), exact tests, and what cheap extra cull tests would save.
//   g++ -O2 -std=c++17 -ffp-contract=off -mfma -fopenmp tools/analysis/wave_sim.cpp -Lrust-raytracer_amd -lrt_host -o /tmp/wave_sim
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../../rust-raytracer_amd/csrc/hip/rt_tables.h"
using namespace rtc;

int main(int argc, char** argv) {
  if (argc < 5) { std::fprintf(stderr, "usage: wave_sim scene.json width height spp\n"); return 1; }
  RtSceneFile* sf = nullptr;
  if (rt_scene_load_file(argv[1], &sf) != RT_OK) { std::fprintf(stderr, "%s\n", rt_host_last_error()); return 1; }
  RtScene* scp = rt_scene_get_mut(sf);
  scp->width = atoi(argv[2]); scp->height = atoi(argv[3]); scp->samples_per_pixel = atoi(argv[4]);
  const RtScene& sc = *scp;
  HostTables t; build_tables(sc, t);
  build_texels(sc, t);
  DevScene ds; fill_dev_scene(sc, t, ds);
  ds.tex4 = t.tex4.data(); ds.sky4 = t.sky4.data();
  ds.geom = t.geom.data(); ds.mat = t.mat.data(); ds.cull = t.cull.data(); ds.lights = t.lights.data(); ds.sky = sc.sky_rgb8;
  std::vector<uint8_t> blob(t.tex_bytes ? t.tex_bytes : 1);
  for (uint32_t i = 0; i < sc.n_textures; ++i) std::memcpy(&blob[t.tex_off[i]], sc.textures[i].rgb8, sc.textures[i].nbytes);
  ds.tex = blob.data();
  const uint32_t tx = (sc.width + 7) / 8, ty = (sc.height + 7) / 8;
  double iters = 0, live_lane_iters = 0, sum_max_cand = 0, sum_cand = 0, sum_max_cand_behind = 0, sum_cand_behind = 0, pair_events = 0;
  double sum_max_attempts = 0, sum_attempts = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : iters, live_lane_iters, sum_max_cand, sum_cand, sum_max_cand_behind, sum_cand_behind, pair_events)
  for (uint32_t tile = 0; tile < tx * ty; ++tile) {
    Lane<false> L[64]; bool alive[64], need_new[64]; uint32_t px[64], py[64];
    for (int l = 0; l < 64; ++l) {
      std::memset(&L[l], 0, sizeof(L[l]));
      px[l] = (tile % tx) * 8 + (l & 7); py[l] = (tile / tx) * 8 + (l >> 3);
      alive[l] = px[l] < sc.width && py[l] < sc.height; need_new[l] = true;
      L[l].ra.pixel = py[l] * sc.width + px[l]; L[l].ra.k0 = ds.seed_lo; L[l].ra.k1 = ds.seed_hi;
    }
    std::vector<uint8_t> pair_hit(t.n_pairs);
    for (;;) {
      int nlive = 0;
      for (int l = 0; l < 64; ++l) {
        if (alive[l] && need_new[l]) { if (L[l].s >= sc.samples_per_pixel) alive[l] = false; else { lane_begin_sample(ds, L[l], px[l], py[l]); need_new[l] = false; } }
        nlive += alive[l];
      }
      if (!nlive) break;
      iters += 1; live_lane_iters += nlive;
      int maxc = 0, maxcb = 0;
      std::fill(pair_hit.begin(), pair_hit.end(), 0);
      for (int l = 0; l < 64; ++l) {
        if (!alive[l]) continue;
        const double a = length_squared(L[l].d);
        const RayF32 rf = make_ray_f32(L[l].o, L[l].d);
        double closest = T_MAX; int best = -1, c = 0, cb = 0;
        for (uint32_t i = 0; i < sc.n_spheres; ++i) {
          const CullPair& cp = t.cull[i / 2];
          if (!cull_pass(cull_disc(rf, cp.cx[i & 1], cp.cy[i & 1], cp.cz[i & 1], cp.R[i & 1]))) continue;
          c++; pair_hit[i / 2] = 1;
          // would an extra "behind the ray" f32 test have removed it?  b>0 (centre behind) and origin clearly outside
          float ocx = rf.ox - cp.cx[i & 1], ocy = rf.oy - cp.cy[i & 1], ocz = rf.oz - cp.cz[i & 1];
          float b = ocx * rf.dx + ocy * rf.dy + ocz * rf.dz, q = ocx * ocx + ocy * ocy + ocz * ocz;
          if (!(b > 0.f && q > cp.R[i & 1] * 1.001f + 1e-3f)) cb++;
          double r = exact_root(L[l].o, L[l].d, a, t.geom[i], T_MIN, closest);
          if (r >= 0.0) { closest = r; best = (int)i; }
        }
        sum_cand += c; sum_cand_behind += cb; maxc = std::max(maxc, c); maxcb = std::max(maxcb, cb);
        need_new[l] = lane_shade(ds, L[l], best, closest);
        if (need_new[l]) L[l].s += 1;
      }
      sum_max_cand += maxc; sum_max_cand_behind += maxcb;
      for (auto h : pair_hit) pair_events += h;
    }
  }
  std::printf("wave-iterations %.0f  live lanes/iter %.2f (util %.1f%%)\n", iters, live_lane_iters / iters, 100 * live_lane_iters / iters / 64);
  std::printf("cull survivors: mean/lane-iter %.2f  max-over-lanes/iter %.2f   with behind-test: mean %.2f max %.2f\n",
              sum_cand / live_lane_iters, sum_max_cand / iters, sum_cand_behind / live_lane_iters, sum_max_cand_behind / iters);
  std::printf("pairs with >=1 surviving lane per iteration: %.1f of %u\n", pair_events / iters, t.n_pairs);
  return 0;
}
