// walk_sim.cpp — development analysis (not product): replay the megakernel's LOCK-STEP grid walk on the CPU.
//
// 64 simulated lanes form a wave exactly as rt_kernel.hip does: a pool of (pixel, sample) items of 4x4-pixel tiles,
// a lane that finishes a sample takes the next one at once, every lane with a ray enters hit_world together.  Per wave
// iteration the walk loop runs max-over-lanes rounds; this tool counts the rounds in which ANY lane moves (step
// rounds) / tests (test rounds) — the two numbers RT_PROFILE builds report as wave_step_iters_per_wave_iter and
// wave_test_iters_per_wave_iter — under alternative walk designs, before a GPU minute is spent on them:
//   * grid resolution (cells per gridded sphere),
//   * a skip field (Chebyshev distance to the nearest non-empty cell) with up to `max_cells` cells per step round,
//   * classes of rays that set the maximum (camera rays vs bounces).
// Results are identical by construction in every variant (the walk only selects candidates); the tool checks the hit
// against the product's per-lane reference walk (hit_world_grid) anyway.
//
//   g++ -O2 -std=c++17 -ffp-contract=off -fopenmp -DRT_DEV_KNOBS tools/analysis/walk_sim.cpp -Lrust-raytracer_amd -lrt_host \
//       -Wl,-rpath,$PWD/rust-raytracer_amd -o /tmp/walk_sim
//   /tmp/walk_sim scenes/cfg2_cover_1200x800_spp128.json 1200 800 8 [cells_per_sphere] [skip 0|1] [max_cells]
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../rust-raytracer_amd/csrc/hip/rt_tables.h"
using namespace rtc;

struct RoundLog {  // per lane: which rounds it moved / tested in
  uint64_t move = 0, test = 0;  // bit r = round r (rounds beyond 63 are folded into bit 63 and counted apart)
  uint32_t rounds = 0, steps = 0, tests = 0, over = 0;
};

struct SimGrid {
  bool dedupe_free = false;   // landing in a cell whose candidate is the sphere tested last drops it in the step block (no test round)
  int cull = 0;               // 0: the product; 1: candidates whose exact test cannot accept are dropped for free when their cell is entered (ideal cull); 2: the conservative f32 cull of rt_core.h; 3 / 4: the same two, realisable form (the two candidates of the cell word only; a culled-out cell ends the step round)
  const DevScene* ds;
  const CullPair* cull_table = nullptr;  // (host table: the device scene no longer carries the round-1 cull table)
  std::vector<uint8_t> skip;  // per padded cell: Chebyshev distance to the nearest non-empty cell or EXIT border, capped (0 for non-empty)
  int max_cells = 2;          // cells a lane may advance per step round
  bool use_skip = false;
};

static void lane_walk_cull(const SimGrid& sg, V3 o, V3 d, double& closest, int& best, RoundLog& log);
// the kernel's walk loop for ONE lane (rt_kernel.hip hit_world, part (3)), logging the rounds
static void lane_walk(const SimGrid& sg, V3 o, V3 d, double& closest, int& best, RoundLog& log) {
  const DevScene& sc = *sg.ds;
  const GridDesc& G = sc.grid;
  const RayK rk = ray_consts(d);
  if (sg.cull) { lane_walk_cull(sg, o, d, closest, best, log); return; }
  for (uint32_t i = 0; i < G.n_large; ++i) exact_hit_any_order(o, d, rk, sc.large_geom[i], sc.large[i], closest, best);
  if (G.n[0] == 0u) return;
  GridWalk w;
  const int mode = rk.fast ? grid_begin(G, o, d, w) : GRID_FALLBACK;
  if (mode == GRID_MISS) return;
  if (mode == GRID_FALLBACK) {
    for (uint32_t idx = 0; idx < sc.n_spheres; ++idx) exact_hit_any_order(o, d, rk, sc.geom[idx], idx, closest, best);
    return;
  }
  struct uint2_ { uint32_t x, y; };
  auto cell = [&](int lin) { uint2_ e; e.x = sc.cell_word[2 * lin]; e.y = sc.cell_word[2 * lin + 1]; return e; };
  float tm0 = w.tmax[0], tm1 = w.tmax[1], tm2 = w.tmax[2];
  const float dt0 = w.delta[0], dt1 = w.delta[1], dt2 = w.delta[2];
  const int dl0 = w.dl[0], dl1 = w.dl[1], dl2 = w.dl[2];
  int lin = w.lin;
  const double t0 = w.t0;
  uint32_t it, end, pend, last = 0xFFFFFFFFu;
  {
    const uint2_ e = cell(lin);
    it = e.x & CELL_START_MASK; end = it + (e.x >> CELL_COUNT_SHIFT); pend = e.y;
  }
  uint32_t r = 0;
  while (it <= end) {
    const uint64_t bit = 1ull << (r < 63 ? r : 63);
    if (it == end) {  // (a) move on
      log.move |= bit;
      float tc = (float)(closest - t0);
      tc = tc + fabsf(tc) * 2.384185791015625e-07f;
      const bool hit = best >= 0;
      if (hit && tc < rt_min3f(tm0, tm1, tm2)) { it = 1; end = 0; }
      else {
        // up to max_cells cells this round: stop at the first non-empty / exit cell, or when the hit lies in the cell
        for (int j = 1;; ++j) {
          const float tmin = rt_min3f(tm0, tm1, tm2);
          const bool ax = tm0 == tmin, ay = !ax && tm1 == tmin;
          tm0 += ax ? dt0 : 0.0f; tm1 += ay ? dt1 : 0.0f; tm2 += (!ax && !ay) ? dt2 : 0.0f;
          lin += ax ? dl0 : (ay ? dl1 : dl2);
          log.steps++;
          const uint2_ e = cell(lin);
          const bool exitc = e.x == CELL_EXIT, empty = (e.x >> CELL_COUNT_SHIFT) == 0u;
          if (exitc) { it = 1; end = 0; break; }
          if (!empty) { it = e.x & CELL_START_MASK; end = it + (e.x >> CELL_COUNT_SHIFT); pend = e.y; break; }
          if (j >= sg.max_cells) { it = 0; end = 0; break; }  // empty cell, out of budget: keep moving next round (which re-checks `done`)
          if (hit && tc < rt_min3f(tm0, tm1, tm2)) { it = 1; end = 0; break; }  // the closest hit lies inside this (empty) cell
        }
      }
    }
    if (it < end) {  // (b) one exact test
      log.test |= bit;
      uint32_t idx = pend & 0xFFFFu;
      if (idx == 0xFFFFu) idx = sc.cell_items[it];
      pend = (pend >> 16) | 0xFFFF0000u;
      it++;
      if (idx != last) { last = idx; log.tests++; exact_hit_any_order(o, d, rk, sc.geom[idx], idx, closest, best); }
    }
    r++;
    if (r > 63) log.over++;
  }
  log.rounds = r;
}

// Variant: an f32 (or ideal) cull of a cell's candidates INSIDE the step block — a cell whose candidates all fail counts as
// empty (the lane moves on within its budget), the survivors take one exact-test round each.
static void lane_walk_cull(const SimGrid& sg, V3 o, V3 d, double& closest, int& best, RoundLog& log) {
  const DevScene& sc = *sg.ds;
  const GridDesc& G = sc.grid;
  const RayK rk = ray_consts(d);
  for (uint32_t i = 0; i < G.n_large; ++i) exact_hit_any_order(o, d, rk, sc.geom[sc.large[i]], sc.large[i], closest, best);
  if (G.n[0] == 0u) return;
  GridWalk w;
  const int mode = rk.fast ? grid_begin(G, o, d, w) : GRID_FALLBACK;
  if (mode == GRID_MISS) return;
  if (mode == GRID_FALLBACK) {
    for (uint32_t idx = 0; idx < sc.n_spheres; ++idx) exact_hit_any_order(o, d, rk, sc.geom[idx], idx, closest, best);
    return;
  }
  const RayF32 rf = make_ray_f32(o, d);
  auto survives = [&](uint32_t idx) {
    if (sg.cull == 8) return true;
    const SphereGeom& g = sc.geom[idx];
    if (sg.cull == 2 || sg.cull == 4) {
      const CullPair& cp = sg.cull_table[idx / 2];
      return cull_pass(cull_disc(rf, cp.cx[idx & 1], cp.cy[idx & 1], cp.cz[idx & 1], cp.R[idx & 1]));
    }
    const V3 oc = sub(o, v3(g.cx, g.cy, g.cz));
    const double half_b = dot(oc, d), c = length_squared(oc) - g.r * g.r;
    return !(c > 0.0 && half_b > 0.0) && (half_b * half_b) - (rk.a * c) >= 0.0;
  };
  float tm0 = w.tmax[0], tm1 = w.tmax[1], tm2 = w.tmax[2];
  const float dt0 = w.delta[0], dt1 = w.delta[1], dt2 = w.delta[2];
  const int dl0 = w.dl[0], dl1 = w.dl[1], dl2 = w.dl[2];
  int lin = w.lin;
  const double t0 = w.t0;
  uint32_t todo[64], n_todo = 0, last = 0xFFFFFFFFu, last_tested = 0xFFFFFFFFu;
  auto enter = [&](int l) {  // candidates of cell l that survive the cull -> todo
    n_todo = 0;
    const uint32_t ex = sc.cell_word[2 * l], ey = sc.cell_word[2 * l + 1];
    if (ex == CELL_EXIT) return;
    const uint32_t first = ex & CELL_START_MASK, cnt = ex >> CELL_COUNT_SHIFT;
    for (uint32_t k = 0; k < cnt && n_todo < 64; ++k) {
      uint32_t idx = k < 2 ? (ey >> (16 * k)) & 0xFFFFu : 0xFFFFu;
      if (idx == 0xFFFFu) idx = sc.cell_items[first + k];
      if (idx == last && (sg.dedupe_free || sg.cull < 6)) continue;   // (modes 1-5 and dedupe_free: a repeated sphere is dropped on landing; else it costs its test round like in the product)
      last = idx;
      if ((sg.cull >= 3 && k >= 2) || sg.cull >= 6 || survives(idx)) todo[n_todo++] = idx;   // (3 / 4: only the two candidates named in the cell word are culled)
    }
  };
  if (sg.cull == 6 || sg.cull == 7) {  // 6 / 7: ONLY the start cell's first candidate (7: first two) is culled, nothing inside the loop
    const uint32_t ex = sc.cell_word[2 * lin], ey = sc.cell_word[2 * lin + 1];
    const uint32_t first = ex & CELL_START_MASK, cnt = ex == CELL_EXIT ? 0u : ex >> CELL_COUNT_SHIFT;
    for (uint32_t k = 0; k < cnt; ++k) {
      uint32_t idx = k < 2 ? (ey >> (16 * k)) & 0xFFFFu : 0xFFFFu;
      if (idx == 0xFFFFu) idx = sc.cell_items[first + k];
      if (k < (sg.cull == 6 ? 1u : 2u) && !survives(idx)) { last_tested = idx; continue; }
      todo[n_todo++] = idx;
    }
  } else if (sg.cull == 5) {  // 5: like 3, but the cell the ray STARTS in is not culled (its candidates all take a test round)
    const uint32_t ex = sc.cell_word[2 * lin], ey = sc.cell_word[2 * lin + 1];
    const uint32_t first = ex & CELL_START_MASK, cnt = ex == CELL_EXIT ? 0u : ex >> CELL_COUNT_SHIFT;
    for (uint32_t k = 0; k < cnt; ++k) {
      uint32_t idx = k < 2 ? (ey >> (16 * k)) & 0xFFFFu : 0xFFFFu;
      if (idx == 0xFFFFu) idx = sc.cell_items[first + k];
      todo[n_todo++] = idx; last = idx;
    }
  } else enter(lin);
  bool walking = true, moving = n_todo == 0;
  uint32_t r = 0;
  while (walking) {
    const uint64_t bit = 1ull << (r < 63 ? r : 63);
    if (moving) {
      log.move |= bit;
      float tc = (float)(closest - t0);
      tc = tc + fabsf(tc) * 2.384185791015625e-07f;
      const bool hit = best >= 0;
      if (hit && tc < rt_min3f(tm0, tm1, tm2)) walking = false;
      else {
        for (int j = 1;; ++j) {
          const float tmin = rt_min3f(tm0, tm1, tm2);
          const bool ax = tm0 == tmin, ay = !ax && tm1 == tmin;
          tm0 += ax ? dt0 : 0.0f; tm1 += ay ? dt1 : 0.0f; tm2 += (!ax && !ay) ? dt2 : 0.0f;
          lin += ax ? dl0 : (ay ? dl1 : dl2);
          log.steps++;
          if (sc.cell_word[2 * lin] == CELL_EXIT) { walking = false; break; }
          enter(lin);
          if (n_todo) { moving = false; break; }
          if (sg.cull >= 3 && sg.cull < 6 && (sc.cell_word[2 * lin] >> CELL_COUNT_SHIFT) != 0u) break;  // a cell whose candidates were all culled ends the round
          if (j >= sg.max_cells) break;  // out of budget: keep moving next round
          if (hit && tc < rt_min3f(tm0, tm1, tm2)) { walking = false; break; }
        }
      }
    }
    if (walking && !moving && n_todo) {  // one exact test (in the round it arrived in, like the product)
      log.test |= bit;
      const uint32_t idx = todo[0]; for (uint32_t q = 1; q < n_todo; ++q) todo[q - 1] = todo[q]; --n_todo;
      if (idx != last_tested) { last_tested = idx; log.tests++; exact_hit_any_order(o, d, rk, sc.geom[idx], idx, closest, best); }
      if (!n_todo) moving = true;
    }
    r++;
    if (r > 63) log.over++;
  }
  log.rounds = r;
}

int main(int argc, char** argv) {
  if (argc < 5) { std::fprintf(stderr, "usage: walk_sim scene.json width height spp [cells_per_sphere] [skip] [max_cells]\n"); return 1; }
  RtSceneFile* sf = nullptr;
  if (rt_scene_load_file(argv[1], &sf) != RT_OK) { std::fprintf(stderr, "%s\n", rt_host_last_error()); return 1; }
  RtScene* scp = rt_scene_get_mut(sf);
  scp->width = atoi(argv[2]); scp->height = atoi(argv[3]); scp->samples_per_pixel = atoi(argv[4]);
  const double cps = argc > 5 ? atof(argv[5]) : 0.0;
  const bool use_skip = argc > 6 && atoi(argv[6]) != 0;
  const int max_cells = argc > 7 ? atoi(argv[7]) : 2;
  const int cull = argc > 8 ? atoi(argv[8]) : 0;
  const RtScene& sc = *scp;
  HostTables t;
  if (cps > 0.0) { char b[64]; snprintf(b, sizeof b, "%g", cps); setenv("RT_GRID_CELLS_PER_SPHERE", b, 1); }
  const std::string why = build_tables(sc, t, true);
  if (!why.empty()) { std::fprintf(stderr, "%s\n", why.c_str()); return 1; }
  build_texels(sc, t);
  DevScene ds; fill_dev_scene(sc, t, ds);
  ds.tex4 = t.tex4.data(); ds.sky4 = t.sky4.data();
  ds.geom = t.geom.data(); ds.mat = t.mat.data(); ds.lights = t.lights.data(); ds.sky = sc.sky_rgb8;
  if (t.grid.wide) { std::fprintf(stderr, "walk_sim: this world builds a WIDE grid (32-bit item lists, four-word cells): the simulator decodes packed cells only\n"); return 2; }
  ds.matc = t.matc.data(); ds.cell_word = t.cell_word.data(); ds.cell_items = t.cell_items.data(); ds.large = t.large.data(); ds.large_geom = t.large_geom.data();
  std::vector<uint8_t> blob(t.tex_bytes ? t.tex_bytes : 1);
  for (uint32_t i = 0; i < sc.n_textures; ++i) std::memcpy(&blob[t.tex_off[i]], sc.textures[i].rgb8, sc.textures[i].nbytes);
  ds.tex = blob.data();
  const GridDesc& G = ds.grid;
  {
    uint32_t nonempty = 0, inner = G.n[0] * G.n[1] * G.n[2];
    for (uint32_t c = 0; c < G.n_cells; ++c) { const uint32_t w = t.cell_word[2 * c]; if (w != CELL_EXIT && (w >> CELL_COUNT_SHIFT)) nonempty++; }
    std::printf("grid %ux%ux%u, %u cells (%u padded), %u items, %u large, non-empty %.1f %%, cell bytes %u\n", G.n[0], G.n[1], G.n[2], inner, G.n_cells,
                G.n_items, G.n_large, 100.0 * nonempty / inner, G.n_cells * 8u);
  }
  SimGrid sg; sg.ds = &ds; sg.cull_table = t.cull.data(); sg.max_cells = max_cells; sg.use_skip = use_skip; sg.cull = cull; sg.dedupe_free = argc > 9 && atoi(argv[9]) != 0;
  const GlobalTables tb{ds.geom, ds.matc};
  const uint32_t TW = 4, TH = 4, NPX = TW * TH;
  const uint32_t tx = (sc.width + TW - 1) / TW, ty = (sc.height + TH - 1) / TH, n_tiles = tx * ty;
  const uint32_t chunk_spp = std::min<uint32_t>(16u, sc.samples_per_pixel), n_chunks = (sc.samples_per_pixel + chunk_spp - 1) / chunk_spp;
  const uint32_t n_items = n_tiles * n_chunks;
  const uint32_t ITEMS_PER_WAVE = 60;
  const uint32_t n_waves = (n_items + ITEMS_PER_WAVE - 1) / ITEMS_PER_WAVE;
  double w_iters = 0, step_rounds = 0, test_rounds = 0, lane_segs = 0, lane_steps = 0, lane_tests = 0, tot_rounds = 0;
  double cls_segs[2] = {0, 0}, cls_rounds[2] = {0, 0}, cls_steps[2] = {0, 0}, cls_tests[2] = {0, 0}, max_by_cls[2] = {0, 0};
  std::vector<double> hist_wave(65, 0.0), hist_lane0(65, 0.0), hist_lane1(65, 0.0);
  unsigned long long mismatches = 0;
  double g_kl_sum[6] = {0, 0, 0, 0, 0, 0}, g_kl_any[6] = {0, 0, 0, 0, 0, 0};
  // (round 6) K RAYS PER LANE: a lane walks K independent segments one after the other inside ONE lock-step loop, so the wave pays
  // max over lanes of the SUM of their rounds instead of the sum over K iterations of the max.  Estimated from K consecutive
  // iterations of the same simulated wave (lane l's K segments = its segments of those iterations): g_multi[k][0] += sum of the K maxima
  // (what the product pays), g_multi[k][1] += max over lanes of the K-sums (what K rays per lane would pay), k = 2, 4.
  double g_multi[5][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}};
#pragma omp parallel
  {
    std::vector<double> hw(65, 0.0), hl0(65, 0.0), hl1(65, 0.0);
    double a_w = 0, a_s = 0, a_t = 0, a_ls = 0, a_lst = 0, a_lt = 0, a_tr = 0, c_s[2] = {0, 0}, c_r[2] = {0, 0}, c_st[2] = {0, 0}, c_t[2] = {0, 0}, m_c[2] = {0, 0};
    unsigned long long mm = 0;
    double kl_sum[6] = {0, 0, 0, 0, 0, 0}, kl_any[6] = {0, 0, 0, 0, 0, 0};
    double multi[5][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}};
#pragma omp for schedule(dynamic, 4)
    for (uint32_t wv = 0; wv < n_waves; ++wv) {
      Lane<false, false> L[64];
      bool has_ray[64];
      for (int l = 0; l < 64; ++l) { std::memset(&L[l], 0, sizeof L[l]); has_ray[l] = false; L[l].ra.k0 = ds.seed_lo; L[l].ra.k1 = ds.seed_hi; fwd_init(L[l].fwd); }
      uint32_t item = wv * ITEMS_PER_WAVE, item_end = std::min(n_items, item + ITEMS_PER_WAVE);
      uint32_t it_next = 0, it_total = 0, it_tile = 0, it_sbeg = 0;
      auto take = [&](int l) -> bool {
        for (;;) {
          while (it_next >= it_total) {
            if (item >= item_end) return false;
            // queue order of the product: bottom of the image first
            const uint32_t q = item / n_chunks, ch = item % n_chunks;
            it_tile = n_tiles - 1 - q; it_sbeg = ch * chunk_spp;
            const uint32_t cnt = std::min(chunk_spp, sc.samples_per_pixel - it_sbeg);
            it_total = NPX * cnt; it_next = 0; item++;
          }
          const uint32_t wq = it_next++;
          const uint32_t p = wq % NPX, s = it_sbeg + wq / NPX;
          const uint32_t px = (it_tile % tx) * TW + p % TW, py = (it_tile / tx) * TH + p / TW;
          if (px >= sc.width || py >= sc.height) continue;
          L[l].s = s; L[l].ra.pixel = py * sc.width + px; L[l].ra.sample = s;
          lane_begin_sample(ds, L[l], px, py);
          return true;
        }
      };
      for (int l = 0; l < 64; ++l) has_ray[l] = take(l);
      uint32_t hist_r[4][64]; uint32_t hist_max[4]; uint32_t n_hist = 0;   // the last 4 iterations' per-lane rounds
      for (;;) {
        int live = 0;
        for (int l = 0; l < 64; ++l) live += has_ray[l];
        if (!live) break;
        uint64_t any_move = 0, any_test = 0;
        uint32_t max_r[2] = {0, 0}, over = 0;
        uint32_t kind_lanes[6] = {0, 0, 0, 0, 0, 0};  // lanes of this iteration by what their segment ends in: [0] miss, [1 + RT_MAT_*] hit material
        uint32_t* cur_r = hist_r[n_hist & 3u];
        for (int l = 0; l < 64; ++l) cur_r[l] = 0;
        for (int l = 0; l < 64; ++l) {
          if (!has_ray[l]) continue;
          RoundLog lg;
          double closest = T_MAX; int best = -1;
          lane_walk(sg, L[l].o, L[l].d, closest, best, lg);
#ifdef WALK_SIM_CHECK
          { double c2 = T_MAX; int b2 = -1; uint32_t ne = 0, ns = 0; hit_world_grid(ds, tb, L[l].o, L[l].d, c2, b2, ne, ns); if (b2 != best || (b2 >= 0 && c2 != closest)) mm++; }
#endif
          any_move |= lg.move; any_test |= lg.test; over = std::max(over, lg.over);
          cur_r[l] = lg.rounds;
          const int cls = L[l].k == 0 ? 0 : 1;
          c_s[cls] += 1; c_r[cls] += lg.rounds; c_st[cls] += lg.steps; c_t[cls] += lg.tests;
          max_r[cls] = std::max(max_r[cls], lg.rounds);
          (cls ? hl1 : hl0)[std::min<uint32_t>(lg.rounds, 64)] += 1;
          a_ls += 1; a_lst += lg.steps; a_lt += lg.tests;
          kind_lanes[best < 0 ? 0 : 1 + std::min<uint32_t>(t.matc[best].kind, 4u)]++;
          const bool fin = lane_shade(ds, tb, L[l], best, closest) == LANE_FINISHED;
          if (fin) has_ray[l] = take(l);
        }
        const uint32_t mr = std::max(max_r[0], max_r[1]);
        hist_max[n_hist & 3u] = mr; n_hist++;
        for (uint32_t K : {2u, 4u})
          if (n_hist % K == 0) {
            uint32_t sum_of_max = 0, max_of_sum = 0;
            for (uint32_t j = 0; j < K; ++j) sum_of_max += hist_max[(n_hist - 1 - j) & 3u];
            for (int l = 0; l < 64; ++l) { uint32_t sm = 0; for (uint32_t j = 0; j < K; ++j) sm += hist_r[(n_hist - 1 - j) & 3u][l]; max_of_sum = std::max(max_of_sum, sm); }
            multi[K][0] += sum_of_max; multi[K][1] += max_of_sum;
          }
        a_w += 1; a_s += __builtin_popcountll(any_move) + over; a_t += __builtin_popcountll(any_test) + over; a_tr += mr;
        hw[std::min<uint32_t>(mr, 64)] += 1;
        for (int q = 0; q < 6; ++q) { kl_sum[q] += kind_lanes[q]; kl_any[q] += kind_lanes[q] != 0; }
        m_c[max_r[1] > max_r[0] ? 1 : 0] += 1;
      }
    }
#pragma omp critical
    {
      w_iters += a_w; step_rounds += a_s; test_rounds += a_t; lane_segs += a_ls; lane_steps += a_lst; lane_tests += a_lt; tot_rounds += a_tr; mismatches += mm;
      for (int c = 0; c < 2; ++c) { cls_segs[c] += c_s[c]; cls_rounds[c] += c_r[c]; cls_steps[c] += c_st[c]; cls_tests[c] += c_t[c]; max_by_cls[c] += m_c[c]; }
      for (int i = 0; i < 65; ++i) { hist_wave[i] += hw[i]; hist_lane0[i] += hl0[i]; hist_lane1[i] += hl1[i]; }
      for (int q = 0; q < 6; ++q) { g_kl_sum[q] += kl_sum[q]; g_kl_any[q] += kl_any[q]; }
      for (int k = 0; k < 5; ++k) { g_multi[k][0] += multi[k][0]; g_multi[k][1] += multi[k][1]; }
    }
  }
  for (int K : {2, 4})
    std::printf("K = %d rays per lane (K consecutive iterations of a wave folded into one lock-step loop): rounds %.0f -> %.0f = %.3f x  (the walk is ~41 %% of a wave iteration)\n", K,
                g_multi[K][0], g_multi[K][1], g_multi[K][1] / std::max(1.0, g_multi[K][0]));
  std::printf("wave iterations %.0f, lanes with a ray per iteration %.2f\n", w_iters, lane_segs / w_iters);
  std::printf("per lane segment: steps %.3f  gridded tests %.3f\n", lane_steps / lane_segs, lane_tests / lane_segs);
  std::printf("per wave iteration: rounds %.3f  step rounds %.3f  test rounds %.3f   (model cost 50*S + 70*T = %.0f instr)\n", tot_rounds / w_iters,
              step_rounds / w_iters, test_rounds / w_iters, (50 * step_rounds + 70 * test_rounds) / w_iters);
  for (int c = 0; c < 2; ++c)
    std::printf("  class %s: %.1f %% of segments, rounds/lane %.2f steps %.2f tests %.2f; sets the wave's maximum in %.1f %% of iterations\n", c ? "bounce (k>=1)" : "camera (k=0)",
                100 * cls_segs[c] / lane_segs, cls_rounds[c] / cls_segs[c], cls_steps[c] / cls_segs[c], cls_tests[c] / cls_segs[c], 100 * max_by_cls[c] / w_iters);
  {
    static const char* names[6] = {"miss (sky)", "kind 0", "kind 1", "kind 2", "kind 3", "kind 4"};
    std::printf("what the segments of a wave iteration end in (RT_MAT_* of include/rt_abi.h): lanes per iteration | share of iterations with at least one such lane\n");
    for (int q = 0; q < 6; ++q)
      if (g_kl_sum[q] > 0) std::printf("  %-12s %6.2f lanes   %5.1f %%\n", names[q], g_kl_sum[q] / w_iters, 100.0 * g_kl_any[q] / w_iters);
  }
  std::printf("rounds histogram (share of wave iterations | camera lanes | bounce lanes):\n");
  for (int i = 0; i < 65; ++i)
    if (hist_wave[i] + hist_lane0[i] + hist_lane1[i] > 0)
      std::printf("  %2d%s  %6.2f %%  %6.2f %%  %6.2f %%\n", i, i == 64 ? "+" : " ", 100 * hist_wave[i] / w_iters, 100 * hist_lane0[i] / std::max(1.0, cls_segs[0]),
                  100 * hist_lane1[i] / std::max(1.0, cls_segs[1]));
#ifdef WALK_SIM_CHECK
  std::printf("hit mismatches against hit_world_grid: %llu\n", mismatches);
#endif
  return 0;
}
