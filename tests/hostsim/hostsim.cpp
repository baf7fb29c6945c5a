// hostsim.cpp — DEVELOPMENT/TEST TOOL, not part of the product.
// Compiles the megakernel's per-lane logic (rust-raytracer_amd/csrc/hip/rt_core.h) for the
// CPU with a trivial one-lane-at-a-time driver, so the state machine, RNG addressing, cull
// margin and forward colour map can be checked against the oracle without a GPU.  The
// product never loads this; librt_hip.so has no CPU fallback.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../rust-raytracer_amd/csrc/hip/rt_tables.h"

using namespace rtc;

namespace {
unsigned long long* g_hist = nullptr;  // [0..63] steps per segment, [64..127] exact tests per segment, [128..] path lengths (diagnostics; 512 entries)
template <bool HL, bool SHORT_MAP = false>
void render_rows(const RtScene& sc, const HostTables& t, const DevScene& ds, const RtRowTiles* tiles, uint8_t* rgb8,
                 float* linear, RtStats* stats, int use_cull_flags) {
  const int use_cull = use_cull_flags & 15;
  const uint32_t rows = rt_tiles_local_rows(sc.height, tiles);
  uint64_t segs = 0, exact = 0, oob = 0, cull_false_reject = 0, steps = 0, cam_steps = 0, cam_segs = 0, cam_exact = 0;
  const GlobalTables tb{ds.geom, ds.matc};
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : segs, exact, oob, cull_false_reject, steps, cam_steps, cam_segs, cam_exact)
  for (uint32_t lr = 0; lr < rows; ++lr) {
    const uint32_t y = rt_tiles_global_row(tiles, lr);
    for (uint32_t x = 0; x < sc.width; ++x) {
      Lane<HL, SHORT_MAP> L;
      std::memset(&L, 0, sizeof L);
      LightStack<HL> light_stack;
      LightParked light_parked;
      lane_attach_light_state(L, light_stack, &light_parked);
      float acc[3] = {0.f, 0.f, 0.f};                 // accum 0: the reference's sequential f32 sum
      unsigned long long facc[3] = {0ull, 0ull, 0ull};  // accum 1: exact fixed point (pooled-sample kernels)
      bool fnan[3] = {false, false, false};             // (+ the NaN flags that go with it)
      L.ra.pixel = y * sc.width + x; L.ra.k0 = ds.seed_lo; L.ra.k1 = ds.seed_hi;
      bool need_new = true;
      uint32_t cur_depth = 0; int first_kind = -1;
      for (;;) {
        if (need_new) {
          if (L.s >= sc.samples_per_pixel || sc.max_depth == 0) break;
          lane_begin_sample(ds, L, x, y);
          need_new = false;
          cur_depth = 0; first_kind = -1;
        }
        // ---- trace: f32 cull + exact confirm, object order
        const double a = length_squared(L.d);
        const RayF32 rf = make_ray_f32(L.o, L.d);
        double closest = T_MAX; int best = -1;
        L.n_segments++;
        if (use_cull == 3 || use_cull == 4) {  // the product's hit_world: `large` list + grid walk
          uint32_t n_steps = 0;
          const uint32_t e0 = L.n_exact;
          hit_world_grid(ds, tb, L.o, L.d, closest, best, L.n_exact, n_steps);
          steps += n_steps;
          if (L.k == 0 && !L.in_light) { cam_steps += n_steps; cam_segs++; cam_exact += L.n_exact - e0; }
          if (g_hist) {
            uint32_t a = n_steps > 63u ? 63u : n_steps, b = (L.n_exact - e0) > 63u ? 63u : (L.n_exact - e0);
#pragma omp atomic
            g_hist[a]++;
#pragma omp atomic
            g_hist[64 + b]++;
          }
          if (use_cull == 4) {  // audit: the reference's brute force must agree on (t, sphere), bit for bit
            double c2 = T_MAX; int b2 = -1;
            for (uint32_t i = 0; i < sc.n_spheres; ++i) {
              double r = exact_root(L.o, L.d, a, t.geom[i], T_MIN, c2);
              if (r >= 0.0) { c2 = r; b2 = (int)i; }
            }
            if (b2 != best || (b2 >= 0 && c2 != closest)) {
              cull_false_reject++;
              if (std::getenv("RT_AUDIT_VERBOSE"))
                std::fprintf(stderr, "AUDIT px %u,%u s %u k %u: o=(%.17g,%.17g,%.17g) d=(%.17g,%.17g,%.17g) grid best %d t %.17g | brute best %d t %.17g\n",
                             x, y, L.s, L.k, L.o.x, L.o.y, L.o.z, L.d.x, L.d.y, L.d.z, best, closest, b2, c2);
            }
          }
        } else
        for (uint32_t i = 0; i < sc.n_spheres; ++i) {
          const CullPair& cp = t.cull[i / 2];
          bool pass = cull_pass(cull_disc(rf, cp.cx[i & 1], cp.cy[i & 1], cp.cz[i & 1], cp.R[i & 1]));
          if (use_cull == 2) {  // audit mode: run the exact test anyway and flag false rejects
            double r = exact_root(L.o, L.d, a, t.geom[i], T_MIN, closest);
            if (r >= 0.0 && !pass) cull_false_reject++;
            pass = true;
          }
          if (!use_cull) pass = true;
          if (!pass) continue;
          L.n_exact++;
          double r = exact_root(L.o, L.d, a, t.geom[i], T_MIN, closest);
          if (r >= 0.0) { closest = r; best = (int)i; }
        }
        if (cur_depth == 0 && best >= 0) first_kind = (int)tb.mat((uint32_t)best).kind;
        cur_depth++;
        need_new = lane_shade(ds, tb, L, best, closest) == LANE_FINISHED;
        if (need_new && g_hist) {  // [128..191] path length, [192 + 64 * kind ..] path length by the first hit's material
          uint32_t dd = cur_depth > 63u ? 63u : cur_depth;
#pragma omp atomic
          g_hist[128 + dd]++;
          if (first_kind >= 0 && first_kind < 5) {
#pragma omp atomic
            g_hist[192 + 64 * first_kind + dd]++;
          }
        }
        if (need_new) {
          for (int k = 0; k < 3; ++k) { acc[k] += L.val[k]; facc[k] += sample_to_fixed(L.val[k]); fnan[k] = fnan[k] || sample_is_nan(L.val[k]); }
          L.s += 1;
        }
      }
      float scale = 1.0f / (float)sc.samples_per_pixel;
      for (int k = 0; k < 3; ++k) {
        float lin = (use_cull_flags & 16) ? (fnan[k] ? rt_nanf() : fixed_to_mean(facc[k], sc.samples_per_pixel)) : scale * acc[k];
        size_t o = ((size_t)lr * sc.width + x) * 3 + k;
        if (linear) linear[o] = lin;
        if (rgb8) rgb8[o] = f32_to_u8(sqrtf(lin));
      }
      segs += L.n_segments; exact += L.n_exact; oob += L.n_tex_oob;
    }
  }
  if (stats) {
    std::memset(stats, 0, sizeof *stats);
    stats->samples = (uint64_t)rows * sc.width * sc.samples_per_pixel;
    stats->segments = segs; stats->sphere_tests = segs * sc.n_spheres; stats->exact_tests = exact;
    stats->tex_oob = oob; stats->kernel_ms = (double)cull_false_reject; stats->frame_ms = 0; stats->grid_steps = steps;
    stats->wave_iters[0] = cam_steps; stats->wave_iters[1] = cam_segs; stats->wave_iters[2] = cam_exact;  // camera-ray share (diagnostics)
  }
}
}  // namespace

// use_cull: 0 = exact test for every sphere, 1 = product behaviour (cull + confirm),
//           2 = audit (stats->kernel_ms returns the number of false rejects; must be 0)
//           3 = grid walk (the default kernel's hit_world); stats->grid_steps = DDA steps
//           4 = grid walk audited against brute force per segment (stats->kernel_ms = mismatches)
//           +16 = accumulate pixels in exact fixed point (the pooled-sample kernels' rule)
//           +32 = the short colour maps (scenes whose albedos all lie in [0, 1]; ignored otherwise)
extern "C" int hostsim_render(const RtScene* scene, const RtRowTiles* tiles, uint8_t* rgb8, float* linear,
                              RtStats* stats, int use_cull) {
  HostTables t;
  if (!scene || !build_tables(*scene, t, true).empty()) return RT_ERR_INVALID;
  build_texels(*scene, t);  // (the 4-byte-texel path the device takes; the RGB8 blob serves the records outside its range)
  DevScene ds;
  fill_dev_scene(*scene, t, ds);
  ds.tex4 = t.tex4.data(); ds.sky4 = t.sky4.data();
  std::vector<uint8_t> blob(t.tex_bytes ? t.tex_bytes : 1);
  for (uint32_t i = 0; i < scene->n_textures; ++i)
    if (scene->textures[i].nbytes) std::memcpy(&blob[t.tex_off[i]], scene->textures[i].rgb8, scene->textures[i].nbytes);
  ds.geom = t.geom.data(); ds.mat = t.mat.data(); ds.lights = t.lights.data();
  ds.tex = blob.data(); ds.sky = scene->sky_rgb8;
  ds.matc = t.matc.data(); ds.cell_word = t.cell_word.data(); ds.cell_items = t.grid.wide ? reinterpret_cast<const uint16_t*>(t.cell_items32.data()) : t.cell_items.data(); ds.large = t.large.data(); ds.large_geom = t.large_geom.data();
  // +32: the SHORT colour maps the product kernel takes when every albedo lies in [0, 1] (rt_core.h FwdT<true>; lit scenes:
  // q in registers + the memory-resident base of lane_compose) instead of the general clamped-affine map — bit-identical
  const bool short_map = (use_cull & 32) != 0 && t.simple_colour;
  use_cull &= ~32;
  if (t.lights.empty()) { if (short_map) render_rows<false, true>(*scene, t, ds, tiles, rgb8, linear, stats, use_cull); else render_rows<false>(*scene, t, ds, tiles, rgb8, linear, stats, use_cull); }
  else { if (short_map) render_rows<true, true>(*scene, t, ds, tiles, rgb8, linear, stats, use_cull); else render_rows<true>(*scene, t, ds, tiles, rgb8, linear, stats, use_cull); }
  return RT_OK;
}

extern "C" float hostsim_cull_disc(const double o[3], const double d[3], const RtSphere* s) {
  float cx, cy, cz, R;
  build_cull_entry(*s, &cx, &cy, &cz, &R);
  RayF32 rf = make_ray_f32(v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]));
  return cull_disc(rf, cx, cy, cz, R);
}

// the kernel's exact root selection for one sphere (rt_core.h exact_root), for property tests
extern "C" double hostsim_exact_root(const double o[3], const double d[3], const RtSphere* s, double t_min, double t_max) {
  SphereGeom g{s->center[0], s->center[1], s->center[2], s->radius};
  V3 dd = v3(d[0], d[1], d[2]);
  return exact_root(v3(o[0], o[1], o[2]), dd, length_squared(dd), g, t_min, t_max);
}

// rt_core.h exact_hit_prefix (the square-root-free part of the hit test; the kernel's RT_START_CELL_PREFIX arm): 1 = may hit
extern "C" int hostsim_hit_prefix(const double o[3], const double d[3], const RtSphere* s) {
  SphereGeom g{s->center[0], s->center[1], s->center[2], s->radius};
  const V3 dd = v3(d[0], d[1], d[2]);
  return exact_hit_prefix(v3(o[0], o[1], o[2]), dd, ray_consts(dd), g).may_hit ? 1 : 0;
}

// grid layout of a scene (for tests): out = n[3], n_large, n_cells, n_items
extern "C" int hostsim_grid_info(const RtScene* scene, uint32_t out[6]) {
  HostTables t;
  if (!scene || !build_tables(*scene, t, true).empty()) return RT_ERR_INVALID;
  out[0] = t.grid.n[0]; out[1] = t.grid.n[1]; out[2] = t.grid.n[2];
  out[3] = t.grid.n_large; out[4] = t.grid.n_cells; out[5] = t.grid.n_items;
  return RT_OK;
}

// 1: the scene's grid is in the wide table format (GridDesc.wide: 32-bit item lists), 0: packed, < 0: error
extern "C" int hostsim_grid_wide(const RtScene* scene) {
  HostTables t;
  if (!scene || !build_tables(*scene, t, true).empty()) return RT_ERR_INVALID;
  return (int)t.grid.wide;
}

// how the walk of one ray begins (rt_core.h grid_begin): 0 = misses the grid, 1 = walks, 2 = numerically unsafe -> full scan; < 0: error / no grid
extern "C" int hostsim_grid_mode(const RtScene* scene, const double o[3], const double d[3]) {
  HostTables t;
  if (!scene || !build_tables(*scene, t, true).empty()) return RT_ERR_INVALID;
  if (t.grid.n[0] == 0u) return -100;
  GridWalk w;
  return grid_begin(t.grid, v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]), w);
}

// one ray against one scene through the grid and by brute force (adversarial tests):
// out = {best_grid, best_brute}, t_out = {t_grid, t_brute}
extern "C" int hostsim_hit_world(const RtScene* scene, const double o[3], const double d[3], int out[2], double t_out[2]) {
  HostTables t;
  if (!scene || !build_tables(*scene, t, true).empty()) return RT_ERR_INVALID;
  DevScene ds;
  fill_dev_scene(*scene, t, ds);
  ds.geom = t.geom.data(); ds.matc = t.matc.data(); ds.cell_word = t.cell_word.data(); ds.cell_items = t.grid.wide ? reinterpret_cast<const uint16_t*>(t.cell_items32.data()) : t.cell_items.data();
  ds.large = t.large.data(); ds.large_geom = t.large_geom.data();
  const GlobalTables tb{ds.geom, ds.matc};
  V3 oo = v3(o[0], o[1], o[2]), dd = v3(d[0], d[1], d[2]);
  const double a = length_squared(dd);
  double c1 = T_MAX; int b1 = -1; uint32_t ne = 0, ns = 0;
  hit_world_grid(ds, tb, oo, dd, c1, b1, ne, ns);
  double c2 = T_MAX; int b2 = -1;
  for (uint32_t i = 0; i < scene->n_spheres; ++i) {
    double r = exact_root(oo, dd, a, t.geom[i], T_MIN, c2);
    if (r >= 0.0) { c2 = r; b2 = (int)i; }
  }
  out[0] = b1; out[1] = b2; t_out[0] = c1; t_out[1] = c2;
  return RT_OK;
}

// rt_core.h div_by_recip over arrays (property test against the IEEE quotient)
// texel of a Texture hit: the fast path's answer (ok, col, row) beside the exact path's (materials.rs:236-254 through
// sphere_uv); n points, 3 doubles each; out: 5 x u64 per point {fast_ok, fast col, fast row, exact col, exact row}
extern "C" void hostsim_texels(const double* points, uint64_t n, const double centre_radius[4], double h_offset, uint64_t tex_w,
                               uint64_t tex_h, uint64_t* out) {
  const SphereGeom g{centre_radius[0], centre_radius[1], centre_radius[2], centre_radius[3]};
#pragma omp parallel for
  for (int64_t i = 0; i < (int64_t)n; ++i) {
    const V3 p = v3(points[3 * i], points[3 * i + 1], points[3 * i + 2]);
    uint64_t col = 0, row = 0;
    const bool ok = texel_fast(p, g, h_offset, tex_w, tex_h, col, row);
    const UV uv = sphere_uv(p, g);
    double rot = uv.u + h_offset;
    if (rot > 1.0) rot = rot - 1.0;
    out[5 * i] = ok; out[5 * i + 1] = col; out[5 * i + 2] = row;
    out[5 * i + 3] = sat_u64(floor(rot * (double)tex_w)); out[5 * i + 4] = sat_u64(floor((1.0 - uv.v) * (double)(tex_h - 1)));
  }
}

// rt_core.h range_m1_1 (fused) and sample_to_fixed over arrays (property tests against the oracle's rounded forms)
extern "C" void hostsim_range_m1_1(const uint32_t* w, double* out, uint64_t n) {
#pragma omp parallel for
  for (int64_t i = 0; i < (int64_t)n; ++i) out[i] = range_m1_1(w[i]);
}
extern "C" void hostsim_u01_53(const uint32_t* lo, const uint32_t* hi, double* out, uint64_t n) {
#pragma omp parallel for
  for (int64_t i = 0; i < (int64_t)n; ++i) out[i] = u01_53(lo[i], hi[i]);
}
extern "C" void hostsim_sample_to_fixed(const float* v, uint64_t* out, uint64_t n) {
#pragma omp parallel for
  for (int64_t i = 0; i < (int64_t)n; ++i) out[i] = sample_to_fixed(v[i]);
}

// rt_core.h rt_div255f over an array (property test against the IEEE quotient x / 255.0f)
extern "C" void hostsim_div255(const float* x, float* out, uint64_t n) {
#pragma omp parallel for
  for (int64_t i = 0; i < (int64_t)n; ++i) out[i] = rt_div255f(x[i]);
}

extern "C" void hostsim_div_by_recip(const double* x, const double* b, double* out, uint64_t n) {
#pragma omp parallel for
  for (int64_t i = 0; i < (int64_t)n; ++i) out[i] = div_by_recip(x[i], b[i], 1.0 / b[i]);
}

extern "C" void hostsim_set_histogram(unsigned long long* h) { g_hist = h; }
