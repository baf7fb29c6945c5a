// rt_tables.h — flatten an RtScene (include/rt_abi.h) into the HBM table layout of
// rt_core.h::DevScene.  Plain host C++; used by rt_hip_api.hip (upload) and tests/hostsim.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "rt_core.h"

namespace rtc {

// a vector whose resize() leaves new elements uninitialised (they are all written right after: no 29 MB zero fill)
template <typename T>
struct DefaultInitAlloc : std::allocator<T> {
  template <typename U> struct rebind { using other = DefaultInitAlloc<U>; };
  template <typename U> void construct(U* p) noexcept { ::new (static_cast<void*>(p)) U; }
  template <typename U, typename... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};
using TexelVec = std::vector<uint32_t, DefaultInitAlloc<uint32_t>>;

struct HostTables {
  std::vector<SphereGeom> geom;
  std::vector<SphereMat> mat;
  std::vector<CullPair> cull;
  std::vector<uint32_t> lights;
  std::vector<uint64_t> tex_off;  // per RtTexture, byte offset in the blob
  uint64_t tex_bytes = 0;
  // the 4-byte-texel copies the device reads (build_texels): every texture whose records are in texels_fast()'s range, and the sky
  TexelVec tex4, sky4;
  bool need_rgb8 = false;            // some Texture sphere takes the general path: the RGB8 blob must be resident too
  bool sky_fast = false;
  uint32_t n_pairs = 0;           // real pairs (cull.size() includes chunk padding)
  bool simple_colour = true;  // every albedo in [0,1] (textures always are): the short colour maps of rt_core.h apply
  // hit_world acceleration (GridDesc, rt_core.h)
  std::vector<MatCore> matc;
  GridDesc grid{};
  std::vector<uint32_t> cell_word;
  std::vector<uint16_t> cell_items;
  std::vector<uint32_t> cell_items32;  // grid.wide: the item lists as 32-bit indices (cell_items is empty then)
  std::vector<uint32_t> large;
  std::vector<SphereGeom> large_geom;
};

// Grid construction knobs (development tunables; the defaults are what ships).
// Bytes the per-segment tables (geometry, material cores, cell entries, item lists) may take so
// that the megakernel can keep them in LDS next to its tile slots (rt_kernel.hip: 160 KB per CU).
constexpr size_t GRID_LDS_TABLE_BUDGET = 88u * 1024u;
struct GridParams {
  double cells_per_sphere = 0.0;   // target cell count = this * gridded spheres; 0 = automatic (below)
  double large_radius_ratio = 4;   // |r| > ratio * median |r|: a candidate for the `large` list (at most `max_large_by_radius`, see build_grid_as)
  uint32_t max_large_by_radius = 8;
  uint32_t large_cell_limit = 512; // a sphere covering more cells than this -> `large` list
  uint32_t min_spheres = 24;       // fewer spheres than this: no grid, test them all
  uint32_t force_n[3] = {0, 0, 0}; // != 0: cells along that axis (development: anisotropic grids)
  bool force_wide = false;         // the wide table format (GridDesc.wide) whatever the sphere count (tests: small worlds through the wide kernels)
};
// The shipped library takes the defaults above and reads NOTHING from the environment.  A/B builds and the analysis
// tools (tools/ab_bench.py, tools/analysis/) compile with -DRT_DEV_KNOBS to sweep the grid's shape.
inline GridParams grid_params_from_env() {
  GridParams p;
#if defined(RT_DEV_KNOBS) || defined(RT_TEST_PROBES)  // (librt_hip_probe.so — tests only — can put any world through the wide format)
  if (const char* e = std::getenv("RT_GRID_WIDE")) p.force_wide = std::atoi(e) != 0;
#endif
#ifdef RT_DEV_KNOBS
  if (const char* e = std::getenv("RT_GRID_CELLS_PER_SPHERE")) p.cells_per_sphere = std::atof(e);
  if (const char* e = std::getenv("RT_GRID_MIN_SPHERES")) p.min_spheres = (uint32_t)std::atoi(e);
  if (const char* e = std::getenv("RT_GRID_LARGE_RATIO")) p.large_radius_ratio = std::atof(e);
  if (const char* e = std::getenv("RT_GRID_LARGE_CELLS")) p.large_cell_limit = (uint32_t)std::atoi(e);
  if (const char* e = std::getenv("RT_GRID_N")) std::sscanf(e, "%u,%u,%u", &p.force_n[0], &p.force_n[1], &p.force_n[2]);
#endif
  return p;
}

// Uniform grid over the ordinary spheres; see GridDesc / 2*GridDesc.pull in rt_core.h for what the
// walk relies on: sphere i is listed in every cell its bounding box, grown by 2*GridDesc.pull
// cells, overlaps (cells farther from the centre than the radius are dropped again).
// `wide`: the table format (GridDesc.wide).  Returns false — with nothing usable in `t` — when the PACKED format cannot hold this
// grid (more than 65 535 spheres: its item indices are u16, 0xFFFF = none; more than 4 095 items in a cell; 2^20 items or more):
// the caller builds it again wide.
inline bool build_grid_as(const RtScene& sc, HostTables& t, const GridParams& gp, bool wide) {
  const uint32_t n = sc.n_spheres;
  GridDesc& G = t.grid;
  std::memset(&G, 0, sizeof G);
  t.cell_word.clear(); t.cell_items.clear(); t.cell_items32.clear(); t.large.clear();
  auto all_large = [&]() {
    std::memset(&G, 0, sizeof G);
    t.cell_word.clear(); t.cell_items.clear(); t.cell_items32.clear(); t.large.resize(n);
    for (uint32_t i = 0; i < n; ++i) t.large[i] = i;
    G.n_large = n;
  };
  if (n < gp.min_spheres) { all_large(); return true; }
  if (!wide && n > 65535u) return false;
  if (n >= 0xFFFFFFFEu) { all_large(); return true; }
  std::vector<uint8_t> is_large(n, 0);
  std::vector<double> radii;
  for (uint32_t i = 0; i < n; ++i) {
    const RtSphere& s = sc.spheres[i];
    const bool finite = std::isfinite(s.center[0]) && std::isfinite(s.center[1]) && std::isfinite(s.center[2]) && std::isfinite(s.radius);
    if (!finite) is_large[i] = 1; else radii.push_back(std::fabs(s.radius));
  }
  if (radii.size() < gp.min_spheres) { all_large(); return true; }
  std::nth_element(radii.begin(), radii.begin() + radii.size() / 2, radii.end());
  const double r_med = radii[radii.size() / 2];
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  uint32_t n_grid = 0;
  // Spheres much bigger than the bulk (the ground, the three r = 1 spheres of the cover scene)
  // are tested by every ray instead of being gridded: without them the grid hugs the bulk (one
  // flat layer of cells around the small spheres), rays from above enter it right where they
  // come down, and a walk is ~2 steps instead of ~10 (measured 21.1 -> 16.1 ms).  Each one costs a
  // wave-uniform exact test per ray, so only the biggest few qualify — and only if taking them out LEAVES a smaller bulk:
  // the k <= max_large_by_radius biggest, k the largest count after which the radii drop by half or more.  (Until round 5: the
  // biggest eight whatever came after them.  A world whose radii spread over decades then paid eight tests per ray for eight
  // spheres no different from the hundreds left in the grid: 15.1 -> 8.1 tests per ray on the log-uniform 10^4-sphere world,
  // 13.9 -> 6.6 on the bimodal one, tools/analysis/levels_estimate.cpp; BASELINE's scenes keep their lists: 1000 | 1 1 1 | 0.2 ...)
  {
    std::vector<std::pair<double, uint32_t>> big;
    double r_rest = 0.0;  // the largest radius that is no candidate
    for (uint32_t i = 0; i < n; ++i) {
      if (is_large[i]) continue;
      const double r = std::fabs(sc.spheres[i].radius);
      if (r > gp.large_radius_ratio * r_med) big.push_back({r, i}); else r_rest = std::max(r_rest, r);
    }
    std::sort(big.begin(), big.end(), [](const std::pair<double, uint32_t>& a, const std::pair<double, uint32_t>& b) { return a.first > b.first || (a.first == b.first && a.second < b.second); });
    size_t k_take = 0;
    for (size_t k = 1; k <= big.size() && k <= gp.max_large_by_radius; ++k) {
      const double next = k < big.size() ? big[k].first : r_rest;  // the biggest sphere left in the grid if k are taken
      if (big[k - 1].first >= 2.0 * next) k_take = k;
    }
    for (size_t k = 0; k < k_take; ++k) is_large[big[k].second] = 1;
  }
  for (uint32_t i = 0; i < n; ++i) {
    const RtSphere& s = sc.spheres[i];
    if (is_large[i]) continue;
    n_grid++;
    for (int k = 0; k < 3; ++k) {
      lo[k] = std::min(lo[k], s.center[k] - std::fabs(s.radius));
      hi[k] = std::max(hi[k], s.center[k] + std::fabs(s.radius));
    }
  }
  if (n_grid < gp.min_spheres) { all_large(); return true; }
  double ext[3], vol = 1.0;
  for (int k = 0; k < 3; ++k) {
    const double pad = 1e-3 * (hi[k] - lo[k]) + 1e-9 * (std::fabs(lo[k]) + std::fabs(hi[k])) + 1e-12;
    lo[k] -= pad; hi[k] += pad;
    ext[k] = hi[k] - lo[k];
    vol *= ext[k];
  }
  const double cell = std::cbrt(vol / std::max(1.0, gp.cells_per_sphere * n_grid));
  for (int k = 0; k < 3; ++k) {
    double c = std::ceil(ext[k] / cell);
    if (!(c >= 1.0)) c = 1.0;
    if (gp.force_n[k]) c = (double)gp.force_n[k];
    if (c > (double)GRID_MAX_AXIS) c = (double)GRID_MAX_AXIS;
    G.n[k] = (uint32_t)c;
  }
  // The walk leaves a cell at exit planes pulled back by `pull` cells, so next to the grid's OUTER
  // faces a sliver of that width is never walked: keep every sphere at least 2*pull (the
  // registration margin) + 1e-3 cells away from them.
  const double pull_cells = 8.0 * grid_walk_eps(std::max(G.n[0], std::max(G.n[1], G.n[2])));
  const double edge = 2.0 * pull_cells + 1e-3;
  for (int k = 0; k < 3; ++k) {
    const double w = ext[k] / ((double)G.n[k] - 2.0 * edge);  // cell width such that the spheres span [edge, n - edge] cells
    G.gmin[k] = lo[k] - edge * w;
    G.inv_cell[k] = 1.0 / w;
    G.nd[k] = (double)G.n[k];
  }
  G.pull = (float)(8.0 * grid_walk_eps(std::max(G.n[0], std::max(G.n[1], G.n[2]))));
  const uint32_t n_inner = G.n[0] * G.n[1] * G.n[2];
  const uint32_t px = G.n[0] + 2, py = G.n[1] + 2, pz = G.n[2] + 2;  // padded with the EXIT border
  G.n_cells = px * py * pz;
  // cell range of each gridded sphere (bounding box grown by the walk's margin)
  struct Range { int a[3], b[3]; };
  std::vector<Range> rng(n);
  const double m = 2.0 * (double)G.pull;  // registration margin (rt_core.h "Margins")
  for (uint32_t i = 0; i < n; ++i) {
    if (is_large[i]) continue;
    const RtSphere& s = sc.spheres[i];
    uint64_t cells = 1;
    for (int k = 0; k < 3; ++k) {
      const double r = std::fabs(s.radius);
      double a = std::floor((s.center[k] - r - G.gmin[k]) * G.inv_cell[k] - m);
      double b = std::floor((s.center[k] + r - G.gmin[k]) * G.inv_cell[k] + m);
      a = std::max(a, 0.0); b = std::min(b, (double)G.n[k] - 1.0);
      rng[i].a[k] = (int)a; rng[i].b[k] = (int)b;
      cells *= (uint64_t)(b >= a ? (int)b - (int)a + 1 : 0);
    }
    if (cells > gp.large_cell_limit) is_large[i] = 1;
  }
  // does the cell (grown by the margin) come within |r| of the centre?  (world units, f64)
  auto overlaps = [&](const RtSphere& s, int ix, int iy, int iz) {
    const int idx[3] = {ix, iy, iz};
    double d2 = 0.0;
    for (int k = 0; k < 3; ++k) {
      const double w = 1.0 / G.inv_cell[k];
      const double c0 = G.gmin[k] + ((double)idx[k] - m) * w, c1 = G.gmin[k] + ((double)idx[k] + 1.0 + m) * w;
      const double d = s.center[k] < c0 ? c0 - s.center[k] : (s.center[k] > c1 ? s.center[k] - c1 : 0.0);
      d2 += d * d;
    }
    const double r = std::fabs(s.radius) * (1.0 + 1e-9);
    return d2 <= r * r;
  };
  std::vector<uint32_t> count(n_inner, 0);
  auto padded = [&](uint32_t c) {
    const uint32_t ix = c % G.n[0], iy = (c / G.n[0]) % G.n[1], iz = c / (G.n[0] * G.n[1]);
    return (ix + 1) + px * ((iy + 1) + py * (iz + 1));
  };
  const uint32_t cw = wide ? 4u : 2u;  // words per cell entry
  for (int pass = 0; pass < 2; ++pass) {
    std::vector<uint32_t> cursor;
    if (pass == 1) {
      uint64_t total = 0;
      t.cell_word.assign((size_t)cw * G.n_cells, 0u);
      for (size_t c = 0; c < G.n_cells; ++c) {  // every padded cell is an EXIT cell until the inner ones are written
        t.cell_word[cw * c] = CELL_EXIT;
        t.cell_word[cw * c + 1] = wide ? 0u : CELL_EXIT;
        if (wide) t.cell_word[cw * c + 2] = CELL_NO_ITEM32;
      }
      cursor.resize(n_inner);
      for (uint32_t c = 0; c < n_inner; ++c) {
        if (!wide && (count[c] > CELL_MAX_COUNT || total >= CELL_START_MASK)) return false;
        if (wide) { t.cell_word[4 * (size_t)padded(c)] = (uint32_t)total; t.cell_word[4 * (size_t)padded(c) + 1] = count[c]; }
        else t.cell_word[2 * (size_t)padded(c)] = (uint32_t)total | (count[c] << CELL_COUNT_SHIFT);
        cursor[c] = (uint32_t)total;
        total += count[c];
        if (total >= 0xFFFFFFFEull) { all_large(); return true; }
      }
      if (!wide && total >= CELL_START_MASK) return false;
      if (wide) t.cell_items32.resize(total); else t.cell_items.resize(total);
      G.n_items = (uint32_t)total;
    }
    for (uint32_t i = 0; i < n; ++i) {
      if (is_large[i]) continue;
      const RtSphere& s = sc.spheres[i];
      for (int iz = rng[i].a[2]; iz <= rng[i].b[2]; ++iz)
        for (int iy = rng[i].a[1]; iy <= rng[i].b[1]; ++iy)
          for (int ix = rng[i].a[0]; ix <= rng[i].b[0]; ++ix) {
            if (!overlaps(s, ix, iy, iz)) continue;
            const uint32_t c = (uint32_t)ix + G.n[0] * ((uint32_t)iy + G.n[1] * (uint32_t)iz);
            if (pass == 0) count[c]++;
            else if (wide) t.cell_items32[cursor[c]++] = i;
            else t.cell_items[cursor[c]++] = (uint16_t)i;
          }
    }
  }
  for (uint32_t c = 0; c < n_inner; ++c) {  // inline copy of each cell's first two items (wide: of its first item)
    if (wide) {
      const uint32_t first = t.cell_word[4 * (size_t)padded(c)], cnt = t.cell_word[4 * (size_t)padded(c) + 1];
      t.cell_word[4 * (size_t)padded(c) + 2] = cnt > 0 ? t.cell_items32[first] : CELL_NO_ITEM32;
      continue;
    }
    const uint32_t word = t.cell_word[2 * padded(c)];
    const uint32_t first = word & CELL_START_MASK, cnt = word >> CELL_COUNT_SHIFT;
    const uint32_t i0 = cnt > 0 ? t.cell_items[first] : 0xFFFFu, i1 = cnt > 1 ? t.cell_items[first + 1] : 0xFFFFu;
    t.cell_word[2 * padded(c) + 1] = i0 | (i1 << 16);
  }
  G.wide = wide ? 1u : 0u;
  for (uint32_t i = 0; i < n; ++i) if (is_large[i]) t.large.push_back(i);
  G.n_large = (uint32_t)t.large.size();
  return true;
}
inline void build_grid(const RtScene& sc, HostTables& t, const GridParams& gp) {
  const bool wide = gp.force_wide || sc.n_spheres > 65535u;
  if (!build_grid_as(sc, t, gp, wide)) build_grid_as(sc, t, gp, true);
}

// returns "" or a description of why the scene is invalid (RT_ERR_INVALID)
// want_cull: also the round-1 cull-pair table (HostTables::cull) — no kernel reads it any more; tests/hostsim's audit mode and
// tools/analysis/walk_sim.cpp do (the product's rt_hip_scene_create leaves it out since round 6).
inline std::string build_tables(const RtScene& sc, HostTables& t, bool want_cull = false) {
  if (sc.abi_version != RT_ABI_VERSION) return "abi_version mismatch";
  if (sc.width == 0 || sc.height == 0) return "empty image";
  if (sc.n_spheres && !sc.spheres) return "null sphere table";
  if (sc.n_textures && !sc.textures) return "null texture table";
  if (sc.sky_mode > RT_SKY_TEXTURE) return "bad sky_mode";
  if (sc.sky_mode == RT_SKY_TEXTURE && (!sc.sky_rgb8 || sc.sky_w == 0 || sc.sky_h == 0)) return "sky texture missing";
  t.tex_off.resize(sc.n_textures);
  t.tex_bytes = 0;
  for (uint32_t i = 0; i < sc.n_textures; ++i) {
    if (sc.textures[i].nbytes && !sc.textures[i].rgb8) return "null texture pixels";
    t.tex_off[i] = t.tex_bytes;
    t.tex_bytes += (sc.textures[i].nbytes + 15) & ~15ull;
  }
  const uint32_t n = sc.n_spheres;
  t.geom.resize(n);
  t.mat.resize(n);
  t.matc.resize(n);
  // pairs, padded to a whole number of CULL_CHUNK-pair chunks plus one chunk of slack so the
  // scan may prefetch one chunk past the end; padding entries can never pass (R = -inf)
  const uint32_t n_pairs = (n + 1) / 2;
  const uint32_t padded = (n_pairs + CULL_CHUNK - 1) / CULL_CHUNK * CULL_CHUNK + CULL_CHUNK;
  CullPair never;
  for (int k = 0; k < 2; ++k) { never.cx[k] = never.cy[k] = never.cz[k] = 0.0f; never.R[k] = -INFINITY; }
  if (want_cull) t.cull.assign(padded, never); else t.cull.clear();
  t.n_pairs = n_pairs;
  t.lights.clear();
  t.simple_colour = true;
  t.need_rgb8 = false;
  for (uint32_t i = 0; i < n; ++i) {
    const RtSphere& s = sc.spheres[i];
    if (s.kind > RT_MAT_LIGHT) return "bad material kind";
    t.geom[i] = SphereGeom{s.center[0], s.center[1], s.center[2], s.radius};
    SphereMat m;
    std::memset(&m, 0, sizeof m);
    m.albedo[0] = s.albedo[0]; m.albedo[1] = s.albedo[1]; m.albedo[2] = s.albedo[2];
    m.kind = s.kind; m.fuzz_or_ior = s.fuzz_or_ior; m.h_offset = s.h_offset;
    m.tex_w = s.tex_w; m.tex_h = s.tex_h;
    if (s.kind == RT_MAT_TEXTURE) {
      if (s.tex_id >= sc.n_textures) return "texture id out of range";
      m.tex_off = t.tex_off[s.tex_id];
      m.tex_nbytes = sc.textures[s.tex_id].nbytes;
      // 4-byte texels: texture k starts at the texel count of textures 0..k-1 (build_texels lays them out the same way)
      uint64_t before = 0;
      for (uint32_t k = 0; k < s.tex_id; ++k) before += sc.textures[k].nbytes / 3;
      if (texels_fast(s.tex_w, s.tex_h, s.h_offset, m.tex_nbytes, before)) {
        m.tex_fast = 1u; m.texel_off = (uint32_t)before; m.texel_last = (uint32_t)(m.tex_nbytes / 3 - 1);
      } else t.need_rgb8 = true;
    }
    t.mat[i] = m;
    MatCore mc;
    mc.albedo[0] = m.albedo[0]; mc.albedo[1] = m.albedo[1]; mc.albedo[2] = m.albedo[2];
    mc.kind = m.kind; mc.fuzz_or_ior = m.fuzz_or_ior;
    mc.inv_r = recip_safe(s.radius) ? 1.0 / s.radius : 0.0;
    mc.r0[0] = mc.r0[1] = 0.0;
    if (s.kind == RT_MAT_GLASS) {
      const double inv_ior = 1.0 / s.fuzz_or_ior;   // materials.rs:181: 1.0 / ir, divided once here
      matcore_set_inv_ior(mc, inv_ior);
      mc.r0[0] = reflectance_r0(inv_ior);           // materials.rs:152-153 for the front-face ratio ...
      mc.r0[1] = reflectance_r0(s.fuzz_or_ior);     // ... and the back-face ratio
    }
    t.matc[i] = mc;
    if (s.kind == RT_MAT_LIGHT) t.lights.push_back(i);
    if (s.kind == RT_MAT_LAMBERTIAN || s.kind == RT_MAT_METAL)
      for (int c = 0; c < 3; ++c)
        if (!(s.albedo[c] >= 0.0f && s.albedo[c] <= 1.0f)) t.simple_colour = false;
    if (want_cull) {
      CullPair& cp = t.cull[i / 2];
      build_cull_entry(s, &cp.cx[i & 1], &cp.cy[i & 1], &cp.cz[i & 1], &cp.R[i & 1]);
    }
  }
  if (want_cull && (n & 1)) {  // odd count: the second half of the last real pair can never pass either
    CullPair& cp = t.cull[n / 2];
    cp.cx[1] = cp.cx[0]; cp.cy[1] = cp.cy[0]; cp.cz[1] = cp.cz[0]; cp.R[1] = -INFINITY;
  }
  auto pack_large = [&]() {
    t.large_geom.resize(t.large.size());
    for (size_t i = 0; i < t.large.size(); ++i) t.large_geom[i] = t.geom[t.large[i]];
  };
  GridParams gp = grid_params_from_env();
  if (gp.cells_per_sphere > 0.0) { build_grid(sc, t, gp); pack_large(); }
  else {
    // Finer cells mean fewer exact tests per ray but more steps (and more lock-step walk rounds per
    // wave).  Measured once the tall spheres had left the grid (profiles/r01_run9_ab_cells.log: 484
    // spheres in LDS, 10 001 spheres out of L2, 4K textured): 2 cells per gridded sphere is best or
    // tied everywhere (2: 15.15 ms, 3: 15.25, 4: 15.9, 8: 16.6), so that is the automatic choice.
    gp.cells_per_sphere = 2.0;
    build_grid(sc, t, gp);
    pack_large();
  }
  return "";
}

// The textures and the sky as the device reads them: R | G << 8 | B << 16 per texel, textures back to back in table order
// (texture k's first texel = the texel count of textures 0..k-1, as build_tables wrote into SphereMat::texel_off).  The
// C ABI takes the caller's RGB8 (materials.rs:213-219, config.rs:36-47); this is a device-resident re-layout of the same
// values.  Only as many textures as fit 2^31 texels are expanded (the records beyond take the RGB8 path).
// Textures and sky as 4-byte texels (one aligned dword load per fetch, rt_core.h).  29 MB for the reference's test scene: the
// conversion was 8 ms of a one-shot run's set-up on one core (round 5) — the buffers are now allocated once, never zero-filled
// (TexelVec) and filled by up to 8 threads in runs of 256 K texels.
inline void build_texels(const RtScene& sc, HostTables& t) {
  t.tex4.clear(); t.sky4.clear();
  struct Span { const uint8_t* src; uint32_t* dst; uint64_t n; };
  std::vector<Span> spans;
  uint64_t total = 0;
  uint32_t n_tex = 0;
  for (; n_tex < sc.n_textures; ++n_tex) {
    const uint64_t n_px = sc.textures[n_tex].nbytes / 3;
    if (total + n_px >= (1ull << 31)) break;
    total += n_px;
  }
  t.tex4.resize(total);
  uint64_t at = 0;
  for (uint32_t k = 0; k < n_tex; ++k) {
    const uint64_t n_px = sc.textures[k].nbytes / 3;
    if (n_px) spans.push_back(Span{sc.textures[k].rgb8, t.tex4.data() + at, n_px});
    at += n_px;
  }
  // (sky_w strictly below 2^24: sky_color() forms y * sky_w with a 24-bit multiply, which keeps only the low 24 bits of each
  //  operand — at sky_w == 2^24 the device product would be 0; y <= sky_h - 1 < 2^24 either way)
  t.sky_fast = sc.sky_mode == RT_SKY_TEXTURE && sc.sky_w < (1ull << 24) && sc.sky_h <= (1ull << 24) && sc.sky_w * sc.sky_h < (1ull << 31);
  if (t.sky_fast) {
    t.sky4.resize(sc.sky_w * sc.sky_h);
    if (!t.sky4.empty()) spans.push_back(Span{sc.sky_rgb8, t.sky4.data(), sc.sky_w * sc.sky_h});
    total += sc.sky_w * sc.sky_h;
  }
  constexpr uint64_t RUN = 1ull << 18;
  std::vector<Span> runs;
  for (const Span& sp : spans)
    for (uint64_t o = 0; o < sp.n; o += RUN) runs.push_back(Span{sp.src + 3 * o, sp.dst + o, std::min(RUN, sp.n - o)});
  std::atomic<size_t> next{0};
  auto work = [&]() {
    for (size_t i; (i = next.fetch_add(1)) < runs.size();) {
      const uint8_t* p = runs[i].src;
      uint32_t* out = runs[i].dst;
      for (uint64_t k = 0; k < runs[i].n; ++k) out[k] = (uint32_t)p[3 * k] | ((uint32_t)p[3 * k + 1] << 8) | ((uint32_t)p[3 * k + 2] << 16);
    }
  };
  unsigned hw = std::thread::hardware_concurrency();
  const unsigned n_threads = (unsigned)std::min<size_t>(std::min<unsigned>(hw ? hw : 1u, 8u), runs.size());
  std::vector<std::thread> pool;
  for (unsigned i = 1; i < n_threads; ++i) pool.emplace_back(work);
  work();
  for (auto& th : pool) th.join();
}

inline void fill_dev_scene(const RtScene& sc, const HostTables& t, DevScene& d) {
  std::memset(&d, 0, sizeof d);
  d.width = sc.width; d.height = sc.height; d.spp = sc.samples_per_pixel; d.max_depth = sc.max_depth;
  d.sky_mode = sc.sky_mode; d.n_spheres = sc.n_spheres; d.n_lights = (uint32_t)t.lights.size();
  d.light_nest_pool = 1u;
  d.seed_lo = (uint32_t)sc.seed; d.seed_hi = (uint32_t)(sc.seed >> 32);
  d.light_thr[0] = 1.0 - (double)d.n_lights * 0.1;   // raytracer.rs:100, the reference's operations (no contraction: -ffp-contract=off)
  d.light_thr[1] = 1.0 - (double)d.n_lights * 0.05;  // Glass (raytracer.rs:94-96)
  for (int i = 0; i < 3; ++i) {
    d.cam_origin[i] = sc.cam_origin[i]; d.cam_ll[i] = sc.cam_lower_left[i];
    d.cam_h[i] = sc.cam_horizontal[i]; d.cam_v[i] = sc.cam_vertical[i];
  }
  d.sky_w = sc.sky_w; d.sky_h = sc.sky_h;
  d.sky_wm1_f = (float)(sc.sky_w - 1); d.sky_hm1_f = (float)(sc.sky_h - 1);
  d.sky_fast = t.sky_fast ? 1u : 0u;  // (build_texels must have run: callers set sky4 / tex4 next to it)
  d.wm1 = (double)sc.width - 1.0; d.hm1 = (double)sc.height - 1.0; d.height_d = (double)sc.height;
  d.inv_wm1 = sc.width > 1 ? 1.0 / d.wm1 : 0.0; d.inv_hm1 = sc.height > 1 ? 1.0 / d.hm1 : 0.0;
  d.cam_fast = (d.inv_wm1 != 0.0 && d.inv_hm1 != 0.0) ? 1u : 0u;
  d.grid = t.grid;
}

}  // namespace rtc
