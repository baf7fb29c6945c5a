// rt_tables.h — flatten an RtScene (include/rt_abi.h) into the HBM table layout of
// rt_core.h::DevScene.  Plain host C++; used by rt_hip_api.hip (upload) and tests/hostsim.
#pragma once
#include <cstring>
#include <string>
#include <vector>

#include "rt_core.h"

namespace rtc {

struct HostTables {
  std::vector<SphereGeom> geom;
  std::vector<SphereMat> mat;
  std::vector<CullPair> cull;
  std::vector<uint32_t> lights;
  std::vector<uint64_t> tex_off;  // per RtTexture, byte offset in the blob
  uint64_t tex_bytes = 0;
  uint32_t n_pairs = 0;           // real pairs (cull.size() includes chunk padding)
  bool simple_colour = true;  // no lights and every albedo in [0,1]
};

// returns "" or a description of why the scene is invalid (RT_ERR_INVALID)
inline std::string build_tables(const RtScene& sc, HostTables& t) {
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
  // pairs, padded to a whole number of CULL_CHUNK-pair chunks plus one chunk of slack so the
  // scan may prefetch one chunk past the end; padding entries can never pass (R = -inf)
  const uint32_t n_pairs = (n + 1) / 2;
  const uint32_t padded = (n_pairs + CULL_CHUNK - 1) / CULL_CHUNK * CULL_CHUNK + CULL_CHUNK;
  CullPair never;
  for (int k = 0; k < 2; ++k) { never.cx[k] = never.cy[k] = never.cz[k] = 0.0f; never.R[k] = -INFINITY; }
  t.cull.assign(padded, never);
  t.n_pairs = n_pairs;
  t.lights.clear();
  t.simple_colour = true;
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
    }
    t.mat[i] = m;
    if (s.kind == RT_MAT_LIGHT) t.lights.push_back(i);
    if (s.kind == RT_MAT_LAMBERTIAN || s.kind == RT_MAT_METAL)
      for (int c = 0; c < 3; ++c)
        if (!(s.albedo[c] >= 0.0f && s.albedo[c] <= 1.0f)) t.simple_colour = false;
    CullPair& cp = t.cull[i / 2];
    build_cull_entry(s, &cp.cx[i & 1], &cp.cy[i & 1], &cp.cz[i & 1], &cp.R[i & 1]);
  }
  if (n & 1) {  // odd count: the second half of the last real pair can never pass either
    CullPair& cp = t.cull[n / 2];
    cp.cx[1] = cp.cx[0]; cp.cy[1] = cp.cy[0]; cp.cz[1] = cp.cz[0]; cp.R[1] = -INFINITY;
  }
  if (!t.lights.empty()) t.simple_colour = false;
  return "";
}

inline void fill_dev_scene(const RtScene& sc, const HostTables& t, DevScene& d) {
  std::memset(&d, 0, sizeof d);
  d.width = sc.width; d.height = sc.height; d.spp = sc.samples_per_pixel; d.max_depth = sc.max_depth;
  d.sky_mode = sc.sky_mode; d.n_spheres = sc.n_spheres; d.n_lights = (uint32_t)t.lights.size();
  d.n_pairs = t.n_pairs;
  d.seed_lo = (uint32_t)sc.seed; d.seed_hi = (uint32_t)(sc.seed >> 32);
  for (int i = 0; i < 3; ++i) {
    d.cam_origin[i] = sc.cam_origin[i]; d.cam_ll[i] = sc.cam_lower_left[i];
    d.cam_h[i] = sc.cam_horizontal[i]; d.cam_v[i] = sc.cam_vertical[i];
  }
  d.sky_w = sc.sky_w; d.sky_h = sc.sky_h;
}

}  // namespace rtc
