// rt_hip_api.hip — extern "C" entry points of librt_hip.so (include/rt_abi.h): scene upload
// to HBM, megakernel launch on a caller-supplied stream, HIP-event timing, counters.
// This is the drop-in for the rayon loop at reference raytracer.rs:260-262.  There is no CPU
// path in this library: without a gfx950 device every call fails with RT_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "rt_kernel.hip"
#include "rt_tables.h"

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& m) { g_err = m; return code; }

#define RT_HIP_TRY(expr)                                                                            \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess)                                                                           \
      return fail(RT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                   \
  } while (0)

}  // namespace

struct RtHipScene {
  int device = 0;
  RtScene host{};          // scalar fields only (pointers are not kept)
  rtc::DevScene dev{};     // device pointers filled in
  bool has_lights = false;
  void* d_geom = nullptr; void* d_mat = nullptr; void* d_cull = nullptr; void* d_lights = nullptr;
  void* d_tex = nullptr; void* d_sky = nullptr;
  unsigned long long* d_counters = nullptr;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  hipStream_t last_stream = nullptr;
  bool launched = false;
  uint32_t last_rows = 0;
  int variant = 0;
  int pool = 1;            // 1: pooled samples + exact fixed-point pixel sums; 0: reference f32 order
  std::chrono::steady_clock::time_point t_launch;
};

extern "C" const char* rt_hip_last_error(void) { return g_err.c_str(); }

extern "C" const char* rt_strerror(int code) {
  switch (code) {
    case RT_OK: return "ok";
    case RT_ERR_INVALID: return "invalid argument or inconsistent scene";
    case RT_ERR_NO_DEVICE: return "no gfx950 GPU visible (this library has no CPU fallback)";
    case RT_ERR_HIP: return "HIP runtime error";
    case RT_ERR_IO: return "Unable to read config file.";
    case RT_ERR_PARSE: return "Unable to parse config json";
    case RT_ERR_TEXTURE: return "failed to open/decode texture";
    case RT_ERR_PNG: return "error writing image";
    case RT_ERR_UNSUPPORTED: return "unsupported";
    default: return "unknown error";
  }
}

extern "C" int rt_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" void rt_hip_scene_destroy(RtHipScene* s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  for (void* p : {s->d_geom, s->d_mat, s->d_cull, s->d_lights, s->d_tex, s->d_sky, (void*)s->d_counters})
    if (p) (void)hipFree(p);
  if (s->ev_start) (void)hipEventDestroy(s->ev_start);
  if (s->ev_stop) (void)hipEventDestroy(s->ev_stop);
  delete s;
}

namespace {
template <typename T>
int upload(void** dst, const std::vector<T>& v) {
  size_t bytes = v.size() * sizeof(T);
  RT_HIP_TRY(hipMalloc(dst, bytes ? bytes : 16));
  if (bytes) RT_HIP_TRY(hipMemcpy(*dst, v.data(), bytes, hipMemcpyHostToDevice));
  return RT_OK;
}
}  // namespace

extern "C" int rt_hip_scene_create(const RtScene* scene, int device, RtHipScene** out) {
  if (!scene || !out) return fail(RT_ERR_INVALID, "null argument");
  *out = nullptr;
  int n = rt_hip_device_count();
  if (n <= 0) return fail(RT_ERR_NO_DEVICE, rt_strerror(RT_ERR_NO_DEVICE));
  if (device < 0 || device >= n) return fail(RT_ERR_INVALID, "device index out of range");
  rtc::HostTables t;
  std::string why = rtc::build_tables(*scene, t);
  if (!why.empty()) return fail(RT_ERR_INVALID, why);
  if (scene->n_spheres > 65535u) return fail(RT_ERR_UNSUPPORTED, "more than 65535 spheres (u16 candidate lists)");
  RT_HIP_TRY(hipSetDevice(device));
  RtHipScene* s = new RtHipScene;
  s->device = device;
  s->host = *scene;
  s->host.spheres = nullptr; s->host.textures = nullptr; s->host.sky_rgb8 = nullptr;
  s->has_lights = !t.lights.empty();
  rtc::fill_dev_scene(*scene, t, s->dev);
  int rc;
  auto bail = [&](int code) { rt_hip_scene_destroy(s); return code; };
  if ((rc = upload(&s->d_geom, t.geom)) != RT_OK) return bail(rc);
  if ((rc = upload(&s->d_mat, t.mat)) != RT_OK) return bail(rc);
  if ((rc = upload(&s->d_cull, t.cull)) != RT_OK) return bail(rc);
  if ((rc = upload(&s->d_lights, t.lights)) != RT_OK) return bail(rc);
  {
    std::vector<uint8_t> blob(t.tex_bytes);
    for (uint32_t i = 0; i < scene->n_textures; ++i)
      if (scene->textures[i].nbytes) std::memcpy(&blob[t.tex_off[i]], scene->textures[i].rgb8, scene->textures[i].nbytes);
    if ((rc = upload(&s->d_tex, blob)) != RT_OK) return bail(rc);
  }
  {
    std::vector<uint8_t> sky;
    if (scene->sky_mode == RT_SKY_TEXTURE) sky.assign(scene->sky_rgb8, scene->sky_rgb8 + scene->sky_w * scene->sky_h * 3);
    if ((rc = upload(&s->d_sky, sky)) != RT_OK) return bail(rc);
  }
  if (hipMalloc((void**)&s->d_counters, 4 * sizeof(unsigned long long)) != hipSuccess ||
      hipEventCreate(&s->ev_start) != hipSuccess || hipEventCreate(&s->ev_stop) != hipSuccess)
    return bail(fail(RT_ERR_HIP, "hipMalloc/hipEventCreate failed"));
  s->dev.geom = (const rtc::SphereGeom*)s->d_geom; s->dev.mat = (const rtc::SphereMat*)s->d_mat;
  s->dev.cull = (const rtc::CullPair*)s->d_cull; s->dev.lights = (const uint32_t*)s->d_lights;
  s->dev.tex = (const uint8_t*)s->d_tex; s->dev.sky = (const uint8_t*)s->d_sky;
  *out = s;
  return RT_OK;
}

extern "C" int rt_hip_set_option(RtHipScene* s, const char* key, int64_t value) {
  if (!s || !key) return fail(RT_ERR_INVALID, "null argument");
  if (!std::strcmp(key, "variant")) { if (value < 0 || value > 2) return fail(RT_ERR_INVALID, "variant must be 0, 1 or 2"); s->variant = (int)value; return RT_OK; }
  if (!std::strcmp(key, "pool")) { s->pool = value != 0; return RT_OK; }
  if (!std::strcmp(key, "samples_per_pixel")) { s->host.samples_per_pixel = s->dev.spp = (uint32_t)value; return RT_OK; }
  if (!std::strcmp(key, "max_depth")) { s->host.max_depth = s->dev.max_depth = (uint32_t)value; return RT_OK; }
  if (!std::strcmp(key, "seed")) { s->host.seed = (uint64_t)value; s->dev.seed_lo = (uint32_t)value; s->dev.seed_hi = (uint32_t)((uint64_t)value >> 32); return RT_OK; }
  return fail(RT_ERR_INVALID, std::string("unknown option ") + key);
}

extern "C" int rt_hip_render(RtHipScene* s, const RtRowTiles* tiles, void* d_rgb8, void* d_linear, void* stream_) {
  if (!s) return fail(RT_ERR_INVALID, "null argument");
  if (!d_rgb8 && rt_tiles_local_rows(s->host.height, tiles) != 0) return fail(RT_ERR_INVALID, "null framebuffer");
  hipStream_t stream = (hipStream_t)stream_;
  RT_HIP_TRY(hipSetDevice(s->device));
  rtk::KArgs ka;
  ka.sc = s->dev;
  ka.out_rgb8 = (uint8_t*)d_rgb8; ka.out_linear = (float*)d_linear; ka.counters = s->d_counters;
  ka.local_rows = rt_tiles_local_rows(s->host.height, tiles);
  const bool tiled = tiles && tiles->tile_rows && tiles->tile_stride;
  ka.tile_rows = tiled ? tiles->tile_rows : 0; ka.first_tile = tiled ? tiles->first_tile : 0;
  ka.tile_stride = tiled ? tiles->tile_stride : 0;
  s->last_rows = ka.local_rows;
  s->last_stream = stream;
  s->t_launch = std::chrono::steady_clock::now();
  RT_HIP_TRY(hipMemsetAsync(s->d_counters, 0, 4 * sizeof(unsigned long long), stream));
  if (ka.local_rows == 0) { s->launched = false; return RT_OK; }
  const uint32_t tiles_x = (s->host.width + rtk::TILE_W - 1) / rtk::TILE_W;
  const uint32_t tiles_y = (ka.local_rows + rtk::TILE_H - 1) / rtk::TILE_H;
  const dim3 grid(tiles_x * tiles_y), block(rtk::BLOCK);
  RT_HIP_TRY(hipEventRecord(s->ev_start, stream));
  const bool geom_lds = s->host.n_spheres <= rtk::LDS_GEOM_MAX_SPHERES && s->variant != 1;
  const size_t lds_bytes = rtk::LDS_GEOM_OFF + (geom_lds ? (size_t)s->host.n_spheres * sizeof(rtc::SphereGeom) : 0);
  const bool pool = s->pool != 0;
#define RT_LAUNCH(HL, V, G, P) hipLaunchKernelGGL((rtk::rt_megakernel<HL, V, G, P>), grid, block, lds_bytes, stream, ka)
#define RT_LAUNCH_P(HL, V, G) do { if (pool) RT_LAUNCH(HL, V, G, true); else RT_LAUNCH(HL, V, G, false); } while (0)
#define RT_LAUNCH_V(HL, G)                                     \
  do {                                                         \
    if (s->variant == 1) RT_LAUNCH_P(HL, 1, false);            \
    else if (s->variant == 2) RT_LAUNCH_P(HL, 2, G);           \
    else RT_LAUNCH_P(HL, 0, G);                                \
  } while (0)
  if (s->has_lights) { if (geom_lds) RT_LAUNCH_V(true, true); else RT_LAUNCH_V(true, false); }
  else { if (geom_lds) RT_LAUNCH_V(false, true); else RT_LAUNCH_V(false, false); }
#undef RT_LAUNCH_V
#undef RT_LAUNCH_P
#undef RT_LAUNCH
  RT_HIP_TRY(hipGetLastError());
  RT_HIP_TRY(hipEventRecord(s->ev_stop, stream));
  s->launched = true;
  return RT_OK;
}

extern "C" int rt_hip_wait(RtHipScene* s, RtStats* stats) {
  if (!s) return fail(RT_ERR_INVALID, "null argument");
  RT_HIP_TRY(hipSetDevice(s->device));
  RT_HIP_TRY(hipStreamSynchronize(s->last_stream));
  if (stats) {
    std::memset(stats, 0, sizeof *stats);
    unsigned long long c[4] = {0, 0, 0, 0};
    RT_HIP_TRY(hipMemcpy(c, s->d_counters, sizeof c, hipMemcpyDeviceToHost));
    float ms = 0.f;
    if (s->launched) RT_HIP_TRY(hipEventElapsedTime(&ms, s->ev_start, s->ev_stop));
    stats->samples = (uint64_t)s->last_rows * s->host.width * s->host.samples_per_pixel;
    stats->segments = c[0];
    stats->sphere_tests = c[0] * (uint64_t)s->host.n_spheres;
    stats->exact_tests = c[1];
    stats->tex_oob = c[2];
    stats->kernel_ms = ms;
    stats->frame_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - s->t_launch).count();
  }
  return RT_OK;
}

// drop-in for the parallel loop of render() (raytracer.rs:254-263): host scene in, host RGB8 out
extern "C" int rt_render_rgb8(const RtScene* scene, uint8_t* out_rgb8, RtStats* stats) {
  if (!scene || !out_rgb8) return fail(RT_ERR_INVALID, "null argument");
  auto t0 = std::chrono::steady_clock::now();
  RtHipScene* s = nullptr;
  int rc = rt_hip_scene_create(scene, 0, &s);
  if (rc != RT_OK) return rc;
  const size_t bytes = (size_t)scene->width * scene->height * 3;
  void* d_out = nullptr;
  if (hipMalloc(&d_out, bytes) != hipSuccess) { rt_hip_scene_destroy(s); return fail(RT_ERR_HIP, "hipMalloc(framebuffer) failed"); }
  rc = rt_hip_render(s, nullptr, d_out, nullptr, nullptr);
  RtStats st;
  if (rc == RT_OK) rc = rt_hip_wait(s, &st);
  if (rc == RT_OK && hipMemcpy(out_rgb8, d_out, bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(RT_ERR_HIP, "hipMemcpy(framebuffer) failed");
  (void)hipFree(d_out);
  rt_hip_scene_destroy(s);
  if (rc == RT_OK && stats) {
    *stats = st;
    stats->frame_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  return rc;
}

// math self-test hook (see rtk::rt_math_probe); all pointers are DEVICE pointers
extern "C" int rt_hip_math_probe(const double* x, const double* y, double* out_sqrt, double* out_div, float* out_sqrtf,
                                 double* out_atan2, uint32_t n, void* stream) {
  hipLaunchKernelGGL(rtk::rt_math_probe, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, y, out_sqrt, out_div,
                     out_sqrtf, out_atan2, n);
  RT_HIP_TRY(hipGetLastError());
  return RT_OK;
}
