// rt_hip_api.hip — extern "C" entry points of librt_hip.so (include/rt_abi.h): scene upload
// to HBM, megakernel launch on a caller-supplied stream, HIP-event timing, counters.
// This is the drop-in for the rayon loop at reference raytracer.rs:260-262.  There is no CPU
// path in this library: without a gfx950 device every call fails with RT_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <mutex>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "rt_kernel.hip"
#include "rt_tables.h"
#ifdef RT_TEST_PROBES  // librt_hip_probe.so: the device probes and debug calls of include/rt_abi_test.h (test infrastructure)
#include "../../../include/rt_abi_test.h"
#endif

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

struct RtHipScene;
namespace { int warm_up(RtHipScene* s); }

// Where set-up time goes (rt_hip_setup_profile): the stages of the most recent scene / group creation and one-shot render of
// this process, in milliseconds — rank 0's scene and the group's own work; other ranks' scenes are created beside it.
namespace rtp {
std::mutex g_mu;
std::vector<std::pair<std::string, double>> g_stages;
thread_local bool tl_record = true;     // (the group's upload threads of ranks >= 1 switch it off)
thread_local bool tl_in_group = false;  // a group is being created on this thread: its scene's stages are appended to the group's
                                        // (a scene created on its own starts a profile of its own: the list never grows without bound)
thread_local std::string tl_text;
void reset() { std::lock_guard<std::mutex> lk(g_mu); g_stages.clear(); }
void add(const char* name, double ms) { if (!tl_record) return; std::lock_guard<std::mutex> lk(g_mu); g_stages.emplace_back(name, ms); }
struct Clock {   // mark("x") books the time since the previous mark under x
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void mark(const char* name) {
    const auto n = std::chrono::steady_clock::now();
    add(name, std::chrono::duration<double, std::milli>(n - t).count());
    t = n;
  }
};
}  // namespace rtp
extern "C" const char* rt_hip_setup_profile(void) {
  std::lock_guard<std::mutex> lk(rtp::g_mu);
  std::string& s = rtp::tl_text;
  s = "{";
  char buf[96];
  for (size_t i = 0; i < rtp::g_stages.size(); ++i) {
    std::snprintf(buf, sizeof buf, "%s\"%s\":%.3f", i ? "," : "", rtp::g_stages[i].first.c_str(), rtp::g_stages[i].second);
    s += buf;
  }
  s += "}";
  return s.c_str();
}

constexpr uint32_t RT_TIMELINE_WAVES = 8192;  // profile builds: {start, end} wall clock per wave behind the counters

struct RtHipScene {
  int device = 0;
  bool owns_tables = true;  // false: a VIEW of another scene (rt_hip_scene_clone_view): it shares that scene's tables, textures and host copies
                            // and owns only what a launch writes — counters, stats slots, queue order, light overflow, framebuffer
  RtScene host{};          // scalar fields only (pointers are not kept)
  rtc::DevScene dev{};     // device pointers filled in
  bool has_lights = false, simple_colour = false;
  void* d_geom = nullptr; void* d_mat = nullptr; void* d_lights = nullptr;
  void* d_tex = nullptr; void* d_sky = nullptr; void* d_tex4 = nullptr; void* d_sky4 = nullptr;
  // The host copies of the BIG uploads (texels: 29 MB for the reference's test scene) live as long as the scene.  hipMemcpy from
  // pageable memory pins the source pages for the device (a userptr mapping the runtime caches); giving such memory back to
  // the OS (free -> munmap) fires the driver's MMU notifier, which EVICTS the process's hardware queues and restores them
  // 10 - 25 ms later — the next kernel launch waits for that.  Found in round 6 as a one-shot frame of the test scene whose
  // first launch started 20 ms late in two runs of three (profiles/r06_run5_first_launch_wait.log); with the buffers kept,
  // 12 of 12 runs start in 0.07 ms.  They are freed with the scene, when nothing waits for the queues.
  rtc::TexelVec keep_tex4, keep_sky4;
  std::vector<uint8_t> keep_tex_rgb8, keep_sky_rgb8;
  size_t texel_bytes = 0;
  void* d_matc = nullptr; void* d_cell_word = nullptr; void* d_cell_items = nullptr; void* d_large = nullptr;
  void* d_all = nullptr;   // 0..n-1: the `large` list of the brute-force arm (variant 1)
  void* d_large_geom = nullptr;
  rtc::GridDesc grid{};    // the product grid (variant 0)
  unsigned long long* d_counters = nullptr;  // 4 counters + the work-queue cursor
  int num_cus = 0;
  int cfg_key = -1; size_t cfg_lds = 0; int cfg_per_cu = 0;  // cached launch configuration
  void* d_frame = nullptr; size_t frame_bytes = 0;           // framebuffer of rt_hip_render_to_host
  // queue order feedback (rt_kernel.hip KArgs::tile_order): depths measured by the last frame of this tile geometry
  uint32_t* d_tile_depth = nullptr; uint32_t* d_tile_order = nullptr; size_t order_cap = 0;
  struct OrderKey {         // tile geometry (+ row tiles) an order belongs to: compared field by field
    uint32_t n_tiles = 0, tile_log2 = 0, tile_shape = 0, aff_group_log2 = 0, tile_rows = 0, first_tile = 0, tile_stride = 0, local_rows = 0;
    bool operator==(const OrderKey& o) const {
      return n_tiles == o.n_tiles && tile_log2 == o.tile_log2 && tile_shape == o.tile_shape && aff_group_log2 == o.aff_group_log2 &&
             tile_rows == o.tile_rows && first_tile == o.first_tile && tile_stride == o.tile_stride && local_rows == o.local_rows;
    }
  } order_key;              // n_tiles == 0: none yet
  bool order_ready = false; // d_tile_order holds an order for order_key
  bool depth_fresh = false; // d_tile_depth holds depths of THIS view (measured by its last frame) that d_tile_order does not reflect yet
  int order_age = 0;        // frames since the order was last invalidated (geometry / camera / option change)
  int tile_affinity = 1;    // "tile_affinity" option: runs of tiles belong to one XCD's queue (framebuffer lines complete in one L2)
  int order_mode = 2;       // "tile_order" option: 0 top row first, 1 bottom row first, 2 deepest tiles of the previous frame first
                            // (a frame without a previous one: bottom row first)
  int force_lit = 0;       // "force_lit" option (diagnostics)
  int light_pool_cap = 0;  // "light_pool" option: cap on the light-frame pool of lit scenes (0 = automatic; tests shrink it to force repeats and overflows)
  int light_base_cap = 0;  // "light_base_pool" option: the same for the pool of colour-map bases
  int light_nest_pool = 1; // "light_nest_pool" option: 0 = nested light activations always go through the HBM overflow (tests)
  void* d_light_overflow = nullptr; size_t light_overflow_bytes = 0;  // lit scenes: 560 B per lane of the largest launch so far (rt_core.h lane_light_begin)
  size_t lds_cap = 0;      // dynamic LDS a workgroup may ask for on this device
  uint32_t last_pool_slots = 0, last_base_slots = 0; size_t last_lds_bytes = 0; bool last_lds_tables = false;  // of the last launch (rt_hip_scene_query)
  int chunk_spp = 0;       // 0 = automatic
  int tile_batch = 0;      // 0 = automatic; else tiles a workgroup takes from the queue per atomic, 1..64
  int tile_log2 = -1;      // -1 = automatic; else tiles of 4^k pixels, k = 0..3
  int tile_shape = 0;      // 0: 2^k x 2^k squares (default: 0.9 % faster); 1: runs of 4^k pixels of one scanline (contiguous
                           // framebuffer bytes: HBM writes 10.9 -> 5.9 MiB per 1200x800 frame, profiles/r02_run8_*)

  // What rt_hip_wait reports about a launch lives in one of two SLOTS, used alternately: its event pair, the rows and waves
  // it covered, and a pinned host copy of its counters that an async copy fills right behind the kernel (stream-ordered:
  // the next launch's counter reset cannot overtake it).  Two launches of a scene may therefore be in flight on its stream
  // — frame i+1 rendering while frame i is gathered, rt_hip_group_submit / _collect — and a wait never issues a synchronous
  // device-to-host copy (it cost every rank of a multi-GPU frame ~25 us on the frame's critical path).
  struct Slot {
    hipEvent_t ev_start = nullptr, ev_stop = nullptr, ev_copied = nullptr;
    unsigned long long* h_counters = nullptr;  // pinned, RT_SLOT_COUNTERS words
    uint32_t rows = 0;
    uint64_t samples = 0;      // rows x width x samples per pixel AS LAUNCHED (an option set between submit and collect must not show)
    uint64_t waves = 0;
    bool launched = false;   // a kernel ran for it (false: an empty shard)
    std::chrono::steady_clock::time_point t_launch;
  } slot[2];
  uint64_t n_launches = 0;   // launch i uses slot[i & 1]
  hipStream_t last_stream = nullptr;
  bool in_flight = false;  // a launch has been enqueued and rt_hip_wait has not returned for it yet
  uint64_t last_waves = 0;
  uint32_t last_tiles_x = 0;
  int variant = 0;
  Slot& last_slot() { return slot[(n_launches + 1) & 1]; }  // the slot of the most recent launch
};
constexpr uint32_t RT_SLOT_COUNTERS = 32;  // segments, exact tests, tex_oob, grid steps, 4 x wave trip counts, 8 x section cycles, profile clocks, (the tile-queue cursors,) [28] repeated segments

extern "C" const char* rt_hip_last_error(void) { return g_err.c_str(); }

extern "C" uint32_t rt_abi_version(void) { return RT_ABI_VERSION; }
extern "C" size_t rt_abi_sizeof(const char* name) {
  if (!name) return 0;
  if (!std::strcmp(name, "RtSphere")) return sizeof(RtSphere);
  if (!std::strcmp(name, "RtTexture")) return sizeof(RtTexture);
  if (!std::strcmp(name, "RtScene")) return sizeof(RtScene);
  if (!std::strcmp(name, "RtRowTiles")) return sizeof(RtRowTiles);
  if (!std::strcmp(name, "RtStats")) return sizeof(RtStats);
  if (!std::strcmp(name, "RtGroupInfo")) return sizeof(RtGroupInfo);
  if (!std::strcmp(name, "RtGroupRank")) return sizeof(RtGroupRank);
  return 0;
}

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

extern "C" int rt_hip_device_warm(int device) {
  const int n = rt_hip_device_count();
  if (n <= 0) return fail(RT_ERR_NO_DEVICE, rt_strerror(RT_ERR_NO_DEVICE));
  if (device < 0 || device >= n) return fail(RT_ERR_INVALID, "device index out of range");
  RT_HIP_TRY(hipSetDevice(device));
  void* p = nullptr;
  RT_HIP_TRY(hipMalloc(&p, 256));  // (the first allocation creates the device's context)
  hipLaunchKernelGGL(rtk::rt_warm_up, dim3(1), dim3(64), 0, nullptr);  // (the first launch loads this library's code object)
  RT_HIP_TRY(hipGetLastError());
  if (!std::getenv("RT_NO_SCRATCH_WARMUP")) {
    hipLaunchKernelGGL(rtk::rt_warm_up_scratch, dim3(1), dim3(64), 0, nullptr, (uint32_t*)nullptr, 1u);  // (... and gives the NULL stream's queue its scratch memory)
    RT_HIP_TRY(hipGetLastError());
  }
  RT_HIP_TRY(hipDeviceSynchronize());
  (void)hipFree(p);
  return RT_OK;
}

extern "C" void rt_hip_scene_destroy(RtHipScene* s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  if (s->owns_tables)
    for (void* p : {s->d_geom, s->d_mat, s->d_lights, s->d_tex, s->d_sky, s->d_tex4, s->d_sky4, s->d_matc, s->d_cell_word, s->d_cell_items, s->d_large,
                    s->d_all, s->d_large_geom})
      if (p) (void)hipFree(p);
  for (void* p : {(void*)s->d_counters, s->d_frame, (void*)s->d_tile_depth, (void*)s->d_tile_order, s->d_light_overflow})
    if (p) (void)hipFree(p);
  for (auto& sl : s->slot) {
    for (hipEvent_t e : {sl.ev_start, sl.ev_stop, sl.ev_copied}) if (e) (void)hipEventDestroy(e);
    if (sl.h_counters) (void)hipHostFree(sl.h_counters);
  }
  delete s;
}

namespace {
template <typename V>
int upload(void** dst, const V& v) {
  size_t bytes = v.size() * sizeof(typename V::value_type);
  RT_HIP_TRY(hipMalloc(dst, bytes ? bytes : 16));
  if (bytes) RT_HIP_TRY(hipMemcpy(*dst, v.data(), bytes, hipMemcpyHostToDevice));
  return RT_OK;
}
}  // namespace

namespace {
// what a LAUNCH of a scene writes: the counter block with the tile-queue cursors, the two stats slots (events + pinned words)
int alloc_launch_state(RtHipScene* s) {
  if (hipMalloc((void**)&s->d_counters, (32 + 4 * RT_TIMELINE_WAVES) * sizeof(unsigned long long)) != hipSuccess)
    return fail(RT_ERR_HIP, "hipMalloc(counters) failed");
  for (auto& sl : s->slot) {
    if (hipEventCreate(&sl.ev_start) != hipSuccess || hipEventCreate(&sl.ev_stop) != hipSuccess ||
        hipEventCreateWithFlags(&sl.ev_copied, hipEventDisableTiming) != hipSuccess ||
        hipHostMalloc((void**)&sl.h_counters, RT_SLOT_COUNTERS * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess)
      return fail(RT_ERR_HIP, "hipEventCreate/hipHostMalloc failed");
    std::memset(sl.h_counters, 0, RT_SLOT_COUNTERS * sizeof(unsigned long long));
  }
  if (hipMemset(s->d_counters, 0, (32 + 4 * RT_TIMELINE_WAVES) * sizeof(unsigned long long)) != hipSuccess)
    return fail(RT_ERR_HIP, "hipMemset failed");
  return RT_OK;
}
}  // namespace

// A second VIEW of a resident scene (internal; the group's overlapped frames): the same tables and textures in HBM, its own
// launch state — so that a launch of the view and a launch of the scene may be in flight on two streams AT ONCE (a scene itself
// is not re-entrant: one tile-queue cursor, one counter block).  The view must be destroyed before the scene it was cloned from.
int rt_hip_scene_clone_view(const RtHipScene* src, RtHipScene** out) {
  if (!src || !out) return fail(RT_ERR_INVALID, "null argument");
  *out = nullptr;
  RT_HIP_TRY(hipSetDevice(src->device));
  RtHipScene* s = new RtHipScene;
  s->device = src->device; s->owns_tables = false;
  s->host = src->host; s->dev = src->dev;
  s->has_lights = src->has_lights; s->simple_colour = src->simple_colour;
  s->d_geom = src->d_geom; s->d_mat = src->d_mat; s->d_lights = src->d_lights; s->d_tex = src->d_tex; s->d_sky = src->d_sky;
  s->d_tex4 = src->d_tex4; s->d_sky4 = src->d_sky4; s->texel_bytes = src->texel_bytes; s->d_matc = src->d_matc;
  s->d_cell_word = src->d_cell_word; s->d_cell_items = src->d_cell_items; s->d_large = src->d_large; s->d_all = src->d_all;
  s->d_large_geom = src->d_large_geom; s->grid = src->grid; s->num_cus = src->num_cus; s->lds_cap = src->lds_cap;
  s->tile_affinity = src->tile_affinity; s->order_mode = src->order_mode; s->force_lit = src->force_lit; s->light_pool_cap = src->light_pool_cap;
  s->light_base_cap = src->light_base_cap; s->light_nest_pool = src->light_nest_pool; s->chunk_spp = src->chunk_spp; s->tile_batch = src->tile_batch;
  s->tile_log2 = src->tile_log2; s->tile_shape = src->tile_shape; s->variant = src->variant;
  s->dev.light_overflow = nullptr;   // (its own: two launches at once park their lanes' records apart)
  int rc = alloc_launch_state(s);
  if (rc == RT_OK) rc = warm_up(s);  // (the launch configuration + the lit kernels' overflow slots, outside any frame)
  if (rc == RT_OK && hipDeviceSynchronize() != hipSuccess) rc = fail(RT_ERR_HIP, "hipDeviceSynchronize failed");
  if (rc != RT_OK) { rt_hip_scene_destroy(s); return rc; }
  *out = s;
  return RT_OK;
}

extern "C" int rt_hip_scene_create(const RtScene* scene, int device, RtHipScene** out) {
  if (!scene || !out) return fail(RT_ERR_INVALID, "null argument");
  *out = nullptr;
  int n = rt_hip_device_count();
  if (n <= 0) return fail(RT_ERR_NO_DEVICE, rt_strerror(RT_ERR_NO_DEVICE));
  if (device < 0 || device >= n) return fail(RT_ERR_INVALID, "device index out of range");
  static const bool trace = std::getenv("RT_GROUP_TRACE") != nullptr;  // (development: where a scene's creation time goes)
  const auto t_create = std::chrono::steady_clock::now();
  auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_create).count(); };
  if (!rtp::tl_in_group) rtp::reset();
  rtp::Clock pc;
  rtc::HostTables t;
  std::string why = rtc::build_tables(*scene, t);
  if (!why.empty()) return fail(RT_ERR_INVALID, why);
  pc.mark("scene.tables_and_grid");
  rtc::build_texels(*scene, t);
  pc.mark("scene.texels_rgbx");
  const double t_tables = since();
  RT_HIP_TRY(hipSetDevice(device));
  RtHipScene* s = new RtHipScene;
  s->device = device;
  s->host = *scene;
  s->host.spheres = nullptr; s->host.textures = nullptr; s->host.sky_rgb8 = nullptr;
  s->has_lights = !t.lights.empty();
  s->simple_colour = t.simple_colour;
  s->grid = t.grid;
  rtc::fill_dev_scene(*scene, t, s->dev);
  {
    hipDeviceProp_t prop;
    RT_HIP_TRY(hipGetDeviceProperties(&prop, device));
    s->num_cus = prop.multiProcessorCount;
    // 160 KB of LDS per CU on gfx950; the unlit layouts stay below LDS_TABLES_MAX_BYTES (156 KB) as before, the light pools may
    // take what the device says is left
    const size_t dev_lds = prop.sharedMemPerBlock > prop.maxSharedMemoryPerMultiProcessor ? prop.sharedMemPerBlock : prop.maxSharedMemoryPerMultiProcessor;
    s->lds_cap = std::max<size_t>(rtk::LDS_TABLES_MAX_BYTES, std::min<size_t>(dev_lds, 160u * 1024u));
  }
  pc.mark("scene.device_properties");
  int rc;
  auto bail = [&](int code) { rt_hip_scene_destroy(s); return code; };
  if ((rc = upload(&s->d_geom, t.geom)) != RT_OK) return bail(rc);
  if ((rc = upload(&s->d_mat, t.mat)) != RT_OK) return bail(rc);
  if ((rc = upload(&s->d_lights, t.lights)) != RT_OK) return bail(rc);
  if ((rc = upload(&s->d_matc, t.matc)) != RT_OK) return bail(rc);
  if ((rc = upload(&s->d_cell_word, t.cell_word)) != RT_OK) return bail(rc);
  if ((rc = t.grid.wide ? upload(&s->d_cell_items, t.cell_items32) : upload(&s->d_cell_items, t.cell_items)) != RT_OK) return bail(rc);  // (wide tables: the 32-bit item lists)
  if ((rc = upload(&s->d_large, t.large)) != RT_OK) return bail(rc);
  if ((rc = upload(&s->d_large_geom, t.large_geom)) != RT_OK) return bail(rc);
  {
    std::vector<uint32_t> all(scene->n_spheres);
    for (uint32_t i = 0; i < scene->n_spheres; ++i) all[i] = i;
    if ((rc = upload(&s->d_all, all)) != RT_OK) return bail(rc);
  }
  pc.mark("scene.upload_tables");
  // textures and sky: resident as 4-byte texels (rt_tables.h build_texels; one dword load per fetch); the caller's RGB8
  // bytes are uploaded too only if some record is outside that path's range (rt_core.h texels_fast)
  s->texel_bytes = (t.tex4.size() + t.sky4.size()) * 4u;
  if ((rc = upload(&s->d_tex4, t.tex4)) != RT_OK) return bail(rc);
  if ((rc = upload(&s->d_sky4, t.sky4)) != RT_OK) return bail(rc);
  {
    std::vector<uint8_t>& blob = s->keep_tex_rgb8;
    blob.assign(t.need_rgb8 ? t.tex_bytes : 0, 0);
    if (t.need_rgb8)
      for (uint32_t i = 0; i < scene->n_textures; ++i)
        if (scene->textures[i].nbytes) std::memcpy(&blob[t.tex_off[i]], scene->textures[i].rgb8, scene->textures[i].nbytes);
    if ((rc = upload(&s->d_tex, blob)) != RT_OK) return bail(rc);
  }
  {
    std::vector<uint8_t>& sky = s->keep_sky_rgb8;
    if (scene->sky_mode == RT_SKY_TEXTURE && !t.sky_fast) sky.assign(scene->sky_rgb8, scene->sky_rgb8 + scene->sky_w * scene->sky_h * 3);
    if ((rc = upload(&s->d_sky, sky)) != RT_OK) return bail(rc);
  }
  if (!std::getenv("RT_FREE_HOST_TEXELS")) { s->keep_tex4.swap(t.tex4); s->keep_sky4.swap(t.sky4); }  // (the variable: the round-5 behaviour, for the A/B)
  pc.mark("scene.upload_texels");
  if ((rc = alloc_launch_state(s)) != RT_OK) return bail(rc);
  s->dev.geom = (const rtc::SphereGeom*)s->d_geom; s->dev.mat = (const rtc::SphereMat*)s->d_mat;
  s->dev.lights = (const uint32_t*)s->d_lights;
  s->dev.tex = (const uint8_t*)s->d_tex; s->dev.sky = (const uint8_t*)s->d_sky;
  s->dev.tex4 = (const uint32_t*)s->d_tex4; s->dev.sky4 = (const uint32_t*)s->d_sky4;
  s->dev.matc = (const rtc::MatCore*)s->d_matc; s->dev.cell_word = (const uint32_t*)s->d_cell_word;
  s->dev.cell_items = (const uint16_t*)s->d_cell_items; s->dev.large = (const uint32_t*)s->d_large;
  s->dev.large_geom = (const rtc::SphereGeom*)s->d_large_geom;
  // Everything above went through the NULL stream — and hipMemset / hipMemcpy from pageable memory return before the device
  // has finished (they are asynchronous to the host: the fill / the DMA out of the staging buffer may still be queued).
  // The frames run on the CALLER's streams, and a non-blocking stream does not order itself behind the NULL stream: a first
  // frame launched right away could start before the tables had landed, or have its tile-queue cursor zeroed under it
  // (round 4: the group launches rank 0 from the creating thread at once — 65 % of fresh 3-rank groups traced tiles twice).
  // ... and nothing a first frame should pay for is left for it: the code object on the device, the default configuration's
  // kernel attribute / occupancy / (lit scenes) overflow slots (a one-shot rt_render_rgb8 — the reference renders one frame per
  // process — reports this under setup_ms, outside its frame_ms window)
  pc.mark("scene.counters_events_pinned_words");
  const double t_uploaded = since();
  if ((rc = warm_up(s)) != RT_OK) return bail(rc);
  pc.mark("scene.kernel_configuration");
  const double t_warm = since();
  if (hipDeviceSynchronize() != hipSuccess) return bail(fail(RT_ERR_HIP, "hipDeviceSynchronize failed"));
  pc.mark("scene.device_idle");
  if (trace) std::fprintf(stderr, "[rt scene] create: tables %.2f ms, uploads + events done %.2f, warm_up (module, configuration, overflow) %.2f, device idle %.2f\n", t_tables, t_uploaded, t_warm, since());
  *out = s;
  return RT_OK;
}

extern "C" int rt_hip_set_option(RtHipScene* s, const char* key, int64_t value) {
  if (!s || !key) return fail(RT_ERR_INVALID, "null argument");
  constexpr int64_t max_variant = 1;
  if (!std::strcmp(key, "variant")) { if (value < 0 || value > max_variant) return fail(RT_ERR_INVALID, "variant must be 0 (grid walk) or 1 (brute force)"); s->variant = (int)value; return RT_OK; }
  if (!std::strcmp(key, "tile_log2")) { if (value < -1 || value > 3) return fail(RT_ERR_INVALID, "tile_log2 must be -1..3"); s->tile_log2 = (int)value; return RT_OK; }
  if (!std::strcmp(key, "tile_shape")) { if (value < 0 || value > 3) return fail(RT_ERR_INVALID, "tile_shape must be 0 (square), 1 (scanline runs), 2 (4:1) or 3 (16:1)"); s->tile_shape = (int)value; return RT_OK; }
  if (!std::strcmp(key, "tile_affinity")) { if (value < 0 || value > 2) return fail(RT_ERR_INVALID, "tile_affinity must be 0 (off), 1 (large frames) or 2 (any frame of 8+ runs: tests)"); s->tile_affinity = (int)value; s->order_ready = false; s->depth_fresh = false; s->order_age = 0; return RT_OK; }
  if (!std::strcmp(key, "tile_order")) { if (value < 0 || value > 2) return fail(RT_ERR_INVALID, "tile_order must be 0, 1 or 2"); s->order_mode = (int)value; s->order_ready = false; s->depth_fresh = false; s->order_age = 0; return RT_OK; }
  if (!std::strcmp(key, "light_pool")) { if (value < 0 || value > 1024 || (value != 0 && value < 32)) return fail(RT_ERR_INVALID, "light_pool must be 0 (automatic) or 32..1024"); s->light_pool_cap = (int)value; return RT_OK; }
  if (!std::strcmp(key, "light_base_pool")) { if (value < 0 || value > 1024 || (value != 0 && value < 32)) return fail(RT_ERR_INVALID, "light_base_pool must be 0 (automatic) or 32..1024"); s->light_base_cap = (int)value; return RT_OK; }
  if (!std::strcmp(key, "light_nest_pool")) { if (value < 0 || value > 1) return fail(RT_ERR_INVALID, "light_nest_pool must be 0 or 1"); s->light_nest_pool = (int)value; return RT_OK; }
  if (!std::strcmp(key, "force_lit")) { if (value < 0 || value > 1) return fail(RT_ERR_INVALID, "force_lit must be 0 or 1"); s->force_lit = (int)value; return RT_OK; }  // (diagnostics: an unlit scene through the lit kernels — what their code costs the ordinary lanes, profiles/r04_run5_lit_sections.log)
  if (!std::strcmp(key, "tile_batch")) { if (value < 0 || value > 64) return fail(RT_ERR_INVALID, "tile_batch must be 0..64"); s->tile_batch = (int)value; return RT_OK; }
  if (!std::strcmp(key, "chunk_spp")) { if (value < 0) return fail(RT_ERR_INVALID, "chunk_spp must be >= 0"); s->chunk_spp = (int)value; return RT_OK; }
  if (!std::strcmp(key, "samples_per_pixel") || !std::strcmp(key, "max_depth")) {
    if (value < 0 || value > (int64_t)0xFFFFFFFFll) return fail(RT_ERR_INVALID, std::string(key) + " must be in 0 .. 2^32-1");
    if (key[0] == 's') s->host.samples_per_pixel = s->dev.spp = (uint32_t)value;
    else s->host.max_depth = s->dev.max_depth = (uint32_t)value;
    return RT_OK;
  }
  if (!std::strcmp(key, "seed")) { s->host.seed = (uint64_t)value; s->dev.seed_lo = (uint32_t)value; s->dev.seed_hi = (uint32_t)((uint64_t)value >> 32); return RT_OK; }
  return fail(RT_ERR_INVALID, std::string("unknown option ") + key);
}

namespace {

// Everything a launch needs that is NOT the launch: the kernel's dynamic-LDS attribute and its occupancy (worked out once per
// configuration: two runtime calls a frame otherwise) and, lit kernels, the lanes' HBM overflow slots.  Runs BEFORE the
// launch's start event is recorded — a first frame used to carry the 147 MB hipMalloc of a lit scene and the runtime's
// first look at the kernel inside its kernel_ms (one-shot CLI frames: 8.4 ms for a 0.9 ms kernel, profiles/r05_run5_cli_stats_before_warmup.log)
// — and once at scene creation for the scene's default configuration (warm_up).
template <bool HL, bool SIMPLE, bool LDS, bool WIDE>
int prepare_grid_t(RtHipScene* s, size_t lds_bytes, hipStream_t stream) {
  auto kern = rtk::rt_megakernel<HL, SIMPLE, LDS, WIDE>;
  const int key = (WIDE ? 8 : 0) | (HL ? 4 : 0) | (SIMPLE ? 2 : 0) | (LDS ? 1 : 0);
  if (s->cfg_key != key || s->cfg_lds != lds_bytes) {
    if (lds_bytes > 48 * 1024) RT_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    int per_cu_q = 0;
    RT_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_q, kern, rtk::BLOCK, lds_bytes));
    if (per_cu_q < 1) return fail(RT_ERR_HIP, "megakernel does not fit on a CU");
    s->cfg_key = key; s->cfg_lds = lds_bytes; s->cfg_per_cu = per_cu_q;
  }
  if (HL) {  // the lanes' overflow slots for suspended light activations (rt_core.h lane_light_begin): for the resident set, kept
    const size_t need_bytes = (size_t)s->cfg_per_cu * (size_t)s->num_cus * rtk::BLOCK * rtc::LIGHT_OVERFLOW_BYTES_PER_LANE;
    if (need_bytes > s->light_overflow_bytes) {
      // (an earlier launch of this scene may still be running on this stream with the old buffer: drain it first — once per
      //  scene and configuration, never in a frame loop)
      if (s->d_light_overflow) { RT_HIP_TRY(hipStreamSynchronize(stream)); (void)hipFree(s->d_light_overflow); s->d_light_overflow = nullptr; s->light_overflow_bytes = 0; }
      RT_HIP_TRY(hipMalloc(&s->d_light_overflow, need_bytes));
      s->light_overflow_bytes = need_bytes;
    }
  }
  return RT_OK;
}
template <bool HL, bool SIMPLE, bool LDS, bool WIDE>
int launch_grid_t(RtHipScene* s, const rtk::KArgs& ka_in, size_t lds_bytes, uint32_t n_items, hipStream_t stream) {
  auto kern = rtk::rt_megakernel<HL, SIMPLE, LDS, WIDE>;
  // persistent: exactly the resident set, never more workgroups than there are wave-sized items
  uint32_t wgs = (uint32_t)s->cfg_per_cu * (uint32_t)s->num_cus;
  const uint32_t need = (n_items + rtk::WAVES - 1) / rtk::WAVES;
  if (wgs > need) wgs = need;
  s->last_waves = (uint64_t)wgs * rtk::WAVES;
  rtk::KArgs ka = ka_in;
  if (HL) ka.sc.light_overflow = (unsigned char*)s->d_light_overflow;
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(rtk::BLOCK), lds_bytes, stream, ka);
  return RT_OK;
}
// (lights, every albedo in [0, 1], tables in LDS, wide cell tables) -> the instantiation's prepare / launch
int dispatch_grid(RtHipScene* s, bool has_lights, bool lds_tables, bool wide, bool prepare_only, const rtk::KArgs* ka, size_t lds_bytes, uint32_t n_items, hipStream_t stream) {
  int rc;
#define RT_GO(HL, SIMPLE, LDS, WIDE) rc = prepare_only ? prepare_grid_t<HL, SIMPLE, LDS, WIDE>(s, lds_bytes, stream) : launch_grid_t<HL, SIMPLE, LDS, WIDE>(s, *ka, lds_bytes, n_items, stream)
  const bool simple = s->simple_colour;
  if (wide) {  // (the launch's grid has 32-bit item lists — more than 65 535 spheres; tables in L2: plan_lds never puts them in LDS)
    if (lds_tables) return fail(RT_ERR_HIP, "wide cell tables cannot be staged in LDS");
    if (has_lights) { if (simple) RT_GO(true, true, false, true); else RT_GO(true, false, false, true); }
    else { if (simple) RT_GO(false, true, false, true); else RT_GO(false, false, false, true); }
  } else if (has_lights) {
    if (lds_tables) { if (simple) RT_GO(true, true, true, false); else RT_GO(true, false, true, false); }
    else { if (simple) RT_GO(true, true, false, false); else RT_GO(true, false, false, false); }
  } else if (lds_tables) { if (simple) RT_GO(false, true, true, false); else RT_GO(false, false, true, false); }
  else { if (simple) RT_GO(false, true, false, false); else RT_GO(false, false, false, false); }
#undef RT_GO
  return rc;
}

// LDS budget of a launch: do the tables fit, how big are the light pools, how much dynamic LDS does a workgroup ask for.
struct LdsPlan { bool lds_tables = false; uint32_t pool_slots = 0, base_slots = 0; size_t lds_bytes = 0; };
int plan_lds(const RtHipScene* s, const rtc::GridDesc& G, bool has_lights, LdsPlan* out) {
  // LDS budget.  Tables + tile slots + (lit scenes) the two pools of light records (rt_core.h): frames — held by a lane while
  // it sums over the lights — and, with the short colour map, bases — held from a sample's first light sampling to its end.
  // Expected demand: a camera path starts summing over the n lights with probability ~0.1 n at each of its first two hits
  // (raytracer.rs:92-102) and then shoots n light rays, so about f = 0.2 n^2 / (2.9 + 0.2 n^2) of the lanes hold a frame at
  // any moment (n = 1: 6.5 % = 66 of 1024 lanes; n = 2: 22 %; n = 3: 38 %) and about b = 0.2 n (n + 2) / (2.9 + 0.2 n^2) a
  // base (n = 1: 19 %).  A lane that finds a pool exhausted repeats its segment, so undersized pools are slow, never wrong
  // (forced pools of 64 / 32 frames on the lit cover scene: +1 % / +64 %, profiles/r03_run*_lit.log).  The pools get what is
  // left beside the tables, in the proportion of their demands, up to one record per lane; if that is less than 1.2 x the
  // demand the TABLES stay in L2 instead and the pools take their room.
  const bool short_map = has_lights && s->simple_colour;
  uint32_t pool_slots = 0, base_slots = 0;
  auto size_pools = [&](size_t avail, double* margin) {  // largest x with frames = x f 1024, bases = x b 1024 (multiples of 32, 32 .. 1024) inside `avail`
    const double n = (double)s->dev.n_lights, den = 2.9 + 0.2 * n * n;
    const double f = std::max(0.2 * n * n / den, 1.0 / 64.0), b = short_map ? std::min(1.0, std::max(0.2 * n * (n + 2.0) / den, 1.0 / 64.0)) : 0.0;
    auto slots = [&](double x, double share) -> uint32_t {
      if (share == 0.0) return 0u;
      const double v = x * share * (double)rtk::BLOCK;
      const uint32_t q = v >= (double)rtc::LIGHT_POOL_MAX_SLOTS ? rtc::LIGHT_POOL_MAX_SLOTS : ((uint32_t)v & ~31u);
      return q < 32u ? 32u : q;
    };
    auto bytes = [&](double x) { return (size_t)((rtk::park_bytes(slots(x, f), slots(x, b)) + 15u) & ~15u); };
    double lo = 0.0, hi = 1.0 / std::min(f, b > 0.0 ? b : f) + 1.0;  // at `hi` both pools hold one record per lane
    if (bytes(hi) <= avail) lo = hi;
    else for (int it = 0; it < 40; ++it) { const double mid = 0.5 * (lo + hi); if (bytes(mid) <= avail) lo = mid; else hi = mid; }
    pool_slots = slots(lo, f); base_slots = slots(lo, b);
    const bool forced_f = s->light_pool_cap > 0, forced_b = s->light_base_cap > 0;  // (tests: small pools on purpose)
    if (forced_f && pool_slots > (uint32_t)s->light_pool_cap) pool_slots = (uint32_t)s->light_pool_cap & ~31u;
    if (forced_b && base_slots > (uint32_t)s->light_base_cap) base_slots = (uint32_t)s->light_base_cap & ~31u;
    if (margin) *margin = (forced_f || forced_b) ? 1e9 : std::min((double)pool_slots / (f * rtk::BLOCK), b > 0.0 ? (double)base_slots / (b * rtk::BLOCK) : 1e9);
    return bytes(0.0) <= avail;  // (the smallest pools fit)
  };
  const rtk::LdsLayout no_pools = rtk::lds_layout(s->host.n_spheres, G.n_cells, G.n_items, true, false);
  bool lds_tables = !G.wide && no_pools.total <= rtk::LDS_TABLES_MAX_BYTES;
  if (has_lights) {
    const size_t fixed = rtc::LIGHT_CENTRES_LDS_MAX * 24u;
    double margin = 0.0;
    bool ok = lds_tables && no_pools.total + fixed < s->lds_cap && size_pools(s->lds_cap - no_pools.total - fixed, &margin) && margin >= 1.2;
    if (!ok) {
      lds_tables = false;
      const size_t bare = rtk::lds_layout(0, 0, 0, false, false).total + fixed;
      if (!size_pools(s->lds_cap - bare, nullptr)) return fail(RT_ERR_UNSUPPORTED, "no room for the light pools in LDS");
    }
  }
  out->lds_tables = lds_tables; out->pool_slots = pool_slots; out->base_slots = base_slots;
  out->lds_bytes = lds_tables ? rtk::lds_layout(s->host.n_spheres, G.n_cells, G.n_items, true, has_lights, pool_slots, base_slots).total
                              : rtk::lds_layout(0, 0, 0, false, has_lights, pool_slots, base_slots).total;
  return RT_OK;
}

int warm_up(RtHipScene* s) {
  // (keeping every CU busy for 0.5 - 20 ms here does NOT make a first frame faster — it is not the clocks:
  //  profiles/r05_run7_cli_warm_spin.log)
  hipLaunchKernelGGL(rtk::rt_warm_up, dim3(1), dim3(64), 0, nullptr);
  RT_HIP_TRY(hipGetLastError());
  LdsPlan plan;
  const int rc = plan_lds(s, s->dev.grid, s->has_lights, &plan);
  if (rc != RT_OK) return rc;
  return dispatch_grid(s, s->has_lights, plan.lds_tables, s->dev.grid.wide != 0u, true, nullptr, plan.lds_bytes, 0, nullptr);
}

}  // namespace

extern "C" int rt_hip_render(RtHipScene* s, const RtRowTiles* tiles, void* d_rgb8, void* d_linear, void* stream_) {
  if (!s) return fail(RT_ERR_INVALID, "null argument");
  const uint32_t local_rows = rt_tiles_local_rows(s->host.height, tiles);
  if (!d_rgb8 && local_rows != 0) return fail(RT_ERR_INVALID, "null framebuffer");
  hipStream_t stream = (hipStream_t)stream_;
  const bool has_lights = s->has_lights || s->force_lit != 0;
  // one tile-queue cursor / counter block per scene: launches of a scene are ordered on ONE stream
  // (a caller that drained the first stream itself — hipStreamSynchronize, an event — need not call rt_hip_wait first:
  //  the scene asks its OWN event, recorded behind the last launch's counter copy — never the caller's stream handle,
  //  which may have been destroyed since.  The abandoned launch's counters stay readable in its slot until two more
  //  launches have reused it; nobody waited for them.)
  if (s->in_flight && stream != s->last_stream && hipEventQuery(s->last_slot().ev_copied) != hipErrorNotReady) s->in_flight = false;
  (void)hipGetLastError();  // (the query's status is not an error of this call)
  if (s->in_flight && stream != s->last_stream)
    return fail(RT_ERR_INVALID, "rt_hip_render: this scene has a launch in flight on another stream (call rt_hip_wait first, "
                                "or use one RtHipScene per concurrent stream)");
  // the exact pixel sums are 2^-40 fixed point in 64 bits: spp * 2^40 must stay below 2^63
  if (s->host.samples_per_pixel > (1u << 22)) return fail(RT_ERR_UNSUPPORTED, "more than 2^22 samples per pixel");
  if (s->host.width > 524280u) return fail(RT_ERR_UNSUPPORTED, "frames wider than 524280 pixels");
  RT_HIP_TRY(hipSetDevice(s->device));
  RtHipScene::Slot& sl = s->slot[s->n_launches & 1];
  s->n_launches++;
  sl.rows = local_rows; sl.waves = 0; sl.launched = false;
  sl.samples = (uint64_t)local_rows * s->host.width * s->host.samples_per_pixel;
  s->last_stream = stream;
  sl.t_launch = std::chrono::steady_clock::now();
  RT_HIP_TRY(hipMemsetAsync(s->d_counters, 0, 32 * sizeof(unsigned long long), stream));
  // behind the kernel(s) of this launch: its counters into the slot's pinned words (what rt_hip_wait reads)
  auto finish_launch = [&](bool launched) -> int {
    sl.launched = launched; sl.waves = s->last_waves;
    if (launched) RT_HIP_TRY(hipMemcpyAsync(sl.h_counters, s->d_counters, RT_SLOT_COUNTERS * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    else std::memset(sl.h_counters, 0, RT_SLOT_COUNTERS * sizeof(unsigned long long));
    RT_HIP_TRY(hipEventRecord(sl.ev_copied, stream));
    s->in_flight = true;
    return RT_OK;
  };
  if (local_rows == 0) return finish_launch(false);

  rtk::KArgs ka;
  ka.sc = s->dev;
  if (s->variant == 1) {  // brute-force arm: no grid, every sphere in the `large` list (object order)
    std::memset(&ka.sc.grid, 0, sizeof ka.sc.grid);
    ka.sc.grid.n_large = s->host.n_spheres;
    ka.sc.large = (const uint32_t*)s->d_all;
    ka.sc.large_geom = (const rtc::SphereGeom*)s->d_geom;
  }
  ka.out_rgb8 = (uint8_t*)d_rgb8; ka.out_linear = (float*)d_linear; ka.counters = s->d_counters;
  ka.queue = (uint32_t*)(s->d_counters + 24);
  ka.local_rows = local_rows;
  const bool tiled = tiles && tiles->tile_rows && tiles->tile_stride;
  ka.tile_rows = tiled ? tiles->tile_rows : 0; ka.first_tile = tiled ? tiles->first_tile : 0;
  ka.tile_stride = tiled ? tiles->tile_stride : 0;
  // Pixel tile: the unit of the global queue, owned by ONE workgroup.  8x8 when the frame has
  // >= ~100 tiles per workgroup; small frames (and the 1/8 shards of the multi-GPU run) use 4x4,
  // 2x2 or 1x1 tiles, so that the heaviest tile (glass: 10x the mean) is a small part of a
  // workgroup's share and the long-path regions spread over many workgroups.
  const uint64_t want_tiles = (uint64_t)s->num_cus * 100u;
  // tile geometry: 4^tl pixels, as a square (default) or a run of one scanline
  // (shape 0: 2^t x 2^t; 1: 4^t x 1; 2 and 3: the square widened / flattened once or twice, 16x4 and 32x2 at t = 3)
  // (default shape: squares — but the 2x2 tiles of a small frame / a multi-GPU shard as 4x1 strips: 12 contiguous bytes
  //  leave in one packed store instead of two rows of byte stores; same time, WRITE_SIZE 2.71 -> 2.17 MB on an 1/8 shard
  //  of the headline frame, profiles/r03_run10_shape_traffic_sweep.log)
  // (shape 2 at t = 1 is the 2x2 square — the A/B arm of the 4x1 default, which no other value selects)
  auto widen = [&](uint32_t t) { const uint32_t k = s->tile_shape == 0 ? (t == 1u ? 1u : 0u) : (s->tile_shape == 1 ? t : (s->tile_shape == 2 && t == 1u ? 0u : (uint32_t)s->tile_shape - 1u)); return k < t ? k : t; };
  auto tiles_xy = [&](uint32_t t, uint32_t& tx, uint32_t& ty) {
    const uint32_t wl = t + widen(t), hl = t - widen(t);
    tx = (s->host.width + (1u << wl) - 1) >> wl; ty = (local_rows + (1u << hl) - 1) >> hl;
  };
  uint32_t tl = (uint32_t)s->tile_log2, tx = 0, ty = 0;
  if (s->tile_log2 < 0) {
    tl = 3;
    for (;;) { tiles_xy(tl, tx, ty); if (tl == 0 || (uint64_t)tx * ty >= want_tiles) break; tl--; }
  }
  // a slot header packs (tile column | tile row << 16), and the queue cursor is 32 bits
  auto too_many = [&](uint32_t t) { tiles_xy(t, tx, ty); return tx > 65535u || ty > 65535u || (uint64_t)tx * ty >= (1ull << 31); };
  while (tl < 3 && too_many(tl)) tl++;
  if (too_many(tl)) return fail(RT_ERR_UNSUPPORTED, "frame too large for the tile queue (more than 65535 tiles on an axis or 2^31 tiles)");
  ka.tile_log2 = tl;
  ka.tile_wl = tl + widen(tl); ka.tile_hl = tl - widen(tl);
  ka.t_slots = rtk::tile_slots(tl);
  ka.tiles_x = tx; s->last_tiles_x = tx;
  ka.n_tiles = tx * ty;
  // A tile's samples are handed out to the waves of its workgroup in chunks: an item's latency
  // is what the last wave of a frame waits for, but below ~128 samples per item the acquire /
  // finish overhead shows (measured, profiles/r01_run4_tiles.log): 8x8 -> 8 samples per pixel,
  // 4x4 -> 16, 2x2 -> 32, 1x1 -> 128 — give or take a factor of two so that a frame has about 60
  // items per wave (measured on the whole frame and its 1/2, 1/4, 1/8 shards, profiles/r01_run11_tiles.log:
  // both fewer, larger items and more, smaller ones lose up to 5 %).
  const uint32_t spp = s->host.samples_per_pixel;
  uint32_t chunk_spp = (uint32_t)s->chunk_spp;
  if (chunk_spp == 0) {
    static const uint32_t by_tile[4] = {128u, 32u, 16u, 8u};
    const uint64_t target_items = (uint64_t)s->num_cus * rtk::WAVES * 60u;
    const uint64_t ideal = ((uint64_t)ka.n_tiles * spp + target_items / 2) / target_items;  // samples per pixel and item
    uint32_t c = 1;
    while ((uint64_t)c * 3u < ideal * 2u) c <<= 1;  // nearest power of two (geometric)
    // (1x1 and 4-pixel tiles never go below their base of 128 samples per item: the 1/8 shard of the headline frame runs 1.678 ms
    //  with chunks of 32 against 1.737 with the 16 the item count alone asks for, profiles/r04_run13_tile_chunk_sweep.log;
    //  16-pixel tiles may: the 800x600 test scene at spp 16 is 3 % faster in two chunks of 8 than in one of 16)
    const uint32_t lo = tl <= 1u ? by_tile[tl] : by_tile[tl] / 2u, hi = by_tile[tl] * 2u;
    chunk_spp = c < lo ? lo : (c > hi ? hi : c);
  }
  if (chunk_spp > spp || spp == 0) chunk_spp = spp ? spp : 1;
  ka.chunk_spp = chunk_spp;
  ka.n_chunks = spp ? (spp + chunk_spp - 1) / chunk_spp : 1;
  const uint32_t n_items = ka.n_tiles * ka.n_chunks;
  // Tiles a workgroup takes from the frame's queue per atomic (rt_kernel.hip: wg_stash), at most, and the taper towards
  // single tiles at the end of the frame: remaining tiles / (workgroups x 16).  Measured (profiles/r03_run26_batch_sweep.log):
  // what counts is that ONE wave of a workgroup asks the queue at a time (cover frame at spp 8 / 32 / 128: 1.49 -> 1.15,
  // 3.71 -> 3.35, 12.92 -> 12.80 ms with batches of one); batches of 4 add 5 % on frames of small tiles (test scene 1.09 ->
  // 1.03 ms).  Not more: a batch is a run of the queue's order, and that order puts the deepest tiles first — 8 or 16 of
  // them in one workgroup are the critical path of a short frame (1/8 shard: 1.76 -> 1.87 -> 2.45 ms).
  ka.tile_batch = s->tile_batch > 0 ? (uint32_t)s->tile_batch : 4u;
  ka.batch_share = 16u;
#ifdef RT_DEV_KNOBS
  if (const char* e = std::getenv("RT_BATCH_SHARE")) { const int v = std::atoi(e); if (v >= 1 && v <= 1024) ka.batch_share = (uint32_t)v; }
#endif
  LdsPlan plan;
  { const int rc_plan = plan_lds(s, ka.sc.grid, has_lights, &plan); if (rc_plan != RT_OK) return rc_plan; }
  const bool lds_tables = plan.lds_tables;
  const size_t lds_bytes = plan.lds_bytes;
  ka.sc.light_pool_slots = plan.pool_slots;
  ka.sc.light_base_slots = plan.base_slots;
  ka.sc.light_nest_pool = (uint32_t)s->light_nest_pool;
  ka.sc.light_overflow = nullptr;  // (launch_grid_t fills it in for the lit kernels)
  s->last_pool_slots = plan.pool_slots; s->last_base_slots = plan.base_slots; s->last_lds_tables = lds_tables; s->last_lds_bytes = lds_bytes;

  // queue order: bottom of the image first; from the second frame of a tile geometry on, the tiles whose samples ran
  // deepest in the previous frame first (their paths are what a frame ends on)
  ka.order_mode = s->order_mode != 0 ? 1u : 0u;
  ka.tile_order = nullptr; ka.tile_depth = nullptr;
  // XCD affinity: runs of 1.5 KB of a scanline's tiles (512 pixels) per XCD, for large frames (smaller ones — the shards
  // of the headline frame — lose more to the coarser balance than the write traffic is worth: 2.10 instead of 1.89 ms,
  // profiles/r02_run29_affinity.log)
  ka.aff_group_log2 = 0xFFFFFFFFu;
  for (int x = 0; x < 8; ++x) { ka.xcd_cnt[x] = 0; ka.xcd_off[x] = 0; }
  {
#ifdef RT_DEV_KNOBS  // (A/B builds only: tools/affinity_sweep.sh)
    static const uint32_t run_px_log2 = [] { const char* e = std::getenv("RT_AFF_RUN_LOG2"); const int v = e ? std::atoi(e) : 9; return (uint32_t)(v < 3 ? 3 : (v > 14 ? 14 : v)); }();
#else
    constexpr uint32_t run_px_log2 = 9;  // runs of 512 pixels of a tile row (profiles/r02_run29_affinity.log)
#endif
    const uint32_t gl = ka.tile_wl >= run_px_log2 ? 0u : run_px_log2 - ka.tile_wl;
    const uint32_t n_groups = (ka.n_tiles + (1u << gl) - 1u) >> gl;
    // on for frames of at least 2^19 pixels (the headline frame: 0.96 M; its 1/2 ... 1/8 shards and the 800x600 test scene are
    // below and lose 6 - 15 % to the coarser balance, profiles/r03_run8_shard_affinity_sweep.log) with at least 8 runs
    const bool big = (uint64_t)local_rows * s->host.width >= (1ull << 19);
    if ((s->tile_affinity == 1 && big && n_groups >= 8u) || (s->tile_affinity == 2 && n_groups >= 8u)) {
      ka.aff_group_log2 = gl;
      for (uint32_t g = 0; g < 8u && g < n_groups; ++g) {  // groups g, g + 8, ...: all full but possibly the frame's last
        const uint32_t mine = (n_groups - 1u - g) / 8u + 1u;
        const uint32_t last = g + 8u * (mine - 1u);
        const uint32_t last_size = last == n_groups - 1u ? ka.n_tiles - (last << gl) : (1u << gl);
        ka.xcd_cnt[g] = ((mine - 1u) << gl) + last_size;
      }
      for (int x = 1; x < 8; ++x) ka.xcd_off[x] = ka.xcd_off[x - 1] + ka.xcd_cnt[x - 1];
    }
  }
  if (s->order_mode >= 2) {
    RtHipScene::OrderKey key;
    key.n_tiles = ka.n_tiles; key.tile_log2 = tl; key.tile_shape = (uint32_t)s->tile_shape; key.aff_group_log2 = ka.aff_group_log2;
    key.tile_rows = ka.tile_rows; key.first_tile = ka.first_tile; key.tile_stride = ka.tile_stride; key.local_rows = local_rows;
    if (ka.n_tiles > s->order_cap) {
      if (s->d_tile_depth) (void)hipFree(s->d_tile_depth);
      if (s->d_tile_order) (void)hipFree(s->d_tile_order);
      s->d_tile_depth = s->d_tile_order = nullptr; s->order_cap = 0;
      RT_HIP_TRY(hipMalloc((void**)&s->d_tile_depth, (size_t)ka.n_tiles * 4));
      RT_HIP_TRY(hipMalloc((void**)&s->d_tile_order, (size_t)ka.n_tiles * 4));
      s->order_cap = ka.n_tiles;
    }
    if (!(key == s->order_key)) { s->order_key = key; s->order_ready = false; s->depth_fresh = false; s->order_age = 0; }
    // The order of THIS frame from the depths the previous frame of the SAME view measured (sorted here, stream-ordered ahead of
    // the launch — until round 5 behind the frame that measured them, which an animation paid every frame for an order it never
    // used).  Which tiles breed deep paths is a property of scene and camera: rebuilt after each of a view's first two frames,
    // then kept; rt_hip_set_camera with a different camera starts over WITHOUT an order (below).
    if (s->depth_fresh) {
      hipLaunchKernelGGL(rtk::rt_order_tiles, dim3(1), dim3(1024), 0, stream, (const uint32_t*)s->d_tile_depth, s->d_tile_order, ka.n_tiles, ka.aff_group_log2);
      RT_HIP_TRY(hipGetLastError());
      s->order_ready = true; s->depth_fresh = false;
    }
    if (s->order_mode == 2 && s->order_age < 2) ka.tile_depth = s->d_tile_depth;  // (measured only while the order is still being built)
    if (s->order_ready) ka.tile_order = s->d_tile_order;
  }

  int rc;
  auto launch = [&](const rtk::KArgs& ka, uint32_t n_items) -> int {
    const int rc = dispatch_grid(s, has_lights, lds_tables, ka.sc.grid.wide != 0u, false, &ka, lds_bytes, n_items, stream);
    if (rc != RT_OK) return rc;
    RT_HIP_TRY(hipGetLastError());
    return RT_OK;
  };
  if ((rc = dispatch_grid(s, has_lights, lds_tables, ka.sc.grid.wide != 0u, true, nullptr, lds_bytes, 0, stream)) != RT_OK) return rc;  // (host-side set-up: before the start event)
  RT_HIP_TRY(hipEventRecord(sl.ev_start, stream));
  // A frame without a measured order (the first of a scene, a one-shot render) leaves the queue bottom row first and ends on
  // whatever deep path started last: 13.3 instead of 12.8 ms on the headline frame.  Two ways to SEED an order — a depth guess
  // from the spheres' projections, a one-sample probe launch — were built and measured in round 4, both slower than no seed
  // (profiles/r04_run3_first_frame_orders.log; the code: profiles/r05_order_seed_removed.patch): what the measured order
  // knows is WHICH tiles hold one of the rare 50-segment paths, which a replay of the same seeds predicts and nothing cheaper does.
  if ((rc = launch(ka, n_items)) != RT_OK) return rc;
  RT_HIP_TRY(hipEventRecord(sl.ev_stop, stream));
  if (ka.tile_depth) { s->order_age++; s->depth_fresh = true; }  // (the NEXT frame of this view sorts them into its order)
  return finish_launch(true);
}

#ifdef RT_TEST_PROBES
// Diagnostics: the per-tile path depths the last measuring frame left behind (tile_order 2, first two frames of a view):
// out[tile] = deepest camera path seen in the tile, tiles in row-major order of the launch's tile grid (*tiles_x wide).
// Returns the number of tiles copied (at most cap), or a negative RtStatus.
extern "C" int rt_hip_debug_tile_depth(RtHipScene* s, uint32_t* out, uint32_t cap, uint32_t* tiles_x) {
  if (!s || !out) return fail(RT_ERR_INVALID, "null argument");
  RT_HIP_TRY(hipSetDevice(s->device));
  if (s->n_launches) RT_HIP_TRY(hipStreamSynchronize(s->last_stream));
  const uint32_t n = s->order_key.n_tiles < cap ? s->order_key.n_tiles : cap;
  if (n && s->d_tile_depth) RT_HIP_TRY(hipMemcpy(out, s->d_tile_depth, (size_t)n * 4, hipMemcpyDeviceToHost));
  if (tiles_x) *tiles_x = s->last_tiles_x;
  return (int)n;
}

extern "C" int rt_hip_debug_timeline(RtHipScene* s, uint64_t* out, uint32_t max_waves) {
  if (!s || !out) return fail(RT_ERR_INVALID, "null argument");
  RT_HIP_TRY(hipSetDevice(s->device));
  RT_HIP_TRY(hipStreamSynchronize(s->last_stream));
  const uint32_t n = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(s->last_waves, max_waves), RT_TIMELINE_WAVES);
  RT_HIP_TRY(hipMemcpy(out, s->d_counters, (32 + (size_t)n * 4) * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return (int)n;
}

#endif  // RT_TEST_PROBES

namespace {
// the report of the launch that used `sl` (its counter copy must have completed: the caller synchronised)
void fill_stats(const RtHipScene* s, const RtHipScene::Slot& sl, RtStats* stats) {
  std::memset(stats, 0, sizeof *stats);
  stats->n_gpus_used = 1;
  const unsigned long long* c = sl.h_counters;  // segments, exact tests, tex_oob, grid steps, 4 x wave trip counts, 8 x section cycles, profile clocks
  float ms = 0.f;
  if (sl.launched && hipEventElapsedTime(&ms, sl.ev_start, sl.ev_stop) != hipSuccess) { ms = 0.f; (void)hipGetLastError(); }
  stats->samples = sl.samples;
  stats->segments = c[0];
  stats->sphere_tests = c[0] * (uint64_t)s->host.n_spheres;
  stats->exact_tests = c[1];
  stats->tex_oob = c[2];
  stats->grid_steps = c[3];
  stats->segments_repeated = c[28] > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)c[28];
  for (int k = 0; k < 4; ++k) stats->wave_iters[k] = c[4 + k];
  for (int k = 0; k < 8; ++k) stats->prof_cycles[k] = c[8 + k];
  if (c[14] && sl.waves) {  // profile builds: longest / shortest wave, waves launched
    stats->prof_cycles[7] = c[15];
    stats->prof_cycles[8] = ~c[17];
    stats->prof_cycles[10] = sl.waves;
  }
  stats->kernel_ms = ms;
  stats->frame_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - sl.t_launch).count();
}
// wait for ONE launch — the one that used slot `which` — and report it (the group's pipelined frames: frame i is collected
// while frame i+1 is in flight on the same stream)
int wait_slot(RtHipScene* s, int which, RtStats* stats) {
  RT_HIP_TRY(hipSetDevice(s->device));
  RtHipScene::Slot& sl = s->slot[which & 1];
  RT_HIP_TRY(hipEventSynchronize(sl.ev_copied));
  if (&sl == &s->last_slot()) s->in_flight = false;
  if (stats) fill_stats(s, sl, stats);
  return RT_OK;
}
}  // namespace

extern "C" int rt_hip_wait(RtHipScene* s, RtStats* stats) {
  if (!s) return fail(RT_ERR_INVALID, "null argument");
  RT_HIP_TRY(hipSetDevice(s->device));
  if (s->n_launches == 0) { if (stats) { std::memset(stats, 0, sizeof *stats); stats->n_gpus_used = 1; } return RT_OK; }
  RT_HIP_TRY(hipStreamSynchronize(s->last_stream));
  s->in_flight = false;
  if (stats) fill_stats(s, s->last_slot(), stats);
  return RT_OK;
}

// what a resident scene was built into (diagnostics; bench.py prices the tables' bytes with it)
extern "C" int64_t rt_hip_scene_query(const RtHipScene* s, const char* key) {
  if (!s || !key) return -1;
  if (!std::strcmp(key, "n_spheres")) return (int64_t)s->host.n_spheres;
  if (!std::strcmp(key, "n_lights")) return (int64_t)s->dev.n_lights;
  if (!std::strcmp(key, "grid_cells")) return (int64_t)s->grid.n_cells;       // padded cell table (8 B each; wide tables: 16 B)
  if (!std::strcmp(key, "grid_items")) return (int64_t)s->grid.n_items;       // u16 each (wide tables: u32)
  if (!std::strcmp(key, "grid_wide")) return (int64_t)s->grid.wide;           // 1: 32-bit item lists (more than 65 535 spheres)
  if (!std::strcmp(key, "grid_large")) return (int64_t)s->grid.n_large;
  if (!std::strcmp(key, "texel_bytes")) return (int64_t)s->texel_bytes;       // 4-byte texels of textures + sky resident in HBM
  if (!std::strcmp(key, "light_pool_slots")) return (int64_t)s->last_pool_slots;  // of the last launch: records in the pools of light frames /
  if (!std::strcmp(key, "light_base_slots")) return (int64_t)s->last_base_slots;  // colour-map bases, the kernel's dynamic LDS, tables staged in LDS
  if (!std::strcmp(key, "lds_bytes")) return (int64_t)s->last_lds_bytes;
  if (!std::strcmp(key, "lds_tables")) return (int64_t)s->last_lds_tables;
  if (!std::strcmp(key, "table_bytes")) return (int64_t)((size_t)s->host.n_spheres * (sizeof(rtc::SphereGeom) + sizeof(rtc::MatCore)) + (size_t)s->grid.n_cells * (s->grid.wide ? 16u : 8u) + (size_t)s->grid.n_items * (s->grid.wide ? 4u : 2u));
  return -1;
}

extern "C" int rt_hip_set_camera(RtHipScene* s, const double origin[3], const double lower_left[3], const double horizontal[3],
                                 const double vertical[3]) {
  if (!s || !origin || !lower_left || !horizontal || !vertical) return fail(RT_ERR_INVALID, "null argument");
  bool same = true;
  for (int i = 0; i < 3; ++i)
    same = same && s->host.cam_origin[i] == origin[i] && s->host.cam_lower_left[i] == lower_left[i] && s->host.cam_horizontal[i] == horizontal[i] &&
           s->host.cam_vertical[i] == vertical[i];
  if (same) return RT_OK;  // (the same view: its order stays)
  for (int i = 0; i < 3; ++i) {
    s->host.cam_origin[i] = s->dev.cam_origin[i] = origin[i];
    s->host.cam_lower_left[i] = s->dev.cam_ll[i] = lower_left[i];
    s->host.cam_horizontal[i] = s->dev.cam_h[i] = horizontal[i];
    s->host.cam_vertical[i] = s->dev.cam_v[i] = vertical[i];
  }
  // A new view starts WITHOUT an order (bottom row first) and measures its own.  Until round 5 the previous view's order stayed
  // in use ("the views of an animation are close") — measured in round 6 on the headline scene turning 3 degrees per frame:
  // the stale order costs 3 - 10 % against the view's own order and is WORSE than no order (+3 - 4 %): the tiles that hold a
  // view's rare 50-segment paths are 4x4 pixels, and a 3 degree turn moves the spheres by tens of pixels
  // (bench.py `animation.same_views`, profiles/r06_run*_bench.json).  RT_STALE_ORDER=1: the round-5 behaviour, for the A/B.
  static const bool keep_stale = std::getenv("RT_STALE_ORDER") != nullptr;
  if (!keep_stale) { s->order_ready = false; s->depth_fresh = false; }   // (stale arm: the previous view's depths are sorted into this frame's order)
  s->order_age = 0;
  return RT_OK;
}

extern "C" int rt_hip_render_to_host(RtHipScene* s, uint8_t* out_rgb8, RtStats* stats) {
  if (!s || !out_rgb8) return fail(RT_ERR_INVALID, "null argument");
  auto t0 = std::chrono::steady_clock::now();
  RT_HIP_TRY(hipSetDevice(s->device));
  const size_t bytes = (size_t)s->host.width * s->host.height * 3;
  if (bytes > s->frame_bytes) {
    if (s->d_frame) { (void)hipFree(s->d_frame); s->d_frame = nullptr; s->frame_bytes = 0; }
    RT_HIP_TRY(hipMalloc(&s->d_frame, bytes));
    s->frame_bytes = bytes;
  }
  int rc = rt_hip_render(s, nullptr, s->d_frame, nullptr, nullptr);
  RtStats st;
  if (rc == RT_OK) rc = rt_hip_wait(s, &st);
  if (rc != RT_OK) return rc;
  RT_HIP_TRY(hipMemcpy(out_rgb8, s->d_frame, bytes, hipMemcpyDeviceToHost));
  if (stats) {
    *stats = st;
    stats->frame_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  return RT_OK;
}

// The scene's own kernel through the stream its frames will use, once, on ONE scanline — whatever the runtime does with the first
// launch of a kernel on a queue (its scratch, its kernarg pool, the instruction cache) happens here, at set-up, instead of inside
// the first frame: the group calls it per rank.  The launch leaves no trace in the scene (slots, queue order, counters of the
// "last launch" are as after creation).
int rt_hip_scene_warm(RtHipScene* s, hipStream_t stream) {
  if (!s || s->host.width == 0 || s->host.height == 0) return RT_OK;
  RT_HIP_TRY(hipSetDevice(s->device));
  void* row = nullptr;
  RT_HIP_TRY(hipMalloc(&row, (size_t)s->host.width * 3 + 16));
  const int saved_order = s->order_mode;
  s->order_mode = 1;  // (nothing measured, nothing sorted)
  const RtRowTiles first_row{1u, 0u, s->host.height};
  int rc = rt_hip_render(s, &first_row, row, nullptr, stream);
  RtStats wst;
  if (rc == RT_OK) rc = rt_hip_wait(s, &wst);
  if (rc == RT_OK) rtp::add("rank0.warm_up_kernel_ms_by_events", wst.kernel_ms);
  s->order_mode = saved_order;
  s->n_launches = 0; s->in_flight = false; s->last_stream = nullptr; s->last_waves = 0;
  s->order_key = RtHipScene::OrderKey(); s->order_ready = false; s->depth_fresh = false; s->order_age = 0;
  for (auto& sl : s->slot) { sl.rows = 0; sl.samples = 0; sl.waves = 0; sl.launched = false; }
  (void)hipFree(row);
  return rc;
}

#include "rt_hip_group.hip"  // rt_hip_group_* and rt_render_rgb8: the frame over 1..G devices

#ifdef RT_TEST_PROBES
// math self-test hook (see rtk::rt_math_probe); all pointers are DEVICE pointers
extern "C" int rt_hip_hit_probe(const double* rays, const double* spheres, double* out_t, uint32_t n, void* stream) {
  hipLaunchKernelGGL(rtk::rt_hit_probe, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, rays, spheres, out_t, n);
  RT_HIP_TRY(hipGetLastError());
  return RT_OK;
}

extern "C" int rt_hip_math_probe(const double* x, const double* y, double* out_sqrt, double* out_div, float* out_sqrtf,
                                 double* out_atan2, uint32_t n, void* stream) {
  hipLaunchKernelGGL(rtk::rt_math_probe, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, y, out_sqrt, out_div,
                     out_sqrtf, out_atan2, n);
  RT_HIP_TRY(hipGetLastError());
  return RT_OK;
}

extern "C" int rt_hip_atan2_probe(const double* d_y, const double* d_x, double* d_out, uint32_t n, void* stream) {
  if (!d_y || !d_x || !d_out) return fail(RT_ERR_INVALID, "null argument");
  hipLaunchKernelGGL(rtk::rt_atan2_probe, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_y, d_x, d_out, n);
  RT_HIP_TRY(hipGetLastError());
  return RT_OK;
}

// device probe of the Texture hit's fast texel path beside the exact one (see rtk::rt_texel_probe); d_* are DEVICE pointers
extern "C" int rt_hip_texel_probe(const double* d_points, const double centre_radius[4], double h_offset, uint64_t tex_w, uint64_t tex_h,
                                  uint64_t* d_out, double* d_uv, uint32_t n, void* stream) {
  if (!d_points || !centre_radius || !d_out) return fail(RT_ERR_INVALID, "null argument");
  hipLaunchKernelGGL(rtk::rt_texel_probe, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_points, centre_radius[0], centre_radius[1],
                     centre_radius[2], centre_radius[3], h_offset, (unsigned long long)tex_w, (unsigned long long)tex_h,
                     (unsigned long long*)d_out, d_uv, n);
  RT_HIP_TRY(hipGetLastError());
  return RT_OK;
}

extern "C" int rt_hip_quot_probe(const double* d_x, const double* d_y, double* d_quot, double* d_rsqrt, double* d_div, uint32_t n, void* stream) {
  if (!d_x || !d_y || !d_quot || !d_rsqrt) return fail(RT_ERR_INVALID, "null argument");
  hipLaunchKernelGGL(rtk::rt_quot_probe, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_x, d_y, d_quot, d_rsqrt, d_div, n);
  RT_HIP_TRY(hipGetLastError());
  return RT_OK;
}
#endif  // RT_TEST_PROBES
