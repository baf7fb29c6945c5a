// rt_hip_group.hip — one frame over the GPUs of a node, inside the C ABI (included by rt_hip_api.hip).
//
// The reference parallelises over scanlines inside one process (rayon, raytracer.rs:255-262).  A group
// does the same across devices, still inside ONE host process: the scene is replicated (<= 22 MB),
// rank r renders scanline tiles r, r+G, ... (RT_GROUP_TILE_ROWS rows each — finely interleaved because
// sky rows cost 1 segment per sample and the rows through the glass ball 10+) on its own host thread and
// stream, the packed tiles meet on rank 0's device through ONE gather over xGMI at frame end
// (`ncclGather` of RCCL, loaded with dlopen so that a single-GPU process never touches RCCL; or G-1
// peer copies with RT_GATHER=peer), a small kernel puts the scanlines in order, and the frame leaves in
// ONE device-to-host copy.  No intra-frame communication.  Philox is addressed by GLOBAL pixel index and
// pixel sums are order-free, so the frame is bit-identical for every G.
//
// Test hooks: RT_GPUS_EMULATE=1 lets ranks share devices (rank r -> device r mod visible devices; peer
// transport only), so the whole path — threads, sharding, gather buffer layout, de-interleave — runs on a
// one-GPU box; RT_GATHER_SELFTEST=1 makes a ONE-rank group go through the gather (RCCL communicator of one
// rank, in-place ncclGather) and the de-interleave kernel too.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is dlopen'ed below, never linked

#include <condition_variable>
#include <mutex>
#include <thread>

#define RT_GROUP_TILE_ROWS 2u  // measured: 2-row interleave balances 8 ranks to +-2 % (8 rows: +-6 %), profiles/r01_run4_shards.log

namespace rtg {

struct RcclApi {
  void* h = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGather) Gather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool load(std::string& err) {
    if (h) return true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (h) break;
    }
    if (!h) { err = std::string("cannot load RCCL: ") + dlerror(); return false; }
#define RT_SYM(field, sym)                                                      \
  field = reinterpret_cast<decltype(field)>(dlsym(h, sym));                     \
  if (!field) { err = std::string("RCCL lacks ") + sym; return false; }
    RT_SYM(CommInitAll, "ncclCommInitAll")
    RT_SYM(CommDestroy, "ncclCommDestroy")
    RT_SYM(GroupStart, "ncclGroupStart")
    RT_SYM(GroupEnd, "ncclGroupEnd")
    RT_SYM(Gather, "ncclGather")
    RT_SYM(GetErrorString, "ncclGetErrorString")
#undef RT_SYM
    return true;
  }
};

// frame[y] <- stacked[rank(y)][local row(y)]: rt_tiles_stacked_row() of include/rt_abi.h, compiled for the device
__host__ __device__ inline uint32_t stacked_row_of(uint32_t y, uint32_t G, uint32_t tile_rows, uint32_t pad_rows) {
  const uint32_t k = y / tile_rows, r = k % G, j = k / G;
  return r * pad_rows + j * tile_rows + y % tile_rows;
}
__global__ void deinterleave_rows(const uint8_t* __restrict__ stacked, uint8_t* __restrict__ frame, uint32_t height, uint32_t row_bytes,
                                  uint32_t G, uint32_t tile_rows, uint32_t pad_rows) {
  const uint32_t y = blockIdx.x;
  if (y >= height) return;
  const uint8_t* src = stacked + (size_t)stacked_row_of(y, G, tile_rows, pad_rows) * row_bytes;
  uint8_t* dst = frame + (size_t)y * row_bytes;
  if ((row_bytes & 15u) == 0u) {  // every row starts 16-byte aligned (buffers are hipMalloc'ed)
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    uint4* d4 = reinterpret_cast<uint4*>(dst);
    for (uint32_t i = threadIdx.x; i < row_bytes / 16u; i += blockDim.x) d4[i] = s4[i];
  } else {
    for (uint32_t i = threadIdx.x; i < row_bytes; i += blockDim.x) dst[i] = src[i];
  }
}

}  // namespace rtg

struct RtHipGroup {
  uint32_t G = 1, width = 0, height = 0, pad_rows = 0;
  size_t row_bytes = 0, pad_bytes = 0;
  bool rccl = false;
  bool shared_device = false;        // RT_GPUS_EMULATE: some ranks share a device
  bool gather = false;               // G > 1 (or the one-rank self-test): gather + de-interleave after the kernels
  std::vector<int> device;
  std::vector<RtHipScene*> scene;
  std::vector<hipStream_t> stream;
  std::vector<hipEvent_t> ev_done;   // rank r: its tiles are in `stacked` (peer transport) / its kernel was enqueued (rccl)
  std::vector<void*> d_tiles;        // rank r's packed tiles on ITS device (rank 0: a slice of `stacked`)
  std::vector<RtRowTiles> tiles;
  void* d_stacked = nullptr;         // device 0: G x pad_rows rows
  void* d_frame = nullptr;           // device 0: height rows (G == 1: the same buffer)
  hipEvent_t ev_assembled = nullptr;
  rtg::RcclApi api;
  std::vector<ncclComm_t> comm;
  // one persistent host thread per rank: the launches of a frame go out in parallel (a serial launch loop
  // would start rank 7 ~8 x 40 us late on a 2 ms shard)
  std::vector<std::thread> worker;
  std::mutex mu;
  std::condition_variable cv_go, cv_done;
  uint64_t generation = 0;
  uint32_t n_done = 0;
  bool quit = false;
  std::vector<int> rc;
  std::vector<std::string> err;
};

namespace rtg {

int resolve_gpus(const RtScene* scene, uint32_t n_gpus, uint32_t* out) {
  uint32_t g = n_gpus ? n_gpus : (scene ? scene->n_gpus : 0u);
  if (g == 0) {
    if (const char* e = std::getenv("RT_GPUS")) {
      const long v = std::strtol(e, nullptr, 10);
      if (v < 1 || v > 1024) return fail(RT_ERR_INVALID, "RT_GPUS must be a positive device count");
      g = (uint32_t)v;
    } else g = 1;
  }
  *out = g;
  return RT_OK;
}

void worker_main(RtHipGroup* g, uint32_t r) {
  uint64_t seen = 0;
  for (;;) {
    {
      std::unique_lock<std::mutex> lk(g->mu);
      g->cv_go.wait(lk, [&] { return g->quit || g->generation != seen; });
      if (g->quit) return;
      seen = g->generation;
    }
    int rc = rt_hip_render(g->scene[r], g->G > 1 ? &g->tiles[r] : nullptr, g->d_tiles[r], nullptr, g->stream[r]);
    std::string err;
    if (rc != RT_OK) err = rt_hip_last_error();
    if (rc == RT_OK && g->gather && !g->rccl && r != 0) {  // peer transport: this rank's slice of the gather
      const hipError_t e = hipMemcpyPeerAsync(static_cast<uint8_t*>(g->d_stacked) + (size_t)r * g->pad_bytes, g->device[0], g->d_tiles[r],
                                              g->device[r], g->pad_bytes, g->stream[r]);
      if (e != hipSuccess) { rc = RT_ERR_HIP; err = std::string("hipMemcpyPeerAsync: ") + hipGetErrorString(e); }
    }
    if (rc == RT_OK) {
      const hipError_t e = hipEventRecord(g->ev_done[r], g->stream[r]);
      if (e != hipSuccess) { rc = RT_ERR_HIP; err = std::string("hipEventRecord: ") + hipGetErrorString(e); }
    }
    {
      std::lock_guard<std::mutex> lk(g->mu);
      g->rc[r] = rc; g->err[r] = err;
      g->n_done++;
    }
    g->cv_done.notify_one();
  }
}

}  // namespace rtg

extern "C" uint32_t rt_hip_group_size(const RtHipGroup* g) { return g ? g->G : 0u; }

extern "C" uint32_t rt_hip_group_stacked_row(uint32_t y, uint32_t n_ranks, uint32_t pad_rows, uint32_t* tile_rows_out) {
  if (tile_rows_out) *tile_rows_out = RT_GROUP_TILE_ROWS;
  return n_ranks ? rtg::stacked_row_of(y, n_ranks, RT_GROUP_TILE_ROWS, pad_rows) : 0u;
}

extern "C" void rt_hip_group_destroy(RtHipGroup* g) {
  if (!g) return;
  {
    std::lock_guard<std::mutex> lk(g->mu);
    g->quit = true;
  }
  g->cv_go.notify_all();
  for (auto& t : g->worker) if (t.joinable()) t.join();
  for (uint32_t r = 0; r < g->comm.size(); ++r)
    if (g->comm[r]) { (void)hipSetDevice(g->device[r]); (void)g->api.CommDestroy(g->comm[r]); }
  for (uint32_t r = 0; r < g->scene.size(); ++r) {
    (void)hipSetDevice(g->device[r]);
    if (r < g->stream.size() && g->stream[r]) (void)hipStreamSynchronize(g->stream[r]);
    if (r != 0 && r < g->d_tiles.size() && g->d_tiles[r]) (void)hipFree(g->d_tiles[r]);
    if (r < g->ev_done.size() && g->ev_done[r]) (void)hipEventDestroy(g->ev_done[r]);
    if (g->scene[r]) rt_hip_scene_destroy(g->scene[r]);
    if (r < g->stream.size() && g->stream[r]) (void)hipStreamDestroy(g->stream[r]);
  }
  if (!g->device.empty()) (void)hipSetDevice(g->device[0]);
  if (g->d_frame && g->d_frame != g->d_stacked) (void)hipFree(g->d_frame);
  if (g->d_stacked) (void)hipFree(g->d_stacked);
  if (g->ev_assembled) (void)hipEventDestroy(g->ev_assembled);
  delete g;
}

extern "C" int rt_hip_group_create(const RtScene* scene, uint32_t n_gpus, RtHipGroup** out) {
  if (!scene || !out) return fail(RT_ERR_INVALID, "null argument");
  *out = nullptr;
  const int ndev = rt_hip_device_count();
  if (ndev <= 0) return fail(RT_ERR_NO_DEVICE, rt_strerror(RT_ERR_NO_DEVICE));
  uint32_t G = 1;
  int rc = rtg::resolve_gpus(scene, n_gpus, &G);
  if (rc != RT_OK) return rc;
  const char* emu = std::getenv("RT_GPUS_EMULATE");
  const bool emulate = emu && emu[0] == '1';
  if (G > (uint32_t)ndev && !emulate)
    return fail(RT_ERR_INVALID, "n_gpus = " + std::to_string(G) + " but only " + std::to_string(ndev) + " device(s) visible");
  RtHipGroup* g = new RtHipGroup;
  g->G = G; g->width = scene->width; g->height = scene->height;
  g->row_bytes = (size_t)scene->width * 3;
  g->device.resize(G); g->scene.assign(G, nullptr); g->stream.assign(G, nullptr); g->ev_done.assign(G, nullptr);
  g->d_tiles.assign(G, nullptr); g->tiles.resize(G); g->rc.assign(G, RT_OK); g->err.resize(G);
  bool& shared_device = g->shared_device;
  for (uint32_t r = 0; r < G; ++r) {
    g->device[r] = (int)(r % (uint32_t)ndev);
    if (r >= (uint32_t)ndev) shared_device = true;
    g->tiles[r] = RtRowTiles{RT_GROUP_TILE_ROWS, r, G};
    const uint32_t rows = G > 1 ? rt_tiles_local_rows(scene->height, &g->tiles[r]) : scene->height;
    if (rows > g->pad_rows) g->pad_rows = rows;
  }
  g->pad_bytes = (size_t)g->pad_rows * g->row_bytes;
  const char* tr = std::getenv("RT_GATHER");
  if (tr && std::strcmp(tr, "rccl") && std::strcmp(tr, "peer")) { delete g; return fail(RT_ERR_INVALID, "RT_GATHER must be rccl or peer"); }
  const char* st = std::getenv("RT_GATHER_SELFTEST");
  g->gather = G > 1 || (st && st[0] == '1');
  g->rccl = g->gather && (tr ? !std::strcmp(tr, "rccl") : !shared_device);
  if (g->rccl && shared_device) { delete g; return fail(RT_ERR_INVALID, "RT_GATHER=rccl needs one device per rank (RT_GPUS_EMULATE shares devices)"); }
  auto bail = [&](int code, const std::string& m) { rt_hip_group_destroy(g); return fail(code, m); };
  // scene replicas: one thread per rank (table upload + texture copy run in parallel)
  {
    std::vector<std::thread> th;
    for (uint32_t r = 0; r < G; ++r)
      th.emplace_back([g, scene, r]() {
        g->rc[r] = rt_hip_scene_create(scene, g->device[r], &g->scene[r]);
        if (g->rc[r] != RT_OK) { g->err[r] = rt_hip_last_error(); return; }
        if (hipStreamCreateWithFlags(&g->stream[r], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&g->ev_done[r], hipEventDisableTiming) != hipSuccess) {
          g->rc[r] = RT_ERR_HIP; g->err[r] = "hipStreamCreate/hipEventCreate failed";
          return;
        }
        if (r != 0 && hipMalloc(&g->d_tiles[r], g->pad_bytes ? g->pad_bytes : 16) != hipSuccess) { g->rc[r] = RT_ERR_HIP; g->err[r] = "hipMalloc(tiles) failed"; }
      });
    for (auto& t : th) t.join();
    for (uint32_t r = 0; r < G; ++r)
      if (g->rc[r] != RT_OK) return bail(g->rc[r], "rank " + std::to_string(r) + ": " + g->err[r]);
  }
  if (hipSetDevice(g->device[0]) != hipSuccess) return bail(RT_ERR_HIP, "hipSetDevice failed");
  if (hipMalloc(&g->d_stacked, g->pad_bytes * G ? g->pad_bytes * G : 16) != hipSuccess) return bail(RT_ERR_HIP, "hipMalloc(gather buffer) failed");
  g->d_tiles[0] = g->d_stacked;  // rank 0 renders into its own slice: the gather is in place on the root
  if (g->gather) {
    if (hipMalloc(&g->d_frame, (size_t)g->height * g->row_bytes ? (size_t)g->height * g->row_bytes : 16) != hipSuccess) return bail(RT_ERR_HIP, "hipMalloc(frame) failed");
  } else g->d_frame = g->d_stacked;
  if (hipEventCreate(&g->ev_assembled) != hipSuccess) return bail(RT_ERR_HIP, "hipEventCreate failed");
  if (g->rccl) {
    std::string why;
    if (!g->api.load(why)) return bail(RT_ERR_HIP, why);
    g->comm.assign(G, nullptr);
    const ncclResult_t nr = g->api.CommInitAll(g->comm.data(), (int)G, g->device.data());
    if (nr != ncclSuccess) { g->comm.clear(); return bail(RT_ERR_HIP, std::string("ncclCommInitAll: ") + g->api.GetErrorString(nr)); }
  } else if (g->gather) {
    for (uint32_t r = 1; r < G; ++r)
      if (g->device[r] != g->device[0]) { (void)hipSetDevice(g->device[r]); (void)hipDeviceEnablePeerAccess(g->device[0], 0); }
    (void)hipGetLastError();  // (already enabled / not supported: the copy is staged instead)
  }
  for (uint32_t r = 0; r < G; ++r) g->worker.emplace_back(rtg::worker_main, g, r);
  *out = g;
  return RT_OK;
}

extern "C" int rt_hip_group_set_camera(RtHipGroup* g, const double origin[3], const double lower_left[3], const double horizontal[3],
                                       const double vertical[3]) {
  if (!g) return fail(RT_ERR_INVALID, "null argument");
  for (uint32_t r = 0; r < g->G; ++r) {
    const int rc = rt_hip_set_camera(g->scene[r], origin, lower_left, horizontal, vertical);
    if (rc != RT_OK) return rc;
  }
  return RT_OK;
}

extern "C" int rt_hip_group_set_option(RtHipGroup* g, const char* key, int64_t value) {
  if (!g) return fail(RT_ERR_INVALID, "null argument");
  for (uint32_t r = 0; r < g->G; ++r) {
    const int rc = rt_hip_set_option(g->scene[r], key, value);
    if (rc != RT_OK) return rc;
  }
  return RT_OK;
}

extern "C" int rt_hip_group_info(const RtHipGroup* g, RtGroupInfo* info) {
  if (!g || !info) return fail(RT_ERR_INVALID, "null argument");
  std::memset(info, 0, sizeof *info);
  info->n_ranks = g->G;
  info->transport = !g->gather ? RT_GATHER_NONE : (g->rccl ? RT_GATHER_RCCL : RT_GATHER_PEER);
  info->rccl_comms = (uint32_t)g->comm.size();
  info->tile_rows = RT_GROUP_TILE_ROWS; info->pad_rows = g->pad_rows;
  info->emulated = g->shared_device ? 1u : 0u;
  uint64_t seen[16] = {0};  // device ordinals < 1024
  for (uint32_t r = 0; r < RT_GROUP_INFO_MAX_RANKS; ++r) info->device[r] = -1;
  for (uint32_t r = 0; r < g->G; ++r) {
    const int d = g->device[r];
    if (r < RT_GROUP_INFO_MAX_RANKS) info->device[r] = d;
    if (d >= 0 && d < 1024 && !((seen[d >> 6] >> (d & 63)) & 1ull)) { seen[d >> 6] |= 1ull << (d & 63); info->n_devices++; }
  }
  return RT_OK;
}

extern "C" const void* rt_hip_group_frame(const RtHipGroup* g, int* device_out) {
  if (!g) return nullptr;
  if (device_out) *device_out = g->device.empty() ? 0 : g->device[0];
  return g->d_frame;
}

namespace rtg {
// One frame: G parallel launches, ONE gather, de-interleave; the frame is left in scanline order on the group's first
// device (rt_hip_group_frame) and, when out_rgb8 is given, leaves in ONE device-to-host copy.  Blocking.
int group_frame(RtHipGroup* g, uint8_t* out_rgb8, RtStats* stats) {
  const auto t0 = std::chrono::steady_clock::now();
  const uint32_t G = g->G;
  {
    std::lock_guard<std::mutex> lk(g->mu);
    g->n_done = 0;
    g->generation++;
  }
  g->cv_go.notify_all();
  {
    std::unique_lock<std::mutex> lk(g->mu);
    g->cv_done.wait(lk, [&] { return g->n_done == G; });
  }
  // whatever fails from here on, every stream is drained and every scene is released before the error is returned
  // (a scene left "in flight" would refuse the caller's next frame)
  auto drain = [&]() {
    const std::string keep = g_err;
    for (uint32_t q = 0; q < G; ++q) { (void)hipSetDevice(g->device[q]); (void)hipStreamSynchronize(g->stream[q]); g->scene[q]->in_flight = false; }
    (void)hipGetLastError();
    g_err = keep;
  };
  for (uint32_t r = 0; r < G; ++r)
    if (g->rc[r] != RT_OK) {
      drain();
      return fail(g->rc[r], "rank " + std::to_string(r) + ": " + g->err[r]);
    }
  RtStats total;
  std::memset(&total, 0, sizeof total);
  double frame_ms = 0.0;
  auto assemble = [&]() -> int {
    RT_HIP_TRY(hipSetDevice(g->device[0]));
    hipStream_t s0 = g->stream[0];
    if (g->gather) {
      if (g->rccl) {  // ONE gather over xGMI: every rank's packed tiles -> rank 0's `stacked`, stream-ordered after its kernel
        ncclResult_t nr = g->api.GroupStart();
        for (uint32_t r = 0; r < G && nr == ncclSuccess; ++r)
          nr = g->api.Gather(g->d_tiles[r], g->d_stacked, g->pad_bytes, ncclUint8, 0, g->comm[r], g->stream[r]);
        const ncclResult_t ne = g->api.GroupEnd();
        if (nr == ncclSuccess) nr = ne;
        if (nr != ncclSuccess) return fail(RT_ERR_HIP, std::string("ncclGather: ") + g->api.GetErrorString(nr));
      } else {
        for (uint32_t r = 1; r < G; ++r) RT_HIP_TRY(hipStreamWaitEvent(s0, g->ev_done[r], 0));
      }
      hipLaunchKernelGGL(rtg::deinterleave_rows, dim3(g->height), dim3(256), 0, s0, static_cast<const uint8_t*>(g->d_stacked),
                         static_cast<uint8_t*>(g->d_frame), g->height, (uint32_t)g->row_bytes, G, RT_GROUP_TILE_ROWS, g->pad_rows);
      RT_HIP_TRY(hipGetLastError());
    }
    RT_HIP_TRY(hipEventRecord(g->ev_assembled, s0));
    if (out_rgb8) RT_HIP_TRY(hipMemcpyAsync(out_rgb8, g->d_frame, (size_t)g->height * g->row_bytes, hipMemcpyDeviceToHost, s0));
    RT_HIP_TRY(hipStreamSynchronize(s0));
    frame_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (uint32_t r = 0; r < G; ++r) {  // (also drains the other ranks' streams: with RCCL their gather kernels)
      RtStats st;
      const int rc = rt_hip_wait(g->scene[r], &st);
      if (rc != RT_OK) return rc;
      total.samples += st.samples; total.segments += st.segments; total.sphere_tests += st.sphere_tests;
      total.exact_tests += st.exact_tests; total.tex_oob += st.tex_oob; total.grid_steps += st.grid_steps;
      for (int k = 0; k < 4; ++k) total.wave_iters[k] += st.wave_iters[k];
      if (st.kernel_ms > total.kernel_ms) total.kernel_ms = st.kernel_ms;  // the slowest rank
    }
    if (stats) {
      *stats = total;
      stats->n_gpus_used = G;
      stats->frame_ms = frame_ms;
      if (g->gather && g->scene[0]->launched) {
        RT_HIP_TRY(hipSetDevice(g->device[0]));
        float ms = 0.f;
        RT_HIP_TRY(hipEventElapsedTime(&ms, g->scene[0]->ev_stop, g->ev_assembled));
        stats->gather_ms = ms;  // rank 0's kernel end -> frame in scanline order on device 0 (includes waiting for slower ranks)
      }
    }
    return RT_OK;
  };
  const int rc = assemble();
  if (rc != RT_OK) drain();
  return rc;
}
}  // namespace rtg

extern "C" int rt_hip_group_render_to_host(RtHipGroup* g, uint8_t* out_rgb8, RtStats* stats) {
  if (!g || !out_rgb8) return fail(RT_ERR_INVALID, "null argument");
  return rtg::group_frame(g, out_rgb8, stats);
}
extern "C" int rt_hip_group_render(RtHipGroup* g, RtStats* stats) {
  if (!g) return fail(RT_ERR_INVALID, "null argument");
  return rtg::group_frame(g, nullptr, stats);
}

// drop-in for the parallel loop of render() (raytracer.rs:254-263): host scene in, host RGB8 out
extern "C" int rt_render_rgb8(const RtScene* scene, uint8_t* out_rgb8, RtStats* stats) {
  if (!scene || !out_rgb8) return fail(RT_ERR_INVALID, "null argument");
  const auto t0 = std::chrono::steady_clock::now();
  RtHipGroup* g = nullptr;
  int rc = rt_hip_group_create(scene, 0, &g);
  if (rc != RT_OK) return rc;
  // one frame per scene: no later frame could use a queue order learned from this one (tile_order 2 would measure
  // the tile depths and run rt_order_tiles inside frame_ms for nothing) — bottom row first
  (void)rt_hip_group_set_option(g, "tile_order", 1);
  const double setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  RtStats st;
  rc = rt_hip_group_render_to_host(g, out_rgb8, &st);
  const std::string keep = g_err;
  rt_hip_group_destroy(g);
  if (rc != RT_OK) { g_err = keep; return rc; }
  if (stats) {
    *stats = st;
    stats->setup_ms = setup_ms;  // HIP context, table build, scene upload (RCCL communicators with n_gpus > 1): before the window
  }
  return RT_OK;
}
