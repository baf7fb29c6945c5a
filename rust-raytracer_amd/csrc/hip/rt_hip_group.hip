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
// Frames are PIPELINED two deep (rt_hip_group_submit / rt_hip_group_collect; the blocking calls are submit + collect):
// everything a frame touches exists twice (gather buffer, frame buffer, per-rank tile buffers, events, the scenes' stats
// slots), a rank's kernels run on its render stream and its tiles travel on a second, TRANSFER stream that waits for the
// kernel's event — so frame i's gather, de-interleave and device-to-host copy run while frame i+1 renders, and an
// animation's step is the slowest rank's kernel, not kernel + gather + copy.
//
// The transport can never cost the frame: RCCL is the default with one device per rank, but if its library does not load
// (RT_RCCL_LIB names another path), ncclCommInitAll fails, or the self-test gather run at creation fails, times out or
// delivers wrong bytes, the communicators are torn down and the group runs on peer copies — and says so (RtGroupInfo.
// transport_fallback, rt_hip_group_fallback_reason).  A gather that fails to enqueue in a later frame switches the same way
// and re-sends that frame's tiles.  Rank threads are pinned to the CPUs of their device's NUMA node (RT_GROUP_PIN=0: not).
// A frame handed to a pageable host buffer leaves the device into a pinned staging buffer of the group (an asynchronous
// copy into pageable memory is synchronous in HIP: submit used to block until the frame was done) and collect moves it on.
//
// Test hooks: RT_GPUS_EMULATE=1 lets ranks share devices (rank r -> device r mod visible devices; peer
// transport only), so the whole path — threads, sharding, gather buffer layout, de-interleave — runs on a
// one-GPU box; RT_GATHER_SELFTEST=1 makes a ONE-rank group go through the gather (RCCL communicator of one
// rank, in-place ncclGather) and the de-interleave kernel too.
#include <dlfcn.h>
#include <pthread.h>
#include <sched.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is dlopen'ed below, never linked

#include <atomic>
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
  decltype(&ncclCommAbort) CommAbort = nullptr;  // (optional: only used to abandon communicators whose self-test hangs)
  bool load(std::string& err) {
    if (h) return true;
    if (const char* forced = std::getenv("RT_RCCL_LIB")) {  // (a site's own build of the library; tests: a path that does not exist)
      h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
      if (!h) { const char* e = dlerror(); err = std::string("cannot load RCCL (RT_RCCL_LIB=") + forced + "): " + (e ? e : "?"); return false; }
    } else {
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
      }
      if (!h) { const char* e = dlerror(); err = std::string("cannot load RCCL: ") + (e ? e : "?"); return false; }
    }
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
    CommAbort = reinterpret_cast<decltype(CommAbort)>(dlsym(h, "ncclCommAbort"));
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
  bool wanted_rccl = false;          // the transport asked for (environment / default) was RCCL
  std::string fallback_reason;       // why it is not the one in use ("" when it is)
  bool pin_threads = true;           // RT_GROUP_PIN != 0
  struct RankPlace { int numa_node = -1; int pinned_cpus = 0; int peer_to_root = 1; char pci[16] = {0}; std::vector<int> cpus; };
  std::vector<RankPlace> place;      // rank r: where its device sits (filled at creation)
  std::vector<double> kernel_ms_last;  // rank r: its kernel of the frame collected last
  bool shared_device = false;        // RT_GPUS_EMULATE: some ranks share a device
  bool gather = false;               // G > 1 (or the one-rank self-test): gather + de-interleave after the kernels
  std::vector<int> device;
  std::vector<RtHipScene*> scene;
  // OVERLAPPED FRAMES (round 6).  A frame ends on its deepest paths: while the last waves of frame i trace them, the CUs whose
  // persistent workgroups found the queue empty sit idle — 3 % of a whole headline frame without a learned queue order, 14 % of
  // an 1/8 shard (tools/experiments/overlap_tail.py, profiles/r06_run11_overlap_tail.log).  Kernels on different HIP streams are
  // independent, so odd frames run on a second VIEW of the rank's scene (rt_hip_scene_clone_view: the same tables in HBM, its
  // own tile queue / counters / stats slots) through a second render stream: frame i + 1's workgroups start on every CU frame i
  // has left.  Only with two frames in flight (rt_hip_group_submit / _collect); a blocking frame and a one-shot group never
  // touch the second view.  RT_GROUP_OVERLAP=0: one scene, one render stream (rounds 4 - 5).
  std::vector<RtHipScene*> scene2;   // rank r: the view odd frames render through (empty: no overlap)
  std::vector<hipStream_t> stream2;  // ... and its render stream
  std::vector<hipEvent_t> prev_stop; // rank r: ev_stop of the kernel of the frame submitted before the current one (its effective time starts there at the earliest)
  std::vector<RtRowTiles> tiles;     // rank r renders RtRowTiles{RT_GROUP_TILE_ROWS, r, G}
  std::vector<hipStream_t> stream;   // rank r: its kernels
  std::vector<hipStream_t> xstream;  // rank r: the transfer of its tiles (peer copy / its side of the gather); xstream[0] also
                                     // assembles: de-interleave + the device-to-host copy
  // One frame in flight: its buffers and events (two of them, used alternately).
  struct Frame {
    void* d_stacked = nullptr;         // device 0: G x pad_rows rows
    void* d_frame = nullptr;           // device 0: height rows (no gather: the same buffer)
    std::vector<void*> d_tiles;        // rank r's packed tiles on ITS device (rank 0: a slice of `d_stacked`)
    std::vector<hipEvent_t> ev_done;   // rank r: its kernel finished (recorded on stream[r])
    std::vector<hipEvent_t> ev_sent;   // peer transport, r > 0: its tiles are in `d_stacked` (recorded on xstream[r])
    std::vector<int> slot;             // the stats slot of scene[r] (or scene2[r]) this frame's launch used
    int which = 0;                     // 0: the ranks' scenes, 1: their second views (overlapped frames)
    std::vector<hipEvent_t> ev_prev_stop;  // rank r: the previous frame's kernel-end event at the time this frame was submitted (null: none)
    hipEvent_t ev_assembled = nullptr; // frame in scanline order on device 0 (xstream[0]; timed)
    hipEvent_t ev_final = nullptr;     // ... and in the caller's buffer, if one was given
    bool busy = false;                 // submitted, not collected
    uint8_t* out = nullptr;
    uint8_t* h_stage = nullptr;        // pinned staging buffer of the frame (height rows), allocated when a pageable `out` is first seen
    bool staged = false;               // this frame's device-to-host copy went into h_stage: collect moves it into `out`
    std::chrono::steady_clock::time_point t0;
    double us[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // RtStats.group_us
  } frame[2];
  uint64_t n_submitted = 0, n_collected = 0;
  const Frame* last = nullptr;       // the frame rt_hip_group_frame points at (the last one collected)
  rtg::RcclApi api;
  std::vector<ncclComm_t> comm;
  // one persistent host thread per rank r >= 1 (rank 0 is launched by the submitting thread itself, which is awake
  // anyway): the launches of a frame go out in parallel — a serial loop would start rank 7 ~8 x 40 us late on a 2 ms shard
  std::vector<std::thread> worker;
  std::mutex mu;
  std::condition_variable cv_go, cv_done;
  std::atomic<uint64_t> generation{0};
  std::atomic<uint32_t> n_done{0};
  int cur = 0;                       // the Frame the workers are enqueuing
  bool quit = false;
  bool own_streams = true;           // false: a one-shot group of one rank runs on the device's NULL stream (rt_render_rgb8: no queue of its own to create)
  int inject = 0;                    // probe build only (RT_RCCL_INJECT, read once at creation): 1 = the self-test gather delivers wrong bytes, 2 = the second frame's gather fails to enqueue
  std::atomic<int> spin_us{0};       // "spin_us" option: a worker polls for the next frame this long before it sleeps on the condition variable
  std::vector<int> rc;
  std::vector<std::string> err;
  std::vector<double> t_wake_us, t_enq_us;  // per rank: since Frame::t0 — thread running; its launch (+ transfer) enqueued
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

inline double us_since(std::chrono::steady_clock::time_point t0) {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}

// where device `dev` sits: PCI bus id, the NUMA node sysfs reports for it, the CPUs of that node
void locate_device(int dev, RtHipGroup::RankPlace& p) {
  char id[32] = {0};
  if (hipDeviceGetPCIBusId(id, (int)sizeof id, dev) != hipSuccess) { (void)hipGetLastError(); return; }
  for (char* c = id; *c; ++c) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');  // sysfs names are lower case
  std::snprintf(p.pci, sizeof p.pci, "%s", id);
  auto slurp = [](const std::string& path, std::string& out) {
    FILE* f = std::fopen(path.c_str(), "r");
    if (!f) return false;
    char buf[4096];
    const size_t n = std::fread(buf, 1, sizeof buf - 1, f);
    std::fclose(f);
    buf[n] = 0; out = buf;
    return true;
  };
  std::string t;
  if (!slurp(std::string("/sys/bus/pci/devices/") + id + "/numa_node", t)) return;
  const long node = std::strtol(t.c_str(), nullptr, 10);
  if (node < 0) return;  // (-1: the platform does not say — VMs, single-socket boxes)
  p.numa_node = (int)node;
  if (!slurp("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist", t)) return;
  for (const char* c = t.c_str(); *c;) {  // "0-31,64-95"
    char* e = nullptr;
    const long a = std::strtol(c, &e, 10);
    if (e == c) break;
    long b = a;
    if (*e == '-') { const char* c2 = e + 1; b = std::strtol(c2, &e, 10); if (e == c2) break; }
    for (long k = a; k <= b && k < CPU_SETSIZE; ++k) p.cpus.push_back((int)k);
    c = *e == ',' ? e + 1 : e;
    if (*e != ',' ) break;
  }
}
// the calling thread runs on the CPUs of rank r's device from now on (memory it touches first lands on that node)
void pin_to_rank(RtHipGroup* g, uint32_t r) {
  RtHipGroup::RankPlace& p = g->place[r];
  if (!g->pin_threads || p.cpus.empty()) return;
  cpu_set_t set;
  CPU_ZERO(&set);
  for (int c : p.cpus) CPU_SET(c, &set);
  if (pthread_setaffinity_np(pthread_self(), sizeof set, &set) == 0) p.pinned_cpus = (int)p.cpus.size();
}

// RCCL as the gather's transport: library, communicators, and ONE gather of known bytes through the frame's own buffers
// before any frame depends on it (it also takes RCCL's lazy first-collective setup out of the first frame).  Returns ""
// when the transport is usable; otherwise everything it created is gone again and the text says what failed.
std::string try_rccl(RtHipGroup* g) {
  std::string why;
  if (!g->api.load(why)) return why;
  const uint32_t G = g->G;
  g->comm.assign(G, nullptr);
  ncclResult_t nr = g->api.CommInitAll(g->comm.data(), (int)G, g->device.data());
  if (nr != ncclSuccess) { g->comm.clear(); return std::string("ncclCommInitAll: ") + g->api.GetErrorString(nr); }
  auto drop_comms = [&](bool abort) {
    for (uint32_t r = 0; r < g->comm.size(); ++r)
      if (g->comm[r]) { (void)hipSetDevice(g->device[r]); if (abort && g->api.CommAbort) (void)g->api.CommAbort(g->comm[r]); else (void)g->api.CommDestroy(g->comm[r]); }
    g->comm.clear();
    (void)hipGetLastError();
  };
  if (g->pad_bytes == 0) return "";
  RtHipGroup::Frame& f = g->frame[0];
  const size_t probe = g->pad_bytes < 4096 ? g->pad_bytes : 4096;  // bytes per rank the self-test sends (the buffers are pad_bytes each)
  for (uint32_t r = 0; r < G; ++r) {
    if (hipSetDevice(g->device[r]) != hipSuccess || hipMemsetAsync(f.d_tiles[r], (int)(r + 1u), probe, g->xstream[r]) != hipSuccess) {
      drop_comms(false);
      return "self-test: hipMemsetAsync failed";
    }
  }
  nr = g->api.GroupStart();
  for (uint32_t r = 0; r < G && nr == ncclSuccess; ++r) nr = g->api.Gather(f.d_tiles[r], f.d_stacked, probe, ncclUint8, 0, g->comm[r], g->xstream[r]);
  const ncclResult_t ne = g->api.GroupEnd();
  if (nr == ncclSuccess) nr = ne;
  if (nr != ncclSuccess) { drop_comms(false); return std::string("self-test ncclGather: ") + g->api.GetErrorString(nr); }
  long timeout_ms = 20000;
  if (const char* e = std::getenv("RT_RCCL_TIMEOUT_MS")) { const long v = std::strtol(e, nullptr, 10); if (v > 0) timeout_ms = v; }
  const auto until = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms);
  for (uint32_t r = 0; r < G; ++r) {
    (void)hipSetDevice(g->device[r]);
    for (;;) {
      const hipError_t q = hipStreamQuery(g->xstream[r]);
      if (q == hipSuccess) break;
      if (q != hipErrorNotReady) { (void)hipGetLastError(); drop_comms(true); return std::string("self-test gather: ") + hipGetErrorString(q); }
      if (std::chrono::steady_clock::now() > until) {
        (void)hipGetLastError();
        drop_comms(true);  // abandon the communicators; the transfer streams may be wedged behind them: fresh ones for the peer copies
        for (uint32_t q2 = 0; q2 < G; ++q2) {
          (void)hipSetDevice(g->device[q2]);
          hipStream_t fresh = nullptr;
          if (hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) == hipSuccess) g->xstream[q2] = fresh;  // (the old stream is leaked on purpose)
        }
        return "self-test gather timed out after " + std::to_string(timeout_ms) + " ms";
      }
      std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
  }
  (void)hipSetDevice(g->device[0]);
  std::vector<uint8_t> got(G, 0);
  for (uint32_t r = 0; r < G; ++r)
    if (hipMemcpy(&got[r], static_cast<const uint8_t*>(f.d_stacked) + (size_t)r * probe + (probe - 1), 1, hipMemcpyDeviceToHost) != hipSuccess) {
      (void)hipGetLastError(); drop_comms(false); return "self-test: reading the gathered bytes back failed";
    }
#ifdef RT_TEST_PROBES  // fault injection exists in the probe build only: a stray environment variable cannot switch the product's transport
  if (g->inject == 1) got[G - 1] = 0;  // (test hook RT_RCCL_INJECT=selftest: the self-test sees wrong bytes)
#endif
  for (uint32_t r = 0; r < G; ++r)
    if (got[r] != (uint8_t)(r + 1u)) { drop_comms(false); return "self-test gather delivered rank " + std::to_string(r) + "'s bytes as " + std::to_string((int)got[r]); }
  return "";
}
// peer copies instead (and from now on): peer access towards the root where the devices allow it, else the runtime stages
void use_peer_transport(RtHipGroup* g, const std::string& reason) {
  g->rccl = false;
  g->fallback_reason = reason;
  for (uint32_t r = 1; r < g->G; ++r)
    if (g->device[r] != g->device[0]) { (void)hipSetDevice(g->device[r]); (void)hipDeviceEnablePeerAccess(g->device[0], 0); }
  (void)hipGetLastError();  // (already enabled / not supported: the copy is staged instead)
  (void)hipSetDevice(g->device[0]);
}

// rank r's part of frame `f`: kernel on its render stream, then — on its transfer stream, behind the kernel's event —
// its slice of the gather (peer transport; RCCL's gather is enqueued for all ranks together by the submitting thread)
void enqueue_rank(RtHipGroup* g, RtHipGroup::Frame& f, uint32_t r) {
  g->t_wake_us[r] = us_since(f.t0);
  RtHipScene* sc = f.which ? g->scene2[r] : g->scene[r];
  hipStream_t rs = f.which ? g->stream2[r] : g->stream[r];
  f.slot[r] = (int)(sc->n_launches & 1);
  f.ev_prev_stop[r] = g->prev_stop[r];
  int rc = rt_hip_render(sc, g->G > 1 ? &g->tiles[r] : nullptr, f.d_tiles[r], nullptr, rs);
  g->prev_stop[r] = sc->slot[f.slot[r] & 1].ev_stop;
  std::string err;
  if (rc != RT_OK) err = rt_hip_last_error();
  auto hip = [&](hipError_t e, const char* what) { if (rc == RT_OK && e != hipSuccess) { rc = RT_ERR_HIP; err = std::string(what) + ": " + hipGetErrorString(e); } };
  if (rc == RT_OK) hip(hipEventRecord(f.ev_done[r], rs), "hipEventRecord");
  if (rc == RT_OK && g->gather) {
    hip(hipStreamWaitEvent(g->xstream[r], f.ev_done[r], 0), "hipStreamWaitEvent");
    if (!g->rccl && r != 0) {
      hip(hipMemcpyPeerAsync(static_cast<uint8_t*>(f.d_stacked) + (size_t)r * g->pad_bytes, g->device[0], f.d_tiles[r], g->device[r], g->pad_bytes,
                             g->xstream[r]), "hipMemcpyPeerAsync");
      hip(hipEventRecord(f.ev_sent[r], g->xstream[r]), "hipEventRecord");
    }
  }
  g->rc[r] = rc; g->err[r] = err;
  g->t_enq_us[r] = us_since(f.t0);
}

void worker_main(RtHipGroup* g, uint32_t r) {
  pin_to_rank(g, r);
  uint64_t seen = 0;
  for (;;) {
    if (const int spin = g->spin_us.load(std::memory_order_relaxed)) {  // (frames of an animation follow each other closely: poll before sleeping)
      const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(spin);
      while (g->generation.load(std::memory_order_acquire) == seen && std::chrono::steady_clock::now() < until) __builtin_ia32_pause();
    }
    {
      std::unique_lock<std::mutex> lk(g->mu);
      g->cv_go.wait(lk, [&] { return g->quit || g->generation.load(std::memory_order_acquire) != seen; });
      if (g->quit) return;
      seen = g->generation.load(std::memory_order_acquire);
    }
    enqueue_rank(g, g->frame[g->cur], r);
    if (g->n_done.fetch_add(1, std::memory_order_acq_rel) + 1 == g->G - 1) {
      std::lock_guard<std::mutex> lk(g->mu);  // (the waiter re-checks n_done under this mutex: no lost wake-up)
      g->cv_done.notify_one();
    }
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
  for (uint32_t r = 0; r < g->scene.size(); ++r) {  // nothing of ours may be running when buffers and communicators go
    (void)hipSetDevice(g->device[r]);
    if (r < g->stream.size() && (g->stream[r] || !g->own_streams)) (void)hipStreamSynchronize(g->stream[r]);
    if (r < g->stream2.size() && g->stream2[r]) (void)hipStreamSynchronize(g->stream2[r]);
    if (r < g->xstream.size() && g->xstream[r]) (void)hipStreamSynchronize(g->xstream[r]);
  }
  for (uint32_t r = 0; r < g->comm.size(); ++r)
    if (g->comm[r]) { (void)hipSetDevice(g->device[r]); (void)g->api.CommDestroy(g->comm[r]); }
  for (uint32_t r = 0; r < g->scene.size(); ++r) {
    (void)hipSetDevice(g->device[r]);
    for (auto& f : g->frame) {
      if (r != 0 && r < f.d_tiles.size() && f.d_tiles[r]) (void)hipFree(f.d_tiles[r]);
      if (r < f.ev_done.size() && f.ev_done[r]) (void)hipEventDestroy(f.ev_done[r]);
      if (r < f.ev_sent.size() && f.ev_sent[r]) (void)hipEventDestroy(f.ev_sent[r]);
    }
    if (r < g->scene2.size() && g->scene2[r]) rt_hip_scene_destroy(g->scene2[r]);  // (the views first: they share their scene's tables)
    if (g->scene[r]) rt_hip_scene_destroy(g->scene[r]);
    if (r < g->stream2.size() && g->stream2[r]) (void)hipStreamDestroy(g->stream2[r]);
    if (g->own_streams && r < g->stream.size() && g->stream[r]) (void)hipStreamDestroy(g->stream[r]);
    if (g->own_streams && r < g->xstream.size() && g->xstream[r]) (void)hipStreamDestroy(g->xstream[r]);
  }
  if (!g->device.empty()) (void)hipSetDevice(g->device[0]);
  for (auto& f : g->frame) {
    if (f.h_stage) (void)hipHostFree(f.h_stage);
    if (f.d_frame && f.d_frame != f.d_stacked) (void)hipFree(f.d_frame);
    if (f.d_stacked) (void)hipFree(f.d_stacked);
    if (f.ev_assembled) (void)hipEventDestroy(f.ev_assembled);
    if (f.ev_final) (void)hipEventDestroy(f.ev_final);
  }
  delete g;
}

namespace rtg {
// one_shot: the group renders ONE frame and goes (rt_render_rgb8 — the reference's one frame per process, main.rs:7-20).  Every
// stream of its own is a hardware queue the runtime has to create (7.8 ms each on MI355X, serialised: tools/microbench/
// setup_costs.hip, profiles/r06_run3_setup_costs.log) and buys a single blocking frame nothing: a one-rank one-shot group
// renders, assembles and copies on the device's NULL stream, whose queue exists since the device was warmed.
int group_create(const RtScene* scene, uint32_t n_gpus, RtHipGroup** out, bool one_shot);
}
extern "C" int rt_hip_group_create(const RtScene* scene, uint32_t n_gpus, RtHipGroup** out) {
  return rtg::group_create(scene, n_gpus, out, false);
}
int rtg::group_create(const RtScene* scene, uint32_t n_gpus, RtHipGroup** out, bool one_shot) {
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
  const auto t_group = std::chrono::steady_clock::now();
  rtp::reset();
  rtp::Clock pc;
  RtHipGroup* g = new RtHipGroup;
  g->G = G; g->width = scene->width; g->height = scene->height;
  g->own_streams = !(one_shot && G == 1 && !std::getenv("RT_ONE_SHOT_STREAMS"));
  g->row_bytes = (size_t)scene->width * 3;
  g->device.resize(G); g->scene.assign(G, nullptr); g->stream.assign(G, nullptr); g->xstream.assign(G, nullptr);
  g->tiles.resize(G); g->rc.assign(G, RT_OK); g->err.resize(G); g->t_wake_us.assign(G, 0.0); g->t_enq_us.assign(G, 0.0);
  g->place.resize(G); g->kernel_ms_last.assign(G, 0.0);
  { const char* pin = std::getenv("RT_GROUP_PIN"); g->pin_threads = !(pin && pin[0] == '0'); }
#ifdef RT_TEST_PROBES
  if (const char* inj = std::getenv("RT_RCCL_INJECT")) g->inject = !std::strcmp(inj, "selftest") ? 1 : (!std::strcmp(inj, "gather") ? 2 : 0);
#endif
  for (auto& f : g->frame) { f.d_tiles.assign(G, nullptr); f.ev_done.assign(G, nullptr); f.ev_sent.assign(G, nullptr); f.slot.assign(G, 0); f.ev_prev_stop.assign(G, nullptr); }
  g->prev_stop.assign(G, nullptr);
  const char* ov = std::getenv("RT_GROUP_OVERLAP");
  const bool overlap = !one_shot && !(ov && ov[0] == '0');
  if (overlap) { g->scene2.assign(G, nullptr); g->stream2.assign(G, nullptr); }
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
  g->wanted_rccl = g->rccl;
  for (uint32_t r = 0; r < G; ++r) {
    rtg::locate_device(g->device[r], g->place[r]);
    int can = 1;
    if (g->device[r] != g->device[0] && hipDeviceCanAccessPeer(&can, g->device[r], g->device[0]) != hipSuccess) { can = 0; (void)hipGetLastError(); }
    g->place[r].peer_to_root = can;
  }
  auto bail = [&](int code, const std::string& m) { rt_hip_group_destroy(g); return fail(code, m); };
  pc.mark("group.locate_devices");
  // scene replicas: one thread per rank (table upload + texture copy run in parallel)
  {
    std::vector<std::thread> th;
    for (uint32_t r = 0; r < G; ++r)
      th.emplace_back([g, scene, r]() {
        rtp::tl_record = r == 0;
        rtp::tl_in_group = true;
        rtp::Clock rc_clock;
        rtg::pin_to_rank(g, r);  // (the replica's pinned counter words and staging copies are first touched on the device's node)
        g->rc[r] = rt_hip_scene_create(scene, g->device[r], &g->scene[r]);
        if (g->rc[r] != RT_OK) { g->err[r] = rt_hip_last_error(); return; }
        rc_clock.t = std::chrono::steady_clock::now();   // (the scene's own stages are booked by rt_hip_scene_create)
        bool ok = !g->own_streams || (hipStreamCreateWithFlags(&g->stream[r], hipStreamNonBlocking) == hipSuccess &&
                                      hipStreamCreateWithFlags(&g->xstream[r], hipStreamNonBlocking) == hipSuccess);
        for (auto& f : g->frame) {
          ok = ok && hipEventCreateWithFlags(&f.ev_done[r], hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&f.ev_sent[r], hipEventDisableTiming) == hipSuccess;
          if (ok && r != 0 && hipMalloc(&f.d_tiles[r], g->pad_bytes ? g->pad_bytes : 16) != hipSuccess) ok = false;
        }
        if (!ok) { g->rc[r] = RT_ERR_HIP; g->err[r] = "hipStreamCreate / hipEventCreate / hipMalloc(tiles) failed"; return; }
        rc_clock.mark("rank0.streams_events_tile_buffers");
        // the rank's kernel once through the rank's own render stream (one scanline): the first frame finds a warm queue
        if (!std::getenv("RT_NO_KERNEL_WARMUP")) {
          const auto tw = std::chrono::steady_clock::now();
          g->rc[r] = rt_hip_scene_warm(g->scene[r], g->stream[r]);
          if (std::getenv("RT_GROUP_TRACE")) std::fprintf(stderr, "[rt group] rank %u kernel warm-up %.2f ms\n", r, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw).count());
          if (g->rc[r] != RT_OK) g->err[r] = rt_hip_last_error();
        }
        rc_clock.mark("rank0.kernel_warm_up_one_scanline");
        if (!g->scene2.empty() && g->rc[r] == RT_OK) {  // the second view + its render stream (overlapped frames)
          g->rc[r] = rt_hip_scene_clone_view(g->scene[r], &g->scene2[r]);
          if (g->rc[r] == RT_OK && hipStreamCreateWithFlags(&g->stream2[r], hipStreamNonBlocking) != hipSuccess) { g->rc[r] = RT_ERR_HIP; g->err[r] = "hipStreamCreate (second render stream) failed"; }
          else if (g->rc[r] != RT_OK) g->err[r] = rt_hip_last_error();
          if (g->rc[r] == RT_OK && !std::getenv("RT_NO_KERNEL_WARMUP")) { g->rc[r] = rt_hip_scene_warm(g->scene2[r], g->stream2[r]); if (g->rc[r] != RT_OK) g->err[r] = rt_hip_last_error(); }
          rc_clock.mark("rank0.second_view_and_stream");
        }
      });
    for (auto& t : th) t.join();
    pc.mark("group.replicas_total_incl_scene_and_rank0_stages");
    for (uint32_t r = 0; r < G; ++r)
      if (g->rc[r] != RT_OK) return bail(g->rc[r], "rank " + std::to_string(r) + ": " + g->err[r]);
  }
  if (hipSetDevice(g->device[0]) != hipSuccess) return bail(RT_ERR_HIP, "hipSetDevice failed");
  for (auto& f : g->frame) {
    if (hipMalloc(&f.d_stacked, g->pad_bytes * G ? g->pad_bytes * G : 16) != hipSuccess) return bail(RT_ERR_HIP, "hipMalloc(gather buffer) failed");
    f.d_tiles[0] = f.d_stacked;  // rank 0 renders into its own slice: the gather is in place on the root
    if (g->gather) {
      if (hipMalloc(&f.d_frame, (size_t)g->height * g->row_bytes ? (size_t)g->height * g->row_bytes : 16) != hipSuccess) return bail(RT_ERR_HIP, "hipMalloc(frame) failed");
    } else f.d_frame = f.d_stacked;
    if (hipEventCreate(&f.ev_assembled) != hipSuccess || hipEventCreateWithFlags(&f.ev_final, hipEventDisableTiming) != hipSuccess)
      return bail(RT_ERR_HIP, "hipEventCreate failed");
  }
  g->last = &g->frame[0];
  pc.mark("group.frame_buffers_events");
  if (g->rccl) {  // never fatal: whatever RCCL cannot do here, peer copies can (the frame is the same bytes either way)
    const std::string why = rtg::try_rccl(g);
    if (!why.empty()) rtg::use_peer_transport(g, why);
  } else if (g->gather) rtg::use_peer_transport(g, "");
  pc.mark("group.transport");
  for (uint32_t r = 1; r < G; ++r) g->worker.emplace_back(rtg::worker_main, g, r);
  pc.mark("group.rank_threads");
  if (std::getenv("RT_GROUP_TRACE")) std::fprintf(stderr, "[rt group] create: %.2f ms in all\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_group).count());
  *out = g;
  return RT_OK;
}

extern "C" int rt_hip_group_set_camera(RtHipGroup* g, const double origin[3], const double lower_left[3], const double horizontal[3],
                                       const double vertical[3]) {
  if (!g) return fail(RT_ERR_INVALID, "null argument");
  for (uint32_t r = 0; r < g->G; ++r) {
    int rc = rt_hip_set_camera(g->scene[r], origin, lower_left, horizontal, vertical);
    if (rc == RT_OK && !g->scene2.empty()) rc = rt_hip_set_camera(g->scene2[r], origin, lower_left, horizontal, vertical);
    if (rc != RT_OK) return rc;
  }
  return RT_OK;
}

namespace rtg { void prepare_staging(RtHipGroup* g, int n_frames); }
extern "C" int rt_hip_group_set_option(RtHipGroup* g, const char* key, int64_t value) {
  if (!g || !key) return fail(RT_ERR_INVALID, "null argument");
  if (!std::strcmp(key, "prepare_host_output")) {  // the caller WILL pass host buffers to submit / render_to_host: make the pinned staging buffers (1 or 2
    // frames) and bring the device-to-host copy path up now, at set-up, instead of inside the first submit (~9 ms: the copy engine's queue)
    if (value < 1 || value > 2) return fail(RT_ERR_INVALID, "prepare_host_output must be 1 or 2 (frames in flight)");
    rtg::prepare_staging(g, (int)value);
    return RT_OK;
  }
  if (!std::strcmp(key, "spin_us")) {  // the group's own option: how long an idle rank thread polls before it sleeps
    if (value < 0 || value > 1000000) return fail(RT_ERR_INVALID, "spin_us must be 0 .. 1000000");
    g->spin_us.store((int)value, std::memory_order_relaxed);
    return RT_OK;
  }
  for (uint32_t r = 0; r < g->G; ++r) {
    int rc = rt_hip_set_option(g->scene[r], key, value);
    if (rc == RT_OK && !g->scene2.empty()) rc = rt_hip_set_option(g->scene2[r], key, value);
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
  info->transport_fallback = (g->wanted_rccl && !g->rccl) ? 1u : 0u;
  uint64_t seen[16] = {0};  // device ordinals < 1024
  for (uint32_t r = 0; r < RT_GROUP_INFO_MAX_RANKS; ++r) info->device[r] = -1;
  for (uint32_t r = 0; r < g->G; ++r) {
    const int d = g->device[r];
    if (r < RT_GROUP_INFO_MAX_RANKS) info->device[r] = d;
    if (d >= 0 && d < 1024 && !((seen[d >> 6] >> (d & 63)) & 1ull)) { seen[d >> 6] |= 1ull << (d & 63); info->n_devices++; }
  }
  return RT_OK;
}

extern "C" uint32_t rt_hip_group_ranks(const RtHipGroup* g, RtGroupRank* out, uint32_t cap) {
  if (!g) return 0u;
  for (uint32_t r = 0; out && r < g->G && r < cap; ++r) {
    RtGroupRank& o = out[r];
    std::memset(&o, 0, sizeof o);
    const RtHipGroup::RankPlace& p = g->place[r];
    o.device = g->device[r]; o.numa_node = p.numa_node; o.pinned_cpus = p.pinned_cpus; o.peer_to_root = p.peer_to_root;
    std::memcpy(o.pci_bus_id, p.pci, sizeof o.pci_bus_id);
    o.pci_bus_id[sizeof o.pci_bus_id - 1] = 0;
    o.kernel_ms = g->kernel_ms_last[r]; o.t_wake_us = g->t_wake_us[r]; o.t_enq_us = g->t_enq_us[r];
  }
  return g->G;
}
extern "C" const char* rt_hip_group_fallback_reason(const RtHipGroup* g) { return g ? g->fallback_reason.c_str() : ""; }

extern "C" const void* rt_hip_group_frame(const RtHipGroup* g, int* device_out) {
  if (!g) return nullptr;
  if (device_out) *device_out = g->device.empty() ? 0 : g->device[0];
  return g->last ? g->last->d_frame : nullptr;
}

namespace rtg {
// whatever failed, every stream is drained and every scene is released before the error is returned (a scene left "in
// flight" would refuse the caller's next frame); frames in flight are dropped
void drain(RtHipGroup* g) {
  const std::string keep = g_err;
  for (uint32_t q = 0; q < g->G; ++q) {
    (void)hipSetDevice(g->device[q]);
    (void)hipStreamSynchronize(g->stream[q]); (void)hipStreamSynchronize(g->xstream[q]);
    g->scene[q]->in_flight = false;
    if (!g->scene2.empty()) { (void)hipStreamSynchronize(g->stream2[q]); g->scene2[q]->in_flight = false; }
    g->prev_stop[q] = nullptr;
  }
  (void)hipGetLastError();
  for (auto& f : g->frame) f.busy = false;
  g->n_collected = g->n_submitted;
  g_err = keep;
}

// is `p` pinned host memory (hipHostMalloc / hipHostRegister)?  Pageable memory is unknown to the runtime: an error, cleared.
bool out_is_pinned(const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeHost;
}
// the staging buffers of both frames in flight, ahead of time (rt_render_rgb8: outside its frame_ms window)
void prepare_staging(RtHipGroup* g, int n_frames) {
  const size_t bytes = (size_t)g->height * g->row_bytes;
  if (!bytes) return;
  (void)hipSetDevice(g->device[0]);
  for (int i = 0; i < n_frames && i < 2; ++i) {
    auto& f = g->frame[i];
    if (!f.h_stage && hipHostMalloc((void**)&f.h_stage, bytes, hipHostMallocDefault) != hipSuccess) { f.h_stage = nullptr; (void)hipGetLastError(); }
  }
  // ... and the copy path itself: the runtime sets its device-to-host machinery up with the first copy of a process (measured:
  // the first hipMemcpyAsync of a one-shot frame kept submit for ~8 ms — seven times the reference's test-scene kernel,
  // profiles/r05_run7_cli_warm_spin.log): one copy of the frame's size through the frame's own stream and buffers, here
  if (g->frame[0].h_stage && g->frame[0].d_frame && !std::getenv("RT_NO_COPY_WARMUP")) {
    (void)hipMemcpyAsync(g->frame[0].h_stage, g->frame[0].d_frame, bytes, hipMemcpyDeviceToHost, g->xstream[0]);
    (void)hipStreamSynchronize(g->xstream[0]);
    (void)hipGetLastError();
  }
}

// Enqueue one frame: G parallel launches, ONE gather, de-interleave, (optionally) ONE device-to-host copy.  Returns as
// soon as everything is in the streams.  At most two frames may be in flight.
int group_submit(RtHipGroup* g, uint8_t* out_rgb8) {
  if (g->n_submitted - g->n_collected >= 2) return fail(RT_ERR_INVALID, "rt_hip_group_submit: two frames are in flight already (collect one first)");
  const uint32_t G = g->G;
  const int b = (int)(g->n_submitted & 1);
  RtHipGroup::Frame& f = g->frame[b];
  f.t0 = std::chrono::steady_clock::now();
  f.out = out_rgb8;
  // a frame submitted while another one is in flight goes through the ranks' OTHER view and render stream: the two kernels overlap
  // (a frame submitted onto an idle group uses the first: blocking frames keep their scene's learned queue order)
  {
    const RtHipGroup::Frame& other = g->frame[b ^ 1];
    f.which = (!g->scene2.empty() && other.busy) ? (other.which ^ 1) : 0;
  }
  for (double& u : f.us) u = 0.0;
  if (G > 1) {
    {
      std::lock_guard<std::mutex> lk(g->mu);
      g->cur = b;
      g->n_done.store(0, std::memory_order_release);
      g->generation.fetch_add(1, std::memory_order_acq_rel);
    }
    g->cv_go.notify_all();
  }
  enqueue_rank(g, f, 0);  // (this thread is awake: rank 0's launch goes out while the other ranks' threads wake up)
  if (G > 1) {  // the other ranks finish within microseconds of this thread: poll briefly, then sleep
    const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(200);
    while (g->n_done.load(std::memory_order_acquire) != G - 1 && std::chrono::steady_clock::now() < until) __builtin_ia32_pause();
    if (g->n_done.load(std::memory_order_acquire) != G - 1) {
      std::unique_lock<std::mutex> lk(g->mu);
      g->cv_done.wait(lk, [&] { return g->n_done.load(std::memory_order_acquire) == G - 1; });
    }
  }
  for (uint32_t r = 0; r < G; ++r) {
    if (g->t_wake_us[r] > f.us[0]) f.us[0] = g->t_wake_us[r];
    if (g->t_enq_us[r] > f.us[1]) f.us[1] = g->t_enq_us[r];
  }
  f.us[2] = us_since(f.t0);
  for (uint32_t r = 0; r < G; ++r)
    if (g->rc[r] != RT_OK) {
      drain(g);
      return fail(g->rc[r], "rank " + std::to_string(r) + ": " + g->err[r]);
    }
  auto assemble = [&]() -> int {
    RT_HIP_TRY(hipSetDevice(g->device[0]));
    hipStream_t x0 = g->xstream[0];
    if (g->gather) {
      if (g->rccl) {  // ONE gather over xGMI: every rank's packed tiles -> rank 0's `stacked`, each rank's side behind its kernel's event
        ncclResult_t nr = ncclSuccess;
#ifdef RT_TEST_PROBES
        if (g->inject == 2 && g->n_submitted == 1) nr = ncclInternalError;  // (test hook RT_RCCL_INJECT=gather: the SECOND frame's gather fails to enqueue)
        else
#endif
        {
          nr = g->api.GroupStart();
          for (uint32_t r = 0; r < G && nr == ncclSuccess; ++r)
            nr = g->api.Gather(f.d_tiles[r], f.d_stacked, g->pad_bytes, ncclUint8, 0, g->comm[r], g->xstream[r]);
          const ncclResult_t ne = g->api.GroupEnd();
          if (nr == ncclSuccess) nr = ne;
        }
        if (nr != ncclSuccess) {
          // the gather did not go out: the frame's kernels are enqueued and their events recorded — send the tiles by peer
          // copies instead (this frame and every later one), each behind its kernel's event like the worker would have
          for (uint32_t r = 0; r < g->comm.size(); ++r) if (g->comm[r]) { (void)hipSetDevice(g->device[r]); if (g->api.CommAbort) (void)g->api.CommAbort(g->comm[r]); }
          g->comm.clear();
          use_peer_transport(g, std::string("ncclGather: ") + g->api.GetErrorString(nr));
          for (uint32_t r = 1; r < G; ++r) {
            RT_HIP_TRY(hipSetDevice(g->device[r]));
            RT_HIP_TRY(hipMemcpyPeerAsync(static_cast<uint8_t*>(f.d_stacked) + (size_t)r * g->pad_bytes, g->device[0], f.d_tiles[r], g->device[r], g->pad_bytes, g->xstream[r]));
            RT_HIP_TRY(hipEventRecord(f.ev_sent[r], g->xstream[r]));
          }
        }
        RT_HIP_TRY(hipSetDevice(g->device[0]));
      }
      if (!g->rccl) {
        for (uint32_t r = 1; r < G; ++r) RT_HIP_TRY(hipStreamWaitEvent(x0, f.ev_sent[r], 0));
      }
      f.us[3] = us_since(f.t0);
      hipLaunchKernelGGL(rtg::deinterleave_rows, dim3(g->height), dim3(256), 0, x0, static_cast<const uint8_t*>(f.d_stacked),
                         static_cast<uint8_t*>(f.d_frame), g->height, (uint32_t)g->row_bytes, G, RT_GROUP_TILE_ROWS, g->pad_rows);
      RT_HIP_TRY(hipGetLastError());
    } else {
      RT_HIP_TRY(hipStreamWaitEvent(x0, f.ev_done[0], 0));
      f.us[3] = us_since(f.t0);
    }
    RT_HIP_TRY(hipEventRecord(f.ev_assembled, x0));
    f.staged = false;
    if (f.out) {
      // An asynchronous copy into PAGEABLE memory is synchronous in HIP (submit blocked until the frame was rendered: 1.8 ms,
      // profiles/r04_run1_group_overhead.json — the two-deep pipeline gone for exactly the call a drop-in host makes): the frame
      // goes into a pinned staging buffer of the group and collect moves it on.  A destination that is pinned itself is written directly.
      const size_t bytes = (size_t)g->height * g->row_bytes;
      uint8_t* dst = f.out;
      static const bool trace = std::getenv("RT_GROUP_TRACE") != nullptr;  // (development: where a submit's host time goes)
      const double t_a = trace ? us_since(f.t0) : 0.0;
      const bool pinned_out = bytes != 0 && out_is_pinned(f.out);
      const double t_b = trace ? us_since(f.t0) : 0.0;
      if (bytes != 0 && !pinned_out) {
        if (!f.h_stage && hipHostMalloc((void**)&f.h_stage, bytes, hipHostMallocDefault) != hipSuccess) { f.h_stage = nullptr; (void)hipGetLastError(); }
        if (f.h_stage) { dst = f.h_stage; f.staged = true; }  // (no pinned memory to be had: the old, blocking copy)
      }
      if (bytes != 0) RT_HIP_TRY(hipMemcpyAsync(dst, f.d_frame, bytes, hipMemcpyDeviceToHost, x0));
      if (trace) std::fprintf(stderr, "[rt group] submit: before pointer query %.1f us, after %.1f, after hipMemcpyAsync %.1f (staged %d)\n", t_a, t_b, us_since(f.t0), (int)f.staged);
    }
    RT_HIP_TRY(hipEventRecord(f.ev_final, x0));
    return RT_OK;
  };
  const int rc = assemble();
  if (rc != RT_OK) { drain(g); return rc; }
  f.us[4] = us_since(f.t0);
  f.busy = true;
  g->n_submitted++;
  return RT_OK;
}

// Wait for the OLDEST frame in flight; its stats.  The frame is then what rt_hip_group_frame points at.
int group_collect(RtHipGroup* g, RtStats* stats) {
  if (g->n_collected == g->n_submitted) return fail(RT_ERR_INVALID, "rt_hip_group_collect: no frame in flight");
  const uint32_t G = g->G;
  RtHipGroup::Frame& f = g->frame[g->n_collected & 1];
  auto finish = [&]() -> int {
    RT_HIP_TRY(hipSetDevice(g->device[0]));
    RT_HIP_TRY(hipEventSynchronize(f.ev_assembled));
    f.us[5] = us_since(f.t0);
    RT_HIP_TRY(hipEventSynchronize(f.ev_final));
    if (f.staged && f.out) std::memcpy(f.out, f.h_stage, (size_t)g->height * g->row_bytes);
    f.us[6] = us_since(f.t0);
    const double frame_ms = f.us[6] * 1e-3;
    RtStats total;
    std::memset(&total, 0, sizeof total);
    for (uint32_t r = 0; r < G; ++r) {  // (with RCCL the other ranks' sides of the gather are behind ev_copied of their launch only on the render stream: their transfer streams are drained by the root's receive)
      RtStats st;
      RtHipScene* fsc = f.which ? g->scene2[r] : g->scene[r];
      const int rc = wait_slot(fsc, f.slot[r], &st);
      if (rc != RT_OK) return rc;
      // overlapped frames: a kernel's start event fires when ITS stream reaches it — while the previous frame's kernel (other
      // stream) still holds most CUs.  The time this frame can be charged with starts at the later of its own start and the
      // previous frame's kernel end: min(start -> stop, previous stop -> stop).  Serial frames: unchanged (start >= previous stop).
      if (f.ev_prev_stop[r] && fsc->slot[f.slot[r] & 1].launched) {
        float since_prev = 0.f;
        if (hipEventElapsedTime(&since_prev, f.ev_prev_stop[r], fsc->slot[f.slot[r] & 1].ev_stop) == hipSuccess) {
          if (since_prev > 0.f && (double)since_prev < st.kernel_ms) st.kernel_ms = (double)since_prev;
        } else (void)hipGetLastError();
      }
      total.samples += st.samples; total.segments += st.segments; total.sphere_tests += st.sphere_tests;
      total.exact_tests += st.exact_tests; total.tex_oob += st.tex_oob; total.grid_steps += st.grid_steps;
      total.segments_repeated = (uint32_t)std::min<uint64_t>((uint64_t)total.segments_repeated + st.segments_repeated, 0xFFFFFFFFull);
      for (int k = 0; k < 4; ++k) total.wave_iters[k] += st.wave_iters[k];
      for (int k = 0; k < 12; ++k) total.prof_cycles[k] += st.prof_cycles[k];
      g->kernel_ms_last[r] = st.kernel_ms;
      if (st.kernel_ms > total.kernel_ms) total.kernel_ms = st.kernel_ms;  // the slowest rank
    }
    f.us[7] = us_since(f.t0);
    if (stats) {
      *stats = total;
      stats->n_gpus_used = G;
      stats->frame_ms = frame_ms;
      const RtHipScene::Slot& s0 = (f.which ? g->scene2[0] : g->scene[0])->slot[f.slot[0] & 1];
      if (s0.launched) {
        RT_HIP_TRY(hipSetDevice(g->device[0]));
        float ms = 0.f;  // device 0's clock: rank 0's kernel start -> frame in scanline order; minus the slowest rank's kernel =
        RT_HIP_TRY(hipEventElapsedTime(&ms, s0.ev_start, f.ev_assembled));  // what the frame spent NOT rendering (start skew, gather, de-interleave)
        if (f.ev_prev_stop[0]) {  // (overlapped frames: the span starts at the previous frame's kernel end at the earliest, like kernel_ms above)
          float ms2 = 0.f;
          if (hipEventElapsedTime(&ms2, f.ev_prev_stop[0], f.ev_assembled) == hipSuccess) { if (ms2 > 0.f && ms2 < ms) ms = ms2; } else (void)hipGetLastError();
        }
        stats->gather_ms = (double)ms > total.kernel_ms ? (double)ms - total.kernel_ms : 0.0;
      }
      for (int k = 0; k < 8; ++k) stats->group_us[k] = f.us[k];
    }
    return RT_OK;
  };
  const int rc = finish();
  if (rc != RT_OK) { drain(g); return rc; }
  f.busy = false;
  g->last = &f;
  g->n_collected++;
  return RT_OK;
}

int group_frame(RtHipGroup* g, uint8_t* out_rgb8, RtStats* stats) {  // blocking: one frame, nothing else in flight
  while (g->n_collected != g->n_submitted) { const int rc = group_collect(g, nullptr); if (rc != RT_OK) return rc; }
  const int rc = group_submit(g, out_rgb8);
  return rc != RT_OK ? rc : group_collect(g, stats);
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
extern "C" int rt_hip_group_submit(RtHipGroup* g, uint8_t* out_rgb8) {
  if (!g) return fail(RT_ERR_INVALID, "null argument");
  return rtg::group_submit(g, out_rgb8);
}
extern "C" int rt_hip_group_collect(RtHipGroup* g, RtStats* stats) {
  if (!g) return fail(RT_ERR_INVALID, "null argument");
  return rtg::group_collect(g, stats);
}
// drop-in for the parallel loop of render() (raytracer.rs:254-263): host scene in, host RGB8 out
extern "C" int rt_render_rgb8(const RtScene* scene, uint8_t* out_rgb8, RtStats* stats) {
  if (!scene || !out_rgb8) return fail(RT_ERR_INVALID, "null argument");
  const auto t0 = std::chrono::steady_clock::now();
  RtHipGroup* g = nullptr;
  int rc = rtg::group_create(scene, 0, &g, true);
  if (rc != RT_OK) return rc;
  // one frame per scene: no later frame could use a queue order learned from this one (tile_order 2 would measure
  // the tile depths and run rt_order_tiles inside frame_ms for nothing) — bottom row first
  (void)rt_hip_group_set_option(g, "tile_order", 1);
  rtp::Clock pc;
  rtg::prepare_staging(g, 1);  // (the caller's buffer is pageable as a rule: its pinned staging buffer — ONE: one frame — is made before the window opens)
  pc.mark("render.pinned_staging_and_copy_warm_up");
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
