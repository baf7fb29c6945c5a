// main.cpp — `raytracer <config_file> <output_file>`: the reference CLI (main.rs:7-20) with
// the rayon loop of render() (raytracer.rs:250-266) replaced by one rt_render_rgb8() call
// into librt_hip.so.  Same argv, same two stdout lines.  Where the reference panics
// (unreadable / unparsable config, texture or PNG failure) this prints the same message to
// stderr and exits 101, the exit status of a Rust panic.
//
// `Frame time` is the window the reference times (raytracer.rs:259-263: the parallel loop until the pixels are in the
// host buffer) — HIP start-up, table build and scene upload happen before it and are reported under RT_STATS=1.
// RT_GPUS=N (or "n_gpus" in RtScene) shards the frame over N GPUs inside librt_hip.so (rt_hip_group_*).
//
// Superset (SURVEY §8f "animation driver"): `raytracer <config_file> <output_prefix> --frames N
// [--orbit DEG]` renders N frames to `<output_prefix>_%03d.png` — the file naming main.rs:17
// keeps commented out and README.md:43-57 / "Make animation" feed to ffmpeg — turning the camera
// around look_at by DEG per frame (default 360/N).  The scene is uploaded once and stays in HBM;
// the PNG of frame i is encoded on writer threads (RT_ANIM_WRITERS, default 4) while the GPU renders frame i+1.  With RT_GPUS=G every frame is sharded
// over the G devices and frames are pipelined two deep (RT_ANIM=sharded, default), or the frames are distributed over the
// devices, each rendering whole frames (RT_ANIM=frames); RT_STATS=1 prints frames per second and the per-frame kernel / frame / PNG
// times to stderr (one JSON line).
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/rt_abi.h"

namespace {
// camera of frame f: look_from turned about vup around look_at by orbit_deg * f (Rodrigues), then camera.rs:45-77
void orbit_camera(const double cam[11], double orbit_deg, int f, double out[13]) {
  const double *lf = cam, *la = cam + 3, *up = cam + 6;
  double k[3];
  const double kl = std::sqrt(up[0] * up[0] + up[1] * up[1] + up[2] * up[2]);
  for (int i = 0; i < 3; ++i) k[i] = up[i] / kl;
  const double th = orbit_deg * f * (3.14159265358979323846264338327950288 / 180.0), c = std::cos(th), s = std::sin(th);
  const double v[3] = {lf[0] - la[0], lf[1] - la[1], lf[2] - la[2]};
  const double kv = k[0] * v[0] + k[1] * v[1] + k[2] * v[2];
  const double kx[3] = {k[1] * v[2] - k[2] * v[1], k[2] * v[0] - k[0] * v[2], k[0] * v[1] - k[1] * v[0]};
  double from[3];
  for (int i = 0; i < 3; ++i) from[i] = la[i] + v[i] * c + kx[i] * s + k[i] * kv * (1.0 - c);
  rt_camera_derive(from, la, up, cam[9], cam[10], out);
}
std::string frame_name(const char* prefix, int f) {
  char name[4096];
  std::snprintf(name, sizeof name, "%s_%03d.png", prefix, f);  // main.rs:17
  return name;
}
double ms_between(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }

// What an animation run reports under RT_STATS=1 (one JSON line on stderr): frames per second disk to disk, and per frame the
// kernel (slowest rank), the frame as the reference times it, and its PNG — which of the two bounds the run is then on the line.
struct AnimStats {
  std::mutex mu;
  std::vector<double> kernel_ms, frame_ms, png_ms;
  double host_us[4] = {0, 0, 0, 0};   // the submitting thread's time, summed over the frames: waiting for a free buffer, camera + submit, collect, stdout + hand-over
  explicit AnimStats(int n) : kernel_ms(n, 0.0), frame_ms(n, 0.0), png_ms(n, 0.0) {}
  void report(const char* mode, int frames, unsigned gpus, unsigned writers, double wall_s, double setup_ms) {
    if (!std::getenv("RT_STATS")) return;
    std::string s;
    char buf[256];
    std::snprintf(buf, sizeof buf, "{\"animation\":\"%s\",\"frames\":%d,\"n_gpus\":%u,\"png_writers\":%u,\"setup_ms\":%.3f,\"wall_s\":%.4f,\"frames_per_s\":%.3f", mode, frames, gpus,
                  writers, setup_ms, wall_s, frames / wall_s);
    s = buf;
    auto arr = [&](const char* key, const std::vector<double>& v) {
      s += std::string(",\"") + key + "\":[";
      for (int i = 0; i < frames && i < (int)v.size(); ++i) { std::snprintf(buf, sizeof buf, "%s%.3f", i ? "," : "", v[i]); s += buf; }
      s += "]";
    };
    arr("kernel_ms", kernel_ms); arr("frame_ms", frame_ms); arr("png_ms", png_ms);
    std::snprintf(buf, sizeof buf, ",\"host_us_per_frame\":{\"wait_for_buffer\":%.1f,\"camera_and_submit\":%.1f,\"collect\":%.1f,\"stdout_and_hand_over\":%.1f}",
                  host_us[0] / frames, host_us[1] / frames, host_us[2] / frames, host_us[3] / frames);
    s += buf;
    s += "}\n";
    std::fputs(s.c_str(), stderr);
  }
};

// PNG writers of an animation: W threads (RT_ANIM_WRITERS, default 4) take finished frames off a queue; a frame's host buffer
// goes back to the free list when its file is on disk.  A frame's PNG is itself deflated in parallel bands (scene.cpp), so one
// writer keeps up with the headline frame; the second one is for frames whose kernel is shorter than their PNG (the
// reference's 1 ms test scene).
struct PngWriters {
  struct Job { int frame; uint8_t* px; };
  std::mutex mu;
  std::condition_variable cv_job, cv_free;
  std::vector<Job> jobs;            // FIFO (few entries)
  std::vector<uint8_t*> free_bufs;
  std::vector<std::thread> th;
  bool quit = false;
  int write_rc = RT_OK;
  std::string write_err;
  PngWriters(unsigned W, const char* prefix, uint32_t w, uint32_t h, AnimStats* stats) {
    for (unsigned i = 0; i < W; ++i)
      th.emplace_back([this, prefix, w, h, stats]() {
        for (;;) {
          Job j;
          {
            std::unique_lock<std::mutex> lk(mu);
            cv_job.wait(lk, [&] { return quit || !jobs.empty(); });
            if (jobs.empty()) return;
            j = jobs.front(); jobs.erase(jobs.begin());
          }
          const auto t0 = std::chrono::steady_clock::now();
          const int rc = rt_png_write_rgb8(frame_name(prefix, j.frame).c_str(), j.px, w, h);
          const double ms = ms_between(t0, std::chrono::steady_clock::now());
          { std::lock_guard<std::mutex> lk(stats->mu); if (j.frame < (int)stats->png_ms.size()) stats->png_ms[j.frame] = ms; }
          std::lock_guard<std::mutex> lk(mu);
          if (rc != RT_OK && write_rc == RT_OK) { write_rc = rc; write_err = rt_host_last_error(); }
          free_bufs.push_back(j.px);
          cv_free.notify_one();
        }
      });
  }
  uint8_t* take_buffer() {  // blocks while every buffer is in flight or waiting for its PNG
    std::unique_lock<std::mutex> lk(mu);
    cv_free.wait(lk, [&] { return !free_bufs.empty(); });
    uint8_t* b = free_bufs.back(); free_bufs.pop_back();
    return b;
  }
  void give_buffer(uint8_t* b) { std::lock_guard<std::mutex> lk(mu); free_bufs.push_back(b); }
  void push(int frame, uint8_t* px) { { std::lock_guard<std::mutex> lk(mu); jobs.push_back({frame, px}); } cv_job.notify_one(); }
  bool failed() { std::lock_guard<std::mutex> lk(mu); return write_rc != RT_OK; }
  void finish() {
    { std::lock_guard<std::mutex> lk(mu); quit = true; }
    cv_job.notify_all();
    for (auto& t : th) t.join();
    th.clear();
  }
};
// Several frames' PNGs are written at once, each by a FEW band threads: four writers x 8 threads keep up with the reference's 0.9 ms
// test-scene frames (one writer x 32 threads: 1.2 ms per PNG = slower than the kernel; three writers x 32 threads each: slower
// still, the threads of three PNGs at once — profiles/r06_run1_anim_writers.log).  RT_ANIM_WRITERS / RT_PNG_THREADS override.
unsigned anim_writers() {
  const char* e = std::getenv("RT_ANIM_WRITERS");
  const long v = e ? std::strtol(e, nullptr, 10) : 4;
  const unsigned w = v < 1 ? 1u : (v > 16 ? 16u : (unsigned)v);
  if (w > 1) setenv("RT_PNG_THREADS", "8", 0);
  return w;
}

// Every frame SHARDED over the RT_GPUS devices (rt_hip_group_*), frames pipelined two deep: frame f+1 is submitted before
// frame f is collected, so f's gather + de-interleave + device-to-host copy run under f+1's kernels, and f's PNG is encoded
// on the writer threads meanwhile.  2 + W host buffers: two frames in flight + one per writer.
int animate_sharded(RtSceneFile* sf, const char* prefix, int frames, double orbit_deg) {
  RtScene* sc = rt_scene_get_mut(sf);
  RtHipGroup* hs = nullptr;  // the scene resident on RT_GPUS devices (default 1)
  const auto t_create = std::chrono::steady_clock::now();
  int rc = rt_hip_group_create(sc, 0, &hs);
  if (rc != RT_OK) { std::fprintf(stderr, "render failed: %s: %s\n", rt_strerror(rc), rt_hip_last_error()); return 101; }
  (void)rt_hip_group_set_option(hs, "prepare_host_output", 2);  // (pinned staging for the two frames in flight + the copy path, at set-up: not inside the first submit)
  const auto t_begin = std::chrono::steady_clock::now();
  double cam[11];
  rt_scene_camera(sf, cam);
  const size_t bytes = (size_t)sc->width * sc->height * 3;
  const unsigned W = anim_writers();
  std::vector<std::vector<uint8_t>> store(2 + W, std::vector<uint8_t>(bytes));
  AnimStats stats(frames);
  PngWriters writers(W, prefix, sc->width, sc->height, &stats);
  for (auto& b : store) writers.give_buffer(b.data());
  int status = 0;
  std::vector<uint8_t*> in_flight;  // oldest first
  auto finish = [&](int f) -> bool {  // collect frame f, print its two lines, hand its pixels to the PNG writers
    RtStats st{};
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = rt_hip_group_collect(hs, &st);
    const auto t1 = std::chrono::steady_clock::now();
    stats.host_us[2] += ms_between(t0, t1) * 1e3;
    if (rc != RT_OK) { std::fprintf(stderr, "render failed: %s: %s\n", rt_strerror(rc), rt_hip_last_error()); status = 101; return false; }
    std::printf("\nRendering %s\nFrame time: %lldms\n", frame_name(prefix, f).c_str(), (long long)st.frame_ms);
    stats.kernel_ms[f] = st.kernel_ms; stats.frame_ms[f] = st.frame_ms;
    uint8_t* px = in_flight.front();
    in_flight.erase(in_flight.begin());
    writers.push(f, px);
    stats.host_us[3] += ms_between(t1, std::chrono::steady_clock::now()) * 1e3;
    return !writers.failed();
  };
  int submitted = 0, collected = 0;
  for (int f = 0; f < frames && status == 0 && !writers.failed(); ++f) {
    double out[13];
    const auto t0 = std::chrono::steady_clock::now();
    uint8_t* buf = writers.take_buffer();
    const auto t1 = std::chrono::steady_clock::now();
    orbit_camera(cam, orbit_deg, f, out);
    rt_hip_group_set_camera(hs, out, out + 3, out + 6, out + 9);
    rc = rt_hip_group_submit(hs, buf);
    stats.host_us[0] += ms_between(t0, t1) * 1e3;
    stats.host_us[1] += ms_between(t1, std::chrono::steady_clock::now()) * 1e3;
    if (rc != RT_OK) { std::fprintf(stderr, "render failed: %s: %s\n", rt_strerror(rc), rt_hip_last_error()); status = 101; break; }
    in_flight.push_back(buf);
    submitted++;
    if (submitted - collected == 2) { if (!finish(collected)) break; collected++; }
  }
  while (status == 0 && !writers.failed() && collected < submitted) { if (!finish(collected)) break; collected++; }
  writers.finish();
  stats.report("sharded", collected, rt_hip_group_size(hs), W, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(), ms_between(t_create, t_begin));
  rt_hip_group_destroy(hs);
  if (writers.write_rc != RT_OK) { std::fprintf(stderr, "error writing image: %s\n", writers.write_err.c_str()); status = 101; }
  return status;
}

// RT_ANIM=frames: the frames DISTRIBUTED over the devices — device g renders whole frames g, g + G, ... on its own resident
// scene and host thread (README.md:43-57 renders an animation one process per frame; this is that, with the scene loaded
// once per device).  No gather, no shard penalty, no per-frame synchronisation between devices; every frame is the bytes the
// sharded mode produces (Philox is addressed by pixel).  Per device the PNG of a frame is encoded while the next one renders.
int animate_frames(RtSceneFile* sf, const char* prefix, int frames, double orbit_deg, unsigned G) {
  RtScene* sc = rt_scene_get_mut(sf);
  const int ndev = rt_hip_device_count();
  const char* emu = std::getenv("RT_GPUS_EMULATE");
  if (ndev <= 0) { std::fprintf(stderr, "render failed: %s\n", rt_strerror(RT_ERR_NO_DEVICE)); return 101; }
  if (G > (unsigned)ndev && !(emu && emu[0] == '1')) {
    std::fprintf(stderr, "render failed: RT_GPUS = %u but only %d device(s) visible\n", G, ndev);
    return 101;
  }
  double cam[11];
  rt_scene_camera(sf, cam);
  const size_t bytes = (size_t)sc->width * sc->height * 3;
  const uint32_t w = sc->width, h = sc->height;
  std::mutex out_mu;
  std::vector<int> status(G, 0);
  AnimStats stats(frames);
  const auto t_begin = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (unsigned g = 0; g < G; ++g)
    th.emplace_back([&, g]() {
      RtHipScene* hs = nullptr;
      int rc = rt_hip_scene_create(sc, (int)(g % (unsigned)ndev), &hs);
      if (rc != RT_OK) { std::lock_guard<std::mutex> lk(out_mu); std::fprintf(stderr, "render failed: %s: %s\n", rt_strerror(rc), rt_hip_last_error()); status[g] = 101; return; }
      std::vector<uint8_t> buf[2] = {std::vector<uint8_t>(bytes), std::vector<uint8_t>(bytes)};
      std::thread writer;
      int write_rc = RT_OK, i = 0;
      for (int f = (int)g; f < frames && write_rc == RT_OK; f += (int)G, ++i) {
        double out[13];
        orbit_camera(cam, orbit_deg, f, out);
        rt_hip_set_camera(hs, out, out + 3, out + 6, out + 9);
        RtStats st{};
        rc = rt_hip_render_to_host(hs, buf[i & 1].data(), &st);
        const std::string fname = frame_name(prefix, f);
        {
          std::lock_guard<std::mutex> lk(out_mu);
          if (rc != RT_OK) { std::fprintf(stderr, "render failed: %s: %s\n", rt_strerror(rc), rt_hip_last_error()); status[g] = 101; }
          else std::printf("\nRendering %s\nFrame time: %lldms\n", fname.c_str(), (long long)st.frame_ms);
        }
        if (rc != RT_OK) break;
        stats.kernel_ms[f] = st.kernel_ms; stats.frame_ms[f] = st.frame_ms;  // (distinct elements per thread)
        if (writer.joinable()) writer.join();
        const uint8_t* px = buf[i & 1].data();
        writer = std::thread([fname, px, w, h, f, &write_rc, &stats]() {
          const auto t0 = std::chrono::steady_clock::now();
          write_rc = rt_png_write_rgb8(fname.c_str(), px, w, h);
          stats.png_ms[f] = ms_between(t0, std::chrono::steady_clock::now());
        });
      }
      if (writer.joinable()) writer.join();
      if (write_rc != RT_OK) { std::lock_guard<std::mutex> lk(out_mu); std::fprintf(stderr, "error writing image: %s\n", rt_host_last_error()); status[g] = 101; }
      rt_hip_scene_destroy(hs);
    });
  for (auto& t : th) t.join();
  stats.report("frames", frames, G, 1, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(), 0.0);
  for (int s : status) if (s) return s;
  return 0;
}

int animate(RtSceneFile* sf, const char* prefix, int frames, double orbit_deg) {
  const char* mode = std::getenv("RT_ANIM");
  if (mode && !std::strcmp(mode, "frames")) {
    const RtScene* sc = rt_scene_get(sf);
    unsigned G = sc->n_gpus;
    if (G == 0) { const char* e = std::getenv("RT_GPUS"); G = e ? (unsigned)std::strtoul(e, nullptr, 10) : 1u; }
    if (G < 1 || G > 1024) { std::fprintf(stderr, "render failed: RT_GPUS must be a positive device count\n"); return 101; }
    return animate_frames(sf, prefix, frames, orbit_deg, G);
  }
  if (mode && std::strcmp(mode, "sharded")) { std::fprintf(stderr, "RT_ANIM must be sharded (default) or frames\n"); return 101; }
  return animate_sharded(sf, prefix, frames, orbit_deg);
}
}  // namespace

// The HIP runtime takes 50 - 200 ms to come up (its first call) — as long as the reference's three 2 - 3 MP JPEG textures take to
// decode.  A thread makes that first call while the main thread reads and parses the scene (at 13 ms a frame the host
// pipeline IS the wall time).
std::thread g_hip_init;
double g_hip_init_ms = 0.0;

int run(int argc, char** argv) {
  const auto t_main = std::chrono::steady_clock::now();
  int frames = 0;
  double orbit = 0.0;
  bool orbit_given = false, bad_args = argc < 3;
  for (int i = 3; i < argc && !bad_args; ++i) {
    if (!std::strcmp(argv[i], "--frames") && i + 1 < argc) frames = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--orbit") && i + 1 < argc) { orbit = std::atof(argv[++i]); orbit_given = true; }
    else bad_args = true;
  }
  if (bad_args || (argc > 3 && frames <= 0)) {  // main.rs:9-12: usage line, normal return
    std::printf("Usage: %s <config_file> <output_file>\n", argv[0]);
    return 0;
  }
  // One frame per process (the reference's way, main.rs:7-20): the runtime's copy engines are hardware queues it creates at
  // their FIRST use — 7.8 ms for the first host-to-device copy, 7.8 ms for the first device-to-host copy on MI355X
  // (tools/microbench/setup_costs.hip) — to move 80 KB of tables in and 2.9 MB of pixels out once.  With HSA_ENABLE_SDMA=0 the
  // runtime copies with kernels on the compute queue that exists anyway (same copy times at these sizes, measured).  Only here:
  // an animation keeps the engines — its copies run UNDER the next frame's kernel, which leaves a copy kernel no registers.
  // A value the user set is left alone.
  if (frames == 0) setenv("HSA_ENABLE_SDMA", "0", 0);
  g_hip_init = std::thread([]() {
    const auto t0 = std::chrono::steady_clock::now();
    const int n = rt_hip_device_count();
    // ... and the context + code object of every device the run will use (RT_GPUS, default 1): rt_hip_scene_create finds them up
    const char* e = std::getenv("RT_GPUS");
    long want = e ? std::strtol(e, nullptr, 10) : 1;
    if (want < 1) want = 1;
    std::vector<std::thread> per_device;  // (side by side: a context takes ~30 ms each)
    for (int d = 1; d < n && d < want; ++d) per_device.emplace_back([d]() { (void)rt_hip_device_warm(d); });
    if (n > 0) (void)rt_hip_device_warm(0);
    for (auto& t : per_device) t.join();
    g_hip_init_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  });
  auto ms_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
  RtSceneFile* sf = nullptr;
  int rc = rt_scene_load_file(argv[1], &sf);
  if (rc != RT_OK) {
    std::fprintf(stderr, "%s\n", rt_host_last_error());
    return 101;
  }
  RtScene* sc = rt_scene_get_mut(sf);
  if (const char* seed = std::getenv("RT_SEED")) sc->seed = std::strtoull(seed, nullptr, 0);
  if (frames > 0) {
    if (g_hip_init.joinable()) g_hip_init.join();
    const int status = animate(sf, argv[2], frames, orbit_given ? orbit : 360.0 / frames);
    rt_scene_free(sf);
    return status;
  }
  const char* filename = argv[2];
  const double load_done_ms = ms_since(t_main);
  std::printf("\nRendering %s\n", filename);  // main.rs:18
  std::vector<uint8_t> pixels((size_t)sc->width * sc->height * 3);  // raytracer.rs:254
  RtStats st{};
  const auto t_hip = std::chrono::steady_clock::now();
  if (g_hip_init.joinable()) g_hip_init.join();  // (what of the runtime's start-up the load did not cover)
  const double hip_wait_ms = ms_since(t_hip), hip_init_ms = g_hip_init_ms;
  rc = rt_render_rgb8(sc, pixels.data(), &st);
  if (rc != RT_OK) {
    std::fprintf(stderr, "render failed: %s: %s\n", rt_strerror(rc), rt_hip_last_error());
    rt_scene_free(sf);
    return 101;
  }
  std::printf("Frame time: %lldms\n", (long long)st.frame_ms);  // raytracer.rs:263
  const auto t_png = std::chrono::steady_clock::now();
  rc = rt_png_write_rgb8(filename, pixels.data(), sc->width, sc->height);  // raytracer.rs:265
  const double png_ms = ms_since(t_png);
  if (std::getenv("RT_STATS")) {  // where a drop-in user's wall time goes: main() entered -> PNG on disk
    double lt[4] = {0, 0, 0, 0};
    rt_scene_load_timings(sf, lt);
    std::fprintf(stderr, "{\"samples\":%llu,\"segments\":%llu,\"sphere_tests\":%llu,\"exact_tests\":%llu,\"n_gpus\":%u,\"kernel_ms\":%.3f,"
                         "\"gather_ms\":%.3f,\"frame_ms\":%.3f,\"setup_ms\":%.3f,\"msamples_per_s\":%.3f,"
                         "\"load_ms\":%.3f,\"read_ms\":%.3f,\"json_ms\":%.3f,\"jpeg_ms\":%.3f,\"hip_init_ms\":%.3f,\"hip_wait_ms\":%.3f,\"png_ms\":%.3f,\"main_ms\":%.3f,"
                         "\"group_us\":[%.1f,%.1f,%.1f,%.1f,%.1f,%.1f,%.1f,%.1f],\"setup_profile\":%s}\n",
                 (unsigned long long)st.samples, (unsigned long long)st.segments, (unsigned long long)st.sphere_tests,
                 (unsigned long long)st.exact_tests, st.n_gpus_used, st.kernel_ms, st.gather_ms, st.frame_ms, st.setup_ms,
                 st.samples / (st.kernel_ms * 1e3), load_done_ms, lt[0], lt[1], lt[2], hip_init_ms, hip_wait_ms, png_ms, ms_since(t_main),
                 st.group_us[0], st.group_us[1], st.group_us[2], st.group_us[3], st.group_us[4], st.group_us[5], st.group_us[6], st.group_us[7],
                 rt_hip_setup_profile());
  }
  rt_scene_free(sf);
  if (rc != RT_OK) {
    std::fprintf(stderr, "error writing image: %s\n", rt_host_last_error());
    return 101;
  }
  return 0;
}

// Everything the process owes the world is on disk or in the pipe when run() returns: stdout / stderr are flushed and the process
// ends WITHOUT the HIP runtime's tear-down (static destructors: tens of milliseconds the reference's binary does not have).
// RT_FAST_EXIT=0 takes the ordinary way out.
int main(int argc, char** argv) {
  const int status = run(argc, argv);
  if (g_hip_init.joinable()) g_hip_init.join();
  std::fflush(stdout);
  std::fflush(stderr);
  const char* fe = std::getenv("RT_FAST_EXIT");
  if (!(fe && fe[0] == '0')) std::_Exit(status);
  return status;
}
