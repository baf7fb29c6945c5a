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
// the PNG of frame i is encoded on a host thread while the GPU renders frame i+1.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/rt_abi.h"

namespace {
int animate(RtSceneFile* sf, const char* prefix, int frames, double orbit_deg) {
  RtScene* sc = rt_scene_get_mut(sf);
  RtHipGroup* hs = nullptr;  // the scene resident on RT_GPUS devices (default 1)
  int rc = rt_hip_group_create(sc, 0, &hs);
  if (rc != RT_OK) { std::fprintf(stderr, "render failed: %s: %s\n", rt_strerror(rc), rt_hip_last_error()); return 101; }
  double cam[11];
  rt_scene_camera(sf, cam);
  const double *lf = cam, *la = cam + 3, *up = cam + 6;
  double k[3], kl = std::sqrt(up[0] * up[0] + up[1] * up[1] + up[2] * up[2]);
  for (int i = 0; i < 3; ++i) k[i] = up[i] / kl;
  const size_t bytes = (size_t)sc->width * sc->height * 3;
  std::vector<uint8_t> buf[2] = {std::vector<uint8_t>(bytes), std::vector<uint8_t>(bytes)};
  std::thread writer;
  int write_rc = RT_OK;
  int status = 0;
  for (int f = 0; f < frames; ++f) {
    // Rodrigues rotation of (look_from - look_at) about vup
    const double th = orbit_deg * f * (3.14159265358979323846264338327950288 / 180.0), c = std::cos(th), s = std::sin(th);
    const double v[3] = {lf[0] - la[0], lf[1] - la[1], lf[2] - la[2]};
    const double kv = k[0] * v[0] + k[1] * v[1] + k[2] * v[2];
    const double kx[3] = {k[1] * v[2] - k[2] * v[1], k[2] * v[0] - k[0] * v[2], k[0] * v[1] - k[1] * v[0]};
    double from[3], out[13];
    for (int i = 0; i < 3; ++i) from[i] = la[i] + v[i] * c + kx[i] * s + k[i] * kv * (1.0 - c);
    rt_camera_derive(from, la, up, cam[9], cam[10], out);
    rt_hip_group_set_camera(hs, out, out + 3, out + 6, out + 9);
    char name[4096];
    std::snprintf(name, sizeof name, "%s_%03d.png", prefix, f);  // main.rs:17
    std::printf("\nRendering %s\n", name);
    RtStats st{};
    rc = rt_hip_group_render_to_host(hs, buf[f & 1].data(), &st);
    if (rc != RT_OK) { std::fprintf(stderr, "render failed: %s: %s\n", rt_strerror(rc), rt_hip_last_error()); status = 101; break; }
    std::printf("Frame time: %lldms\n", (long long)st.frame_ms);
    if (writer.joinable()) writer.join();
    if (write_rc != RT_OK) break;
    const std::string fname(name);
    const uint8_t* px = buf[f & 1].data();
    const uint32_t w = sc->width, h = sc->height;
    writer = std::thread([fname, px, w, h, &write_rc]() { write_rc = rt_png_write_rgb8(fname.c_str(), px, w, h); });
  }
  if (writer.joinable()) writer.join();
  rt_hip_group_destroy(hs);
  if (write_rc != RT_OK) { std::fprintf(stderr, "error writing image: %s\n", rt_host_last_error()); status = 101; }
  return status;
}
}  // namespace

int main(int argc, char** argv) {
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
  RtSceneFile* sf = nullptr;
  int rc = rt_scene_load_file(argv[1], &sf);
  if (rc != RT_OK) {
    std::fprintf(stderr, "%s\n", rt_host_last_error());
    return 101;
  }
  RtScene* sc = rt_scene_get_mut(sf);
  if (const char* seed = std::getenv("RT_SEED")) sc->seed = std::strtoull(seed, nullptr, 0);
  if (frames > 0) {
    const int status = animate(sf, argv[2], frames, orbit_given ? orbit : 360.0 / frames);
    rt_scene_free(sf);
    return status;
  }
  const char* filename = argv[2];
  std::printf("\nRendering %s\n", filename);  // main.rs:18
  std::vector<uint8_t> pixels((size_t)sc->width * sc->height * 3);  // raytracer.rs:254
  RtStats st{};
  rc = rt_render_rgb8(sc, pixels.data(), &st);
  if (rc != RT_OK) {
    std::fprintf(stderr, "render failed: %s: %s\n", rt_strerror(rc), rt_hip_last_error());
    rt_scene_free(sf);
    return 101;
  }
  std::printf("Frame time: %lldms\n", (long long)st.frame_ms);  // raytracer.rs:263
  if (std::getenv("RT_STATS"))
    std::fprintf(stderr, "{\"samples\":%llu,\"segments\":%llu,\"sphere_tests\":%llu,\"exact_tests\":%llu,\"n_gpus\":%u,\"kernel_ms\":%.3f,"
                         "\"gather_ms\":%.3f,\"frame_ms\":%.3f,\"setup_ms\":%.3f,\"msamples_per_s\":%.3f}\n",
                 (unsigned long long)st.samples, (unsigned long long)st.segments, (unsigned long long)st.sphere_tests,
                 (unsigned long long)st.exact_tests, st.n_gpus_used, st.kernel_ms, st.gather_ms, st.frame_ms, st.setup_ms,
                 st.samples / (st.kernel_ms * 1e3));
  rc = rt_png_write_rgb8(filename, pixels.data(), sc->width, sc->height);  // raytracer.rs:265
  rt_scene_free(sf);
  if (rc != RT_OK) {
    std::fprintf(stderr, "error writing image: %s\n", rt_host_last_error());
    return 101;
  }
  return 0;
}
