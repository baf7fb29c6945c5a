// main.cpp — `raytracer <config_file> <output_file>`: the reference CLI (main.rs:7-20) with
// the rayon loop of render() (raytracer.rs:250-266) replaced by one rt_render_rgb8() call
// into librt_hip.so.  Same argv, same two stdout lines.  Where the reference panics
// (unreadable / unparsable config, texture or PNG failure) this prints the same message to
// stderr and exits 101, the exit status of a Rust panic.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../../include/rt_abi.h"

int main(int argc, char** argv) {
  if (argc != 3) {  // main.rs:9-12: usage line, normal return
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
    std::fprintf(stderr, "{\"samples\":%llu,\"segments\":%llu,\"sphere_tests\":%llu,\"exact_tests\":%llu,\"kernel_ms\":%.3f,\"frame_ms\":%.3f,\"msamples_per_s\":%.3f}\n",
                 (unsigned long long)st.samples, (unsigned long long)st.segments, (unsigned long long)st.sphere_tests,
                 (unsigned long long)st.exact_tests, st.kernel_ms, st.frame_ms, st.samples / (st.kernel_ms * 1e3));
  rc = rt_png_write_rgb8(filename, pixels.data(), sc->width, sc->height);  // raytracer.rs:265
  rt_scene_free(sf);
  if (rc != RT_OK) {
    std::fprintf(stderr, "error writing image: %s\n", rt_host_last_error());
    return 101;
  }
  return 0;
}
