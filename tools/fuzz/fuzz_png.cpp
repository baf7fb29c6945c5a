// fuzz_png.cpp — the banded PNG writer (scene.cpp: rt_png_write_rgb8) under ASan / UBSan: frame shapes around the band and thread
// boundaries (one row, one column, fewer rows than bands, more bands than 8 per thread ...) x deflate strategies x thread caps;
// every file is inflated again here (zlib) and compared with the Sub-filtered scanlines.  tests/test_sanitizers.py builds and runs it.
#include <zlib.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
extern "C" int rt_png_write_rgb8(const char* path, const uint8_t* rgb8, uint32_t w, uint32_t h);
extern "C" const char* rt_host_last_error(void);
static uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }
int main(int argc, char** argv) {
  const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1u;
  const char* path = argc > 2 ? argv[2] : "/tmp/fuzz_png.png";
  std::mt19937 g(seed);
  const uint32_t shapes[][2] = {{1, 1}, {1, 9}, {7, 1}, {53, 37}, {421, 150}, {1200, 17}, {3, 4000}, {16384, 3}, {640, 360}};
  int n_ok = 0;
  for (const char* strat : {"rle", "huffman", "default"})
    for (const char* thr : {"1", "3", "32"})
      for (auto& s : shapes) {
        setenv("RT_PNG_DEFLATE", strat, 1); setenv("RT_PNG_THREADS", thr, 1);
        const uint32_t w = s[0], h = s[1];
        std::vector<uint8_t> img(size_t(w) * h * 3);
        const int kind = g() % 3;   // noise | flat | gradient + noise
        for (size_t i = 0; i < img.size(); ++i) img[i] = kind == 0 ? uint8_t(g()) : (kind == 1 ? 77 : uint8_t((i / 3) % w * 255 / w + g() % 5));
        if (rt_png_write_rgb8(path, img.data(), w, h) != 0) { fprintf(stderr, "write failed: %s\n", rt_host_last_error()); return 1; }
        FILE* f = fopen(path, "rb"); fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
        std::vector<uint8_t> file(n); if (fread(file.data(), 1, n, f) != size_t(n)) return 1; fclose(f);
        std::vector<uint8_t> z; size_t p = 8; bool iend = false;
        while (p + 12 <= file.size()) {
          const uint32_t len = be32(&file[p]);
          if (p + 12 + len > file.size()) { fprintf(stderr, "chunk runs past the file\n"); return 1; }
          if (uint32_t(crc32(0L, &file[p + 4], 4 + len)) != be32(&file[p + 8 + len])) { fprintf(stderr, "bad chunk CRC\n"); return 1; }
          if (!memcmp(&file[p + 4], "IDAT", 4)) z.insert(z.end(), &file[p + 8], &file[p + 8] + len);
          if (!memcmp(&file[p + 4], "IEND", 4)) iend = true;
          p += 12 + len;
        }
        if (!iend || p != file.size()) { fprintf(stderr, "no IEND / trailing bytes\n"); return 1; }
        const size_t stride = size_t(w) * 3;
        std::vector<uint8_t> raw((stride + 1) * h); uLongf rl = raw.size();
        if (uncompress(raw.data(), &rl, z.data(), z.size()) != Z_OK || rl != raw.size()) { fprintf(stderr, "inflate failed (%s, %s threads, %ux%u)\n", strat, thr, w, h); return 1; }
        for (uint32_t y = 0; y < h; ++y) {
          if (raw[(stride + 1) * y] != 1) { fprintf(stderr, "filter byte\n"); return 1; }
          for (size_t i = 0; i < stride; ++i) {
            const uint8_t want = uint8_t(img[stride * y + i] - (i >= 3 ? img[stride * y + i - 3] : 0));
            if (raw[(stride + 1) * y + 1 + i] != want) { fprintf(stderr, "pixel mismatch at row %u byte %zu (%s, %s threads, %ux%u)\n", y, i, strat, thr, w, h); return 1; }
          }
        }
        n_ok++;
      }
  printf("ok %d files\n", n_ok);
  return 0;
}
