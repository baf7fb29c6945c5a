/* Offline search behind tests' LIGHT_DRAW_OPEN_SEEDS (test infrastructure; links the CPU oracle's Philox).
 *
 * The light-sampling draw of a hit that is not Glass (raytracer.rs:100) is u01_53(low = slot 0 word 2, high = slot 1 word 3)
 * (oracle/rt_oracle.c, "RNG addressing").  The HIP kernel decides `draw > 1 - n_lights * 0.1` from the high word alone and fetches
 * the low word only when the high word leaves the comparison open — one high word in 2^32, which no rendered scene ever meets.
 * This program finds seeds for which pixel 0, sample 0, node 0 (the first camera-path hit of the first pixel) has exactly that
 * high word for one light (threshold 0.9), so that a test can put a Lambertian sphere under pixel 0 and compare kernel and oracle
 * on the path that needs the low word.  For threshold t: draw > t  <=>  U >= K with U = high * 2^21 + (low >> 11) and
 * K = t * 2^53 + 1 (an integer for t in [0.5, 1)); open high word = K >> 21.
 *
 *   gcc -O3 -fopenmp tests/golden/find_light_draw_seeds.c -Loracle -lrt_oracle -Wl,-rpath,$PWD/oracle -o /tmp/find_seeds && /tmp/find_seeds 4
 * ~10 s per seed found on 8 cores (2^32 Philox calls expected each). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
void rt_oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

int main(int argc, char** argv) {
  const int want = argc > 1 ? atoi(argv[1]) : 4;
  const double t = 1.0 - 1.0 * 0.1;
  const uint64_t K = (uint64_t)(t * 9007199254740992.0) + 1u;
  const uint32_t open_high = (uint32_t)(K >> 21), k_low = (uint32_t)(K & 0x1FFFFFu);
  printf("threshold %.17g  K %llu  open high word %u  low-part bound %u\n", t, (unsigned long long)K, open_high, k_low);
  int found = 0;
  for (uint64_t base = 1; found < want; base += 1ull << 28) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)(1ull << 28); ++i) {
      const uint64_t seed = base + (uint64_t)i;
      const uint32_t ctr[4] = {0u, 0u, 0u, 1u}, key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
      uint32_t w[4];
      rt_oracle_philox4x32_10(ctr, key, w);
      if (w[3] == open_high) {
        const uint32_t c0[4] = {0u, 0u, 0u, 0u};
        uint32_t l[4];
        rt_oracle_philox4x32_10(c0, key, l);
#pragma omp critical
        {
          printf("seed %llu  low word %u  low part %u  -> %s\n", (unsigned long long)seed, l[2], l[2] >> 11, (l[2] >> 11) >= k_low ? "samples the lights" : "does not");
          fflush(stdout);
          ++found;
        }
      }
    }
  }
  return 0;
}
