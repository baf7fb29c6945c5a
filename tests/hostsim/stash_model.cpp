// Model of the workgroup tile stash of rt_kernel.hip::acquire() on host threads: W "workgroups" of T
// "waves" drain one frame queue (or eight per-XCD queues) through per-workgroup stash words, with the kernel's batch
// taper.  Checked: every queue position is opened exactly once, by exactly one wave, whatever the interleaving; nobody
// retires while tiles remain.  (Test infrastructure: the product protocol is the HIP code; this restates it with
// std::atomic so that thread sanitizers and plain stress can run it on the CPU.)
//   g++ -O2 -std=c++17 -pthread stash_model.cpp -o stash_model && ./stash_model n_tiles workgroups waves batch share queues seed
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

struct Frame {
  uint32_t n_tiles, batch, share, n_queues;
  std::vector<uint32_t> cnt, off;           // per-queue tile counts and offsets into queue order
  std::atomic<uint32_t> queue[8];
  std::vector<std::atomic<uint32_t>> opened;  // per queue position: how often it was opened
  explicit Frame(uint32_t n) : opened(n) {}
};
struct Group {
  std::atomic<unsigned long long> stash{0};  // end << 32 | next
  std::atomic<uint32_t> next_batch{1};
  std::atomic<uint32_t> dry{0};
  std::atomic<uint32_t> frame_empty{0};
};

// one acquire attempt of a wave: >= 0: a queue position; -1: the frame is empty; -2: a batch is on its way, ask again
static long take_tile(Frame& f, Group& g, uint32_t my_xcd, uint32_t n_groups, std::mt19937& rng) {
  if (g.frame_empty.load()) return -1;
  auto jitter = [&]() { if ((rng() & 7u) == 0u) std::this_thread::yield(); };
  const unsigned long long old = g.stash.fetch_add(1ull);
  const uint32_t s_next = (uint32_t)old, s_end = (uint32_t)(old >> 32);
  if (s_next < s_end) return (long)s_next;
  if (s_next != s_end) return -2;
  jitter();
  const uint32_t B = g.next_batch.load();
  uint32_t pos = f.n_tiles, end = 0, rem = 0, share = n_groups * f.share;
  if (f.n_queues == 1) {
    const uint32_t j = f.queue[0].fetch_add(B);
    if (j < f.n_tiles) { pos = j; end = j + B < f.n_tiles ? j + B : f.n_tiles; rem = f.n_tiles - end; }
  } else {
    uint32_t dry = g.dry.load();
    for (uint32_t q = 0; q < 8u && pos == f.n_tiles; ++q) {
      const uint32_t x = (my_xcd + q) & 7u;
      if ((dry >> x) & 1u) continue;
      const uint32_t j = f.queue[x].fetch_add(B);
      if (j >= f.cnt[x]) { dry |= 1u << x; continue; }
      const uint32_t e = j + B < f.cnt[x] ? j + B : f.cnt[x];
      pos = f.off[x] + j; end = f.off[x] + e; rem = f.cnt[x] - e;
    }
    share = (n_groups + 7u) / 8u * f.share;
    g.dry.fetch_or(dry);
  }
  jitter();
  if (pos >= f.n_tiles) { g.frame_empty.store(1); return -1; }
  const uint32_t nb = rem / share;
  g.next_batch.store(nb < 1u ? 1u : (nb > f.batch ? f.batch : nb));
  g.stash.exchange(((unsigned long long)end << 32) | (unsigned long long)(pos + 1u));
  return (long)pos;
}

int main(int argc, char** argv) {
  if (argc < 8) { std::fprintf(stderr, "usage: stash_model n_tiles workgroups waves batch share queues seed\n"); return 2; }
  const uint32_t n_tiles = (uint32_t)std::atoi(argv[1]), W = (uint32_t)std::atoi(argv[2]), T = (uint32_t)std::atoi(argv[3]);
  Frame f(n_tiles);
  f.n_tiles = n_tiles; f.batch = (uint32_t)std::atoi(argv[4]); f.share = (uint32_t)std::atoi(argv[5]);
  f.n_queues = (uint32_t)std::atoi(argv[6]);
  const uint32_t seed = (uint32_t)std::atoi(argv[7]);
  f.cnt.assign(8, 0); f.off.assign(8, 0);
  if (f.n_queues == 8) {  // uneven queues (the last one short, one empty when the frame is tiny)
    uint32_t left = n_tiles;
    for (int x = 0; x < 8; ++x) { const uint32_t c = x < 7 ? (left < n_tiles / 7 ? left : n_tiles / 7) : left; f.cnt[x] = c; left -= c; }
    for (int x = 1; x < 8; ++x) f.off[x] = f.off[x - 1] + f.cnt[x - 1];
  }
  for (auto& q : f.queue) q.store(0);
  for (auto& o : f.opened) o.store(0);
  std::vector<Group> groups(W);
  for (auto& g : groups) g.next_batch.store(f.batch);
  std::atomic<uint32_t> asks{0}, waits{0};
  std::vector<std::thread> th;
  for (uint32_t w = 0; w < W; ++w)
    for (uint32_t t = 0; t < T; ++t)
      th.emplace_back([&, w, t]() {
        std::mt19937 rng(seed * 7919u + w * 131u + t);
        for (;;) {
          const long r = take_tile(f, groups[w], w & 7u, W, rng);
          asks.fetch_add(1);
          if (r == -1) break;
          if (r == -2) { waits.fetch_add(1); std::this_thread::yield(); continue; }
          f.opened[(size_t)r].fetch_add(1);
          if ((rng() & 3u) == 0u) std::this_thread::yield();  // "render the tile"
        }
      });
  for (auto& t : th) t.join();
  uint32_t bad = 0;
  for (uint32_t i = 0; i < n_tiles; ++i) bad += f.opened[i].load() != 1u;
  std::printf("{\"n_tiles\": %u, \"bad\": %u, \"asks\": %u, \"waits\": %u}\n", n_tiles, bad, asks.load(), waits.load());
  return bad ? 1 : 0;
}
