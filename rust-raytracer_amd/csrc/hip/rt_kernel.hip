// rt_kernel.hip — the gfx950 megakernel: render_line + ray_color + hit_world of the reference
// (raytracer/src/raytracer.rs:191-218, 71-165, 44-59) as ONE persistent launch.
//
// Shape (CDNA4-first):
//   * persistent workgroups (one resident set per CU) pull work items from a global queue.
//     Item = (8x8-pixel wave tile, chunk of samples); items are small (a few thousand
//     samples), so the frame has no tail even though paths differ 50x in length;
//   * the scene tables a ray touches per segment — f64 sphere geometry, material cores, the
//     uniform grid's cell words and item lists — are staged ONCE per workgroup into LDS and
//     gathered from there by lane (ds_read_b64), never from HBM;
//   * inside an item the wave's 64 x chunk samples form one pool: a lane that finishes a
//     sample takes the next (pixel, sample) by ballot + prefix rank, so all 64 lanes enter
//     hit_world together every iteration (active-ray compaction without moving any state);
//   * hit_world = `large` spheres (the ground) tested by every lane through scalar loads, then
//     a per-lane 3D-DDA through the uniform grid (f32, conservative) whose cells list the
//     spheres that get the reference's exact f64 Sphere::hit.  The closest hit is the
//     lexicographic minimum of (t, object index), i.e. bit-identical to the reference's
//     object-order scan (rt_core.h exact_hit_any_order);
//   * pixel sums are exact 2^-40 fixed point: in LDS per item, then (when a pixel's samples are
//     split over several items) u64 atomics into an HBM accumulator that a trivial epilogue
//     kernel turns into RGB8.  Order-free, hence bit-reproducible for any schedule / GPU count;
//   * Philox4x32-10 per lane, addressed by (pixel, sample, node, slot).
#include <hip/hip_runtime.h>

#include "rt_core.h"

namespace rtk {
using namespace rtc;

struct KArgs {
  DevScene sc;
  uint8_t* out_rgb8;
  float* out_linear;
  unsigned long long* counters;  // [0] segments, [1] exact tests, [2] tex_oob, [3] grid steps, [4..7] wave trip counts
  unsigned long long* accum;     // [local pixel][3] fixed-point sums; used when n_chunks > 1
  uint32_t* queue;               // work-item cursor (zeroed before the launch)
  uint32_t local_rows, tile_rows, first_tile, tile_stride;
  uint32_t tiles_x, n_tiles, n_chunks, chunk_spp;
};

#ifndef RT_BLOCK
#define RT_BLOCK 1024
#endif

#ifndef RT_WAVES_PER_EU
#define RT_WAVES_ATTR
#else
#define RT_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(RT_WAVES_PER_EU, RT_WAVES_PER_EU)))
#endif

#ifdef RT_PROFILE
#define RT_PROF_DECL unsigned long long prof_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long prof_last = __builtin_readcyclecounter(); const unsigned long long prof_begin = prof_last;
#define RT_PROF(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); prof_t[k] += now_ - prof_last; prof_last = now_; } while (0)
#define RT_PROF_COUNT(c) do { (c)++; } while (0)
#else
#define RT_PROF_COUNT(c) do { } while (0)
#define RT_PROF_DECL
#define RT_PROF(k) do { } while (0)
#endif

constexpr int BLOCK = RT_BLOCK;
constexpr int WAVES = BLOCK / 64;
constexpr int TILE = 8;  // wave tile = 8x8 pixels

// constant-address-space views: a wave-uniform index into these becomes an s_load
typedef const double __attribute__((address_space(4))) * F64PtrK;
typedef const uint32_t __attribute__((address_space(4))) * U32PtrK;

// ---- dynamic LDS layout: [pixel sums: WAVES x 2 item slots x 64 x 3 u64][geom][matc][cell entries][cell items]
// one resident set of workgroups per CU must fit 160 KB of LDS
constexpr uint32_t LDS_TABLES_MAX_BYTES = BLOCK >= 1024 ? 156u * 1024u : (BLOCK >= 512 ? 78u * 1024u : 52u * 1024u);
struct LdsLayout {
  uint32_t geom_off, matc_off, cell_off, item_off, total;
};
__host__ __device__ inline LdsLayout lds_layout(uint32_t n_spheres, uint32_t n_cells, uint32_t n_items, bool tables) {
  LdsLayout l;
  uint32_t o = WAVES * 2u * 64u * 3u * (uint32_t)sizeof(unsigned long long);
  l.geom_off = o; if (tables) o += n_spheres * (uint32_t)sizeof(SphereGeom);
  l.matc_off = o; if (tables) o += n_spheres * (uint32_t)sizeof(MatCore);
  l.cell_off = o; if (tables) o += n_cells * 8u;
  l.item_off = o; if (tables) o += (n_items * 2u + 7u) & ~7u;
  l.total = o;
  return l;
}

// The kernel arguments live in the kernarg segment (constant memory).  Reading them through a
// pointer the compiler cannot see through makes every section of the path loop RE-LOAD the few
// fields it needs (s_load, scalar cache) instead of keeping all ~80 argument SGPRs live across
// the hot walk loop, where they were being spilled to VGPR lanes.
typedef const KArgs __attribute__((address_space(4)))* KArgsK;
__device__ __forceinline__ const KArgs& fresh_args() {
  KArgsK p = (KArgsK)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return *(const KArgs*)p;
}

struct LdsTables {  // per-lane gathers from the workgroup's LDS copies
  const double* g;
  const MatCore* m;
  __device__ __forceinline__ SphereGeom geom(uint32_t i) const {
    const double* p = g + 4u * i;
    SphereGeom r; r.cx = p[0]; r.cy = p[1]; r.cz = p[2]; r.r = p[3];
    return r;
  }
  __device__ __forceinline__ MatCore mat(uint32_t i) const { return m[i]; }
};

template <bool HL, bool SIMPLE, bool LDS_TABLES>
__global__ __launch_bounds__(BLOCK) RT_WAVES_ATTR void rt_megakernel(const KArgs ka) {
  const DevScene& sc = ka.sc;
  const GridDesc& G = sc.grid;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  unsigned long long* const wave_acc = reinterpret_cast<unsigned long long*>(lds_raw) + wave * 384u;  // 2 item slots x 64 px x 3
  const LdsLayout lay = lds_layout(sc.n_spheres, G.n_cells, G.n_items, LDS_TABLES);

  if constexpr (LDS_TABLES) {  // stage the tables once per (persistent) workgroup
    {
      double* dst = reinterpret_cast<double*>(lds_raw + lay.geom_off);
      const double* src = reinterpret_cast<const double*>(sc.geom);
      for (uint32_t i = threadIdx.x; i < sc.n_spheres * 4u; i += BLOCK) dst[i] = src[i];
    }
    {
      double* dst = reinterpret_cast<double*>(lds_raw + lay.matc_off);
      const double* src = reinterpret_cast<const double*>(sc.matc);
      for (uint32_t i = threadIdx.x; i < sc.n_spheres * (uint32_t)(sizeof(MatCore) / 8u); i += BLOCK) dst[i] = src[i];
    }
    {
      uint32_t* dst = reinterpret_cast<uint32_t*>(lds_raw + lay.cell_off);
      for (uint32_t i = threadIdx.x; i < 2u * G.n_cells; i += BLOCK) dst[i] = sc.cell_word[i];
    }
    {
      uint16_t* dst = reinterpret_cast<uint16_t*>(lds_raw + lay.item_off);
      for (uint32_t i = threadIdx.x; i < G.n_items; i += BLOCK) dst[i] = sc.cell_items[i];
    }
    __syncthreads();
  }
  using Tables = typename std::conditional<LDS_TABLES, LdsTables, GlobalTables>::type;
  Tables tb;
  const uint2* cell_word;
  const uint16_t* cell_items;
  if constexpr (LDS_TABLES) {
    tb.g = reinterpret_cast<const double*>(lds_raw + lay.geom_off);
    tb.m = reinterpret_cast<const MatCore*>(lds_raw + lay.matc_off);
    cell_word = reinterpret_cast<const uint2*>(lds_raw + lay.cell_off);
    cell_items = reinterpret_cast<const uint16_t*>(lds_raw + lay.item_off);
  } else {
    tb.g = sc.geom; tb.m = sc.matc;
    cell_word = reinterpret_cast<const uint2*>(sc.cell_word); cell_items = sc.cell_items;
  }

  typedef Lane<HL, SIMPLE> LaneT;
  LaneT L;
  L.s = 0; L.k = 0; L.node = 0; L.in_light = 0;
  L.val[0] = L.val[1] = L.val[2] = 0.0f;
  L.n_segments = L.n_exact = L.n_tex_oob = 0;
  L.ra.pixel = 0; L.ra.sample = 0; L.ra.k0 = sc.seed_lo; L.ra.k1 = sc.seed_hi;
  L.o = v3(0, 0, 0); L.d = v3(0, 0, 1);
  fwd_init(L.fwd);
  if constexpr (HL) L.ls.top = 0;
  uint32_t n_segments = 0, n_exact = 0, n_steps = 0;  // per-lane counters (one exec-masked add each)
  uint32_t cnt_w_iter = 0, cnt_w_step = 0, cnt_w_test = 0, cnt_items = 0;  // wave trip counts (RT_PROFILE builds)

  RT_PROF_DECL
  const uint32_t n_items = ka.n_tiles * ka.n_chunks;  // (one SGPR across the loop)
  auto fetch_item = [&]() -> uint32_t {
    uint32_t v = 0;
    if (lane == 0) v = atomicAdd(fresh_args().queue, 1u);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
  };

  // ---- streaming work items.  A wave holds up to two items at a time: the CURRENT one hands out
  // samples, the PREVIOUS one only waits for its last paths to finish; when the current item has
  // no samples left, lanes go straight on with the next item from the queue, so no lane idles
  // while a neighbour finishes a long path (only at the very end of the frame).  Each item slot
  // has its own set of pixel sums in LDS.
  uint32_t s_bx[2] = {0, 0}, s_by[2] = {0, 0}, s_sbeg[2] = {0, 0}, s_total[2] = {0, 0}, s_next[2] = {0, 0}, s_out[2] = {0, 0};
  bool s_active[2] = {false, false};
  uint32_t cur = 0;
  bool q_empty = false;
  uint32_t py_slot[2] = {0, 0};   // per lane: global scanline of this lane's pixel slot in item slot k
  bool ok_slot[2] = {false, false};  // per lane: that pixel slot is inside the image
  uint32_t my_slot = 0, cur_p = lane;
  bool has_ray = false;

  auto open_item = [&](uint32_t k, uint32_t item) {
    const KArgs& ka = fresh_args();
    const DevScene& sc = ka.sc;
    // chunk-major order: the last items of the frame are spread over the whole image
    const uint32_t chunk = item / ka.n_tiles, tile = item - chunk * ka.n_tiles;
    const uint32_t by = tile / ka.tiles_x, bx = tile - by * ka.tiles_x;
    const uint32_t px = bx * TILE + (lane & 7u);
    const uint32_t lr = by * TILE + (lane >> 3);  // local (packed) row
    uint32_t py = lr;  // global scanline (raytracer.rs:255: band index, 0 = top)
    if (ka.tile_rows != 0u) py = (ka.first_tile + (lr / ka.tile_rows) * ka.tile_stride) * ka.tile_rows + lr % ka.tile_rows;
    const uint32_t s_begin = chunk * ka.chunk_spp;
    const uint32_t s_count = sc.spp - s_begin < ka.chunk_spp ? sc.spp - s_begin : ka.chunk_spp;
    s_bx[k] = bx; s_by[k] = by; s_sbeg[k] = s_begin; s_next[k] = 0; s_out[k] = 0; s_active[k] = true;
    // pool item w = (pixel slot w & 63, sample s_begin + (w >> 6)); max_depth == 0: ray_color
    // returns black before tracing anything (raytracer.rs:80-82), so nothing is handed out
    s_total[k] = sc.max_depth != 0u ? 64u * s_count : 0u;
    py_slot[k] = py; ok_slot[k] = px < sc.width && lr < ka.local_rows;
    unsigned long long* acc = wave_acc + k * 192u;
    acc[lane * 3u] = 0ull; acc[lane * 3u + 1u] = 0ull; acc[lane * 3u + 2u] = 0ull;
    RT_PROF_COUNT(cnt_items);
  };
  auto flush_item = [&](uint32_t k) {  // all samples of slot k are in its pixel sums: write them out
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const KArgs& ka = fresh_args();
    const DevScene& sc = ka.sc;
    const uint32_t px = s_bx[k] * TILE + (lane & 7u), lr = s_by[k] * TILE + (lane >> 3);
    const unsigned long long* acc = wave_acc + k * 192u;
    if (px < sc.width && lr < ka.local_rows) {
      const size_t o = ((size_t)lr * sc.width + px) * 3;
      if (ka.n_chunks == 1u) {  // raytracer.rs:207-216: mean, sqrt gamma, f32 -> u8, store
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float lin = fixed_to_mean(acc[lane * 3u + c], sc.spp);
          if (ka.out_linear) ka.out_linear[o + c] = lin;
          ka.out_rgb8[o + c] = f32_to_u8(__builtin_sqrtf(lin));
        }
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const unsigned long long v = acc[lane * 3u + c];
          if (v) atomicAdd(&ka.accum[o + c], v);
        }
      }
    }
    s_active[k] = false;
  };

  RT_PROF(5);
  for (;;) {
    // ------------------------------------------------------------ refill lanes that hold no path
    {
      const DevScene& sc = fresh_args().sc;
      bool want = !has_ray;
      for (;;) {
        const unsigned long long m = __ballot(want);
        if (!m) break;
        if (s_active[cur] && s_next[cur] < s_total[cur]) {  // hand out samples of the current item
          const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
          const uint32_t w = s_next[cur] + rank;
          const uint32_t left = s_total[cur] - s_next[cur], asked = (uint32_t)__builtin_popcountll(m);
          s_next[cur] += asked < left ? asked : left;
          const uint32_t p = w & 63u;
          const uint32_t p_py = (uint32_t)__shfl((int)py_slot[cur], (int)p);
          const int p_ok = __shfl((int)ok_slot[cur], (int)p);
          bool took = false;
          if (want && w < s_total[cur] && p_ok) {  // (a slot outside the image consumes its index and asks again)
            const uint32_t p_px = s_bx[cur] * TILE + (p & 7u);
            cur_p = p; my_slot = cur; L.s = s_sbeg[cur] + (w >> 6); L.ra.pixel = p_py * sc.width + p_px;
            lane_begin_sample(sc, L, p_px, p_py);
            has_ray = true; want = false; took = true;
          }
          s_out[cur] += (uint32_t)__builtin_popcountll(__ballot(took));
          continue;
        }
        // the current slot has nothing (more) to hand out
        if (s_active[cur] && s_out[cur] == 0u) flush_item(cur);  // and nothing in flight: done with it
        if (s_active[cur]) {             // it still drains: open the next item in the other slot
          if (s_active[cur ^ 1u]) break;  // both slots busy: these lanes wait
          cur ^= 1u;
        }
        if (q_empty) break;
        const uint32_t item = fetch_item();
        if (item >= n_items) { q_empty = true; break; }
        open_item(cur, item);
      }
    }
    if (!__any(has_ray)) break;  // nothing in flight, nothing left to hand out
    RT_PROF_COUNT(cnt_w_iter);
    RT_PROF(0);
    {
      // ---------------------------------------------------------- hit_world (raytracer.rs:44-59)
      const DevScene& sc = fresh_args().sc;
      const GridDesc& G = sc.grid;
      const F64PtrK geom_k = (F64PtrK)(uintptr_t)sc.geom;
      const U32PtrK large_k = (U32PtrK)(uintptr_t)sc.large;
      const uint32_t n_large = G.n_large;
      const bool has_grid = G.n[0] != 0u;
      const RayK rk = ray_consts(L.d);
      double closest = T_MAX;
      int best = -1;
      if (has_ray) n_segments++;
      // (1) spheres outside the grid: every lane tests them; the record is wave-uniform -> SGPRs
      for (uint32_t i = 0; i < n_large; ++i) {
        const uint32_t idx = large_k[i];
        const F64PtrK gp = geom_k + (size_t)idx * 4u;
        SphereGeom g; g.cx = gp[0]; g.cy = gp[1]; g.cz = gp[2]; g.r = gp[3];
        if (has_ray && rk.fast) exact_hit_any_order_t<true>(L.o, L.d, rk, g, idx, closest, best);
      }
      if (has_ray && rk.fast) n_exact += n_large;
      RT_PROF(1);
      // (2) enter the grid
      GridWalk w;
      const int mode = !has_ray ? GRID_MISS : (!rk.fast ? GRID_FALLBACK : (has_grid ? grid_begin(G, L.o, L.d, w) : GRID_MISS));
      if (__any(mode == GRID_FALLBACK)) {  // numerically unsafe ray: the reference's full scan, real divisions
        for (uint32_t idx = 0; idx < sc.n_spheres; ++idx) {
          const F64PtrK gp = geom_k + (size_t)idx * 4u;
          SphereGeom g; g.cx = gp[0]; g.cy = gp[1]; g.cz = gp[2]; g.r = gp[3];
          if (mode == GRID_FALLBACK) exact_hit_slow(L.o, L.d, rk.a, g, idx, closest, best);
        }
        if (mode == GRID_FALLBACK) n_exact += sc.n_spheres;
      }
      if (has_grid) {
        // (3) walk rounds: every walking lane moves on by up to two cells and/or tests one sphere.
        // Per-lane walk state: tm = GridWalk.tmax, dt = GridWalk.delta, dl = GridWalk.dl, lin;
        // the current cell's untested spheres are items [it, end), the next two of them also in `pend`.
        bool walking = mode == GRID_WALK;
        float tm0 = w.tmax[0], tm1 = w.tmax[1], tm2 = w.tmax[2];
        const float dt0 = w.delta[0], dt1 = w.delta[1], dt2 = w.delta[2];
        const int dl0 = w.dl[0], dl1 = w.dl[1], dl2 = w.dl[2];
        int lin = walking ? w.lin : 0;
        const double t0 = w.t0;
        uint32_t it = 0, end = 0, pend = 0xFFFFFFFFu, last = 0xFFFFFFFFu;
        const int lin_max = (int)G.n_cells - 1;
        if (walking) {
          const uint2 e = cell_word[lin];
          it = e.x & CELL_START_MASK; end = it + (e.x >> CELL_COUNT_SHIFT); pend = e.y;
        }
        for (;;) {
          if (!__any(walking)) break;
          // (a) lanes whose cell is exhausted: finished, or on to the next non-empty cell.  The next
          // TWO cells along the ray are computed and fetched together (one LDS round trip), the
          // second one is used only if the first is empty.
          const bool moving = walking && it == end;
          if (__any(moving)) {
            RT_PROF_COUNT(cnt_w_step);
            if (moving) {
              float tc = (float)(closest - t0);
              tc = tc + fabsf(tc) * 2.384185791015625e-07f;  // grid_done
              const bool hit = best >= 0;
              const float tminA = rt_min3f(tm0, tm1, tm2);
              if (hit && tc < tminA) walking = false;
              else {
                // grid_step x 2
                const bool ax = tm0 == tminA, ay = !ax && tm1 == tminA, az = !ax && !ay;
                const float a0 = tm0 + (ax ? dt0 : 0.0f), a1 = tm1 + (ay ? dt1 : 0.0f), a2 = tm2 + (az ? dt2 : 0.0f);
                const int linA = lin + (ax ? dl0 : (ay ? dl1 : dl2));
                const float tminB = rt_min3f(a0, a1, a2);
                const bool bx_ = a0 == tminB, by_ = !bx_ && a1 == tminB, bz_ = !bx_ && !by_;
                const float b0 = a0 + (bx_ ? dt0 : 0.0f), b1 = a1 + (by_ ? dt1 : 0.0f), b2 = a2 + (bz_ ? dt2 : 0.0f);
                int linB = linA + (bx_ ? dl0 : (by_ ? dl1 : dl2));
                linB = linB < 0 ? 0 : (linB > lin_max ? lin_max : linB);  // speculative address: keep it inside the table
                const uint2 eA = cell_word[linA];
                const uint2 eB = cell_word[linB];
                n_steps++;
                const bool exitA = eA.x == CELL_EXIT, emptyA = (eA.x >> CELL_COUNT_SHIFT) == 0u;
                const bool doneA = hit && tc < tminB;  // the closest hit lies inside cell A
                if (exitA || !emptyA || doneA) {  // stay in A (or stop there)
                  tm0 = a0; tm1 = a1; tm2 = a2; lin = linA;
                  it = eA.x & CELL_START_MASK; end = it + (eA.x >> CELL_COUNT_SHIFT); pend = eA.y;
                  if (exitA || emptyA) { walking = false; end = it; }
                } else {                           // A is empty: on to B
                  n_steps++;
                  tm0 = b0; tm1 = b1; tm2 = b2; lin = linB;
                  it = eB.x & CELL_START_MASK; end = it + (eB.x >> CELL_COUNT_SHIFT); pend = eB.y;
                  if (eB.x == CELL_EXIT) { walking = false; end = it; }
                }
              }
            }
          }
          const bool testing = walking && it != end;
          if (__any(testing)) {  // (b) one exact Sphere::hit per lane standing in a cell with spheres left
            RT_PROF_COUNT(cnt_w_test);
            if (testing) {
              uint32_t idx = pend & 0xFFFFu;
              if (idx == 0xFFFFu) idx = cell_items[it];  // third and later items of a cell: from the list
              pend = (pend >> 16) | 0xFFFF0000u;
              it++;
              if (idx != last) {  // a sphere spanning consecutive cells is not re-tested
                last = idx; n_exact++;
                exact_hit_any_order_t<true>(L.o, L.d, rk, tb.geom(idx), idx, closest, best);
              }
            }
          }
        }
      }
      RT_PROF(3);

      // ---------------------------------------------------------- ray_color body
      bool finished = false;
      if (has_ray) {
        finished = lane_shade(fresh_args().sc, tb, L, best, closest);
        if (finished) {  // sample finished: add it to its pixel (raytracer.rs:203-205)
          unsigned long long* acc = wave_acc + my_slot * 192u + cur_p * 3u;
          atomicAdd(&acc[0], sample_to_fixed(L.val[0]));
          atomicAdd(&acc[1], sample_to_fixed(L.val[1]));
          atomicAdd(&acc[2], sample_to_fixed(L.val[2]));
          has_ray = false;
        }
      }
      const unsigned long long mf = __ballot(finished);
      if (mf) {
#pragma unroll
        for (uint32_t k = 0; k < 2u; ++k) {
          s_out[k] -= (uint32_t)__builtin_popcountll(__ballot(finished && my_slot == k));
          if (s_active[k] && s_out[k] == 0u && s_next[k] >= s_total[k]) flush_item(k);
        }
      }
      RT_PROF(4);
    }
  }

  // counters: wave reduction, one atomic per wave
  unsigned long long c0 = n_segments, c1 = n_exact, c2 = L.n_tex_oob, c3 = n_steps;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    c0 += __shfl_down(c0, off); c1 += __shfl_down(c1, off); c2 += __shfl_down(c2, off); c3 += __shfl_down(c3, off);
  }
  if (lane == 0) {
    const KArgs& ka = fresh_args();
    atomicAdd(&ka.counters[0], c0);
    atomicAdd(&ka.counters[1], c1);
    if (c2) atomicAdd(&ka.counters[2], c2);
    atomicAdd(&ka.counters[3], c3);
    atomicAdd(&ka.counters[4], (unsigned long long)cnt_w_iter);
    atomicAdd(&ka.counters[5], (unsigned long long)cnt_w_step);
    atomicAdd(&ka.counters[6], (unsigned long long)cnt_w_test);
    atomicAdd(&ka.counters[7], (unsigned long long)cnt_items);
#ifdef RT_PROFILE
    RT_PROF(5);
    prof_t[6] = prof_last - prof_begin;
    for (int k = 0; k < 7; ++k) atomicAdd(&ka.counters[8 + k], prof_t[k]);
#endif
  }
}

// Epilogue when a pixel's samples were split over several work items: fixed-point sums ->
// mean, sqrt gamma, RGB8 (raytracer.rs:207-216).  One thread per pixel channel triple.
__global__ void rt_resolve(const unsigned long long* accum, uint8_t* out_rgb8, float* out_linear, uint32_t n_pixels, uint32_t spp) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pixels) return;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float lin = fixed_to_mean(accum[(size_t)i * 3 + k], spp);
    if (out_linear) out_linear[(size_t)i * 3 + k] = lin;
    out_rgb8[(size_t)i * 3 + k] = f32_to_u8(__builtin_sqrtf(lin));
  }
}

// --------------------------------------------------------------------------- device self-test
// f64 sqrt / divide / f32 sqrt must be correctly rounded on the GPU for bit-parity with the CPU
// oracle; tests/test_gpu_parity.py checks these against numpy.
__global__ void rt_math_probe(const double* x, const double* y, double* out_sqrt, double* out_div, float* out_sqrtf,
                              double* out_atan2, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out_sqrt[i] = sqrt(x[i]);
  out_div[i] = x[i] / y[i];
  out_sqrtf[i] = __builtin_sqrtf((float)x[i]);
  out_atan2[i] = atan2(x[i] - 0.5, y[i] - 0.5);
}

}  // namespace rtk
