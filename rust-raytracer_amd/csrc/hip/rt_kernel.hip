// rt_kernel.hip — the gfx950 megakernel: render_line + ray_color + hit_world of the reference
// (raytracer/src/raytracer.rs:191-218, 71-165, 44-59) as ONE persistent launch.
//
// Shape (CDNA4-first):
//   * persistent workgroups (one per CU, 16 waves) pull pixel TILES (8x8 ... 1x1, by frame size)
//     from a global queue; a tile's samples are handed out to the workgroup's waves in small
//     chunks through shared LDS slots and counted per tile, so tiles leave no tail of their own
//     although paths differ 50x in length (what a frame ends on is the latency of its deepest
//     paths), and a pixel's samples meet in LDS: the only HBM traffic of a frame is
//     the framebuffer write; launch counters are summed per workgroup in LDS before they go out;
//   * the scene tables a ray touches per segment — f64 sphere geometry, material cores, the
//     uniform grid's cell words and item lists — are staged ONCE per workgroup into LDS and
//     gathered from there by lane (ds_read_b64), never from HBM;
//   * the samples of a wave's current chunk item form a pool: a lane that finishes a sample
//     takes the next (pixel, sample) by ballot + prefix rank — from the next item if this one ran
//     dry — so all 64 lanes enter hit_world together every iteration (active-ray compaction
//     without moving any state);
//   * hit_world = `large` spheres (the ground) tested by every lane through scalar loads, then
//     a per-lane 3D-DDA through the uniform grid (f32, conservative) whose cells list the
//     spheres that get the reference's exact f64 Sphere::hit.  The closest hit is the
//     lexicographic minimum of (t, object index), i.e. bit-identical to the reference's
//     object-order scan (rt_core.h exact_hit_any_order);
//   * pixel sums are exact 2^-40 fixed point (u64 LDS atomics): order-free, hence
//     bit-reproducible for any schedule / chunking / GPU count;
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
  uint32_t* queue;               // tile cursor (zeroed before the launch)
  uint32_t local_rows, tile_rows, first_tile, tile_stride;
  uint32_t tiles_x, n_tiles, n_chunks, chunk_spp;  // a tile's samples are handed out in n_chunks chunks
  uint32_t tile_log2;  // a tile has 4^tile_log2 pixels (64, 16, 4 or 1) ...
  uint32_t tile_wl, tile_hl;  // ... 2^tile_wl wide, 2^tile_hl high (tile_wl + tile_hl == 2 * tile_log2): square (8x8 ... 1x1) by
                              // default, or 64x1, 16x1, 4x1, 1x1 — one tile is one contiguous run of framebuffer bytes
  uint32_t t_slots;    // tile slots per workgroup (tile_slots() of tile_log2)
  uint32_t tile_batch, batch_share;  // tiles a workgroup takes from the queue per atomic (at most), and the taper: rem / (workgroups * batch_share)
  // Queue order.  The frame ends on its deepest paths (50 sequential segments of a lone wave), so the tiles
  // that breed them should leave the queue FIRST: position i of the queue is tile tile_order[i] (null: n_tiles-1-i, bottom
  // of the image first — the sky rows last).  tile_depth[tile] receives the deepest camera path seen in the tile; the
  // next frame's order is sorted by it (rt_order_tiles below).
  const uint32_t* tile_order;
  uint32_t* tile_depth;
  uint32_t order_mode;  // 0: top row first (the round-1 order), 1: reversed / tile_order
  // XCD affinity of the queue (aff_group_log2 != 0xFFFFFFFF).  Each XCD has its own L2, and a framebuffer line (128 B
  // = ten 4-pixel tile rows) written piecemeal by workgroups of different XCDs goes to HBM as partial sectors from every
  // one of them: 3.6 x the framebuffer's bytes.  So runs of 2^aff_group_log2 consecutive tiles (1.5 KB of a scanline)
  // belong to XCD (run mod 8): a workgroup takes tiles from the queue of the XCD it runs on (queue[x], xcd_cnt[x] tiles;
  // position j of it = tile_order[xcd_off[x] + j], or the j-th tile of XCD x by arithmetic) and from the other seven only
  // when its own is empty.  The lines of a run then collect all their bytes in one L2 before they are written back.
  uint32_t aff_group_log2;
  uint32_t xcd_cnt[8], xcd_off[8];
};
// tiles of XCD x in image order: the j-th one (aff_group_log2 = gl)
__host__ __device__ inline uint32_t xcd_tile(uint32_t x, uint32_t j, uint32_t gl) { return ((((j >> gl) << 3) + x) << gl) + (j & ((1u << gl) - 1u)); }
__host__ __device__ inline uint32_t tile_xcd(uint32_t tile, uint32_t gl) { return (tile >> gl) & 7u; }

#ifndef RT_BLOCK
#define RT_BLOCK 1024
#endif

#ifndef RT_WAVES_PER_EU
#define RT_WAVES_ATTR
#else
#define RT_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(RT_WAVES_PER_EU, RT_WAVES_PER_EU)))
#endif

#ifdef RT_PROFILE
#define RT_PROF_DECL unsigned long long prof_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long prof_last = __builtin_readcyclecounter(); const unsigned long long prof_begin = prof_last; const unsigned long long prof_wall0 = wall_clock64(); unsigned long long prof_wall_qdone = 0; uint32_t prof_tail_iters = 0, prof_tail_lanes = 0; bool prof_in_tail = false; unsigned long long prof_tail_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define RT_PROF_RAW(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); prof_t[k] += now_ - prof_last; if (prof_in_tail) prof_tail_t[k] += now_ - prof_last; prof_last = now_; } while (0)
#ifdef RT_PROF_LIT
// the lit kernels' CENSUS build (round 6): the six section slots are re-used — 0: everything outside lane_shade,
// 1: lane_shade up to its decision (surface, scatter, light trigger), 2: the ACT_SAMPLE continuation (pool takes, frame record,
// aim), 3: the ACT_RETURN continuation (accumulate, next light / compose + release + step), 4: the ACT_FINISH / ACT_CONTINUE tail,
// 5: lanes that took ACT_SAMPLE | lanes that took ACT_RETURN << 32; wave_iters[1] / [2]: wave iterations in which SOME lane took them
#define RT_PROF(k) RT_PROF_RAW(0)
#else
#define RT_PROF(k) RT_PROF_RAW(k)
#endif
#define RT_PROF_COUNT(c) do { (c)++; } while (0)
#else
#define RT_PROF_COUNT(c) do { } while (0)
#define RT_PROF_DECL
#define RT_PROF(k) do { } while (0)
#endif

constexpr int BLOCK = RT_BLOCK;
constexpr int WAVES = BLOCK / 64;
constexpr int TILE_MAX = 8;  // largest pixel tile = 8x8 (one pixel per lane); smaller tiles for small frames

// constant-address-space views: a wave-uniform index into these becomes an s_load
typedef const double __attribute__((address_space(4))) * F64PtrK;
typedef const uint32_t __attribute__((address_space(4))) * U32PtrK;

// ---- dynamic LDS layout: [flags][tile slots: headers, then pixel sums][coop exchange][geom][matc][cell entries][cell items]
// one resident set of workgroups per CU must fit 160 KB of LDS
#ifdef RT_LDS_TABLES_MAX
constexpr uint32_t LDS_TABLES_MAX_BYTES = RT_LDS_TABLES_MAX;  // (occupancy experiments)
#else
constexpr uint32_t LDS_TABLES_MAX_BYTES = BLOCK >= 1024 ? 156u * 1024u : (BLOCK >= 512 ? 78u * 1024u : 52u * 1024u);
#endif
// Tile slots are shared by the workgroup.  A slot holds one open tile: header + the exact
// fixed-point sums of its pixels.  It is freed when ALL samples of the tile have been added —
// counted per tile, whichever waves traced them — so a long path only keeps its own tile's
// slot busy; small tiles get proportionally more slots out of the same LDS budget.
struct SlotHdr {
  uint32_t tile_xy;   // tile column | tile row << 16
  uint32_t next;      // next chunk to hand out (may overshoot n_chunks)
  uint32_t finished;  // samples of this tile added to the pixel sums so far
  uint32_t expected;  // valid pixels of the tile x samples_per_pixel
  uint32_t state;     // SLOT_*
  uint32_t max_depth; // deepest camera path among this tile's samples so far (feeds KArgs.tile_depth)
  unsigned long long nan_mask[3];  // per channel: pixel slots of this tile that received a NaN sample (sample_is_nan)
};
static_assert(sizeof(SlotHdr) == 48, "slot header is 48 B");
constexpr uint32_t SLOT_HDR_BYTES = (uint32_t)sizeof(SlotHdr);
enum { SLOT_FREE = 0, SLOT_OPEN = 1, SLOT_OPENING = 2 };
constexpr uint32_t LDS_FLAGS_BYTES = 32u + 32u * 8u;          // {queue_empty, hint, waves retired, dry queues}, the tile stash (u64) and its next batch size, pad; then the workgroup's 32 launch counters
constexpr uint32_t LDS_SLOT_BUDGET = WAVES * 3u * 1024u;     // 48 KB at 16 waves (3 KB per wave)
constexpr uint32_t T_SLOTS_MAX = 512u;
__host__ __device__ inline uint32_t tile_slots(uint32_t tile_log2) {
  const uint32_t per_slot = SLOT_HDR_BYTES + (1u << (2u * tile_log2)) * 24u;
  const uint32_t t = LDS_SLOT_BUDGET / per_slot;
  return t > T_SLOTS_MAX ? T_SLOTS_MAX : t;
}
struct LdsLayout {
  uint32_t hdr_off, coop_off, light_off, park_off, geom_off, matc_off, cell_off, item_off, total;
};
// light records (lit scenes): [frame bitmap][base bitmap][pool of base_slots colour-map bases][pool of frame_slots LightParked] (rt_core.h)
__host__ __device__ constexpr uint32_t park_bytes(uint32_t frame_slots, uint32_t base_slots) {
  return 2u * LIGHT_POOL_BITMAP_BYTES + frame_slots * (uint32_t)sizeof(LightParked) + base_slots * LIGHT_BASE_BYTES;
}
constexpr uint32_t LIGHT_CENTRES_LDS_OFF = LDS_FLAGS_BYTES + LDS_SLOT_BUDGET + WAVES * 64u * 16u;  // lit scenes: centres of the first 32 lights, 3 doubles each
static_assert(LIGHT_CENTRES_LDS_OFF + LIGHT_CENTRES_LDS_MAX * 24u == LIGHT_POOL_LDS_OFF, "rt_core.h LIGHT_POOL_LDS_OFF = park_off of the layout below (for this RT_BLOCK)");
__host__ __device__ inline LdsLayout lds_layout(uint32_t n_spheres, uint32_t n_cells, uint32_t n_items, bool tables, bool lights, uint32_t frame_slots = 0,
                                                uint32_t base_slots = 0) {
  LdsLayout l;
  uint32_t o = LDS_FLAGS_BYTES;
  l.hdr_off = o; o += LDS_SLOT_BUDGET;  // [t_slots headers][t_slots x npx x 3 u64 sums], sized by tile_slots()
  l.coop_off = o; o += WAVES * 64u * 16u;  // per wave: 64 x 16 B exchange slots of coop_random_in_unit_sphere
  l.light_off = o; if (lights) o += LIGHT_CENTRES_LDS_MAX * 24u;
  l.park_off = o; if (lights) o += (park_bytes(frame_slots, base_slots) + 15u) & ~15u;  // rt_core.h: the pools of light frames and colour-map bases
  l.geom_off = o; if (tables) o += n_spheres * (uint32_t)sizeof(SphereGeom);
  l.matc_off = o; if (tables) o += n_spheres * (uint32_t)sizeof(MatCore);
  l.cell_off = o; if (tables) o += n_cells * 8u;
  l.item_off = o; if (tables) o += (n_items * 2u + 7u) & ~7u;
  l.total = o;
  return l;
}

// Wave votes on a bool: the i1 form of the ballot is a plain lane-mask AND (HIP's __ballot / __any
// take an int and go through a VGPR 0/1 and a compare again — two VALU instructions and a
// VALU->SALU dependency per vote, three votes per walk round).
__device__ __forceinline__ unsigned long long wave_ballot(bool b) { return __builtin_amdgcn_ballot_w64(b); }
__device__ __forceinline__ bool wave_any(bool b) { return __builtin_amdgcn_ballot_w64(b) != 0ull; }

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

// random_in_unit_sphere (point3d.rs:31-38) for a whole wave.  Per lane the rejection loop takes
// 1.9 attempts on average, but a wave runs max-over-lanes (~6) rounds of it: measured 18 % of the
// frame.  Here every lane makes attempt 0 for itself; after that ALL 64 lanes (also those that
// need no point) work for the lanes still failing: failing lane #r posts its RNG address in LDS
// slot r, helper lane h makes attempt base + h / nf of failing lane #(h % nf), accepted helpers
// post their three words, and each failing lane takes those of its lowest-numbered accepted
// attempt.  Attempt a is Philox slot 1+a whoever computes it, so the result is bit-identical to
// the sequential loop.  `xch`: this wave's 64 x uint4 exchange slots.
// A/B arms that were measured and NOT adopted are not kept in this file: their code is in the history, their numbers under
// profiles/ — unfused refill (r02_run3), stash spin (r03_run28), start-cell prefix (r03_run34), dedupe on landing (r03_run36),
// direct hand-out (r02_run12), single settle (r03_run14), carried walks (r02_run7), cull on landing (r04_run6, with the patch),
// light migration (r04_run5, with the patch).

#ifndef RT_DEEP_PATH
#define RT_DEEP_PATH 2u  // camera paths at least this many segments long mark their tile (SlotHdr::max_depth); 2 / 3 / 4 / 6 / 8 / 16: 13.96 / 13.98 / 13.98 / 14.01 / 14.05 / 14.5 ms (profiles/r02_run10_ab.log)
#endif
#ifndef RT_COOP_LAYERS
#define RT_COOP_LAYERS 4u  // attempts a failing lane gets per helper round, at most (64 / failing lanes, capped)
#endif
// Lanes whose hit is Glass need no point but one Philox call of their own (slot 0, the reflectance
// draw of materials.rs:189): they make it in round 0, in the instruction stream the others use for
// attempt 0, and get its first two words back in `glass_u`.
// Lanes that just took a new sample (`fresh`; they hold no hit) make the Philox call of its camera jitter (node
// NODE_CAMERA, slot 0, raytracer.rs:199-200) in the same stream and get its four words back in `cam_w`.
// ceil(65536 / n) for n = 1 .. 64: (lane * m) >> 16 == lane / n for every lane < 64 (checked for all 4096 pairs by the generator)
__device__ __constant__ const uint32_t kCeil65536Over[65] = {
    0, 65536, 32768, 21846, 16384, 13108, 10923, 9363, 8192, 7282, 6554, 5958, 5462, 5042, 4682, 4370, 4096, 3856, 3641, 3450, 3277, 3121,
    2979, 2850, 2731, 2622, 2521, 2428, 2341, 2260, 2185, 2115, 2048, 1986, 1928, 1873, 1821, 1772, 1725, 1681, 1639, 1599, 1561, 1525,
    1490, 1457, 1425, 1395, 1366, 1338, 1311, 1286, 1261, 1237, 1214, 1192, 1171, 1150, 1130, 1111, 1093, 1075, 1058, 1041, 1024};
// Glass lanes also get the light-sampling draw of their node back (same Philox call, words z w: raytracer.rs:100) in `glass_lu`;
// every other lane's attempt-0 call leaves its fourth word in `cam_w.w`: the high word of ITS light-sampling draw.
__device__ __forceinline__ V3 coop_random_in_unit_sphere(bool need, bool glass, bool fresh, const RngAddr& ra, uint32_t node, uint32_t lane,
                                                         uint4* xch, double& glass_u, double& glass_lu, U4& cam_w) {
  auto point = [](uint32_t x, uint32_t y, uint32_t z) { return v3(range_m1_1(x), range_m1_1(y), range_m1_1(z)); };
  uint32_t wx = 0, wy = 0, wz = 0, ww = 0;
  bool pending = false;
  if (need || glass || fresh) {
    const U4 w = rng(ra, fresh ? NODE_CAMERA : node, (glass || fresh) ? 0u : 1u);
    wx = w.x; wy = w.y; wz = w.z; ww = w.w;
    pending = need && !(length_squared(point(wx, wy, wz)) < 1.0);
  }
  cam_w.x = wx; cam_w.y = wy; cam_w.z = wz; cam_w.w = ww;
  glass_u = u01_53(wx, wy);
  glass_lu = u01_53(wz, ww);
  uint32_t base = 1;  // next attempt of every lane still pending (wave-uniform)
  for (;;) {
    const unsigned long long F = wave_ballot(pending);
    if (!F) break;
    const uint32_t nf = (uint32_t)__builtin_popcountll(F);
    // Everything that depends on nf alone is wave-uniform and stays on the scalar unit: the compiler turned `64 / nf` and
    // `ceil(65536 / nf)` into ~20 VECTOR instructions of float-reciprocal division per helper round (there is no scalar
    // integer division), and the search for a failing lane's first accepted layer into ~20 more.
#if RT_COOP_LAYERS == 4
    const uint32_t layers = nf <= 16u ? 4u : (nf <= 21u ? 3u : (nf <= 32u ? 2u : 1u));  // min(4, 64 / nf)
#else
    uint32_t layers = 64u / nf;
    if (layers > RT_COOP_LAYERS) layers = RT_COOP_LAYERS;
#endif
    const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(F >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)F, 0u));
    if (pending) xch[r] = make_uint4(ra.pixel, ra.sample, node, 0u);
    const uint32_t m = kCeil65536Over[nf];  // j = lane / nf for lane < 64 by multiplication (scalar load of a 65-entry table)
    const uint32_t j = __umul24(lane, m) >> 16, q = lane - __umul24(j, nf);
    const bool helping = j < layers;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    bool acc = false;
    uint32_t hx = 0, hy = 0, hz = 0;
    if (helping) {
      const uint4 a = xch[q];
      RngAddr ha; ha.pixel = a.x; ha.sample = a.y; ha.k0 = ra.k0; ha.k1 = ra.k1;
      const U4 hw = rng(ha, a.z, 1u + base + j);
      hx = hw.x; hy = hw.y; hz = hw.z;
      acc = length_squared(point(hx, hy, hz)) < 1.0;
    }
    const unsigned long long A = wave_ballot(acc);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // every helper has read its slot: reuse them for the answers
    if (acc) xch[lane] = make_uint4(hx, hy, hz, 0u);
    // failing lane #r: its helpers are lanes r, r + nf, r + 2 nf, ...: the lowest layer that accepted = the lowest set bit
    // of (A >> r) under the (wave-uniform) pattern of bits 0, nf, 2 nf, ... below layers * nf
    unsigned long long pat = 1ull;
    {
      const unsigned long long p1 = 1ull << (nf & 63u);
      if (layers > 1u) pat |= p1;
      if (layers > 2u) pat |= p1 << (nf & 63u);
      if (layers > 3u) pat |= (p1 << (nf & 63u)) << (nf & 63u);
#if RT_COOP_LAYERS > 4
      for (uint32_t l = 4; l < layers; ++l) pat |= 1ull << (l * nf);
#endif
    }
    const unsigned long long mine = pending ? ((A >> r) & pat) : 0ull;
    const bool found = mine != 0ull;
    const uint32_t from = r + (uint32_t)__builtin_ctzll(mine | (1ull << 63));
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (found) {
      const uint4 g = xch[from];
      wx = g.x; wy = g.y; wz = g.z;
      pending = false;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    base += layers;
  }
  return point(wx, wy, wz);
}

// centre of light j from the workgroup's LDS copy (the first LIGHT_CENTRES_LDS_MAX lights), else through HBM
__device__ __forceinline__ V3 light_centre_lds(const unsigned char* lds, const DevScene& sc, uint32_t j) {
  if (j < LIGHT_CENTRES_LDS_MAX) {
    const double* c = reinterpret_cast<const double*>(lds + LIGHT_CENTRES_LDS_OFF) + 3u * j;
    return v3(c[0], c[1], c[2]);
  }
  const SphereGeom lg = sc.geom[sc.lights[j]];
  return v3(lg.cx, lg.cy, lg.cz);
}
struct GlobalTablesK : GlobalTables {  // scenes whose tables stay in HBM / L2: only the light centres come from LDS
  const unsigned char* lds;
  __device__ __forceinline__ V3 light_centre(const DevScene& sc, uint32_t j) const { return light_centre_lds(lds, sc, j); }
};
struct LdsTables {  // per-lane gathers from the workgroup's LDS copies
  const double* g;
  const MatCore* m;
  const unsigned char* lds;
  __device__ __forceinline__ V3 light_centre(const DevScene& sc, uint32_t j) const { return light_centre_lds(lds, sc, j); }
  __device__ __forceinline__ SphereGeom geom(uint32_t i) const {
    const double* p = g + 4u * i;
    SphereGeom r; r.cx = p[0]; r.cy = p[1]; r.cz = p[2]; r.r = p[3];
    return r;
  }
  __device__ __forceinline__ MatCore mat(uint32_t i) const { return m[i]; }
};

// WIDE: the wide cell-table format of rt_core.h (GridDesc.wide: four words per cell, u32 item lists, the first item inline) that
// scenes of more than 65 535 spheres use, instead of the packed one (two words per cell, u16 items, the first TWO items inline).
// The two forms stand side by side as `if constexpr` blocks in hit_world: the packed path is textually what it was before the
// wide one existed, and compiles to the same code (helper functions over a common cell type did not: three instructions and a
// different register assignment in every instantiation, +0.4 % on the headline frame, profiles/r05_run19_ab_wide_tables.log).
template <bool HL, bool SIMPLE, bool LDS_TABLES, bool WIDE = false>
__global__ __launch_bounds__(BLOCK) RT_WAVES_ATTR void rt_megakernel(const KArgs ka) {
  static_assert(!(LDS_TABLES && WIDE), "wide tables (more than 65 535 spheres) never fit LDS");
  const DevScene& sc = ka.sc;
  const GridDesc& G = sc.grid;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  RT_PROF_DECL
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const LdsLayout lay = lds_layout(sc.n_spheres, G.n_cells, G.n_items, LDS_TABLES, HL, sc.light_pool_slots, sc.light_base_slots);
  uint32_t* const wg_flags = reinterpret_cast<uint32_t*>(lds_raw);  // [0] the frame's tile queue is empty, [1] slot opened last
  SlotHdr* const hdr = reinterpret_cast<SlotHdr*>(lds_raw + lay.hdr_off);
  const uint32_t T = ka.t_slots, acc_stride = 3u << (2u * ka.tile_log2);  // u64 words of pixel sums per slot
  unsigned long long* const tile_acc = reinterpret_cast<unsigned long long*>(lds_raw + lay.hdr_off + T * SLOT_HDR_BYTES);
  uint4* const coop_xch = reinterpret_cast<uint4*>(lds_raw + lay.coop_off) + wave * 64u;
  for (uint32_t i = threadIdx.x; i < T; i += BLOCK) {
    SlotHdr h; h.tile_xy = 0; h.next = 0x80000000u; h.finished = 0; h.expected = 0; h.state = SLOT_FREE; h.max_depth = 0;
    h.nan_mask[0] = h.nan_mask[1] = h.nan_mask[2] = 0ull;
    hdr[i] = h;
  }
  if (threadIdx.x == 0) { wg_flags[0] = 0u; wg_flags[1] = 0u; wg_flags[2] = 0u; wg_flags[3] = 0u; }
  // Tiles this workgroup has taken from the frame's queue and not opened yet: (end << 32 | next), positions in queue
  // order; empty when next >= end.  The queue is ONE counter (eight with XCD affinity, in one cache line) that every
  // workgroup of the chip adds to: taken one tile at a time, the opening wave waited ~27 us for it at 15 tiles/us
  // (RT_PROFILE sections against samples per pixel, profiles/r03_run25_spp_prof.log) — a fixed ~0.4 ms of every frame.
  unsigned long long* const wg_stash = reinterpret_cast<unsigned long long*>(lds_raw + 16);
  uint32_t* const wg_batch = reinterpret_cast<uint32_t*>(lds_raw + 24);  // tiles the next batch asks for
  if (threadIdx.x == 0) { *wg_stash = 0ull; *wg_batch = ka.tile_batch; }
  // the XCD this workgroup runs on (HW_REG_XCC_ID, bits 3:0; MI355X_MICROARCH.md): affinity only, never correctness
  const uint32_t my_xcd = (uint32_t)__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u;
  unsigned long long* const wg_counters = reinterpret_cast<unsigned long long*>(lds_raw + 32);
  if (threadIdx.x < 32u) wg_counters[threadIdx.x] = 0ull;
  if constexpr (HL) {  // every record of the two light pools is free
    if (threadIdx.x < 2u * LIGHT_POOL_BITMAP_BYTES / 4u) reinterpret_cast<uint32_t*>(lds_raw + LIGHT_POOL_LDS_OFF)[threadIdx.x] = 0u;
  }
  if constexpr (!LDS_TABLES) __syncthreads();

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
  if constexpr (HL) {  // centres of the lights: read when a light ray is aimed (raytracer.rs:104-105)
    double* lc = reinterpret_cast<double*>(lds_raw + LIGHT_CENTRES_LDS_OFF);
    const uint32_t nl = sc.n_lights < LIGHT_CENTRES_LDS_MAX ? sc.n_lights : LIGHT_CENTRES_LDS_MAX;
    for (uint32_t i = threadIdx.x; i < 3u * nl; i += BLOCK) lc[i] = reinterpret_cast<const double*>(sc.geom + sc.lights[i / 3u])[i % 3u];
    __syncthreads();
  }
  using Tables = typename std::conditional<LDS_TABLES, LdsTables, GlobalTablesK>::type;
  Tables tb;
  tb.lds = lds_raw;
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

  typedef Lane<HL, SIMPLE, HL> LaneT;  // (lit lanes keep their light records in the workgroup's LDS pools)
  LaneT L;
  L.s = 0; L.k = 0; L.node = 0; L.in_light = 0;
  L.val[0] = L.val[1] = L.val[2] = 0.0f;
  L.n_segments = L.n_exact = L.n_tex_oob = 0;  // (hostsim's counters; of these the kernel only passes n_tex_oob on, and empties it after every use)
  L.ra.pixel = 0; L.ra.sample = 0; L.ra.k0 = sc.seed_lo; L.ra.k1 = sc.seed_hi;
  L.o = v3(0, 0, 0); L.d = v3(0, 0, 1);
  fwd_init(L.fwd);
  if constexpr (HL) L.ls.wt = 0u;  // no light record held
  auto flush_oob = [&]() {  // (lit kernels, after every call that may count an out-of-range texel: L.n_tex_oob is 0 again, so nothing is carried)
    if constexpr (!HL) return;
    if (L.n_tex_oob != 0u) { __hip_atomic_fetch_add(&wg_counters[2], (unsigned long long)L.n_tex_oob, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); L.n_tex_oob = 0u; }
  };
  uint32_t n_exact = 0, n_steps = 0;  // per-lane counters (one exec-masked add each)
  // LIT kernels count segments per WAVE on the scalar unit (the lanes of the trace's lane mask: one s_bcnt1 per iteration)
  // and send out-of-range texel fetches — practically never — straight to the workgroup's counter: two registers every
  // lane carried across the walk loop, in the lit kernels exactly the pair that was spilled there (15 scratch round trips
  // per wave iteration, profiles/r03_codeobj.txt -> r04_codeobj.txt).  The unlit kernels have the registers and keep the
  // per-lane counters (the scalar form cost them 0.6 %, profiles/r04_run2_ab.log).
  uint32_t w_segments = 0;
  uint32_t cnt_w_iter = 0, cnt_w_step = 0, cnt_w_test = 0, cnt_items = 0;  // wave trip counts (RT_PROFILE builds)

  RT_PROF(5);  // staging of the tables into LDS (+ item bookkeeping later)
  // ---- work distribution, two levels.
  // Global: a queue of pixel TILES (2^k x 2^k pixels, all their samples).  Workgroup: an open tile
  // lives in one of T shared LDS slots (header + exact fixed-point pixel sums); its samples are
  // handed out to the workgroup's waves in n_chunks chunks, every finished sample is counted per
  // tile, and the wave that adds a tile's last sample converts the sums and writes its pixels —
  // the only HBM traffic of the frame.  Wave: hands out one chunk item at a time; a lane that
  // finishes a sample takes the next one at once (of the next item, if this one ran dry), so no
  // lane idles while a neighbour finishes a long path, and all 16 waves of a workgroup converge
  // on the last tiles of the frame.
  uint32_t it_k = 0, it_bx = 0, it_by = 0, it_sbeg = 0, it_total = 0, it_next = 0;  // the wave's current item (uniform)
  bool q_done = false;            // no more items will ever be available to this wave
  uint32_t py_slot = 0;           // per lane: global scanline of this lane's pixel slot in the current item
  bool ok_slot = false;           // per lane: that pixel slot is inside the image
  uint32_t my_k = 0, cur_p = lane;  // tile slot and pixel slot of the lane's sample
  bool has_ray = false;
  auto bcast = [&](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
  auto lds_load = [&](const uint32_t* p) -> uint32_t { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
  auto lds_store = [&](uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); };

  // every sample of tile slot k is in its pixel sums: mean, sqrt gamma, f32 -> u8, store
  // (raytracer.rs:207-216), then free the slot
  auto flush_tile = [&](uint32_t k) {
    const KArgs& ka = fresh_args();
    const DevScene& sc = ka.sc;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const uint32_t xy = bcast(lds_load(&hdr[k].tile_xy));
    const uint32_t tl = ka.tile_log2, wl = ka.tile_wl, tw = 1u << wl;
    const uint32_t px = ((xy & 0xFFFFu) << wl) + (lane & (tw - 1u)), lr = ((xy >> 16) << ka.tile_hl) + (lane >> wl);
    const unsigned long long* acc = tile_acc + k * (3u << (2u * tl));
    const bool valid = lane < (1u << (2u * tl)) && px < sc.width && lr < ka.local_rows;
    const size_t o = ((size_t)lr * sc.width + px) * 3;
    uint32_t rgb = 0u;  // R | G << 8 | B << 16
    if (valid) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float lin = fixed_to_mean(__hip_atomic_load(&acc[lane * 3u + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), sc.spp);
        if ((__hip_atomic_load(&hdr[k].nan_mask[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> lane) & 1ull) lin = rt_nanf();
        if (ka.out_linear) ka.out_linear[o + c] = lin;
        rgb |= (uint32_t)f32_to_u8(__builtin_sqrtf(lin)) << (8 * c);
      }
    }
    // Framebuffer write.  Tiles at least 4 pixels wide in a frame whose width is a multiple of 4: every aligned group
    // of 4 pixels of a tile row is 12 contiguous, 4-byte-aligned bytes, written as three dwords by its first three
    // lanes (R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3; the neighbour's bytes arrive by a lane shift) — one store
    // instruction per flush instead of three byte stores.  Otherwise: bytes.
    const bool packed = wl >= 2u && (sc.width & 3u) == 0u && (reinterpret_cast<uintptr_t>(ka.out_rgb8) & 3u) == 0u;
    if (packed) {
      const uint32_t nxt = (uint32_t)__shfl_down((int)rgb, 1);
      uint32_t i = lane & 3u;
      // (made HERE: hoisted out of the path loop, i, 8 i, 24 - 8 i and a zero-extended copy hold four registers for good — spilled in
      //  the lit kernels and with the general colour map: <lights=0, simple=0> 12 -> 0 spilled registers.  The unlit short-map
      //  kernel has them to spare; measured both ways three times, it is 0.0 - 0.9 % faster with this form, profiles/r05_run6_ab_takes_and_flush.log)
      asm volatile("" : "+v"(i));
      if (valid && i < 3u) *reinterpret_cast<uint32_t*>(ka.out_rgb8 + o + i) = (rgb >> (8u * i)) | (nxt << (24u - 8u * i));
    } else if (valid) {
      ka.out_rgb8[o] = (uint8_t)rgb; ka.out_rgb8[o + 1] = (uint8_t)(rgb >> 8); ka.out_rgb8[o + 2] = (uint8_t)(rgb >> 16);
    }
    if (lane == 0 && ka.tile_depth) ka.tile_depth[(xy >> 16) * ka.tiles_x + (xy & 0xFFFFu)] = lds_load(&hdr[k].max_depth);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) {
      lds_store(&hdr[k].next, 0x80000000u);  // nothing to hand out from a free slot
      lds_store(&hdr[k].state, (uint32_t)SLOT_FREE);
    }
  };

  // Take a chunk of an open tile, or open the next tile of the frame.  1: got (k, chunk);
  // 0: nothing right now (every slot is busy); -1: the frame has nothing left to hand out.
  auto acquire = [&](uint32_t& k_out, uint32_t& chunk_out) -> int {
    const KArgs& ka = fresh_args();
    const DevScene& sc = ka.sc;
    const uint32_t n_chunks = ka.n_chunks;
    for (int tries = 0; tries < 8; ++tries) {
      {  // fast path: the tile opened last usually still has chunks
        const uint32_t h = bcast(lds_load(&wg_flags[1]));
        uint32_t c = 0;
        if (lane == 0) c = atomicAdd(&hdr[h].next, 1u);
        c = bcast(c);
        if (c < n_chunks) { k_out = h; chunk_out = c; return 1; }
      }
      // any other open tile with chunks left (two waves may have opened tiles at the same moment)
      bool any_opening = false;
      unsigned long long free_mask = 0ull; uint32_t free_base = 0;
      for (uint32_t base = 0; base < T; base += 64u) {
        const uint32_t k = base + lane;
        const bool mine = k < T;
        const uint32_t st = mine ? lds_load(&hdr[mine ? k : 0].state) : (uint32_t)SLOT_OPEN;
        const uint32_t nx = mine ? lds_load(&hdr[mine ? k : 0].next) : 0xFFFFFFFFu;
        unsigned long long mo = wave_ballot(mine && st == SLOT_OPEN && nx < n_chunks);
        while (mo) {
          const uint32_t kk = base + (uint32_t)__builtin_ctzll(mo);
          mo &= mo - 1ull;
          uint32_t c = 0;
          if (lane == 0) c = atomicAdd(&hdr[kk].next, 1u);
          c = bcast(c);
          if (c < n_chunks) { k_out = kk; chunk_out = c; return 1; }
        }
        any_opening = any_opening || wave_any(mine && st == SLOT_OPENING);
        const unsigned long long mf = wave_ballot(mine && st == SLOT_FREE);
        if (mf && !free_mask) { free_mask = mf; free_base = base; }
      }
      if (bcast(lds_load(&wg_flags[0])) != 0u) return any_opening ? 0 : -1;  // (a tile being opened right now will still offer chunks)
      if (!free_mask) return 0;
      const uint32_t k = free_base + (uint32_t)__builtin_ctzll(free_mask);
      uint32_t ok = 0;
      if (lane == 0) ok = atomicCAS(&hdr[k].state, (uint32_t)SLOT_FREE, (uint32_t)SLOT_OPENING) == (uint32_t)SLOT_FREE ? 1u : 0u;
      if (!bcast(ok)) continue;  // another wave claimed it: rescan
      // the next tile: from the workgroup's stash; whoever finds it empty first fetches the next batch from the queue
      uint32_t tile = 0;
      if (lane == 0) {
        uint32_t g = 0xFFFFFFFFu;  // position in queue order; n_tiles: the frame has none left; ~0: a batch is on its way
        unsigned long long old = atomicAdd(wg_stash, 1ull);
        const uint32_t s_next = (uint32_t)old, s_end = (uint32_t)(old >> 32);
        if (s_next < s_end) g = s_next;
        else if (s_next == s_end) {
          const uint32_t B = lds_load(wg_batch);
          uint32_t end = 0, rem = 0, share = gridDim.x * ka.batch_share;
          g = ka.n_tiles;
          if (ka.aff_group_log2 == 0xFFFFFFFFu) {
            const uint32_t j = atomicAdd(ka.queue, B);
            if (j < ka.n_tiles) { g = j; end = j + B < ka.n_tiles ? j + B : ka.n_tiles; rem = ka.n_tiles - end; }
          } else {  // this XCD's queue first, then the others'
            uint32_t dry = lds_load(&wg_flags[3]);  // queues this workgroup has seen empty
            for (uint32_t q = 0; q < 8u && g == ka.n_tiles; ++q) {
              const uint32_t x = (my_xcd + q) & 7u;
              if ((dry >> x) & 1u) continue;
              const uint32_t cnt = ka.xcd_cnt[x];
              const uint32_t j = atomicAdd(&ka.queue[x], B);
              if (j >= cnt) { dry |= 1u << x; continue; }
              const uint32_t e = j + B < cnt ? j + B : cnt;
              g = ka.xcd_off[x] + j; end = ka.xcd_off[x] + e; rem = cnt - e;
            }
            share = (gridDim.x + 7u) / 8u * ka.batch_share;
            __hip_atomic_fetch_or(&wg_flags[3], dry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          if (g < ka.n_tiles) {
            // the next batch: 1 / batch_share (a sixteenth) of a workgroup's fair share of what the queue still holds —
            // single tiles at the end of the frame, where a stashed tile is work no other workgroup can take
            const uint32_t nb = rem / share;
            lds_store(wg_batch, nb < 1u ? 1u : (nb > ka.tile_batch ? ka.tile_batch : nb));
            __hip_atomic_exchange(wg_stash, ((unsigned long long)end << 32) | (unsigned long long)(g + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
        tile = g;
        if (g < ka.n_tiles) {  // queue position -> tile
          if (ka.aff_group_log2 == 0xFFFFFFFFu) {
            if (ka.order_mode != 0u) tile = ka.tile_order ? ka.tile_order[g] : ka.n_tiles - 1u - g;
          } else if (ka.tile_order) tile = ka.tile_order[g];
          else {
            uint32_t x = 0;
            while (x < 7u && g >= ka.xcd_off[x + 1u]) ++x;
            const uint32_t j = g - ka.xcd_off[x];
            tile = xcd_tile(x, ka.order_mode != 0u ? ka.xcd_cnt[x] - 1u - j : j, ka.aff_group_log2);
          }
        }
      }
      tile = bcast(tile);
      if (tile == 0xFFFFFFFFu) {  // another wave is fetching a batch: ask again in the next iteration
        if (lane == 0) lds_store(&hdr[k].state, (uint32_t)SLOT_FREE);
        return 0;
      }
      if (tile >= ka.n_tiles) {  // the frame's queue is empty
        if (lane == 0) { lds_store(&wg_flags[0], 1u); lds_store(&hdr[k].state, (uint32_t)SLOT_FREE); }
        continue;
      }
      const uint32_t tl = ka.tile_log2, wl = ka.tile_wl, tw = 1u << wl, npx = 1u << (2u * tl);
      const uint32_t by = tile / ka.tiles_x, bx = tile - by * ka.tiles_x;
      const uint32_t px = (bx << wl) + (lane & (tw - 1u)), lr = (by << ka.tile_hl) + (lane >> wl);
      const uint32_t n_valid = (uint32_t)__builtin_popcountll(wave_ballot(lane < npx && px < sc.width && lr < ka.local_rows));
      // max_depth == 0: ray_color returns black before tracing anything (raytracer.rs:80-82)
      const uint32_t expected = sc.max_depth != 0u ? n_valid * sc.spp : 0u;
      unsigned long long* acc = tile_acc + k * acc_stride;
      unsigned long long zero;  // (made here: as a plain constant the pair was hoisted out of the path loop and, in the lit kernels, spilled)
      { uint32_t z; asm volatile("v_mov_b32 %0, 0" : "=v"(z)); zero = ((unsigned long long)z << 32) | z; }
      if (lane < npx) { acc[lane * 3u] = zero; acc[lane * 3u + 1u] = zero; acc[lane * 3u + 2u] = zero; }
      if (lane < 3u) hdr[k].nan_mask[lane] = zero;
      if (lane == 3u) hdr[k].max_depth = 0u;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) {  // publish: everything before next, next before state
        lds_store(&hdr[k].tile_xy, bx | (by << 16));
        lds_store(&hdr[k].finished, 0u);
        lds_store(&hdr[k].expected, expected);
        lds_store(&hdr[k].next, expected ? 1u : 0x80000000u);  // this wave takes chunk 0
        lds_store(&hdr[k].state, (uint32_t)SLOT_OPEN);
        lds_store(&wg_flags[1], k);
      }
      if (!expected) { flush_tile(k); continue; }  // nothing to trace: write the (black) pixels now
      k_out = k; chunk_out = 0;
      return 1;
    }
    return 0;
  };
  auto open_item = [&](uint32_t k, uint32_t chunk) {
    const KArgs& ka = fresh_args();
    const DevScene& sc = ka.sc;
    // pairs with the opener's release stores (tile_xy ... before next, next before state): the chunk came from
    // a relaxed atomicAdd on `next`, so order the header reads behind it explicitly
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const uint32_t xy = bcast(lds_load(&hdr[k].tile_xy));
    const uint32_t bx = xy & 0xFFFFu, by = xy >> 16;
    const uint32_t tl = ka.tile_log2, wl = ka.tile_wl, tw = 1u << wl, npx = 1u << (2u * tl);  // pixel slots of the tile: lanes 0..npx-1
    const uint32_t px = (bx << wl) + (lane & (tw - 1u));
    const uint32_t lr = (by << ka.tile_hl) + (lane >> wl);  // local (packed) row
    uint32_t py = lr;  // global scanline (raytracer.rs:255: band index, 0 = top)
    if (ka.tile_rows != 0u) py = (ka.first_tile + (lr / ka.tile_rows) * ka.tile_stride) * ka.tile_rows + lr % ka.tile_rows;
    const uint32_t s_begin = chunk * ka.chunk_spp;
    const uint32_t s_left = sc.spp - s_begin;
    const uint32_t s_count = s_left < ka.chunk_spp ? s_left : ka.chunk_spp;
    it_k = k; it_bx = bx; it_by = by; it_sbeg = s_begin; it_next = 0;
    it_total = npx * s_count;  // pool item w = (pixel slot w % npx, sample s_begin + w / npx)
    py_slot = py; ok_slot = lane < npx && px < sc.width && lr < ka.local_rows;
    RT_PROF_COUNT(cnt_items);
  };

  // ---- hand out (pixel, sample) pairs to the lanes that ask for one: sets the lane's tile slot / pixel slot / sample /
  // RNG pixel and returns its pixel coordinates; true for the lanes that received one.  A lane that finishes a sample
  // takes the next one by ballot + prefix rank — of the next item at once if this one ran dry.
  auto hand_out = [&](bool want, uint32_t& o_px, uint32_t& o_py) -> bool {
    const KArgs& kr = fresh_args();
    const DevScene& sc = kr.sc;
    const uint32_t wl = kr.tile_wl, tw = 1u << wl, pl = 2u * kr.tile_log2, pmask = (1u << pl) - 1u;
    bool got = false;
    for (;;) {
      const unsigned long long m = wave_ballot(want);
      if (!m) break;
      if (it_next < it_total) {  // hand out samples of the current item
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        const uint32_t w = it_next + rank;
        const uint32_t left = it_total - it_next, asked = (uint32_t)__builtin_popcountll(m);
        it_next += asked < left ? asked : left;
        const uint32_t p = w & pmask;
        const uint32_t p_px = (it_bx << wl) + (p & (tw - 1u));
        const uint32_t p_py = (uint32_t)__shfl((int)py_slot, (int)p);
        const int p_ok = __shfl((int)ok_slot, (int)p);
        if (want && w < it_total && p_ok) {  // (a slot outside the image consumes its index and asks again)
          cur_p = p; my_k = it_k; L.s = it_sbeg + (w >> pl); L.ra.pixel = p_py * sc.width + p_px; L.ra.sample = L.s;
          o_px = p_px; o_py = p_py;
          got = true; want = false;
        }
        continue;
      }
      // the current item has nothing (more) to hand out: on to the next one
      if (q_done) break;
      uint32_t k = 0, chunk = 0;
      const int have = acquire(k, chunk);
      if (have < 0) {
        q_done = true;
#ifdef RT_PROFILE
        prof_wall_qdone = wall_clock64();
        prof_in_tail = true;
#endif
        break;
      }
      if (have == 0) break;  // every tile slot is busy: these lanes wait
      open_item(k, chunk);
    }
    return got;
  };

  // ---- a sample finished (its radiance is in L.val): add it to its pixel (raytracer.rs:203-205) ...
  auto add_sample = [&](bool finished) {
    if (finished) {
      unsigned long long* acc = tile_acc + __umul24(my_k, acc_stride) + __umul24(cur_p, 3u);  // (24-bit factors: full-rate multiplies)
      atomicAdd(&acc[0], sample_to_fixed(L.val[0]));
      atomicAdd(&acc[1], sample_to_fixed(L.val[1]));
      atomicAdd(&acc[2], sample_to_fixed(L.val[2]));
      if (L.k >= RT_DEEP_PATH) atomicMax(&hdr[my_k].max_depth, L.k);  // (rare: tiles that breed deep paths go first next frame)
      // NaN samples (frames one pixel wide or high; NaN scene data) added 0 above: flag the pixel instead.  Samples
      // are clamped to [0, 1], so the sum of the channels is NaN iff one of them is.  (A plain divergent branch right
      // here: a wave vote around it, or a cold call, cost 6-12 more spilled registers in the path loop.)
      if (sample_is_nan((L.val[0] + L.val[1]) + L.val[2])) {
        const unsigned long long bit = 1ull << cur_p;
        if (sample_is_nan(L.val[0])) atomicOr(&hdr[my_k].nan_mask[0], bit);
        if (sample_is_nan(L.val[1])) atomicOr(&hdr[my_k].nan_mask[1], bit);
        if (sample_is_nan(L.val[2])) atomicOr(&hdr[my_k].nan_mask[2], bit);
      }
    }
  };
  // ... and count it in its tile (slot `k_of`); whoever adds a tile's last sample converts the sums and writes its pixels
  auto count_tiles = [&](bool finished, uint32_t k_of) {
    unsigned long long mf = wave_ballot(finished);
    if (mf) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      while (mf) {
        const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)k_of, (int)__builtin_ctzll(mf));
        const unsigned long long same = wave_ballot(finished && k_of == k);
        mf &= ~same;
        const uint32_t cnt = (uint32_t)__builtin_popcountll(same);
        uint32_t total = 0, expected = 1;
        if (lane == 0) { total = atomicAdd(&hdr[k].finished, cnt) + cnt; expected = lds_load(&hdr[k].expected); }
        if (bcast(total) == bcast(expected)) flush_tile(k);
      }
    }
  };

  // ---- hit_world (raytracer.rs:44-59) for the lanes that hold a ray
  auto hit_world = [&](double& closest, int& best) {
    // ---------------------------------------------------------- hit_world (raytracer.rs:44-59)
    const DevScene& sc = fresh_args().sc;
    const GridDesc& G = sc.grid;
    const F64PtrK geom_k = (F64PtrK)(uintptr_t)sc.geom;
    const U32PtrK large_k = (U32PtrK)(uintptr_t)sc.large;
    const uint32_t n_large = G.n_large;
    const bool has_grid = G.n[0] != 0u;
    const RayK rk = ray_consts(L.d);
    closest = t_max_fresh();
    best = -1;
    if constexpr (HL) w_segments += (uint32_t)__builtin_popcountll(wave_ballot(has_ray));
    else if (has_ray) L.n_segments++;
    // (1) spheres outside the grid: every lane tests them.  The records are wave-uniform, so they
    // arrive by scalar loads as SGPR operands; the next record is fetched while this one is tested.
    if (n_large != 0u) {
      const F64PtrK lg = (F64PtrK)(uintptr_t)sc.large_geom;
      SphereGeom g; g.cx = lg[0]; g.cy = lg[1]; g.cz = lg[2]; g.r = lg[3];
      uint32_t idx = large_k[0];
      for (uint32_t i = 0; i < n_large; ++i) {
        const uint32_t nx = i + 1u < n_large ? i + 1u : i;  // (the last round re-reads its own record)
        const F64PtrK np = lg + (size_t)nx * 4u;
        SphereGeom gn; gn.cx = np[0]; gn.cy = np[1]; gn.cz = np[2]; gn.r = np[3];
        const uint32_t idxn = large_k[nx];
        if (has_ray && rk.fast) exact_hit_any_order_t<true>(L.o, L.d, rk, g, idx, closest, best);
        g = gn; idx = idxn;
      }
    }
    if (has_ray && rk.fast) n_exact += n_large;
    RT_PROF(1);
    // (2) enter the grid
    GridWalk w;
    const int mode = !has_ray ? GRID_MISS : (!rk.fast ? GRID_FALLBACK : (has_grid ? grid_begin(G, L.o, L.d, w) : GRID_MISS));
    if (wave_any(mode == GRID_FALLBACK)) {  // numerically unsafe ray: the reference's full scan, real divisions
      for (uint32_t idx = 0; idx < sc.n_spheres; ++idx) {
        const F64PtrK gp = geom_k + (size_t)idx * 4u;
        SphereGeom g; g.cx = gp[0]; g.cy = gp[1]; g.cz = gp[2]; g.r = gp[3];
        if (mode == GRID_FALLBACK) { const HitCB r = exact_hit_slow(L.o, L.d, rk.a, g, idx, closest, best); closest = r.closest; best = r.best; }
      }
      if (mode == GRID_FALLBACK) n_exact += sc.n_spheres;
    }
    if (has_grid) {
      // (3) walk rounds: every walking lane moves on by up to two cells and/or tests one sphere.
      // Per-lane walk state: tm = GridWalk.tmax, dt = GridWalk.delta, dl = GridWalk.dl, lin;
      // the current cell's untested spheres are items [it, end), the next two of them also in `pend`.
      const bool walk0 = mode == GRID_WALK;
      float tm0 = w.tmax[0], tm1 = w.tmax[1], tm2 = w.tmax[2];
      const float dt0 = w.delta[0], dt1 = w.delta[1], dt2 = w.delta[2];
      const int dl0 = w.dl[0], dl1 = w.dl[1], dl2 = w.dl[2];
      int lin = walk0 ? w.lin : 0;
      const double t0 = w.t0;
      const int lin_max = (int)G.n_cells - 1;
      // A lane is walking while it <= end (it == end: cell exhausted, move on; it < end: spheres left to
      // test); a lane that stopped has (it, end) = (1, 0).  Each predicate is ONE compare: a separate
      // `walking` flag costs a lane-mask AND per use and a VGPR round trip in the loop's exit vote.
      uint32_t it = 1, end = 0, pend = 0xFFFFFFFFu, last = 0xFFFFFFFFu;
      if (walk0) {
        if constexpr (WIDE) {
          const uint4 e = reinterpret_cast<const uint4*>(cell_word)[lin];
          it = e.x; end = it + e.y; pend = e.z;
        } else {
          const uint2 e = cell_word[lin];
          it = e.x & CELL_START_MASK; end = it + (e.x >> CELL_COUNT_SHIFT); pend = e.y;
        }
      }
      for (;;) {
        if (!wave_any(it <= end)) break;
        // (a) lanes whose cell is exhausted: finished, or on to the next non-empty cell.  The next
        // TWO cells along the ray are computed and fetched together (one LDS round trip), the
        // second one is used only if the first is empty.
        const bool moving = it == end;
        {  // (no wave vote around the block: the lane mask of `if (moving)` already skips it when empty)
#ifdef RT_PROFILE
#ifndef RT_PROF_LIT
          if (wave_any(moving)) cnt_w_step++;
#endif
#endif
          if (moving) {
            float tc = (float)(closest - t0);
            tc = tc + fabsf(tc) * 2.384185791015625e-07f;  // grid_done
            const bool hit = best >= 0;
            const float tminA = rt_min3f(tm0, tm1, tm2);
            if (hit && tc < tminA) { it = 1; end = 0; }
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
              if constexpr (WIDE) {
                const uint4 eA = reinterpret_cast<const uint4*>(cell_word)[linA];
                const uint4 eB = reinterpret_cast<const uint4*>(cell_word)[linB];
                n_steps++;
                const bool exitA = eA.x == CELL_EXIT, emptyA = eA.y == 0u;
                const bool doneA = hit && tc < tminB;  // the closest hit lies inside cell A
                if (exitA || !emptyA || doneA) {  // stay in A (or stop there)
                  tm0 = a0; tm1 = a1; tm2 = a2; lin = linA;
                  it = eA.x; end = it + eA.y; pend = eA.z;
                  if (exitA || emptyA) { it = 1; end = 0; }
                } else {                           // A is empty: on to B
                  n_steps++;
                  tm0 = b0; tm1 = b1; tm2 = b2; lin = linB;
                  it = eB.x; end = it + eB.y; pend = eB.z;
                  if (eB.x == CELL_EXIT) { it = 1; end = 0; }
                }
              } else {
                const uint2 eA = cell_word[linA];
                const uint2 eB = cell_word[linB];
                n_steps++;
                const bool exitA = eA.x == CELL_EXIT, emptyA = (eA.x >> CELL_COUNT_SHIFT) == 0u;
                const bool doneA = hit && tc < tminB;  // the closest hit lies inside cell A
                if (exitA || !emptyA || doneA) {  // stay in A (or stop there)
                  tm0 = a0; tm1 = a1; tm2 = a2; lin = linA;
                  it = eA.x & CELL_START_MASK; end = it + (eA.x >> CELL_COUNT_SHIFT); pend = eA.y;
                  if (exitA || emptyA) { it = 1; end = 0; }
                } else {                           // A is empty: on to B
                  n_steps++;
                  tm0 = b0; tm1 = b1; tm2 = b2; lin = linB;
                  it = eB.x & CELL_START_MASK; end = it + (eB.x >> CELL_COUNT_SHIFT); pend = eB.y;
                  if (eB.x == CELL_EXIT) { it = 1; end = 0; }
                }
              }
            }
          }
        }
        const bool testing = it < end;
        {  // (b) one exact Sphere::hit per lane standing in a cell with spheres left
#ifdef RT_PROFILE
#ifndef RT_PROF_LIT
          if (wave_any(testing)) cnt_w_test++;
#endif
#endif
          if (testing) {
            uint32_t idx;
            if constexpr (WIDE) {  // (one inline item, the rest from the 32-bit list)
              idx = pend;
              if (idx == CELL_NO_ITEM32) idx = reinterpret_cast<const uint32_t*>(cell_items)[it];
              pend = CELL_NO_ITEM32;
            } else {
              idx = pend & 0xFFFFu;
              if (idx == 0xFFFFu) idx = cell_items[it];  // third and later items of a cell: from the list
              pend = (pend >> 16) | 0xFFFF0000u;
            }
            it++;
            if (idx != last) {  // a sphere spanning consecutive cells is not re-tested
              last = idx; n_exact++;
              exact_hit_any_order_t<true>(L.o, L.d, rk, tb.geom(idx), idx, closest, best);
            }
          }
        }
      }
    }
  };

  RT_PROF(5);
  uint32_t idle_spins = 0;
  // Loop order: trace -> rays that left the scene finish at once (sky) -> every lane without a path takes its next
  // sample -> ONE Philox instruction stream serves the hits (unit-sphere point / Glass draw) and the new samples (camera
  // jitter) -> shade the hits -> start the new samples.  The refill used to be a block of its own at the top of the loop
  // with its own Philox stream (~100 of the ~1700 vector instructions of an iteration, executed by the ~38 % of lanes
  // that had finished).  The first iteration of a wave traces nothing and only hands out samples.
  for (;;) {
#ifdef RT_PROFILE
    if (q_done) { prof_tail_iters++; prof_tail_lanes += (uint32_t)__builtin_popcountll(wave_ballot(has_ray)); }
#endif
    RT_PROF_COUNT(cnt_w_iter);
    RT_PROF(0);
    double closest = t_max_fresh();
    int best = -1;
    if (wave_any(has_ray)) hit_world(closest, best);
    RT_PROF(3);
    // (a) a ray that left the scene ends its sample here (raytracer.rs:133-163); light rays return to their parent in (d)
    bool miss = has_ray && best < 0;
    if constexpr (HL) miss = miss && !(L.in_light & 1u);
    const uint32_t k_miss = my_k;
    if (wave_any(miss)) {
      if (miss) {
        { const DevScene& scf = fresh_args().sc; lane_finish_sample(scf, L, sky_color(scf, L.d, L.n_tex_oob)); lane_base_release(scf, L); }
        flush_oob();
        has_ray = false;
      }
      add_sample(miss);
    }
    RT_PROF(4);
    // (b) every lane without a path takes its next sample
    uint32_t n_px = 0, n_py = 0;
    const bool fresh = hand_out(!has_ray, n_px, n_py);
    RT_PROF(0);
    // (c) the random numbers of this iteration, one instruction stream
    const uint32_t hit_kind = has_ray && best >= 0 ? tb.mat((uint32_t)best).kind : 0xFFFFFFFFu;
    double glass_u, light_u;
    U4 cam_w;
    const V3 rnd = coop_random_in_unit_sphere(hit_kind != 0xFFFFFFFFu && material_draws_unit_sphere(hit_kind), hit_kind == RT_MAT_GLASS, fresh,
                                              L.ra, L.node, lane, coop_xch, glass_u, light_u, cam_w);
#ifdef RT_PROF_SPLIT  // (experiment builds: the random draws are booked under "item", lane_shade proper stays under "lane_shade")
    RT_PROF(5);
#endif
    if constexpr (HL) {
      // The light-sampling draw (raytracer.rs:100) of a hit that is not Glass: its HIGH word is the word attempt 0's Philox call
      // left over (slot 1, .w — rt_core.h, RNG addressing), so `draw > threshold` is decided here without a Philox stream of its
      // own (rounds 3/4 ran one in every wave iteration: ~4 % of a lit frame) unless the high word alone leaves it open: 2^-32
      // of the draws fetch their low word through a real call.  (Glass hits: `light_u` came back from their slot-0 call.)
      if (hit_kind != RT_MAT_GLASS) {  // (lanes without a hit compute it too and never read it)
        const double thr = fresh_args().sc.light_thr[0];
        const double u_lo = u01_53(0u, cam_w.w), u_hi = u01_53(0xFFFFFFFFu, cam_w.w);
        light_u = u_lo;
        if (hit_kind != 0xFFFFFFFFu && (u_lo > thr) != (u_hi > thr)) light_u = u01_53(light_draw_low_word(L.ra, L.node), cam_w.w);
      }
    }
#ifdef RT_PROF_SPLIT
    RT_PROF(0);  // (the light draw alone: booked under "refill")
#endif
    // (d) shade the hits
    int status = LANE_CONTINUE;
    if (has_ray) {
#ifdef RT_PROF_LIT
      rtc::ShadeProf shade_prof{prof_t, &prof_last, &cnt_w_step, &cnt_w_test};
      status = lane_shade(fresh_args().sc, tb, L, best, closest, &rnd, &glass_u, HL ? &light_u : nullptr, &shade_prof);
#else
      status = lane_shade(fresh_args().sc, tb, L, best, closest, &rnd, &glass_u, HL ? &light_u : nullptr);
#endif
      if constexpr (HL) { if (status == LANE_REPEAT) L.n_tex_oob = 0u; }  // (the hit is shaded again next iteration: its out-of-range texel counts THEN, once — RtStats.tex_oob equals the oracle's)
      flush_oob();
    }
    const bool finished = status == LANE_FINISHED;
    if constexpr (HL) {  // a segment repeated because a light pool was exhausted is ONE segment of its path (a wave-level count: outside the divergent region)
      const unsigned long long rep = wave_ballot(status == LANE_REPEAT);
      if (rep) {  // (rare: the repeats go straight to the workgroup's counter — RtStats.segments_repeated — no register carries them)
        w_segments -= (uint32_t)__builtin_popcountll(rep);
        if (lane == 0) __hip_atomic_fetch_add(&wg_counters[28], (unsigned long long)__builtin_popcountll(rep), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    RT_PROF(2);
    if (wave_any(finished)) {
      if (finished) { has_ray = false; lane_base_release(fresh_args().sc, L); }
      add_sample(finished);
    }
    count_tiles(miss || finished, miss ? k_miss : my_k);
    RT_PROF(4);
    // (e) the new samples start (raytracer.rs:199-201, camera.rs:79-84)
    if (fresh) { lane_begin_sample_w(fresh_args().sc, L, n_px, n_py, cam_w); has_ray = true; }
    RT_PROF(0);
    if (!wave_any(has_ray)) {
      // nothing in flight.  Done when the frame has nothing left; otherwise (all tile slots are busy with other waves'
      // long paths) wait a little and ask again — bounded, a wave may always retire: the samples it traced are already
      // counted in their tiles.
      if (q_done || ++idle_spins > (1u << 16)) break;
      __builtin_amdgcn_s_sleep(32);
    } else idle_spins = 0;
  }

  // counters: wave reduction, one atomic per wave
  // (segments: per wave, on lane 0; out-of-range texels went to the workgroup's counter as they happened)
  unsigned long long c0 = HL ? (lane == 0 ? w_segments : 0u) : L.n_segments, c1 = n_exact, c2 = HL ? 0u : L.n_tex_oob, c3 = n_steps;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    c0 += __shfl_down(c0, off); c1 += __shfl_down(c1, off); c2 += __shfl_down(c2, off); c3 += __shfl_down(c3, off);
  }
  if (lane == 0) {
    // workgroup totals in LDS; the wave that retires last sends them on (one global atomic per
    // counter and workgroup, not per wave: 4096 waves hitting the same few words at the end of
    // the frame queue up at the memory side)
    auto wg_add = [&](int k, unsigned long long v) { __hip_atomic_fetch_add(&wg_counters[k], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto wg_max = [&](int k, unsigned long long v) { __hip_atomic_fetch_max(&wg_counters[k], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    wg_add(0, c0); wg_add(1, c1); wg_add(2, c2); wg_add(3, c3);
    wg_add(4, cnt_w_iter); wg_add(5, cnt_w_step); wg_add(6, cnt_w_test); wg_add(7, cnt_items);
#ifdef RT_PROFILE
    RT_PROF(5);
    prof_t[6] = prof_last - prof_begin;
    for (int k = 0; k < 7; ++k) wg_add(8 + k, prof_t[k]);
    for (int k = 0; k < 5; ++k) wg_add(19 + k, prof_tail_t[k]);   // sections of the iterations after the queue ran dry
    wg_max(15, prof_last - prof_begin);     // longest / shortest wave (clock bases differ between XCDs)
    wg_max(17, ~(prof_last - prof_begin));  // (max of the complement = min; the counters start at 0)
    {  // timeline on the chip-wide 100 MHz clock, behind the 32 counters
      const KArgs& ka = fresh_args();
      const uint32_t wid = blockIdx.x * WAVES + wave;
      ka.counters[32 + 4 * wid] = prof_wall0;
      ka.counters[32 + 4 * wid + 1] = wall_clock64();
      ka.counters[32 + 4 * wid + 2] = prof_wall_qdone;
      ka.counters[32 + 4 * wid + 3] = (unsigned long long)prof_tail_iters | ((unsigned long long)prof_tail_lanes << 32);
    }
#endif
    const uint32_t retired = __hip_atomic_fetch_add(&wg_flags[2], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (retired == WAVES - 1u) {
      const KArgs& ka = fresh_args();
      for (int k = 0; k < 24; ++k) {
        const unsigned long long v = __hip_atomic_load(&wg_counters[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (k == 15 || k == 17) { if (v) atomicMax(&ka.counters[k], v); }
        else if (v) atomicAdd(&ka.counters[k], v);
      }
      if constexpr (HL) {
        const unsigned long long v = __hip_atomic_load(&wg_counters[28], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (v) atomicAdd(&ka.counters[28], v);
      }
    }
  }
}

// Launched once when a scene is created: the runtime loads a module's code object onto the device with the first launch of ANY
// of its kernels — milliseconds that would otherwise sit inside the first frame.
__global__ void rt_warm_up() {}
// ... and the first launch on a queue of a kernel that needs SCRATCH makes the runtime give that queue its scratch memory (every
// megakernel instantiation has 64 - 112 B per lane: the by-value arguments of its cold calls, a few spilled registers).  Measured
// on one-shot frames (round 6, profiles/r06_run5_first_launch_wait.log): 0.2 ms when it happens on a fresh device, 13 - 25 ms when
// 30 MB of textures were uploaded first — so the device warm-up asks for it, with the largest footprint any instantiation has.
__global__ void rt_warm_up_scratch(uint32_t* out, uint32_t n) {
  volatile uint32_t pad[32];   // 128 B per lane of private memory, indexed at run time: stays in scratch
  for (uint32_t i = 0; i < 32u; ++i) pad[(i + n) & 31u] = i * n;
  if (out && n == 0xFFFFFFFFu) out[threadIdx.x] = pad[threadIdx.x & 31u];   // (never true: keeps the array alive)
}

// --------------------------------------------------------------------------- queue order for the next frame
// tile_order <- the tiles sorted by descending tile_depth (counting sort over 64 depth buckets, one workgroup; the
// histograms and scatter cursors are private to each wave so that the bulk bucket — tiles without a deep path — does not
// serialise the whole workgroup on one LDS word).  Inside a bucket: bottom of the image first, up to the interleaving of
// the 16 waves.  Stream-ordered behind the frame that measured the depths.
__global__ __launch_bounds__(1024) void rt_order_tiles(const uint32_t* __restrict__ tile_depth, uint32_t* __restrict__ tile_order, uint32_t n_tiles,
                                                       uint32_t aff_group_log2) {
  // key = depth bucket (64 of them), or with XCD affinity (XCD of the tile, 32 depth buckets): the XCDs' segments one
  // after the other (they start at KArgs.xcd_off: both are the per-XCD tile counts), deepest first inside each
  __shared__ uint32_t hist[16][256], start[16][256], total[256];
  const uint32_t w = threadIdx.x >> 6;
  const bool aff = aff_group_log2 != 0xFFFFFFFFu;
  auto key_of = [&](uint32_t tile, uint32_t depth) -> uint32_t {
    if (!aff) return 63u - (depth < 63u ? depth : 63u);                                  // ascending key = descending depth
    return (tile_xcd(tile, aff_group_log2) << 5) | (31u - (depth < 31u ? depth : 31u));
  };
  for (uint32_t i = threadIdx.x; i < 16u * 256u; i += blockDim.x) (&hist[0][0])[i] = 0u;
  __syncthreads();
  for (uint32_t j = threadIdx.x; j < n_tiles; j += blockDim.x) {
    const uint32_t i = n_tiles - 1u - j;
    atomicAdd(&hist[w][key_of(i, tile_depth[i])], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 256u) {
    uint32_t o = 0;
    for (uint32_t k = 0; k < 16u; ++k) o += hist[k][threadIdx.x];
    total[threadIdx.x] = o;
  }
  __syncthreads();
  if (threadIdx.x < 256u) {  // key d starts after all smaller keys; inside it wave 0's tiles, then wave 1's, ...
    const uint32_t d = threadIdx.x;
    uint32_t o = 0;
    for (uint32_t e = 0; e < d; ++e) o += total[e];
    for (uint32_t k = 0; k < 16u; ++k) { start[k][d] = o; o += hist[k][d]; }
  }
  __syncthreads();
  for (uint32_t j = threadIdx.x; j < n_tiles; j += blockDim.x) {
    const uint32_t i = n_tiles - 1u - j;
    tile_order[atomicAdd(&start[w][key_of(i, tile_depth[i])], 1u)] = i;
  }
}

#ifdef RT_TEST_PROBES
// --------------------------------------------------------------------------- device self-test
// f64 sqrt / divide / f32 sqrt must be correctly rounded on the GPU for bit-parity with the CPU
// oracle; tests/test_gpu_parity.py checks these against numpy.
__global__ void rt_math_probe(const double* x, const double* y, double* out_sqrt, double* out_div, float* out_sqrtf,
                              double* out_atan2, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out_sqrt[i] = rt_sqrt(x[i]);  // the kernel's square root (RT_FAST_SQRT builds: the short sequence + its cold path)
  out_div[i] = x[i] / y[i];
  out_sqrtf[i] = __builtin_sqrtf((float)x[i]);
  out_atan2[i] = rt_atan2(x[i] - 0.5, y[i] - 0.5);  // the shared routine (csrc/common/rt_atan2.h): must equal its CPU build bit for bit
}

// the shared atan2 (csrc/common/rt_atan2.h) of n (y, x) pairs as the device evaluates it
__global__ void rt_atan2_probe(const double* y, const double* x, double* out, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = rt_atan2(y[i], x[i]);
}

// Sphere::hit on the device, one (ray, sphere) pair per thread, through the kernel's own hit test
// (closest-so-far = f64::MAX): out_t = accepted root or -1.  tests/test_gpu_parity.py compares it
// with the CPU build of the same function on random, tangent (discriminant 0 / denormal-range) and
// degenerate pairs — the cold paths a rendered frame practically never takes.
__global__ void rt_hit_probe(const double* rays, const double* spheres, double* out_t, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const V3 o = v3(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]), d = v3(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
  SphereGeom g; g.cx = spheres[4 * i]; g.cy = spheres[4 * i + 1]; g.cz = spheres[4 * i + 2]; g.r = spheres[4 * i + 3];
  const RayK rk = ray_consts(d);
  double closest = T_MAX;
  int best = -1;
  const bool hit = exact_hit_any_order(o, d, rk, g, 0u, closest, best);
  out_t[i] = hit ? closest : -1.0;
}


// The Texture hit's texel ON THE DEVICE, both ways (materials.rs:236-254 through sphere.rs:35-43): texel_fast — the
// plain-f64 (u, v) through v_rsq_f64 / v_rcp_f64 + Newton steps that only the device build takes — beside the exact path
// (three correctly rounded divisions + the shared double-double atan2).  out = n x {fast_ok, fast col, fast row, exact
// col, exact row}; uv = n x {fast u, fast v, exact u, exact v} (fast u = NaN: the fast path declined).
__global__ void rt_texel_probe(const double* points, double cx, double cy, double cz, double r, double h_offset, unsigned long long tex_w,
                               unsigned long long tex_h, unsigned long long* out, double* uv, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const V3 p = v3(points[3 * i], points[3 * i + 1], points[3 * i + 2]);
  SphereGeom g; g.cx = cx; g.cy = cy; g.cz = cz; g.r = r;
  uint64_t col = 0, row = 0;
  const bool ok = texel_fast(p, g, h_offset, tex_w, tex_h, col, row);
  const UV fa = fast_uv_core(p, g);
  const UV ex = sphere_uv(p, g);
  double rot = ex.u + h_offset;
  if (rot > 1.0) rot = rot - 1.0;
  out[5 * i] = ok ? 1ull : 0ull; out[5 * i + 1] = col; out[5 * i + 2] = row;
  out[5 * i + 3] = sat_u64(floor(rot * (double)tex_w)); out[5 * i + 4] = sat_u64(floor((1.0 - ex.v) * (double)(tex_h - 1)));
  if (uv) { uv[4 * i] = fa.u; uv[4 * i + 1] = fa.v; uv[4 * i + 2] = ex.u; uv[4 * i + 3] = ex.v; }
}
// rt_fast_quot(x, y), rt_fast_rsqrt(x) and rt_div_inrange(x, y) as the device evaluates them (x, y normal and positive)
__global__ void rt_quot_probe(const double* x, const double* y, double* out_quot, double* out_rsqrt, double* out_div, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out_quot[i] = rt_fast_quot(x[i], y[i]);
  out_rsqrt[i] = rt_fast_rsqrt(x[i]);
  if (out_div) out_div[i] = rt_div_inrange(x[i], y[i]);
}
#endif  // RT_TEST_PROBES

}  // namespace rtk
