// rt_core.h — per-lane path-tracing logic of the gfx950 megakernel.
//
// Everything here is a per-lane function: the wave-level orchestration (sphere scan with
// scalar-broadcast tables, LDS candidate lists, ballots, sample refill) lives in
// rt_kernel.hip.  The functions are __host__ __device__ so that tests/hostsim can run the
// very same arithmetic on the CPU for development; the product only ever runs them on the GPU.
//
// Arithmetic contract (what makes the GPU image match the CPU oracle):
//   * geometry is f64 with the reference's exact operation order and NO fused multiply-add
//     (compile with -ffp-contract=off; the only FMAs are the explicit ones in the f32 cull,
//     which never decides a hit on its own);
//   * colour is f32; the per-level `clamp(light + albedo*child)` recursion of
//     raytracer.rs:118-122 is carried forward as a clamped-affine map (struct Fwd), so the
//     f32 products associate differently from the recursion (documented tolerance);
//   * randomness is Philox4x32-10 addressed by (pixel, sample, node, slot) — order-free.
//
// Reference lines are cited as file:line relative to /root/reference/raytracer/src/.
#pragma once
#include <stdint.h>

#include "../../../include/rt_abi.h"

#if defined(__HIPCC__)
#define RT_ATAN2_FN __host__ __device__ inline
#endif
#include "../common/rt_atan2.h"  // the one atan2 kernel and CPU checker share (sphere.rs:39)

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ __forceinline__
// cold paths (texture lookups): a real call, so that their constants (atan2's polynomial) are
// not hoisted into registers that stay live across the whole path loop
#define RT_HD_COLD __host__ __device__ __attribute__((noinline))
#else
#include <cmath>
#define RT_HD inline
#define RT_HD_COLD inline
#endif

namespace rtc {

// ------------------------------------------------------------------ device scene tables
struct SphereGeom {  // 32 B, f64: sphere.rs:18-23 center + radius
  double cx, cy, cz, r;
};
struct SphereMat {  // 80 B: materials.rs:35-42 payloads
  float albedo[3];
  uint32_t kind;
  double fuzz_or_ior;
  double h_offset;
  uint64_t tex_w, tex_h;  // as written in the JSON (materials.rs:206-210)
  uint64_t tex_off;       // byte offset of this texture in the RGB8 blob (DevScene::tex; the general path)
  uint64_t tex_nbytes;
  // the texture as 4-byte texels (DevScene::tex4, texels_fast() below): ONE aligned dword load per Texture hit instead of
  // three byte loads behind 64-bit index arithmetic
  uint32_t texel_off;     // index of its first texel in DevScene::tex4
  uint32_t texel_last;    // tex_nbytes / 3 - 1: the texel an out-of-range index is clamped to
  uint32_t tex_fast;      // 1: take that path (the record's sizes are in texels_fast()'s range)
  uint32_t pad;
};
static_assert(sizeof(SphereMat) == 80, "SphereMat is 80 B");
// When the 4-byte-texel path returns what materials.rs:236-254 returns.  With width, height <= 2^24 and |h_offset| <= 1024
// the reference's index arithmetic cannot wrap: row <= height - 1 < 2^24 (v lies in [0, 1]), col <= 1025 * 2^24 < 2^35, so
// row * width + col < 2^49 — a column beyond 2^32 - 1 may be clamped there (the index is out of range either way) and
// v_mad_u64_u32 forms the index exactly.  Anything else keeps the u64 arithmetic on the RGB8 bytes.
RT_HD bool texels_fast(uint64_t tex_w, uint64_t tex_h, double h_offset, uint64_t nbytes, uint64_t texels_before) {
  // (height and width at least 1: with height 0 the reference's (height - 1) wraps to 2^64 - 1 and its row * width is taken
  //  mod 2^64 — such records keep the general u64 path that mirrors materials.rs word for word)
  return tex_w >= 1 && tex_h >= 1 && tex_w <= (1ull << 24) && tex_h <= (1ull << 24) && h_offset >= -1024.0 && h_offset <= 1024.0 && nbytes >= 3 &&
         texels_before + nbytes / 3 < (1ull << 31);
}
// f32 cull record for TWO spheres (SoA so one packed f32 instruction handles both):
// centre rounded to f32 and R = r^2 inflated by the rounding budget of that sphere.
struct CullPair {  // 32 B
  float cx[2], cy[2], cz[2], R[2];
};

#ifndef RT_CULL_CHUNK
#define RT_CULL_CHUNK 4
#endif
constexpr uint32_t CULL_CHUNK = RT_CULL_CHUNK;  // pairs per scan chunk; the table is padded to this

// Per-sphere fields every hit needs besides the geometry (48 B; the LDS copy of the material
// table).  Texture parameters stay in the 80 B SphereMat and are fetched only when a Texture
// sphere is hit.
struct MatCore {
  float albedo[3];
  uint32_t kind;
  double fuzz_or_ior;
  double inv_r;  // RN(1/radius) for div_by_recip (sphere.rs:60), or 0: divide the slow way
  // Glass: r0 of Schlick's reflectance (materials.rs:152-153: ((1 - ri) / (1 + ri))^2) for the two refraction ratios a
  // hit can have — [0] front face, ri = RN(1/ior); [1] back face, ri = ior — divided and squared once on the host
  // with the reference's own operations instead of one IEEE division per Glass hit
  double r0[2];
};
// Glass ignores its albedo (attenuation is white, materials.rs:178): those bytes carry 1/ior, the
// quotient materials.rs:181 computes at every front-face hit, divided once on the host instead.
RT_HD void matcore_set_inv_ior(MatCore& m, double inv_ior) {
  unsigned long long b;
  __builtin_memcpy(&b, &inv_ior, 8);
  uint32_t lo = (uint32_t)b, hi = (uint32_t)(b >> 32);
  __builtin_memcpy(&m.albedo[0], &lo, 4); __builtin_memcpy(&m.albedo[1], &hi, 4);
}
RT_HD double matcore_inv_ior(const MatCore& m) {
  uint32_t lo, hi;
  __builtin_memcpy(&lo, &m.albedo[0], 4); __builtin_memcpy(&hi, &m.albedo[1], 4);
  const unsigned long long b = ((unsigned long long)hi << 32) | lo;
  double d;
  __builtin_memcpy(&d, &b, 8);
  return d;
}

// Uniform grid over the scene's ordinary spheres (rt_tables.h builds it).  Oversized spheres
// (the r = 1000 ground of cover_scene.json) are kept out of it in the `large` list, which
// every ray tests.  Coordinates inside the walk are in CELL UNITS: x_cell = (x - gmin) * inv_cell.
struct GridDesc {
  double gmin[3];
  double inv_cell[3];
  uint32_t n[3];      // cells per axis; n[0] == 0: no grid (every sphere is in `large`)
  uint32_t n_large;
  uint32_t n_cells;   // PADDED table size (n[0]+2)*(n[1]+2)*(n[2]+2)
  uint32_t n_items;
  float pull;         // 8 * grid_walk_eps(max n): how far the crossing planes are pulled back
  uint32_t wide;      // 1: 32-bit item lists (more than 65 535 spheres, or a cell / item count beyond the packed word): cell entries are four words (below)
  double nd[3];       // n[] as doubles (the slab test's far planes)
};
constexpr uint32_t GRID_MAX_AXIS = 256;        // cells per axis (bounds the f32 error of the walk)
constexpr uint32_t CELL_COUNT_SHIFT = 20;      // cell word = first item | (item count << 20)
constexpr uint32_t CELL_START_MASK = (1u << CELL_COUNT_SHIFT) - 1u;
constexpr uint32_t CELL_MAX_COUNT = 4095;
// The cell table is padded by one layer of EXIT cells on every side: a walk that steps out of
// the grid reads this word and stops — no per-axis range checks in the step.
constexpr uint32_t CELL_EXIT = 0xFFFFFFFFu;
// WIDE tables (GridDesc.wide): a cell entry is {first item, item count, index of the first item (0xFFFFFFFF = none), 0} and the
// item list holds 32-bit sphere indices — no limit short of 2^32 on spheres, items or items per cell; EXIT cells have the
// first word CELL_EXIT as above.  Such scenes never fit LDS: the kernel's instantiations for them (rt_kernel.hip, WIDE) gather
// from L2 like every scene beyond the LDS budget.
constexpr uint32_t CELL_NO_ITEM32 = 0xFFFFFFFFu;
// Margins, in cells (DESIGN.md "Grid walk").  The walk is an incremental f32 DDA whose crossing
// times are off by at most eps(n) = n(n+1)u + 6u(n+3) cells of ray travel (u = 2^-24, n = the
// largest cell count of an axis; < 4.1e-3 for n <= 256).  It runs on crossing planes pulled
// back by GridDesc.pull = 8 eps towards the ray origin, so "closest hit before every crossing
// time" means the hit point really lies inside the current cell.  A sphere is listed in every
// cell its bounding box, grown by 2*pull > pull + eps, overlaps: every cell that contains a true
// hit point is either visited or lies within that margin of a visited cell that lists the
// sphere too.
RT_HD double grid_walk_eps(uint32_t n_max) {
  const double u = 5.9604644775390625e-08, n = (double)n_max;
  return n * (n + 1.0) * u + 6.0 * u * (n + 3.0);
}

struct DevScene {
  uint32_t width, height, spp, max_depth;
  uint32_t sky_mode, n_spheres, n_lights;
  uint32_t light_nest_pool;    // lit scenes: 1 (default) = nested light activations take pool records; 0 = always the HBM overflow ("light_nest_pool" option, tests)
  uint32_t seed_lo, seed_hi;
  uint32_t light_pool_slots;  // lit scenes: records in the workgroup's pool of light frames (LightState<true, true>)
  uint32_t cam_fast;          // inv_wm1 and inv_hm1 are both usable (width, height > 1): divide through them
  // raytracer.rs:92-100: a hit samples the lights when its draw exceeds 1 - n_lights * prob, prob = 0.1 (Glass: 0.05): the two
  // thresholds, computed once on the host with those operations — a wave-uniform scalar operand instead of two f64
  // constants the compiler hoisted into (and spilled from) vector registers of the path loop
  double light_thr[2];        // [0] prob 0.1, [1] Glass
  double cam_origin[3], cam_ll[3], cam_h[3], cam_v[3];
  double wm1, hm1, inv_wm1, inv_hm1, height_d;  // (width-1), (height-1), their RN reciprocals (0: slow divide), height
  const SphereGeom* geom;
  const SphereMat* mat;
  // lit scenes: where a SUSPENDED light activation goes when the workgroup's pool has no record for the nested one that
  // suspends it (80 B x (RT_MAX_LIGHT_NEST - 1) per lane of the launch, in HBM; practically never touched): the guarantee
  // that a lane holding records never waits for one (lane_light_begin).  (This pointer and the flag above sit where the
  // round-1 scan kernel's cull table and pair count sat: the kernel arguments keep the layout the unlit kernels were tuned with.)
  unsigned char* light_overflow;
  const uint32_t* lights;  // sphere indices of Light spheres, object order (raytracer.rs:220-229)
  const uint8_t* tex;      // all textures back to back, RGB8 (null on the device when every texture takes the 4-byte path)
  const uint8_t* sky;      // sky texture RGB8 (null on the device when sky_fast)
  uint64_t sky_w, sky_h;
  const uint32_t* tex4;    // the same texels, R | G << 8 | B << 16, 4 bytes each (SphereMat::texel_off indexes it)
  const uint32_t* sky4;
  uint32_t sky_fast;       // sky_w, sky_h <= 2^24 and sky_w * sky_h < 2^31: 24-bit index arithmetic on sky4
  float sky_wm1_f, sky_hm1_f;  // (float)(sky_w - 1), (float)(sky_h - 1) (raytracer.rs:149-150), converted once
  uint32_t light_base_slots;   // lit scenes with the short colour map: records in the workgroup's pool of colour-map bases (24 B each)
  GridDesc grid;
  const uint32_t* cell_word;   // [n_cells][2]: {first item | count << 20, first two item indices (u16 | u16 << 16, 0xFFFF = none)}; grid.wide: [n_cells][4], see CELL_NO_ITEM32
  const uint16_t* cell_items;  // [n_items] sphere indices, object order inside a cell (grid.wide: the same list as uint32_t, behind this pointer)
  const uint32_t* large;       // [n_large] sphere indices, object order
  const SphereGeom* large_geom;  // [n_large] their geometry, packed in the same order (streamed by scalar loads)
  const MatCore* matc;         // [n_spheres]
};

// ------------------------------------------------------------------ f64 square root
// IEEE sqrt.  The device library's sequence = range scaling (compare, select, ldexp) + v_rsq_f64 + nine
// mul/fma Newton-Goldschmidt steps + unscaling (ldexp) + the 0 / inf / NaN fix-up (class compare, two
// selects).  Arguments in [2^-500, 2^500] need neither scaling nor fix-up: the same nine steps on the
// unscaled value give the same bits (scaling by 2^256 only moves exponents), eight instructions shorter.
// Everything else (0, denormals, huge, negative, NaN) takes the library's sqrt on a cold path.
#ifndef RT_FAST_SQRT
#define RT_FAST_SQRT 1  // (-0.3 % kernel time, profiles/r02_run1_ab.log, r02_run2_ab.log)
#endif
#if defined(__HIP_DEVICE_COMPILE__) && RT_FAST_SQRT
__device__ __attribute__((noinline)) inline double rt_sqrt_cold(double x) { return sqrt(x); }
__device__ __forceinline__ double rt_sqrt(double x) {
  double g;
  if (x >= 0x1p-500 && x <= 0x1p+500) {
    const double y = __builtin_amdgcn_rsq(x);
    g = x * y;
    double h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
  } else g = rt_sqrt_cold(x);
  return g;
}
#else
RT_HD double rt_sqrt(double x) { return sqrt(x); }
#endif

// ------------------------------------------------------------------ point3d.rs
struct V3 {
  double x, y, z;
};
RT_HD V3 v3(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_HD V3 add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }           // :89-99
RT_HD V3 sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }           // :101-111
RT_HD V3 neg(V3 a) { return v3(-a.x, -a.y, -a.z); }                                // :113-123
RT_HD V3 muls(V3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }            // :137-147
RT_HD V3 divs(V3 a, double s) { return v3(a.x / s, a.y / s, a.z / s); }            // :161-171
RT_HD double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }         // :72-74
RT_HD double length_squared(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }    // :59-61
RT_HD double length(V3 a) { return rt_sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }   // :52-57, :63-65
RT_HD V3 unit_vector(V3 a) { double l = length(a); return v3(a.x / l, a.y / l, a.z / l); }  // :67-70
RT_HD bool near_zero(V3 a) {                                                       // :84-86
  const double eps = 2.220446049250313e-16;
  return fabs(a.x) < eps && fabs(a.y) < eps && fabs(a.z) < eps;
}

// x / b, correctly rounded, from y = RN(1/b): Markstein's division — q0 = RN(x*y) is within
// an ulp or so of x/b, and each fused correction q' = RN(q + RN(x - q*b) * y) (the residual is
// exact in one FMA) lands on RN(x/b) once q is faithful; two corrections are applied.  Bit-equal
// to the IEEE quotient for normal, finite x, b and x/b (tests/test_core_cpu.py checks 10^7 cases
// per run; the grid-walk audit re-checks every ray against true divisions).  Callers route
// anything outside that range (see RayK::fast, MatCore::inv_r == 0) to real divisions.
RT_HD double div_by_recip(double x, double b, double y) {
  double q = x * y;
  q = __builtin_fma(__builtin_fma(-q, b, x), y, q);
  return __builtin_fma(__builtin_fma(-q, b, x), y, q);
}
RT_HD bool recip_safe(double b) { double m = fabs(b); return m > 1e-150 && m < 1e150; }
// x / b, correctly rounded, for |b| in [1e-150, 1e150] and a quotient that is zero or of magnitude in [1e-290, 1e290]: the
// device library's own division — v_rcp_f64, two Newton steps, one residual correction — WITHOUT its range scaling
// (v_div_scale_f64 x 2, identity for such operands) and special-case fix-up (v_div_fixup_f64: zero / infinite / NaN
// operands): the same instructions on the same values, three fewer of them.  The CPU build divides.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double rt_recip_refined(double b) {  // the division's refined reciprocal estimate (NOT RN(1/b): only for rt_div_by_refined)
  double y = __builtin_amdgcn_rcp(b);
  y = __builtin_fma(y, __builtin_fma(-b, y, 1.0), y);
  return __builtin_fma(y, __builtin_fma(-b, y, 1.0), y);
}
__device__ __forceinline__ double rt_div_by_refined(double x, double b, double y) {  // x / b given y = rt_recip_refined(b): several quotients by one divisor share y
  const double q = x * y;
  return __builtin_fma(__builtin_fma(-b, q, x), y, q);
}
__device__ __forceinline__ double rt_div_inrange(double x, double b) { return rt_div_by_refined(x, b, rt_recip_refined(b)); }
#else
RT_HD double rt_recip_refined(double b) { return b; }
RT_HD double rt_div_by_refined(double x, double b, double) { return x / b; }
RT_HD double rt_div_inrange(double x, double b) { return x / b; }
#endif

struct Rgb {
  float r, g, b;
};
RT_HD Rgb rgb(float r, float g, float b) { Rgb c; c.r = r; c.g = g; c.b = b; return c; }
RT_HD float clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }  // raytracer.rs:61-69

// ------------------------------------------------------------------ Philox4x32-10
struct U4 {
  uint32_t x, y, z, w;
};
RT_HD U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int round = 0; round < 10; ++round) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;  // (the compiler emits one v_mad_u64_u32 per product on gfx950)
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  U4 r; r.x = c0; r.y = c1; r.z = c2; r.w = c3;
  return r;
}
// RNG addressing: counter = (pixel, sample, node, slot), key = seed.
//   node NODE_CAMERA, slot 0 : camera jitter                      (raytracer.rs:199-200)
//   node n, slot 0           : .x.y Glass reflectance draw         (materials.rs:189)
//                              .z.w light-sampling draw of a Glass hit (raytracer.rs:100); .z = the LOW word of every other hit's
//   node n, slot 1+a         : .x.y.z attempt a of random_in_unit_sphere (point3d.rs:31-38)
//   node n, slot 1           : .w = the HIGH word of the light-sampling draw of a hit that is not Glass: the word attempt 0's
//                              call leaves over, so a lit kernel decides `draw > threshold` without a second Philox stream —
//                              the low word (slot 0) matters only when the high word alone leaves the comparison open (2^-32)
constexpr uint32_t NODE_CAMERA = 0xFFFFFFFFu;
RT_HD double u01_53(uint32_t lo, uint32_t hi) {  // rand 0.8 Standard f64: (u64 >> 11) * 2^-53
  // u >> 11 = h * 2^32 + l with h = hi >> 11 (21 bits), l = the 32 bits below: the value h * 2^-21 + l * 2^-53 is a 53-bit
  // number < 1, so both products and their sum are exact — the same bits as converting the 64-bit integer and scaling it,
  // without the emulated u64 -> f64 conversion (the GPU has none)
  const uint32_t h = hi >> 11, l = (hi << 21) | (lo >> 11);
  return __builtin_fma((double)h, 1.0 / 2097152.0, (double)l * (1.0 / 9007199254740992.0));
}
RT_HD double range_m1_1(uint32_t w) {  // gen_range(-1.0..1.0) on a 2^-32 grid: v*(hi-lo)+lo with v = w * 2^-32
  // = (w * 2^-32) * 2.0 + (-1.0).  Every intermediate is exact in f64 (w < 2^32; the sum has at most 33 significant
  // bits), so ONE fused w * 2^-31 - 1 returns the same bits as the three rounded operations (the oracle's form): two
  // instructions fewer per coordinate, three coordinates per attempt of random_in_unit_sphere.
  return __builtin_fma((double)w, 1.0 / 2147483648.0, -1.0);
}
RT_HD uint32_t child_node(uint32_t node, uint32_t light_j) {
  uint32_t x = node * 0x9E3779B1u + (light_j + 1u) * 0x85EBCA77u;
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15;
  return 0x80000000u | (x & 0x7FFFFFFEu);
}

struct RngAddr {
  uint32_t pixel, sample, k0, k1;
};
RT_HD U4 rng(const RngAddr& a, uint32_t node, uint32_t slot) {
  uint32_t k0 = a.k0, k1 = a.k1;
#if defined(__HIP_DEVICE_COMPILE__)
  // the key is wave-uniform; without this the compiler hoists all 20 round keys out of the
  // path loop and pins 20 SGPRs for the whole kernel (they are 18 s_adds to recompute)
  k0 = __builtin_amdgcn_readfirstlane(k0); k1 = __builtin_amdgcn_readfirstlane(k1);
  asm volatile("" : "+s"(k0), "+s"(k1));
#endif
  return philox4x32_10(a.pixel, a.sample, node, slot, k0, k1);
}
// point3d.rs:22-38
RT_HD V3 random_in_unit_sphere(const RngAddr& a, uint32_t node) {
  for (uint32_t attempt = 0;; ++attempt) {
    U4 w = rng(a, node, 1u + attempt);
    V3 p = v3(range_m1_1(w.x), range_m1_1(w.y), range_m1_1(w.z));
    if (length_squared(p) < 1.0) return p;
  }
}

// ------------------------------------------------------------------ conservative f32 cull
// A sphere can only be hit (sphere.rs:51-53, discriminant >= 0) if the f32 quantity below is
// not negative.  With unit direction dh, oc = o - c:   disc/a = (oc.dh)^2 - |oc|^2 + r^2.
// Every f32 rounding (inputs o, c, dh rounded to f32, then ~20 operations) is covered by the
// margin  A*|oc|^2 + 2u|o|^2 + 2u|c|^2  with u = 2^-24, A = 2^-17  (derivation: DESIGN.md
// "Cull margin").  2u|c|^2 and r^2 are folded into CullPair.R on the host, 2u|o|^2 is the
// per-ray constant `Ko`.  The test is written `!(disc < 0)` so NaN/inf (overflowing or
// degenerate rays) PASS and fall through to the exact test — the cull can only over-accept.
constexpr float CULL_A = 7.62939453125e-06f;    // 2^-17
constexpr float CULL_2U = 1.1920928955078125e-07f;  // 2 * 2^-24

struct RayF32 {  // per-ray constants of the cull
  float ox, oy, oz, dx, dy, dz, Ko;
};
RT_HD RayF32 make_ray_f32(V3 o, V3 d) {
  RayF32 r;
  r.ox = (float)o.x; r.oy = (float)o.y; r.oz = (float)o.z;
  float fx = (float)d.x, fy = (float)d.y, fz = (float)d.z;
  float l2 = __builtin_fmaf(fz, fz, __builtin_fmaf(fy, fy, fx * fx));
#if defined(__HIP_DEVICE_COMPILE__)
  float inv = __builtin_amdgcn_rsqf(l2);  // 1 ulp; budgeted in CULL_A
#else
  float inv = 1.0f / sqrtf(l2);
#endif
  r.dx = fx * inv; r.dy = fy * inv; r.dz = fz * inv;
  r.Ko = CULL_2U * __builtin_fmaf(r.oz, r.oz, __builtin_fmaf(r.oy, r.oy, r.ox * r.ox));
  return r;
}
// scalar form (one sphere); the kernel evaluates the same sequence on packed pairs
RT_HD float cull_disc(const RayF32& r, float cx, float cy, float cz, float R) {
  float ocx = r.ox - cx, ocy = r.oy - cy, ocz = r.oz - cz;
  float b = __builtin_fmaf(ocz, r.dz, __builtin_fmaf(ocy, r.dy, ocx * r.dx));
  float q = __builtin_fmaf(ocz, ocz, __builtin_fmaf(ocy, ocy, ocx * ocx));
  float t = __builtin_fmaf(q, CULL_A - 1.0f, R + r.Ko);
  return __builtin_fmaf(b, b, t);
}
RT_HD bool cull_pass(float disc) { return !(disc < 0.0f); }

// host-side table construction (f64 -> f32 with outward rounding)
inline float f32_round_up(double x) {
  float f = (float)x;
  if ((double)f < x) f = nextafterf(f, INFINITY);
  return f;
}
inline void build_cull_entry(const RtSphere& s, float* cx, float* cy, float* cz, float* R) {
  *cx = (float)s.center[0]; *cy = (float)s.center[1]; *cz = (float)s.center[2];
  double c2 = s.center[0] * s.center[0] + s.center[1] * s.center[1] + s.center[2] * s.center[2];
  double r2 = s.radius * s.radius;
  *R = f32_round_up(r2 * (1.0 + 1e-6) + 2.0 * (double)CULL_2U * c2 + 1e-30);
}

// ------------------------------------------------------------------ exact hit (sphere.rs:46-58)
// Returns the root Sphere::hit would accept for this sphere given (t_min, t_max), or a
// negative number when it rejects.  Bit-identical arithmetic to the oracle.
RT_HD double exact_root(V3 o, V3 d, double a, const SphereGeom& g, double t_min, double t_max) {
  V3 oc = sub(o, v3(g.cx, g.cy, g.cz));
  double half_b = dot(oc, d);
  double c = length_squared(oc) - g.r * g.r;
  // Exact shortcut (no tolerance involved): origin outside the sphere (c > 0) and the centre
  // behind the ray (half_b > 0).  Then a*c >= 0, so sqrtd <= sqrt(fl(half_b^2)) == half_b and
  // both roots (-half_b -/+ sqrtd)/a are <= 0 < t_min: Sphere::hit rejects.  Half of the cull
  // survivors of a bounced ray are of this kind; this skips their sqrt and two divisions.
  if (c > 0.0 && half_b > 0.0) return -1.0;
  double discriminant = (half_b * half_b) - (a * c);
  if (discriminant >= 0.0) {
    double sqrtd = rt_sqrt(discriminant);
    double root_a = ((-half_b) - sqrtd) / a;
    if (root_a < t_max && root_a > t_min) return root_a;
    double root_b = ((-half_b) + sqrtd) / a;
    if (root_b < t_max && root_b > t_min) return root_b;
  }
  return -1.0;
}
constexpr double T_MIN = 0.001;                     // raytracer.rs:83
constexpr double T_MAX = 1.7976931348623157e308;    // f64::MAX
// T_MAX for the per-iteration `closest = f64::MAX` of the path loop: two moves made where they are needed.  As a plain
// constant the compiler hoists the pair out of the loop into registers that stay live across everything — and, in the lit
// kernels, spills them and reloads them from scratch memory every iteration.
RT_HD double t_max_fresh() {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t lo, hi;
  asm volatile("v_mov_b32 %0, -1" : "=v"(lo));
  asm volatile("v_mov_b32 %0, 0x7fefffff" : "=v"(hi));
  const unsigned long long b = ((unsigned long long)hi << 32) | lo;
  double d;
  __builtin_memcpy(&d, &b, 8);
  return d;
#else
  return T_MAX;
#endif
}

// Order-free form of the closest-hit scan (raytracer.rs:52-57).  The reference walks the
// spheres in object order and accepts sphere i iff its first root f_i in (t_min, inf) is
// < closest-so-far (sphere.rs:57-58: the far root is tried only when the near one is outside
// (t_min, t_max), and far >= near, so a far root is accepted only when near <= t_min).  The
// scan therefore returns the lexicographic minimum of (f_i, i).  Testing spheres in ANY order
// with "root < closest, or root == closest and i < best" reaches the same (t, sphere).
// Per-ray constants of the hit tests: a = |d|^2 (sphere.rs:48) and its reciprocal.
struct RayK {
  double a, inv_a;
  bool fast;  // the roots may be divided through inv_a (div_by_recip); else real divisions
};
RT_HD RayK ray_consts(V3 d) {
  RayK k;
  k.a = length_squared(d);
  k.fast = recip_safe(k.a);  // false for NaN too
  k.inv_a = rt_div_inrange(1.0, k.a);  // RN(1 / a) when `fast` (the only case it is used in)
  return k;
}
#ifndef RT_FLAT_HIT
#define RT_FLAT_HIT 1  // (-1.5 % kernel time, profiles/r02_run9_ab.log)
#endif
template <bool FAST>
RT_HD bool exact_hit_any_order_t(V3 o, V3 d, const RayK& rk, const SphereGeom& g, uint32_t idx, double& closest, int& best) {
  V3 oc = sub(o, v3(g.cx, g.cy, g.cz));
  double half_b = dot(oc, d);
  double c = length_squared(oc) - g.r * g.r;
#if RT_FLAT_HIT
  // one divergent region instead of four nested ones: both roots are always formed and the winner is taken by selects
  // (the nested form's merge points each copy closest / best; the far root costs 6 instructions more)
  const double discriminant = (half_b * half_b) - (rk.a * c);
  bool hit = false;
  if (!(c > 0.0 && half_b > 0.0) && discriminant >= 0.0) {  // (exact shortcut, see exact_root)
    const bool tie_ok = best >= 0 && idx < (uint32_t)best;
    const double sqrtd = rt_sqrt(discriminant);
    const double num_a = (-half_b) - sqrtd, num_b = (-half_b) + sqrtd;
    const double root_a = FAST ? div_by_recip(num_a, rk.a, rk.inv_a) : num_a / rk.a;
    const double root_b = FAST ? div_by_recip(num_b, rk.a, rk.inv_a) : num_b / rk.a;
    const bool ok_a = root_a > T_MIN && (root_a < closest || (tie_ok && root_a == closest));
    const bool ok_b = root_b > T_MIN && (root_b < closest || (tie_ok && root_b == closest));
    hit = ok_a || ok_b;
    const double root = ok_a ? root_a : root_b;
    closest = hit ? root : closest;
    best = hit ? (int)idx : best;
  }
  return hit;
#else
  if (c > 0.0 && half_b > 0.0) return false;  // exact shortcut, see exact_root
  double discriminant = (half_b * half_b) - (rk.a * c);
  if (discriminant >= 0.0) {
    const bool tie_ok = best >= 0 && idx < (uint32_t)best;
    double sqrtd = rt_sqrt(discriminant);
    double num = (-half_b) - sqrtd;
    double root = FAST ? div_by_recip(num, rk.a, rk.inv_a) : num / rk.a;
    if (!(root > T_MIN && (root < closest || (tie_ok && root == closest)))) {
      num = (-half_b) + sqrtd;
      root = FAST ? div_by_recip(num, rk.a, rk.inv_a) : num / rk.a;
      if (!(root > T_MIN && (root < closest || (tie_ok && root == closest)))) return false;
    }
    closest = root; best = (int)idx;
    return true;
  }
  return false;
#endif
}
// The part of the test above that decides "cannot be accepted" without a square root: the same operations in the same
// order (sphere.rs:47-53), so `may_hit == false` exactly when exact_hit_any_order_t returns false at its first branch.
struct HitPrefix {
  bool may_hit;
};
RT_HD HitPrefix exact_hit_prefix(V3 o, V3 d, const RayK& rk, const SphereGeom& g) {
  V3 oc = sub(o, v3(g.cx, g.cy, g.cz));
  double half_b = dot(oc, d);
  double c = length_squared(oc) - g.r * g.r;
  const double discriminant = (half_b * half_b) - (rk.a * c);
  HitPrefix p;
  p.may_hit = !(c > 0.0 && half_b > 0.0) && discriminant >= 0.0;
  return p;
}
// a ray whose |d|^2 is outside div_by_recip's range: the reference's own arithmetic (cold).
// Everything travels BY VALUE: a reference parameter of a real call would pin the caller's
// closest/best (the hottest variables of the kernel) to stack memory.
struct HitCB {
  double closest;
  int best;
};
RT_HD_COLD HitCB exact_hit_slow(V3 o, V3 d, double a, SphereGeom g, uint32_t idx, double closest, int best) {
  RayK rk; rk.a = a; rk.inv_a = 0.0; rk.fast = false;
  exact_hit_any_order_t<false>(o, d, rk, g, idx, closest, best);
  HitCB r; r.closest = closest; r.best = best;
  return r;
}
RT_HD bool exact_hit_any_order(V3 o, V3 d, const RayK& rk, const SphereGeom& g, uint32_t idx, double& closest, int& best) {
  if (rk.fast) return exact_hit_any_order_t<true>(o, d, rk, g, idx, closest, best);
  const HitCB r = exact_hit_slow(o, d, rk.a, g, idx, closest, best);
  const bool hit = r.best != best || r.closest != closest;
  closest = r.closest; best = r.best;
  return hit;
}

// ------------------------------------------------------------------ grid walk (hit_world, raytracer.rs:44-59)
// Incremental 3D-DDA over GridDesc in f32 cell units.  It only decides WHICH spheres get the
// exact f64 test; it can visit too many cells but never too few (DESIGN.md "Grid walk").
struct GridWalk {
  float tmax[3];   // time at which the ray crosses the (pulled-back) exit plane of the current cell, per axis
  float delta[3];  // time per cell, per axis (clamped to 1e30)
  int dl[3];       // change of the linear (padded) cell index per step, per axis
  int lin;         // linear index of the current cell in the padded table
  double t0;       // the walk's clock starts at the grid entry: t_ray = t0 + t_walk
};
enum { GRID_MISS = 0, GRID_WALK = 1, GRID_FALLBACK = 2 };

RT_HD int rt_mul24(int a, int b) {  // a * b for 0 <= a, b < 2^24
#if defined(__HIP_DEVICE_COMPILE__)
  return (int)__umul24((unsigned)a, (unsigned)b);
#else
  return a * b;
#endif
}
RT_HD float rt_rcpf(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcpf(x);  // 1 ulp; the walk's margins budget several
#else
  return 1.0f / x;
#endif
}
RT_HD float rt_clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }  // NaN stays NaN
RT_HD double rt_mind(double a, double b) { return a < b ? a : b; }
RT_HD double rt_maxd(double a, double b) { return a > b ? a : b; }
// The slab test of grid_begin on the device: one v_min_f64 / v_max_f64 / v_med3_f32 each instead of compare + selects (12 +
// 3 of them per ray).  They differ from the forms above only when an operand is NaN (minNum drops it; med3 returns a finite
// bound): a ray with a non-finite origin or direction — whose entry point is then NaN, fails the `sane` test and takes the
// full scan, or whose other axes already miss the grid's slab while no sphere can be hit through a NaN anyway.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double rt_slab_min(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ double rt_slab_max(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ float rt_slab_clamp(float x) { return __builtin_amdgcn_fmed3f(x, -1e30f, 1e30f); }
#else
inline double rt_slab_min(double a, double b) { return rt_mind(a, b); }
inline double rt_slab_max(double a, double b) { return rt_maxd(a, b); }
inline float rt_slab_clamp(float x) { return rt_clampf(x, -1e30f, 1e30f); }
#endif
RT_HD float rt_min3f(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_fminf(__builtin_fminf(a, b), c);  // v_min3_f32
#else
  return fminf(fminf(a, b), c);
#endif
}

// Prepare the walk of ray o + t*d.  GRID_MISS: the ray cannot touch any gridded sphere.
// GRID_FALLBACK: numerically unsafe (non-finite input, or entry too far away for f32): the
// caller tests every sphere exactly instead.
RT_HD int grid_begin(const GridDesc& G, V3 o, V3 d, GridWalk& w) {
  // (This function only SELECTS candidates — the margins below absorb its roundings — so it may fuse freely: the slab
  //  test and the entry point are 12 fused operations shorter than their literal form, every ray pays for them.)
  const double ol[3] = {(o.x - G.gmin[0]) * G.inv_cell[0], (o.y - G.gmin[1]) * G.inv_cell[1], (o.z - G.gmin[2]) * G.inv_cell[2]};
  const double dl[3] = {d.x * G.inv_cell[0], d.y * G.inv_cell[1], d.z * G.inv_cell[2]};
  float inv[3];
  double tn = 0.0, tf = T_MAX;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    inv[k] = rt_slab_clamp(rt_rcpf((float)dl[k]));
    const double invd = (double)inv[k];
    const double t1 = (-ol[k]) * invd, t2 = __builtin_fma(G.nd[k], invd, t1);  // (0 - ol) / dl, (n - ol) / dl
    tn = rt_slab_max(tn, rt_slab_min(t1, t2));
    tf = rt_slab_min(tf, rt_slab_max(t1, t2));
  }
  // the slab parameters carry the f32 reciprocal's relative error (conversion + 1 ulp: < 2.4e-7 = 2^-22): decide with 2^-16 slack.
  // (2^-12 until round 5: the entry point is taken that fraction of the way back towards the origin, and a ray whose origin is
  //  more than 2 * 4096 cells from the grid — back from the far side of the r = 1000 ground into a grid with thin cells — then
  //  lands more than two cells outside it, fails the `sane` test below and takes the FULL SCAN: one such ray in a frame of a
  //  2 x 10^5-sphere world cost 100 ms, profiles/r05_run22_grid_shape_probe.log.  Now: 2 * 65 536 cells.)
  const double slack = 1.0 / 65536.0;
  w.t0 = __builtin_fma(-tn, slack, tn);  // never later than the true entry; >= 0
  if (__builtin_fma(fabs(tf), slack, tf) < w.t0) return GRID_MISS;
  bool sane = true;
  int lin = 0, stride = 1;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double op = __builtin_fma(w.t0, dl[k], ol[k]);
    sane = sane && op >= -2.0 && op <= G.nd[k] + 2.0;
    const float of = sane ? (float)op : 0.0f;
    int i = (int)floorf(of);
    i = i < 0 ? 0 : (i > (int)G.n[k] - 1 ? (int)G.n[k] - 1 : i);
    const bool pos = inv[k] > 0.0f;
    const float b = (float)(i + (pos ? 1 : 0));       // exit plane of cell i along this axis
    w.delta[k] = fabsf(inv[k]);
    w.tmax[k] = __builtin_fmaf(b - of, inv[k], -G.pull * w.delta[k]);
    w.dl[k] = pos ? stride : -stride;
    lin += rt_mul24(i + 1, stride);                   // +1: the EXIT border (both factors < 2^24: v_mul_u32_u24 is full rate, v_mul_lo_u32 a quarter)
    stride *= (int)G.n[k] + 2;
  }
  if (!sane) return GRID_FALLBACK;
  w.lin = lin;
  return GRID_WALK;
}
// Move to the next cell along the ray (the caller reads its cell word; CELL_EXIT ends the walk).
// One axis moves by one cell per call and the table has an EXIT border, so a walk ends after
// at most n[0]+n[1]+n[2]+3 steps whatever the arithmetic does (NaN falls through to the z axis).
RT_HD void grid_step(GridWalk& w) {
  const float tmin = rt_min3f(w.tmax[0], w.tmax[1], w.tmax[2]);
  const bool sx = w.tmax[0] == tmin;
  const bool sy = !sx && w.tmax[1] == tmin;
  const bool sz = !sx && !sy;
  w.tmax[0] += sx ? w.delta[0] : 0.0f;
  w.tmax[1] += sy ? w.delta[1] : 0.0f;
  w.tmax[2] += sz ? w.delta[2] : 0.0f;
  w.lin += sx ? w.dl[0] : (sy ? w.dl[1] : w.dl[2]);
}
// True when the closest hit found so far lies inside the current cell, at least GridDesc.pull cells
// before every exit face: no sphere listed only in later cells can be closer.
RT_HD bool grid_done(const GridWalk& w, double closest) {
  float tc = (float)(closest - w.t0);
  tc = tc + fabsf(tc) * 2.384185791015625e-07f;  // round up past the conversion (2^-22)
  return tc < rt_min3f(w.tmax[0], w.tmax[1], w.tmax[2]);
}

// hit_world through the grid for ONE ray — the per-lane reference form of what the megakernel
// does with 64 lanes in lock-step (rt_kernel.hip); tests/hostsim runs this one on the CPU.
template <class Tables>
RT_HD void hit_world_grid(const DevScene& sc, const Tables& tb, V3 o, V3 d, double& closest, int& best,
                          uint32_t& n_exact, uint32_t& n_steps) {
  const GridDesc& G = sc.grid;
  const RayK a = ray_consts(d);
  for (uint32_t i = 0; i < G.n_large; ++i) {
    const uint32_t idx = sc.large[i];
    n_exact++;
    exact_hit_any_order(o, d, a, tb.geom(idx), idx, closest, best);
  }
  if (G.n[0] == 0u) return;
  GridWalk w;
  const int mode = grid_begin(G, o, d, w);
  if (mode == GRID_MISS) return;
  if (mode == GRID_FALLBACK) {
    for (uint32_t idx = 0; idx < sc.n_spheres; ++idx) { n_exact++; exact_hit_any_order(o, d, a, tb.geom(idx), idx, closest, best); }
    return;
  }
  uint32_t last = 0xFFFFFFFFu;
  const uint32_t* const items32 = reinterpret_cast<const uint32_t*>(sc.cell_items);
  for (;;) {
    const uint32_t word = sc.cell_word[(G.wide ? 4 : 2) * w.lin];
    if (word == CELL_EXIT) return;
    const uint32_t first = G.wide ? word : word & CELL_START_MASK, count = G.wide ? sc.cell_word[4 * w.lin + 1] : word >> CELL_COUNT_SHIFT;
    for (uint32_t k = 0; k < count; ++k) {
      const uint32_t idx = G.wide ? items32[first + k] : sc.cell_items[first + k];
      if (idx == last) continue;  // the sphere tested last (large spheres span consecutive cells)
      last = idx;
      n_exact++;
      exact_hit_any_order(o, d, a, tb.geom(idx), idx, closest, best);
    }
    if (best >= 0 && grid_done(w, closest)) return;
    n_steps++;
    grid_step(w);
  }
}

// ------------------------------------------------------------------ materials.rs
RT_HD V3 reflect(V3 v, V3 n) { return sub(v, muls(n, 2.0 * dot(v, n))); }  // :111-113
RT_HD V3 refract(V3 uv, V3 n, double etai_over_etat) {                     // :144-149
  double cos_theta = fmin(dot(neg(uv), n), 1.0);
  V3 r_out_perp = muls(add(uv, muls(n, cos_theta)), etai_over_etat);
  V3 r_out_parallel = muls(n, -1.0 * rt_sqrt(fabs(1.0 - length_squared(r_out_perp))));
  return add(r_out_perp, r_out_parallel);
}
RT_HD double reflectance_r0(double ref_idx) {                              // :152-153
  double r0 = (1.0 - ref_idx) / (1.0 + ref_idx);
  return r0 * r0;
}
RT_HD double reflectance_from_r0(double cosine, double r0) {               // :154
  double x = 1.0 - cosine;
  double x2 = x * x, x4 = x2 * x2;
  return r0 + (1.0 - r0) * (x * x4);  // powi(5)
}
RT_HD uint64_t sat_u64(double x) {  // Rust `f64 as u64`
  if (!(x > 0.0)) return 0;
  if (x >= 18446744073709551616.0) return ~0ull;
  return (uint64_t)x;
}
RT_HD uint64_t sat_u64_f32(float x) {  // Rust `f32 as usize`
  if (!(x > 0.0f)) return 0;
  if (x >= 18446744073709551616.0f) return ~0ull;
  return (uint64_t)x;
}
// x / 255.0f (materials.rs:247-252, raytracer.rs:153-158), correctly rounded, through y = RN(1/255) and ONE Markstein
// correction: 3 instructions instead of the ~10 of an IEEE f32 division, three times per texel.  Equal to the IEEE quotient
// for EVERY float in [0, 256] (tools/analysis/div255_check.cpp: all 1.13e9 of them) — the kernel divides bytes and 0.7f * bytes.
RT_HD float rt_div255f(float x) {
  const float y = 0.003921568859368563f;  // RN(1 / 255) = 0x1.010102p-8
  const float q = x * y;
  return __builtin_fmaf(__builtin_fmaf(-q, 255.0f, x), y, q);
}
// materials.rs:236-254 (out-of-range index: clamp + count; the reference panics)
RT_HD Rgb texel_fetch(const DevScene& sc, const SphereMat& m, uint64_t col, uint64_t row, uint32_t& tex_oob) {
  uint64_t base_pixel = 3 * (row * m.tex_w + col);
  if (m.tex_nbytes < 3) { tex_oob++; return rgb(0.f, 0.f, 0.f); }
  if (base_pixel > m.tex_nbytes - 3) { tex_oob++; base_pixel = (m.tex_nbytes / 3 - 1) * 3; }
  const uint8_t* px = sc.tex + m.tex_off + base_pixel;
  return rgb(rt_div255f((float)px[0]), rt_div255f((float)px[1]), rt_div255f((float)px[2]));
}
RT_HD uint32_t sat_u32(double x) {  // min(Rust's `f64 as u64`, 2^32 - 1): NaN and negatives -> 0
  const double c = fmin(fmax(x, 0.0), 4294967295.0);  // (fmax / fmin drop a NaN operand)
  return (uint32_t)c;
}
RT_HD Rgb rgb_of_texel(uint32_t w) {  // bytes -> colour / 255 (materials.rs:247-252): v_cvt_f32_ubyte0/1/2 + rt_div255f
  return rgb(rt_div255f((float)(w & 0xFFu)), rt_div255f((float)((w >> 8) & 0xFFu)), rt_div255f((float)((w >> 16) & 0xFFu)));
}
RT_HD Rgb texture_albedo(const DevScene& sc, const SphereMat& m, double u, double v, uint32_t& tex_oob) {
  double rot = u + m.h_offset;
  if (rot > 1.0) rot = rot - 1.0;
  double uu = rot * (double)m.tex_w;
  double vv = (1.0 - v) * (double)(m.tex_h - 1);
  const double fu = floor(uu), fv = floor(vv);
  if (m.tex_fast) {  // (texels_fast: the same texel by construction — one multiply-add, one compare, one dword load)
    const unsigned long long pi = (unsigned long long)sat_u32(fv) * (unsigned long long)(uint32_t)m.tex_w + (unsigned long long)sat_u32(fu);
    uint32_t idx = (uint32_t)pi;
    if (pi > (unsigned long long)m.texel_last) { tex_oob++; idx = m.texel_last; }
    return rgb_of_texel(sc.tex4[m.texel_off + idx]);
  }
  return texel_fetch(sc, m, sat_u64(fu), sat_u64(fv), tex_oob);
}

// ------------------------------------------------------------------ colour: forward form
// ray_color returns clamp01(light + albedo * child) at every level (raytracer.rs:118-122).
// Composing those maps outermost-first gives, per channel, G(x) = min(max(p + q*x, lo), hi):
//   G o f (x),  f(x) = clamp01(L + a*x)   =>   p' = p + q*L,  q' = q*a,
//                                              [lo', hi'] = sorted {G(0), G(1)}
// (G is monotone, so G(clamp01(z)) = clamp(G(z), G(0), G(1))).  Exact in real arithmetic for
// any sign of a; in f32 the products associate outermost-first instead of innermost-first.
// SIMPLE scenes (no lights, every albedo in [0,1]; rt_tables.h decides): L is always 0 and the
// clamps never bind, so the map degenerates to G(x) = q*x — three floats instead of twelve,
// with bit-identical results (p stays +0, lo = 0, hi = previous q >= q*x).
template <bool SIMPLE>
struct FwdT {
  float p[3], q[3], lo[3], hi[3];
};
template <>
struct FwdT<true> {
  float q[3];
};
typedef FwdT<false> Fwd;
RT_HD void fwd_init(FwdT<false>& f) {
#pragma unroll
  for (int i = 0; i < 3; ++i) { f.p[i] = 0.0f; f.q[i] = 1.0f; f.lo[i] = -3.4028234663852886e38f; f.hi[i] = 3.4028234663852886e38f; }
}
RT_HD void fwd_init(FwdT<true>& f) {
#if defined(__HIP_DEVICE_COMPILE__)
  // three moves where a sample begins: as plain constants the compiler kept a (1.0f, 1.0f) register pair alive across the
  // whole path loop for this — and, in the lit kernels, spilled and reloaded it every iteration
  float one;
  asm volatile("v_mov_b32 %0, 1.0" : "=v"(one));
  f.q[0] = one; f.q[1] = one; f.q[2] = one;
#else
  f.q[0] = f.q[1] = f.q[2] = 1.0f;
#endif
}
RT_HD float fwd_eval1(const FwdT<false>& f, int i, float x) {
  float y = f.p[i] + f.q[i] * x;
  y = y < f.lo[i] ? f.lo[i] : y;
  y = y > f.hi[i] ? f.hi[i] : y;
  return y;
}
RT_HD float fwd_eval1(const FwdT<true>& f, int i, float x) { return 0.0f + f.q[i] * x; }
RT_HD void fwd_compose(FwdT<false>& f, const float L[3], const float a[3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float g0 = fwd_eval1(f, i, 0.0f), g1 = fwd_eval1(f, i, 1.0f);
    f.p[i] = f.p[i] + f.q[i] * L[i];
    f.q[i] = f.q[i] * a[i];
    f.lo[i] = g0 < g1 ? g0 : g1;
    f.hi[i] = g0 < g1 ? g1 : g0;
  }
}
RT_HD void fwd_compose(FwdT<true>& f, const float*, const float a[3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i) f.q[i] = f.q[i] * a[i];
}

// raytracer.rs:134-160: colour of a ray that left the scene
RT_HD Rgb sky_color(const DevScene& sc, V3 d, uint32_t& tex_oob) {
  if (sc.sky_mode == RT_SKY_NONE) return rgb(0.0f, 0.0f, 0.0f);
  // unit(dir).y and .x as f32 (raytracer.rs:135-136).  |d|^2 in [2^-500, 2^500] (every ordinary ray): the square root's short
  // form and the division's core without range scaling, the two quotients sharing one refined reciprocal — the f64
  // quotients are the IEEE ones unless they are so small (< 2^-300) that their f32 conversion is a signed zero either way.
  const double a = length_squared(d);
  const bool want_x = sc.sky_mode != RT_SKY_GRADIENT;
  double qy, qx = 0.0;
  if (a >= 0x1p-500 && a <= 0x1p+500) {
    const double l = rt_sqrt(a), yl = rt_recip_refined(l);
    qy = rt_div_by_refined(d.y, l, yl);
    if (want_x) qx = rt_div_by_refined(d.x, l, yl);
  } else {
    const double l = rt_sqrt(a);
    qy = d.y / l;
    if (want_x) qx = d.x / l;
  }
  float t = clamp01(0.5f * ((float)qy + 1.0f));
  if (!want_x)
    return rgb((1.0f - t) * 1.0f + t * 0.5f, (1.0f - t) * 1.0f + t * 0.7f, (1.0f - t) * 1.0f + t * 1.0f);
  float u = clamp01(0.5f * ((float)qx + 1.0f));
  const float xf = u * sc.sky_wm1_f, yf = (1.0f - t) * sc.sky_hm1_f;  // raytracer.rs:149-150
  if (sc.sky_fast) {
    // u and 1 - t lie in [0, 1] (or are NaN), so xf <= sky_w - 1 < 2^24 and yf <= sky_h - 1 < 2^24 exactly: `as usize` is a
    // plain conversion, the index a 24-bit multiply-add below 2^31, and it cannot leave the texture (the check stays)
    const uint32_t x = !(xf > 0.0f) ? 0u : (uint32_t)xf, y = !(yf > 0.0f) ? 0u : (uint32_t)yf;
    const uint32_t last = (uint32_t)sc.sky_w * (uint32_t)sc.sky_h - 1u;
    uint32_t idx = (uint32_t)rt_mul24((int)y, (int)(uint32_t)sc.sky_w) + x;
    if (idx > last) { tex_oob++; idx = last; }
    const uint32_t w = sc.sky4[idx];
    return rgb(rt_div255f(0.7f * (float)(w & 0xFFu)), rt_div255f(0.7f * (float)((w >> 8) & 0xFFu)), rt_div255f(0.7f * (float)((w >> 16) & 0xFFu)));
  }
  uint64_t x = sat_u64_f32(xf);
  uint64_t y = sat_u64_f32(yf);
  uint64_t base = (y * sc.sky_w + x) * 3;
  if (base + 2 >= sc.sky_w * sc.sky_h * 3) { tex_oob++; base = (sc.sky_w * sc.sky_h - 1) * 3; }
  const uint8_t* px = sc.sky + base;
  return rgb(rt_div255f(0.7f * (float)px[0]), rt_div255f(0.7f * (float)px[1]), rt_div255f(0.7f * (float)px[2]));
}

// ------------------------------------------------------------------ table access
// Where a lane reads the per-sphere records from: the HBM tables (this struct; also what
// tests/hostsim uses) or the workgroup's LDS copies (rt_kernel.hip::LdsTables).
struct GlobalTables {
  const SphereGeom* g;
  const MatCore* m;
  RT_HD SphereGeom geom(uint32_t i) const { return g[i]; }
  RT_HD MatCore mat(uint32_t i) const { return m[i]; }
  // centre of light j (raytracer.rs:104-105 aims at it): the kernel's table types answer from a small LDS copy instead of
  // two dependent global loads (a light ray starts in nearly every wave iteration of a lit scene)
  RT_HD V3 light_centre(const DevScene& sc, uint32_t j) const { const SphereGeom lg = g[sc.lights[j]]; return v3(lg.cx, lg.cy, lg.cz); }
};

// ------------------------------------------------------------------ scatter (materials.rs:44-54)
enum { SCATTER_ABSORBED = 0, SCATTER_EMIT = 1, SCATTER_RAY = 2 };
struct Surface {  // what Sphere::hit records for the accepted root (sphere.rs:59-75)
  V3 point, normal;
  bool front_face;
};
RT_HD_COLD V3 divs_slow(V3 a, double s) { return divs(a, s); }
RT_HD Surface surface_at(V3 o, V3 d, double t, const SphereGeom& g, double inv_r) {
  Surface s;
  s.point = add(o, muls(d, t));                                  // ray.rs:18-20
  V3 pc = sub(s.point, v3(g.cx, g.cy, g.cz));
  V3 outward;                                                    // sphere.rs:60: (p - c) / r
  if (inv_r != 0.0) outward = v3(div_by_recip(pc.x, g.r, inv_r), div_by_recip(pc.y, g.r, inv_r), div_by_recip(pc.z, g.r, inv_r));
  else outward = divs_slow(pc, g.r);
  s.front_face = dot(d, outward) < 0.0;                          // :61
  s.normal = s.front_face ? outward : neg(outward);              // :68
  return s;
}
// sphere.rs:35-43, evaluated only when the closest hit is a Texture (a pure function of the
// accepted hit, so skipping it for the other candidates changes nothing)
struct UV {
  double u, v;
};
RT_HD_COLD UV sphere_uv(V3 point, SphereGeom g) {  // by value: see exact_hit_slow
  V3 n = unit_vector(sub(point, v3(g.cx, g.cy, g.cz)));
  UV r;
  r.u = (rt_atan2(n.x, n.z) / (2.0 * 3.14159265358979323846264338327950288)) + 0.5;
  r.v = n.y * 0.5 + 0.5;
  return r;
}
// ---- the texel of a Texture hit without the exact (u, v) ---------------------------------------------------------
// sphere_uv + texture_albedo only ever use (u, v) through floor(rot * width) and floor((1 - v) * (height - 1)): three
// correctly-rounded divisions, a correctly-rounded double-double atan2 and a fourth division (270 instructions) to pick
// one of a few thousand columns.  texel_fast computes the same two numbers in plain f64 with |error| < 1e-14 (unit
// vector through one reciprocal square root; atan through two quotients, a five-point table and eight Taylor terms)
// and reports a texel only when both lie further than TEXEL_EPS x size from the next integer and the wrap test
// `rot > 1` is further than TEXEL_EPS from deciding differently: then the exact values floor to the same texel.
// Anything else — a boundary closer than that, a non-finite or negative coordinate, a texture wider than 1e11 pixels —
// answers false and the caller takes the exact path.  (tests/test_core_cpu.py: 10^7 hit points against the exact path,
// with the boundary cases aimed at; audited again per Texture hit by hostsim mode 4.)
RT_HD double rt_nan() { return __builtin_nan(""); }
constexpr double TEXEL_EPS = 4e-12;  // >> the fast path's error in u and v (< 1e-14), << a texel of any real texture
RT_HD double rt_fast_quot(double x, double b) {  // x / b to ~1.5 ulp, b normal and positive
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rcp(b);
  y = __builtin_fma(__builtin_fma(-b, y, 1.0), y, y);
  y = __builtin_fma(__builtin_fma(-b, y, 1.0), y, y);
  const double q = x * y;
  return __builtin_fma(__builtin_fma(-q, b, x), y, q);
#else
  return x / b;
#endif
}
// 1 / sqrt(x) to a few ulp, x normal and positive (v_rsq_f64 + two Newton steps on the device; the device sequences of
// this and of rt_fast_quot are probed ON THE GPU by rt_hip_texel_probe / tests/test_gpu_parity.py: the CPU build
// takes the library's operations instead, so CPU tests of texel_fast do not exercise them)
RT_HD double rt_fast_rsqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double rs = __builtin_amdgcn_rsq(x);
  rs = rs * __builtin_fma(__builtin_fma(-x * rs, rs, 1.0), 0.5, 1.0);
  rs = rs * __builtin_fma(__builtin_fma(-x * rs, rs, 1.0), 0.5, 1.0);
  return rs;
#else
  return 1.0 / sqrt(x);
#endif
}
// (Three small real calls instead of one function: interprocedural register allocation lets the kernel keep its lane
//  state in whatever registers its callees leave alone; sphere_uv takes 29, and a cold function that needs 52 costs
//  the path loop 17 spilled registers — 4 % of the whole frame, textures or not.)
RT_HD UV fast_uv_core(V3 point, const SphereGeom& g) {  // (u, v) of sphere_uv to ~1e-15, or u = NaN: ask the exact path
  UV out; out.u = rt_nan(); out.v = 0.0;
  V3 pc = sub(point, v3(g.cx, g.cy, g.cz));
  const double l2 = length_squared(pc);
  if (!(l2 > 1e-280 && l2 < 1e280)) return out;
  const double rs = rt_fast_rsqrt(l2);
  const double nx = pc.x * rs, nz = pc.z * rs;
  out.v = __builtin_fma(pc.y * rs, 0.5, 0.5);
  // angle = atan2(nx, nz) in [-pi, pi]
  const double ax = fabs(nx), az = fabs(nz);
  const double mx = ax > az ? ax : az, mn = ax > az ? az : ax;
  if (!(mx > 1e-3)) return out;                 // (a unit vector has max(|x|, |z|) > 1e-3 unless it points along y)
  const double t = rt_fast_quot(mn, mx);          // in [0, 1]
  const double kf = floor(t * 4.0 + 0.5);         // nearest of c = 0, 1/4, 1/2, 3/4, 1
  const double c = kf * 0.25;
  const double atan_c = kf < 0.5 ? 0.0 : (kf < 1.5 ? 0.24497866312686414 : (kf < 2.5 ? 0.4636476090008061 : (kf < 3.5 ? 0.6435011087932844 : 0.7853981633974483)));
  const double z = rt_fast_quot(t - c, __builtin_fma(t, c, 1.0));  // |z| <= 1/8: atan(t) = atan(c) + atan(z)
  const double s = z * z;
  double p = -1.0 / 15.0;
  p = __builtin_fma(p, s, 1.0 / 13.0);
  p = __builtin_fma(p, s, -1.0 / 11.0);
  p = __builtin_fma(p, s, 1.0 / 9.0);
  p = __builtin_fma(p, s, -1.0 / 7.0);
  p = __builtin_fma(p, s, 1.0 / 5.0);
  p = __builtin_fma(p, s, -1.0 / 3.0);
  double r = atan_c + __builtin_fma(p * s, z, z);  // atan(mn / mx) in [0, pi/4]
  if (ax > az) r = 1.5707963267948966 - r;         // atan(|nx| / |nz|)
  if (nz < 0.0) r = 3.141592653589793 - r;
  if (nx < 0.0) r = -r;
  out.u = __builtin_fma(r, 0.15915494309189535, 0.5);
  return out;
}
// does (u, v) +- TEXEL_EPS name one texel?  Then (col, row) is what the exact (u, v) floors to as well.
RT_HD bool texel_sure(double u, double v, double h_offset, uint64_t tex_w, uint64_t tex_h, uint64_t& col, uint64_t& row) {
  const double wd = (double)tex_w, hd = (double)(tex_h - 1);
  double rot = u + h_offset;
  if (!(fabs(rot - 1.0) > TEXEL_EPS)) return false;  // (false for NaN as well: every comparison below is, too)
  if (rot > 1.0) rot = rot - 1.0;
  const double uu = rot * wd, vv = (1.0 - v) * hd;
  const double fu = floor(uu), fv = floor(vv);
  const double du = uu - fu, dv = vv - fv;
  const double eu = wd * TEXEL_EPS, ev = hd * TEXEL_EPS;
  const bool row_sure = (dv > ev && 1.0 - dv > ev) || (hd == 0.0 && vv == 0.0);  // (a one-row texture: vv = (1 - v) * 0)
  if (!(du > eu && 1.0 - du > eu && row_sure && fu >= 0.0 && fv >= 0.0 && fu < 4.5e15 && fv < 4.5e15)) return false;
  col = (uint64_t)fu; row = (uint64_t)fv;
  return true;
}
RT_HD bool texel_fast(V3 point, const SphereGeom& g, double h_offset, uint64_t tex_w, uint64_t tex_h, uint64_t& col, uint64_t& row) {
  const UV a = fast_uv_core(point, g);
  return texel_sure(a.u, a.v, h_offset, tex_w, tex_h, col, row);
}
// (u, v) for texture_albedo: the fast pair when it is sure of its texel (texture_albedo repeats texel_sure's arithmetic
// on it, bit for bit, and so floors to that texel), else u = NaN: take sphere_uv.
// A LEAF on purpose: as a function that may call sphere_uv itself it had to keep its return address in a VGPR lane and
// spill that register around the call — one 256-byte scratch store and load per wave and Texture hit, 5.65 GB of HBM
// writes per 4K textured frame (profiles/r03_run40_pmc_cfg3.json).  The caller asks sphere_uv when u comes back NaN.
RT_HD_COLD UV fast_uv_for_texel(V3 point, SphereGeom g, const SphereMat* mats, uint32_t idx) {
  UV a = fast_uv_core(point, g);
  uint64_t col, row;
  if (!texel_sure(a.u, a.v, mats[idx].h_offset, mats[idx].tex_w, mats[idx].tex_h, col, row)) a.u = rt_nan();
  return a;
}
RT_HD UV sphere_uv_for_texel(V3 point, const SphereGeom& g, const SphereMat* mats, uint32_t idx) {
  const UV a = fast_uv_for_texel(point, g, mats, idx);
  if (a.u == a.u) return a;
  return sphere_uv(point, g);  // ~1e-8 of the hits: a texel boundary closer than TEXEL_EPS, the poles, non-finite input
}
// unit_vector (point3d.rs:67-70) with one real division: 1/l, then div_by_recip per component
RT_HD V3 unit_vector_fast(V3 a) {
  double l = length(a);
  if (!recip_safe(l)) return v3(a.x / l, a.y / l, a.z / l);
  double y = rt_div_inrange(1.0, l);
  return v3(div_by_recip(a.x, l, y), div_by_recip(a.y, l, y), div_by_recip(a.z, l, y));
}
RT_HD bool material_draws_unit_sphere(uint32_t kind) {
  return kind == RT_MAT_LAMBERTIAN || kind == RT_MAT_TEXTURE || kind == RT_MAT_METAL;
}
// rnd_pre / glass_u_pre: the random_in_unit_sphere(ra, node) point and the Glass reflectance draw
// (slot 0, .x.y) if the caller already drew them, or null
RT_HD int scatter(const DevScene& sc, const RngAddr& ra, uint32_t node, V3 in_dir, const Surface& h,
                  const SphereGeom& g, const MatCore& m, uint32_t idx, V3& out_dir, float att[3], uint32_t& tex_oob,
                  const V3* rnd_pre = nullptr, const double* glass_u_pre = nullptr) {
  // Lambertian, Texture and Metal all draw random_in_unit_sphere (Metal even with fuzz = 0,
  // materials.rs:120); one shared rejection loop instead of one per material branch.
  V3 rnd = v3(0.0, 0.0, 0.0);
  if (material_draws_unit_sphere(m.kind)) rnd = rnd_pre ? *rnd_pre : random_in_unit_sphere(ra, node);
  att[0] = m.albedo[0]; att[1] = m.albedo[1]; att[2] = m.albedo[2];
  switch (m.kind) {
    case RT_MAT_LIGHT:  // :65-69
      att[0] = att[1] = att[2] = 1.0f;
      return SCATTER_EMIT;
    case RT_MAT_LAMBERTIAN:  // :84-95
    case RT_MAT_TEXTURE: {   // :256-267
      V3 sd = add(h.normal, rnd);
      if (near_zero(sd)) sd = h.normal;
      V3 target = add(h.point, sd);
      out_dir = sub(target, h.point);  // (p + d) - p, as the reference computes it
      if (m.kind == RT_MAT_TEXTURE) {
        const UV uv = sphere_uv_for_texel(h.point, g, sc.mat, idx);
        Rgb a = texture_albedo(sc, sc.mat[idx], uv.u, uv.v, tex_oob);
        att[0] = a.r; att[1] = a.g; att[2] = a.b;
      }
      return SCATTER_RAY;
    }
    case RT_MAT_METAL: {  // :115-129
      V3 reflected = reflect(in_dir, h.normal);
      out_dir = add(reflected, muls(rnd, m.fuzz_or_ior));
      return dot(out_dir, h.normal) > 0.0 ? SCATTER_RAY : SCATTER_ABSORBED;
    }
    case RT_MAT_GLASS: {  // :176-199
      att[0] = att[1] = att[2] = 1.0f;
      double refraction_ratio = h.front_face ? matcore_inv_ior(m) : m.fuzz_or_ior;  // :180-184 (1/ior precomputed)
      V3 unit_direction = unit_vector_fast(in_dir);
      double cos_theta = fmin(dot(neg(unit_direction), h.normal), 1.0);
      // :187-188  cannot_refract = refraction_ratio * sqrt(1 - cos^2) > 1.  The square root is only compared: with
      // s2 = 1 - cos^2 in [0, 1], ratio < 1 makes RN(ratio * sqrt(s2)) <= ratio < 1 (false, also for NaN); otherwise
      // ratio^2 * s2 decides unless it lies within 1e-9 of 1 (the two roundings of either form are ~1e-16), and only
      // that sliver (or a NaN) evaluates the reference's expression — one library sqrt less per Glass hit.
      const double s2 = 1.0 - cos_theta * cos_theta;
      bool do_reflect = false;
      if (!(refraction_ratio < 1.0)) {
        const double t2 = (refraction_ratio * refraction_ratio) * s2;
        if (t2 > 1.000000001) do_reflect = true;
        else if (!(t2 < 0.999999999)) do_reflect = refraction_ratio * rt_sqrt(s2) > 1.0;
      }
      if (!do_reflect) {  // (the draw is addressed by counter: taking it early or not at all changes nothing else)
        double u;
        if (glass_u_pre) u = *glass_u_pre;
        else { U4 w = rng(ra, node, 0); u = u01_53(w.x, w.y); }
        do_reflect = reflectance_from_r0(cos_theta, m.r0[h.front_face ? 0 : 1]) > u;
      }
      if (do_reflect) out_dir = reflect(unit_direction, h.normal);
      else out_dir = refract(unit_direction, h.normal, refraction_ratio);
      return SCATTER_RAY;
    }
    default:
      att[0] = att[1] = att[2] = 0.0f;
      return SCATTER_ABSORBED;
  }
}

// ------------------------------------------------------------------ per-lane path state machine
// One lane owns one pixel and walks its samples one ray segment at a time.  Each call of
// lane_shade() consumes the closest hit of the lane's CURRENT ray (camera-path segment or
// nested light ray, raytracer.rs:103-110) and leaves the NEXT ray in (o, d), so that the
// expensive part — the sphere scan — is always executed by all live lanes together.
struct LightFrame {  // one ray_color activation that is summing over the lights
  V3 P;              // hit point = origin of its light rays
  float a[3];        // its albedo
  float acc[3];      // light_red/green/blue so far (raytracer.rs:89-91)
  uint32_t j;        // next light
  uint32_t node;     // its RNG node (children are child_node(node, j))
};
// Where the light state of a lane lives.
//
// HOST form (LightState<true, false>: tests/hostsim, the CPU build of this header): one LightParked per lane (`pk`, a local
// of the simulator) for the activation that is summing over the lights right now, a LightStack for the suspended outer
// activations (nesting levels below the active one) and for the base of the short colour map (lane_compose).
//
// DEVICE form (LightState<true, true>: every lit kernel): nothing of it is a per-lane object in memory.  Until round 5 the
// stack and the base were one — 416 B per lane, indexed by the nesting level and so in SCRATCH: 480 – 528 B per lane, a write
// to HBM whenever some lane of a wave returned from its light loop (27 – 38 x the framebuffer's bytes per frame,
// profiles/r04_run20_pmc_litcover.json).  Now the workgroup shares two POOLS in LDS, each a bitmap + records:
//   * FRAMES (LightParked, 80 B): the activation summing over the lights — hit point, albedo, sum, next light, RNG node —
//     and the camera path's pending direction.  Taken (one LDS atomic) when a camera-path hit starts sampling the lights,
//     given back when that activation returns to the camera path: ~6 % of the lanes hold one at any moment (one light).
//     A NESTED activation (a light ray's own hit samples the lights again, probability 0.1 n_lights) takes a record of
//     its own and LINKS it to the suspended one's (the link overlays `saved_d`, which only a camera-path activation
//     needs): nothing is copied, push and pop are one word each.
//   * BASES (24 B): p[3], h[3] of the short colour map once a light contributed (lane_compose) — taken together with the
//     sample's first frame, given back when the SAMPLE ends (~20 % of the lanes of a one-light scene).  The lane carries
//     its record's offset in the spare high half of `in_light`.
// A camera-path hit that finds either pool exhausted does NOT shade its hit: it leaves its ray as it is and traces the
// same segment again in the next iteration (LANE_REPEAT; every draw is addressed by counter, so the repeat decides the
// same).  For that to end, a lane that HOLDS records must never wait for one: a nested activation that finds the frame
// pool exhausted does not wait — the suspended record is copied out to the lane's own overflow slot in HBM
// (DevScene::light_overflow, 80 B x 7 levels per lane, written practically never) and its LDS record is reused for the
// nested activation; the pop copies it back.  The host sizes the pools for 1.2 x the expected demand or keeps the tables
// out of LDS to make room (rt_hip_api.hip), so a repeat is rare and an overflow rarer still; tests force both.
// The pools sit at a FIXED offset of the kernel's dynamic LDS ([frame bitmap][base bitmap][frames][bases]; rt_kernel.hip's
// layout asserts it); their sizes travel in DevScene.light_pool_slots / light_base_slots.
template <bool HAS_LIGHTS>
struct LightStack {};
template <>
struct LightStack<true> {  // (host form only)
  double P[RT_MAX_LIGHT_NEST - 1][3];
  float a[RT_MAX_LIGHT_NEST - 1][6];  // a[0..2], acc[0..2]
  uint32_t j[RT_MAX_LIGHT_NEST - 1][2];  // j, node
  float base[6];  // lit scenes with every albedo in [0, 1]: p[3], h[3] of the sample's colour map once a light contributed (lane_compose)
};
struct LightParked {
  LightFrame cur;
  V3 saved_d;  // camera-path activations: the scattered direction the camera path resumes with; nested ones: the link (first word)
};
static_assert(sizeof(LightFrame) == 56 && sizeof(LightParked) == 80, "a light frame record is 80 B, its link sits behind the 56 B frame");
#ifndef RT_BLOCK
#define RT_BLOCK 1024  // threads of the megakernel's workgroup (rt_kernel.hip): the offset below depends on its wave count
#endif
constexpr uint32_t LIGHT_CENTRES_LDS_MAX = 32u;       // light centres staged in LDS (24 B each); further lights are read from HBM
// = rt_kernel.hip lds_layout().park_off of a lit scene (asserted there, for every RT_BLOCK): flag block, tile slots (3 KB per
// wave), exchange slots of the cooperative draw (1 KB per wave), light centres
constexpr uint32_t LIGHT_POOL_LDS_OFF = (32u + 32u * 8u) + (RT_BLOCK / 64u) * 3u * 1024u + (RT_BLOCK / 64u) * 64u * 16u + LIGHT_CENTRES_LDS_MAX * 24u;
constexpr uint32_t LIGHT_POOL_BITMAP_BYTES = 128u;   // 1024 slots at most, per pool
constexpr uint32_t LIGHT_POOL_MAX_SLOTS = 1024u;
constexpr uint32_t LIGHT_BASE_BYTES = 32u;  // p[3], h[3] + 8 B of padding: a base is addressed by its INDEX (a shift), not by a byte offset
constexpr uint32_t LIGHT_BASE_BITMAP_LDS_OFF = LIGHT_POOL_LDS_OFF + LIGHT_POOL_BITMAP_BYTES;
constexpr uint32_t LIGHT_BASES_LDS_OFF = LIGHT_POOL_LDS_OFF + 2u * LIGHT_POOL_BITMAP_BYTES;  // the bases first (fixed offset), the frames behind them
RT_HD uint32_t light_frames_lds_off(uint32_t base_slots) { return LIGHT_BASES_LDS_OFF + base_slots * LIGHT_BASE_BYTES; }
constexpr uint32_t LIGHT_LINK_OVERFLOW = 1u;  // link of a record whose suspended parent is in the HBM overflow (record offsets are multiples of 16)
constexpr uint32_t LIGHT_OVERFLOW_BYTES_PER_LANE = (RT_MAX_LIGHT_NEST - 1u) * (uint32_t)sizeof(LightParked);
#if defined(__HIP_DEVICE_COMPILE__)
extern __shared__ __attribute__((aligned(16))) unsigned char rt_lds_dyn[];  // the kernel's dynamic LDS (aliases its own declaration)
#endif
template <bool HAS_LIGHTS, bool POOLED = false>
struct LightState {};
template <>
struct LightState<true, false> {
  LightParked* pk;
  LightStack<true>* stack; // levels 0..top-1: touched only when a light ray's own hit starts sampling the lights again
                           // (probability 0.1 n_lights) and when that nested activation returns
  int top;
};
template <>
struct LightState<true, true> {
  // LDS byte offset of the record of the activation that is summing over the lights, else 0 — a multiple of 16 — with the
  // nesting level (`top`, 0 .. RT_MAX_LIGHT_NEST - 1 < 16) in its low four bits: one register for both (the lit kernels
  // sit on the 128-register edge)
  uint32_t wt;
};
static_assert(RT_MAX_LIGHT_NEST <= 16u && LIGHT_BASES_LDS_OFF % 16u == 0u && sizeof(LightParked) % 16u == 0u && LIGHT_BASE_BYTES % 16u == 0u,
              "the nesting level shares a register with the record offset: offsets are multiples of 16");
RT_HD uint32_t ls_where(const LightState<true, true>& ls) { return ls.wt & ~15u; }
RT_HD int ls_top(const LightState<true, true>& ls) { return (int)(ls.wt & 15u); }
RT_HD void ls_set_top(LightState<true, true>& ls, int t) { ls.wt = (ls.wt & ~15u) | (uint32_t)t; }
RT_HD int ls_top(const LightState<true, false>& ls) { return ls.top; }
RT_HD void ls_set_top(LightState<true, false>& ls, int t) { ls.top = t; }
RT_HD LightParked& light_frame(LightState<true, false>& ls) { return *ls.pk; }
RT_HD void light_frame_release(const DevScene&, LightState<true, false>&) {}
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ LightParked& light_frame(LightState<true, true>& ls) { return *reinterpret_cast<LightParked*>(rt_lds_dyn + ls_where(ls)); }
__device__ __forceinline__ uint32_t& light_link(uint32_t where) { return *reinterpret_cast<uint32_t*>(rt_lds_dyn + where + (uint32_t)sizeof(LightFrame)); }
// Take a record of a pool (a bitmap of `words` 32-bit words): its index, or ~0: none free — every word once, starting at w.
// One flat loop of full-rate arithmetic.  (Inlined twice, frames and bases: as a real call it cost the lit kernels 1 - 5 spilled
// registers around the call site.)
__device__ __forceinline__ uint32_t light_pool_take(uint32_t* bitmap, uint32_t words, uint32_t w) {
  uint32_t full = 0;
  uint32_t cur = __hip_atomic_load(&bitmap[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  for (;;) {
    if (cur == 0xFFFFFFFFu) {  // this word is full (as far as this lane has seen): the next one, all of them once
      if (++full >= words) return 0xFFFFFFFFu;
      w = w + 1u == words ? 0u : w + 1u;
      cur = __hip_atomic_load(&bitmap[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      continue;
    }
    const uint32_t b = (uint32_t)__builtin_ctz(~cur);
    const uint32_t old = __hip_atomic_fetch_or(&bitmap[w], 1u << b, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (!((old >> b) & 1u)) return (w << 5) + b;
    cur = old | (1u << b);
  }
}
// 16-bit multiplicative hash of a seed; scaled to a word count by a 24-bit multiply (full-rate arithmetic: the round-4 form
// took `hash % words` — ~20 instructions of emulated division, five of them quarter-rate multiplies)
__device__ __forceinline__ uint32_t light_pool_hash(uint32_t seed) { return __umul24(seed & 0xFFFFu, 0x9E3Bu) & 0xFFFFu; }
__device__ __forceinline__ void light_pool_give(uint32_t bitmap_off, uint32_t slot) {  // (after the last read of the record)
  uint32_t* const bitmap = reinterpret_cast<uint32_t*>(rt_lds_dyn + bitmap_off);
  __hip_atomic_fetch_and(&bitmap[slot >> 5], ~(1u << (slot & 31u)), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// index of the frame record at LDS offset `where`: (where - first record) / 80 without a division — offsets are multiples
// of 16, y / 5 = (y * ceil(2^16 / 5)) >> 16 exactly for y < 2^14 (an LDS offset / 16)
__device__ __forceinline__ uint32_t light_frame_slot(const DevScene& sc, uint32_t where) {
  return __umul24((where - light_frames_lds_off(sc.light_base_slots)) >> 4, 13108u) >> 16;
}
// the camera-path activation is back on the camera path: give its frame record back
__device__ __forceinline__ void light_frame_release(const DevScene& sc, LightState<true, true>& ls) {
  light_pool_give(LIGHT_POOL_LDS_OFF, light_frame_slot(sc, ls_where(ls)));
  ls.wt = 0u;
}
__device__ __forceinline__ unsigned char* light_overflow_slot(const DevScene& sc, int level) {
  // (the lane's index is made HERE: as a plain expression of blockIdx / threadIdx the compiler hoists lane * 7 out of the path
  //  loop into a register pair that stays live across everything — for a path taken practically never)
  uint32_t tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  return sc.light_overflow + ((size_t)(blockIdx.x * (uint32_t)RT_BLOCK + tid) * (RT_MAX_LIGHT_NEST - 1u) + (uint32_t)level) * sizeof(LightParked);
}
// ... and a nested activation has returned its colour: the suspended one is current again
__device__ __forceinline__ void light_frame_pop(const DevScene& sc, LightState<true, true>& ls) {
  const int t = ls_top(ls) - 1;
  const uint32_t cur = ls_where(ls);
  const uint32_t link = light_link(cur);
  if (link != LIGHT_LINK_OVERFLOW) {
    light_pool_give(LIGHT_POOL_LDS_OFF, light_frame_slot(sc, cur));
    ls.wt = link | (uint32_t)t;
  } else {
    const uint4* src = reinterpret_cast<const uint4*>(light_overflow_slot(sc, t));
    uint4* dst = reinterpret_cast<uint4*>(rt_lds_dyn + cur);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(LightParked) / 16u); ++i) dst[i] = src[i];
    ls.wt = cur | (uint32_t)t;
  }
}
#else  // (host passes: the pooled form exists on the device only; the host simulator runs the direct form)
inline LightParked& light_frame(LightState<true, true>&) { static LightParked never; return never; }
inline void light_frame_release(const DevScene&, LightState<true, true>&) {}
inline void light_frame_pop(const DevScene&, LightState<true, true>&) {}
#endif

RT_HD void light_frame_push(const DevScene&, LightState<true, false>& ls, uint32_t) {  // stack[top] <- cur; ++top
  const int t = ls_top(ls);
  LightStack<true>& k = *ls.stack;
  const LightFrame& c = light_frame(ls).cur;
  k.P[t][0] = c.P.x; k.P[t][1] = c.P.y; k.P[t][2] = c.P.z;
  k.a[t][0] = c.a[0]; k.a[t][1] = c.a[1]; k.a[t][2] = c.a[2];
  k.a[t][3] = c.acc[0]; k.a[t][4] = c.acc[1]; k.a[t][5] = c.acc[2];
  k.j[t][0] = c.j; k.j[t][1] = c.node;
  ls_set_top(ls, t + 1);
}
RT_HD void light_frame_pop(const DevScene&, LightState<true, false>& ls) {  // --top; cur <- stack[top]
  const int t = ls_top(ls) - 1;
  const LightStack<true>& k = *ls.stack;
  LightFrame& c = light_frame(ls).cur;
  c.P.x = k.P[t][0]; c.P.y = k.P[t][1]; c.P.z = k.P[t][2];
  c.a[0] = k.a[t][0]; c.a[1] = k.a[t][1]; c.a[2] = k.a[t][2];
  c.acc[0] = k.a[t][3]; c.acc[1] = k.a[t][4]; c.acc[2] = k.a[t][5];
  c.j = k.j[t][0]; c.node = k.j[t][1];
  ls_set_top(ls, t);
}
// Lane setup (host form): `stk` and `*pk` must outlive the lane.
template <class LaneT> RT_HD void lane_attach_light_state(LaneT&, LightStack<false>&, LightParked*) {}
template <class LaneT> RT_HD void lane_attach_light_state(LaneT& L, LightStack<true>& stk, LightParked* pk) { L.ls.stack = &stk; L.ls.pk = pk; L.ls.top = 0; }

template <bool HAS_LIGHTS, bool SIMPLE = false, bool POOLED = false>
struct Lane {
  static constexpr bool kLights = HAS_LIGHTS;
  static constexpr bool kSimple = SIMPLE;
  static constexpr bool kPooled = POOLED;
  V3 o, d;        // current ray
  uint32_t node;  // RNG node of the current ray
  uint32_t k;     // camera-path segment index of the current (or suspended) camera ray
  uint32_t s;     // current sample
  uint32_t in_light;  // bit 0: the current ray is a nested light ray; bit 1 (LANE_HAS_BASE): the sample's colour map has a base (lane_compose);
                      // bits 16..31 (device): index + 1 of the sample's base record in the workgroup's pool while it holds one, else 0
  FwdT<SIMPLE> fwd;
  float val[3];   // radiance of the sample that just finished (valid when lane_shade returned true)
  RngAddr ra;
  LightState<HAS_LIGHTS, POOLED> ls;
  // counters
  uint32_t n_segments, n_exact, n_tex_oob;
};

// raytracer.rs:199-201 + camera.rs:79-84
// (the four Philox words of the camera jitter — rng(L.ra, NODE_CAMERA, 0) with L.ra.sample = L.s — come from the caller)
template <class LaneT>
RT_HD void lane_begin_sample_w(const DevScene& sc, LaneT& L, uint32_t px, uint32_t py, U4 w) {
  L.ra.sample = L.s;
  double un = (double)px + u01_53(w.x, w.y), vn = sc.height_d - ((double)py + u01_53(w.z, w.w));
  double u, v;  // raytracer.rs:199-200: un / (width - 1), vn / (height - 1)
  if (sc.cam_fast) { u = div_by_recip(un, sc.wm1, sc.inv_wm1); v = div_by_recip(vn, sc.hm1, sc.inv_hm1); }  // (a flag of its own: testing the reciprocals themselves kept one of them in a — spilled — vector register across the branch)
  else { u = un / sc.wm1; v = vn / sc.hm1; }
  V3 origin = v3(sc.cam_origin[0], sc.cam_origin[1], sc.cam_origin[2]);
  V3 llc = v3(sc.cam_ll[0], sc.cam_ll[1], sc.cam_ll[2]);
  V3 hor = v3(sc.cam_h[0], sc.cam_h[1], sc.cam_h[2]);
  V3 ver = v3(sc.cam_v[0], sc.cam_v[1], sc.cam_v[2]);
  L.o = origin;
  L.d = sub(add(add(llc, muls(hor, u)), muls(ver, v)), origin);
  L.node = 0; L.k = 0; L.in_light = 0;
  fwd_init(L.fwd);
}

template <class LaneT>
RT_HD void lane_begin_sample(const DevScene& sc, LaneT& L, uint32_t px, uint32_t py) {
  L.ra.sample = L.s;
  lane_begin_sample_w(sc, L, px, py, rng(L.ra, NODE_CAMERA, 0));
}

// ---- the colour map of LIT scenes whose albedos all lie in [0, 1] (Lane<true, true, *>) ------------------------------
// With a, L >= 0 every level is f(x) = min(1, L + a x), and the composition is G(x) = min(h, p + q x) — FwdT<false>'s
// (p, q, hi); its `lo` never binds (lo <= p by induction, and fl(p + q x) >= p).  Only the first two levels of a sample can
// carry light (raytracer.rs:99-102), so:
//   * a sample that never sampled the lights has p = 0 and h >= q: G(x) = q x — the three floats of the unlit map;
//   * once a light contributed, (p, h) — the BASE — are final after level 1: later levels only multiply q (p + q*0 = p,
//     and their h-terms fl(p + q_j) >= fl(p + fl(q x)) cannot bind).
// So the lane keeps q[3] in registers like an unlit lane, and the base — written at most twice per sample and read once
// when the sample ends, for ~20 % of the samples of a one-light scene — lives in memory (the device: a pool record in LDS,
// lane_base; the host form: LightStack::base), not in nine registers the 128-register kernel does not have (the general
// map cost the lit kernels 30 spilled registers, some reloaded inside the walk loop).  Bit-identical to FwdT<false>
// (tests/test_core_cpu.py).
constexpr uint32_t LANE_HAS_BASE = 2u;
template <class LaneT>
RT_HD float* lane_base(LaneT& L) {
  if constexpr (LaneT::kPooled) {
#if defined(__HIP_DEVICE_COMPILE__)
    return reinterpret_cast<float*>(rt_lds_dyn + (LIGHT_BASES_LDS_OFF - LIGHT_BASE_BYTES) + (L.in_light >> 16) * LIGHT_BASE_BYTES);  // (index + 1 in the high half)
#else
    (void)L;
    return nullptr;  // (host pass of the kernel's instantiations: never run)
#endif
  } else return L.ls.stack->base;
}
// A hit starts summing over the lights (raytracer.rs:99-111): make room for its activation.
//   camera path (light_ray false): its frame and — short colour map — the sample's base unless an earlier level took one
//     already.  Both or nothing: false = a pool is exhausted and NOTHING has changed (the caller repeats the segment).
//   a light ray's own hit (light_ray true): the activation whose light ray it is gets suspended (level t), a record for
//     the nested one (level t + 1) becomes current — from the pool, or by moving the suspended one out to the lane's HBM
//     overflow slot and reusing its record: never false (a lane that holds records does not wait for one).
// ONE take of each pool in the instruction stream (a wave runs every inlined copy some lane reaches).
template <class LaneT>
RT_HD bool lane_light_begin(const DevScene& sc, LaneT& L, bool light_ray) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (LaneT::kPooled) {
    const uint32_t h = light_pool_hash(L.ra.pixel + L.ra.sample + L.node);
    uint32_t* const fbits = reinterpret_cast<uint32_t*>(rt_lds_dyn + LIGHT_POOL_LDS_OFF);
    uint32_t* const bbits = reinterpret_cast<uint32_t*>(rt_lds_dyn + LIGHT_BASE_BITMAP_LDS_OFF);
    const uint32_t fwords = sc.light_pool_slots >> 5, bwords = sc.light_base_slots >> 5;
    const bool direct_base = sc.light_base_slots == (uint32_t)RT_BLOCK;  // (a base per lane fits — small scenes, the reference's test_scene: the lane's own, no bitmap)
    bool want_base = false;
    if constexpr (LaneT::kSimple) want_base = !light_ray && (L.in_light >> 16) == 0u;
    const bool pool_base = want_base && !direct_base;
    const bool want_frame = !light_ray || sc.light_nest_pool != 0u;
    const uint32_t wf = __umul24(h, fwords) >> 16, wb = __umul24(h, bwords) >> 16;
    // (The two takes one after the other.  Overlapped — both bitmap words loaded together, both bits claimed back to back: two
    //  dependent LDS round trips instead of four — was built and measured: lit cover 4.465 against 4.483 ms, the reference's
    //  test scene 0.936 against 0.930, and one spilled register in the path loop: not adopted, profiles/r05_run6_ab_takes_and_flush.log.)
    uint32_t base_slot = 0xFFFFFFFFu, slot = 0xFFFFFFFFu;
    if (pool_base) {
      base_slot = light_pool_take(bbits, bwords, wb);
      if (base_slot == 0xFFFFFFFFu) return false;
    }
    if (want_frame) slot = light_pool_take(fbits, fwords, wf);
    if (want_base && direct_base) base_slot = threadIdx.x;
    const uint32_t nw = light_frames_lds_off(sc.light_base_slots) + __umul24(slot, (uint32_t)sizeof(LightParked));
    if (!light_ray) {
      if (slot == 0xFFFFFFFFu) {
        if (pool_base) light_pool_give(LIGHT_BASE_BITMAP_LDS_OFF, base_slot);
        return false;
      }
      if (want_base) L.in_light |= (base_slot + 1u) << 16;
      L.ls.wt = nw;  // (nesting level 0)
      return true;
    }
    const int t = ls_top(L.ls);
    const uint32_t cur = ls_where(L.ls);
    if (slot != 0xFFFFFFFFu) {
      light_link(nw) = cur;
      L.ls.wt = nw | (uint32_t)(t + 1);
    } else {  // no record free: the suspended activation moves out to HBM, its record serves the nested one
      const uint4* src = reinterpret_cast<const uint4*>(rt_lds_dyn + cur);
      uint4* dst = reinterpret_cast<uint4*>(light_overflow_slot(sc, t));
#pragma unroll
      for (int i = 0; i < (int)(sizeof(LightParked) / 16u); ++i) dst[i] = src[i];
      light_link(cur) = LIGHT_LINK_OVERFLOW;
      L.ls.wt = cur | (uint32_t)(t + 1);
    }
    return true;
  } else
#endif
  {
    if constexpr (!LaneT::kPooled) {
      if (light_ray) light_frame_push(sc, L.ls, 0u);
      else ls_set_top(L.ls, 0);
    }
    (void)sc; (void)light_ray;
    return true;
  }
}
// ... and when the sample has ended: its base record goes back (device form; after lane_finish_sample, the last read of it).
// Called by the KERNEL at the two places a sample can end in an iteration (a ray left the scene; lane_shade returned
// LANE_FINISHED) rather than from each of lane_finish_sample's five call sites: a wave pays for every inlined copy.
template <class LaneT>
RT_HD void lane_base_release(const DevScene& sc, LaneT& L) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (LaneT::kPooled && LaneT::kSimple) {
    const uint32_t held = L.in_light >> 16;  // index + 1
    if (held != 0u) {
      if (sc.light_base_slots != (uint32_t)RT_BLOCK) light_pool_give(LIGHT_BASE_BITMAP_LDS_OFF, held - 1u);
      L.in_light &= 0xFFFFu;
    }
  }
#endif
  (void)sc; (void)L;
}
// the six floats of a base as one 16-byte and one 8-byte LDS access (records are 32-byte aligned) instead of six dwords
struct Base6 { float p[3], h[3]; };
template <class LaneT>
RT_HD Base6 lane_base_load(LaneT& L) {
  Base6 r;
  const float* b = lane_base(L);
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (LaneT::kPooled) {
    const float4 lo = *reinterpret_cast<const float4*>(b);
    const float2 hi = *reinterpret_cast<const float2*>(b + 4);
    r.p[0] = lo.x; r.p[1] = lo.y; r.p[2] = lo.z; r.h[0] = lo.w; r.h[1] = hi.x; r.h[2] = hi.y;
    return r;
  }
#endif
  r.p[0] = b[0]; r.p[1] = b[1]; r.p[2] = b[2]; r.h[0] = b[3]; r.h[1] = b[4]; r.h[2] = b[5];
  return r;
}
template <class LaneT>
RT_HD void lane_base_store(LaneT& L, const Base6& v) {
  float* b = lane_base(L);
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (LaneT::kPooled) {
    *reinterpret_cast<float4*>(b) = make_float4(v.p[0], v.p[1], v.p[2], v.h[0]);
    *reinterpret_cast<float2*>(b + 4) = make_float2(v.h[1], v.h[2]);
    return;
  }
#endif
  b[0] = v.p[0]; b[1] = v.p[1]; b[2] = v.p[2]; b[3] = v.h[0]; b[4] = v.h[1]; b[5] = v.h[2];
}
template <class LaneT>
RT_HD void lane_compose(LaneT& L, const float light[3], const float att[3], bool has_light) {
  if constexpr (LaneT::kLights && LaneT::kSimple) {
    if (has_light) {  // level L.k (0 or 1) contributes `light`: create / update the base
      Base6 v;
      if (L.in_light & LANE_HAS_BASE) v = lane_base_load(L);
      else { v.p[0] = v.p[1] = v.p[2] = 0.0f; v.h[0] = v.h[1] = v.h[2] = 3.4028234663852886e38f; }  // (a level 0 without light before this one: its h-term 0 + 1 >= q)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float g1 = v.p[i] + L.fwd.q[i] * 1.0f;
        v.h[i] = g1 < v.h[i] ? g1 : v.h[i];
        v.p[i] = v.p[i] + L.fwd.q[i] * light[i];
        L.fwd.q[i] = L.fwd.q[i] * att[i];
      }
      lane_base_store(L, v);
      L.in_light |= LANE_HAS_BASE;
    } else {
      fwd_compose(L.fwd, light, att);
    }
  } else {
    (void)has_light;
    fwd_compose(L.fwd, light, att);
  }
}
// the sample's radiance is known: fold the leaf colour through the forward map.  The caller
// adds L.val to the pixel (raytracer.rs:203-205) and picks the lane's next sample.
template <class LaneT>
RT_HD void lane_finish_sample(const DevScene& sc, LaneT& L, Rgb leaf) {
  const float x[3] = {leaf.r, leaf.g, leaf.b};
  if constexpr (LaneT::kLights && LaneT::kSimple) {
    if (L.in_light & LANE_HAS_BASE) {
      const Base6 v = lane_base_load(L);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        float y = v.p[i] + L.fwd.q[i] * x[i];
        y = y > v.h[i] ? v.h[i] : y;
        L.val[i] = y;
      }
      return;  // (the kernel gives the base record back: lane_base_release)
    }
  }
  (void)sc;
  L.val[0] = fwd_eval1(L.fwd, 0, x[0]);
  L.val[1] = fwd_eval1(L.fwd, 1, x[1]);
  L.val[2] = fwd_eval1(L.fwd, 2, x[2]);
}

// Pooled-sample mode: lanes of a wave take (pixel, sample) items from a shared counter, so
// several lanes add samples of the same pixel in an order the reference does not have.  To
// stay deterministic the pixel sum is kept in exact 2^-40 fixed point (integer adds are
// associative): independent of the schedule, of tiling and of the GPU count.  It differs from
// the reference's sequential f32 sum only by that sum's own rounding (a few 1e-7 relative).
constexpr double FIX_SCALE = 1099511627776.0;  // 2^40; sample values are in [0,1]
RT_HD unsigned long long sample_to_fixed(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  // clamp in f32, one v_med3_f32: NaN and negatives -> 0 (min3 of a NaN operand set), above 1 -> 1 — the same values as the
  // compares below (the conversion to f64 is exact and monotone, so clamping before it changes nothing)
  double x = (double)__builtin_amdgcn_fmed3f(v, 0.0f, 1.0f);
#else
  double x = (double)v;
  if (!(x > 0.0)) return 0ull;
  if (x > 1.0) x = 1.0;  // every sample is clamp01'ed at the root level already
#endif
  // (unsigned long long)(x * 2^40 + 0.5), written so that the GPU needs no 64-bit float->int
  // conversion (it has none: 10 instructions per channel): the truncated value is an integer
  // below 2^41, so adding 2^52 is exact and leaves it in the low mantissa bits.
  const double y = __builtin_trunc(__builtin_fma(x, FIX_SCALE, 0.5)) + 4503599627370496.0;  // (x * 2^40 is exact: the fused form rounds once, to the same value)
  unsigned long long bits;
  __builtin_memcpy(&bits, &y, sizeof bits);
  return bits & 0x000FFFFFFFFFFFFFull;
}
RT_HD float fixed_to_mean(unsigned long long sum, uint32_t spp) {
  return (float)(((double)sum * (1.0 / FIX_SCALE)) / (double)spp);
}
// A NaN sample makes the reference's f32 pixel sum NaN (raytracer.rs:203-205).  Fixed point has no NaN: such a
// sample adds 0 and sets the pixel's bit in a per-tile, per-channel mask instead; the pixel then reads NaN.
RT_HD bool sample_is_nan(float v) { return v != v; }
RT_HD float rt_nanf() {
  const uint32_t b = 0x7FC00000u;
  float f;
  __builtin_memcpy(&f, &b, 4);
  return f;
}

// Continue the camera path after its hit at level k has been fully evaluated:
// compose clamp(light + albedo*child) and step to the scattered ray.  Returns true when the
// sample finished (depth exhausted: the child is ray_color(.., depth 0) = black, :80-82).
template <class LaneT>
RT_HD bool lane_continue_main(const DevScene& sc, LaneT& L, V3 point, V3 out_dir, const float light[3], const float att[3], bool has_light = false) {
  lane_compose(L, light, att, has_light);
  L.k += 1;
  if (L.k >= sc.max_depth) { lane_finish_sample(sc, L, rgb(0.0f, 0.0f, 0.0f)); return true; }
  L.o = point; L.d = out_dir; L.node = L.k; L.in_light &= ~1u;
  return false;
}

// aim the current ray at light j of frame `top` (raytracer.rs:104-106)
template <class Tables, class LaneT>
RT_HD void lane_aim_light(const DevScene& sc, const Tables& tb, LaneT& L) {
  LightFrame& f = light_frame(L.ls).cur;
  L.o = f.P;
  L.d = sub(tb.light_centre(sc, f.j), f.P);
  L.node = child_node(f.node, f.j);
  L.in_light |= 1u;
}
// a nested light ray produced colour tc: hand it to its parent activation(s) (:107-113)
template <class Tables, class LaneT>
RT_HD bool lane_light_return(const DevScene& sc, const Tables& tb, LaneT& L, Rgb tc) {
  for (;;) {
    LightFrame& f = light_frame(L.ls).cur;
    f.acc[0] += f.a[0] * tc.r; f.acc[1] += f.a[1] * tc.g; f.acc[2] += f.a[2] * tc.b;
    f.j += 1;
    if (f.j < sc.n_lights) { lane_aim_light(sc, tb, L); return false; }
    float light[3] = {f.acc[0], f.acc[1], f.acc[2]};
    if (sc.n_lights != 1u) {  // raytracer.rs:112-114 `/= lights.len() as f32`: x / 1.0 is x — one light (every shipped scene) skips three IEEE divisions (wave-uniform branch)
      const float nl = (float)sc.n_lights;
      light[0] = f.acc[0] / nl; light[1] = f.acc[1] / nl; light[2] = f.acc[2] / nl;
    }
    if (ls_top(L.ls) == 0) {  // back on the camera path: clamp(light + albedo*child), child = scattered ray
      const bool fin = lane_continue_main(sc, L, f.P, light_frame(L.ls).saved_d, light, f.a, true);
      light_frame_release(sc, L.ls);  // (after the last read of the frame)
      return fin;
    }
    // a nested activation (max_depth 2, depth 1): its own child is depth 0 = black (:117-122)
    tc = rgb(clamp01(light[0] + f.a[0] * 0.0f), clamp01(light[1] + f.a[1] * 0.0f), clamp01(light[2] + f.a[2] * 0.0f));
    light_frame_pop(sc, L.ls);  // the activation that shot the light ray goes on
  }
}

// The light-sampling draw of the current node (raytracer.rs:100; addressing above), and whether a hit of this lane's
// current ray could need it at all: a camera-path hit of the first two levels, or a light ray's hit below the nesting cap.
// The kernel hands it in (light_u_pre) from the words its one Philox stream of the iteration has already made — two Philox
// streams inlined in two branches that a few lanes take cost every wave iteration ~160 instructions, one stream of its own ~4 %.
template <class LaneT>
RT_HD double lane_light_draw(const LaneT& L, bool glass) {
  const U4 w = rng(L.ra, L.node, 0);
  return u01_53(w.z, glass ? w.w : rng(L.ra, L.node, 1).w);
}
// The low word of a light-sampling draw whose high word alone does not decide it (a real call: 2^-32 of the draws)
RT_HD_COLD uint32_t light_draw_low_word(RngAddr ra, uint32_t node) { return rng(ra, node, 0).z; }
template <class LaneT>
RT_HD bool lane_may_sample_lights(const DevScene& sc, const LaneT& L) {
  if constexpr (LaneT::kLights) {
    return (L.in_light & 1u) ? (uint32_t)ls_top(L.ls) + 1u < RT_MAX_LIGHT_NEST : (sc.max_depth >= 2 && L.k < 2);
  }
  return false;
}

// Consume the closest hit (idx < 0: miss) of the lane's current ray.  Returns LANE_FINISHED when the
// lane's current sample finished (its radiance is in L.val; the caller starts the next one), LANE_REPEAT when nothing was
// consumed (pooled lit kernels: no light frame free — the same ray is traced again next iteration), else LANE_CONTINUE.
enum { LANE_CONTINUE = 0, LANE_FINISHED = 1, LANE_REPEAT = 2 };
// the census build of the lit kernels (-DRT_PROFILE -DRT_PROF_LIT, rt_kernel.hip): shader-clock cycles of lane_shade's parts
struct ShadeProf { unsigned long long* t; unsigned long long* last; uint32_t* n_iter_sample; uint32_t* n_iter_return; };
#if defined(RT_PROF_LIT) && defined(__HIP_DEVICE_COMPILE__)
#define RT_SHADE_MARK(k) do { if (sp) { const unsigned long long now_ = __builtin_readcyclecounter(); sp->t[k] += now_ - *sp->last; *sp->last = now_; } } while (0)
#define RT_SHADE_COUNT(act_) do { if (sp) { const unsigned long long sm_ = __builtin_amdgcn_ballot_w64((act_) == ACT_SAMPLE), rm_ = __builtin_amdgcn_ballot_w64((act_) == ACT_RETURN); \
    if (sm_) (*sp->n_iter_sample)++; if (rm_) (*sp->n_iter_return)++; sp->t[5] += (unsigned long long)__builtin_popcountll(sm_) | ((unsigned long long)__builtin_popcountll(rm_) << 32); } } while (0)
#else
#define RT_SHADE_MARK(k) do { } while (0)
#define RT_SHADE_COUNT(act_) do { } while (0)
#endif
template <class LaneT, class Tables>
RT_HD int lane_shade(const DevScene& sc, const Tables& tb, LaneT& L, int idx, double t, const V3* rnd_pre = nullptr,
                      const double* glass_u_pre = nullptr, const double* light_u_pre = nullptr, ShadeProf* sp = nullptr) {
  (void)sp;
  constexpr bool HL = LaneT::kLights;
  const float zero3[3] = {0.0f, 0.0f, 0.0f};
  if constexpr (!HL) {
    (void)light_u_pre;
    if (idx < 0) {  // raytracer.rs:133-163
      lane_finish_sample(sc, L, sky_color(sc, L.d, L.n_tex_oob));
      return LANE_FINISHED;
    }
    const SphereGeom g = tb.geom((uint32_t)idx);
    const MatCore m = tb.mat((uint32_t)idx);
    Surface h = surface_at(L.o, L.d, t, g, m.inv_r);
    V3 out_dir = v3(0, 0, 0);
    float att[3];
    int st = scatter(sc, L.ra, L.node, L.d, h, g, m, (uint32_t)idx, out_dir, att, L.n_tex_oob, rnd_pre, glass_u_pre);
    if (st == SCATTER_ABSORBED) { lane_finish_sample(sc, L, rgb(0.f, 0.f, 0.f)); return LANE_FINISHED; }       // :127-131
    if (st == SCATTER_EMIT) { lane_finish_sample(sc, L, rgb(att[0], att[1], att[2])); return LANE_FINISHED; }  // :124
    return lane_continue_main(sc, L, h.point, out_dir, zero3, att) ? LANE_FINISHED : LANE_CONTINUE;
  } else {
    // Lit scenes.  First decide what the hit MEANS for the lane; the heavy continuations — hand a colour back to the
    // activation that shot this light ray, start summing over the lights — then exist once each: in a wave some lane
    // takes nearly every branch, and every inlined copy of lane_light_return a different lane reaches is paid by all 64.
    enum { ACT_FINISH = 0, ACT_CONTINUE = 1, ACT_RETURN = 2, ACT_SAMPLE = 3 };
    const bool light_ray = (L.in_light & 1u) != 0u;   // a nested activation ray_color(light_ray, 2, 1) at nesting level top+1
    Rgb col = rgb(0.f, 0.f, 0.f);  // the sample's leaf colour (ACT_FINISH) / the colour of the light ray (ACT_RETURN)
    V3 point = v3(0, 0, 0), out_dir = v3(0, 0, 0);
    float att[3] = {0.f, 0.f, 0.f};
    int act;
    if (idx < 0) {  // raytracer.rs:133-163
      col = sky_color(sc, L.d, L.n_tex_oob);
      act = light_ray ? ACT_RETURN : ACT_FINISH;
    } else {
      const SphereGeom g = tb.geom((uint32_t)idx);
      const MatCore m = tb.mat((uint32_t)idx);
      Surface h = surface_at(L.o, L.d, t, g, m.inv_r);
      point = h.point;
      const int st = scatter(sc, L.ra, L.node, L.d, h, g, m, (uint32_t)idx, out_dir, att, L.n_tex_oob, rnd_pre, glass_u_pre);
      if (st != SCATTER_RAY) {  // :124 Light: its colour; :127-131 absorbed: black
        if (st == SCATTER_EMIT) col = rgb(att[0], att[1], att[2]);
        act = light_ray ? ACT_RETURN : ACT_FINISH;
      } else {
        // raytracer.rs:92-102: sample the lights?  Camera path: depth > max_depth-2 <=> k < 2 (and max_depth >= 2, usize
        // wrap); a light ray's own hit: the same test one level down, below the nesting cap
        bool sample = false;
        if (lane_may_sample_lights(sc, L)) {
          const double lu = light_u_pre ? *light_u_pre : lane_light_draw(L, m.kind == RT_MAT_GLASS);
          sample = lu > sc.light_thr[m.kind == RT_MAT_GLASS ? 1 : 0];  // 1.0 - n_lights as f64 * prob (fill_dev_scene)
        }
        if (sample) act = ACT_SAMPLE;
        else if (light_ray) {  // no light sampling: clamp(0 + albedo * black) (:117-122, the child is depth 0)
          col = rgb(clamp01(0.0f + att[0] * 0.0f), clamp01(0.0f + att[1] * 0.0f), clamp01(0.0f + att[2] * 0.0f));
          act = ACT_RETURN;
        } else act = ACT_CONTINUE;
      }
    }
#if defined(RT_PROF_LIT) && defined(__HIP_DEVICE_COMPILE__)
    // (census build: the continuations one after the other, each closed by a wave-level mark; no lane leaves early, so that the
    //  lane whose clock is reported — lane 0 — passes every mark.  The product's form is below.)
    RT_SHADE_MARK(1);
    RT_SHADE_COUNT(act);
    int result = LANE_CONTINUE;
    if (act == ACT_SAMPLE) {
      if (!lane_light_begin(sc, L, light_ray)) result = LANE_REPEAT;
      else {
        if (!light_ray) light_frame(L.ls).saved_d = out_dir;
        LightFrame& f = light_frame(L.ls).cur;
        f.P = point; f.a[0] = att[0]; f.a[1] = att[1]; f.a[2] = att[2];
        f.acc[0] = f.acc[1] = f.acc[2] = 0.0f; f.j = 0; f.node = L.node;
        lane_aim_light(sc, tb, L);
      }
    }
    __builtin_amdgcn_wave_barrier();
    RT_SHADE_MARK(2);
    if (act == ACT_RETURN) result = lane_light_return(sc, tb, L, col) ? LANE_FINISHED : LANE_CONTINUE;
    __builtin_amdgcn_wave_barrier();
    RT_SHADE_MARK(3);
    if (act == ACT_FINISH) { lane_finish_sample(sc, L, col); result = LANE_FINISHED; }
    else if (act == ACT_CONTINUE) result = lane_continue_main(sc, L, point, out_dir, zero3, att) ? LANE_FINISHED : LANE_CONTINUE;
    __builtin_amdgcn_wave_barrier();
    RT_SHADE_MARK(4);
    return result;
#else
    if (act == ACT_SAMPLE) {
      // (a light ray's hit suspends the activation whose light ray it is; a camera-path hit that finds a pool of the device
      //  form exhausted has changed nothing yet: the same segment is traced again; the caller counts the segment once — its
      //  exact tests and grid steps are the work actually done, and counted as such)
      if (!lane_light_begin(sc, L, light_ray)) return LANE_REPEAT;
      if (!light_ray) light_frame(L.ls).saved_d = out_dir;
      LightFrame& f = light_frame(L.ls).cur;
      f.P = point; f.a[0] = att[0]; f.a[1] = att[1]; f.a[2] = att[2];
      f.acc[0] = f.acc[1] = f.acc[2] = 0.0f; f.j = 0; f.node = L.node;
      lane_aim_light(sc, tb, L);
      return LANE_CONTINUE;
    }
    if (act == ACT_RETURN) return lane_light_return(sc, tb, L, col) ? LANE_FINISHED : LANE_CONTINUE;
    if (act == ACT_FINISH) { lane_finish_sample(sc, L, col); return LANE_FINISHED; }
    return lane_continue_main(sc, L, point, out_dir, zero3, att) ? LANE_FINISHED : LANE_CONTINUE;
#endif
  }
}

// raytracer.rs:207-213: mean, sqrt gamma, palette f32 -> u8: round-half-even of min(x*255, 255), negatives -> 0,
// NaN -> 255 (Rust's f32::min drops a NaN operand; third-party, unpinned — see oracle/rt_oracle.c)
RT_HD uint8_t f32_to_u8(float x) {
  float scaled = x * 255.0f;
  if (scaled != scaled) return 255;
  if (!(scaled > 0.0f)) return 0;
  if (scaled > 255.0f) scaled = 255.0f;
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint8_t)__builtin_rintf(scaled);
#else
  return (uint8_t)nearbyintf(scaled);
#endif
}

}  // namespace rtc
