/*
 * rt_oracle.c — CPU restatement of the reference's ray_color hot path (see rt_oracle.h).
 * TEST INFRASTRUCTURE ONLY — never linked into, loaded by, or called from the product.
 *
 * Build: gcc -O3 -march=x86-64-v3 -std=c11 -ffp-contract=off -fno-fast-math -fopenmp (oracle/Makefile; x86-64-v3 — AVX2, no FMA
 * contraction — instead of -march=native: the library is built in the CPU container and SHIPS to the GPU box, whose host CPU differs).
 * -ffp-contract=off matters: rustc never fuses a*b+c, so neither may we.
 *
 * Every function cites the reference lines it follows (paths relative to
 * /root/reference/raytracer/src/).
 */
#include "rt_oracle.h"
/* the one atan2 shared with the kernel (see its header for why libm's cannot be used) */
#include "../rust-raytracer_amd/csrc/common/rt_atan2.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ point3d.rs:10-177 */
typedef struct { double x, y, z; } P3;

static inline P3 p3(double x, double y, double z) { P3 r = {x, y, z}; return r; }
static inline P3 p3_add(P3 a, P3 b) { return p3(a.x + b.x, a.y + b.y, a.z + b.z); }   /* :89-99  */
static inline P3 p3_sub(P3 a, P3 b) { return p3(a.x - b.x, a.y - b.y, a.z - b.z); }   /* :101-111 */
static inline P3 p3_neg(P3 a) { return p3(-a.x, -a.y, -a.z); }                        /* :113-123 */
static inline P3 p3_muls(P3 a, double s) { return p3(a.x * s, a.y * s, a.z * s); }    /* :137-147 */
static inline P3 p3_divs(P3 a, double s) { return p3(a.x / s, a.y / s, a.z / s); }    /* :161-171 */
static inline double p3_dot(P3 a, P3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; } /* :72-74  */
static inline double p3_length_squared(P3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; } /* :59-61 */
/* :52-57 distance, :63-65 length = distance to the origin */
static inline double p3_length(P3 a) {
  double dx = a.x - 0.0, dy = a.y - 0.0, dz = a.z - 0.0;
  return sqrt(dx * dx + dy * dy + dz * dz);
}
/* :67-70 */
static inline P3 p3_unit(P3 a) { double l = p3_length(a); return p3(a.x / l, a.y / l, a.z / l); }
/* :76-82 */
static inline P3 p3_cross(P3 a, P3 b) {
  return p3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* :84-86, f64::EPSILON = 2^-52 */
static inline int p3_near_zero(P3 a) {
  const double eps = 2.220446049250313e-16;
  return fabs(a.x) < eps && fabs(a.y) < eps && fabs(a.z) < eps;
}

typedef struct { P3 origin, direction; } Ray;                                          /* ray.rs:7-21 */
static inline P3 ray_at(Ray r, double t) { return p3_add(r.origin, p3_muls(r.direction, t)); }

typedef struct { float r, g, b; } Rgb; /* palette::Srgb<f32> used as a plain 3-float box */
static inline Rgb rgb(float r, float g, float b) { Rgb c = {r, g, b}; return c; }

/* ------------------------------------------------------------------ Philox4x32-10 (Random123) */
void rt_oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int round = 0; round < 10; ++round) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* RNG addressing: counter = (pixel, sample, node, slot), key = seed.
 *   node 0xFFFFFFFF, slot 0 : camera jitter xi1, xi2                 (raytracer.rs:199-200)
 *   node n, slot 0          : words 0,1 = Glass reflectance draw     (materials.rs:189)
 *                             words 2,3 = light-sampling draw of a Glass hit (raytracer.rs:100);
 *                             word 2 = the LOW word of every other hit's light-sampling draw
 *   node n, slot 1+a        : words 0..2 = attempt a of random_in_unit_sphere (point3d.rs:31-38)
 *   node n, slot 1          : word 3 = the HIGH word of the light-sampling draw of a hit that is not Glass
 *                             (the word attempt 0 leaves over: one Philox call serves both draws of such a hit)
 * node = k for the k-th segment of the camera path, rt_child_node() for nested light rays. */
#define NODE_CAMERA 0xFFFFFFFFu

typedef struct {
  const RtScene* scene;
  const uint32_t* lights;
  uint32_t n_lights;
  uint32_t pixel, sample;
  uint64_t segments, tex_oob;
  /* bookkeeping only (no effect on the image): > 0 while tracing light rays whose sum the caller will
   * throw away (raytracer.rs:124) — the GPU kernel skips exactly those segments */
  uint32_t discarding;
  uint64_t segments_discarded;
} Ctx;

static void rng_words(const Ctx* c, uint32_t node, uint32_t slot, uint32_t w[4]) {
  uint32_t ctr[4] = {c->pixel, c->sample, node, slot};
  uint32_t key[2] = {(uint32_t)c->scene->seed, (uint32_t)(c->scene->seed >> 32)};
  rt_oracle_philox4x32_10(ctr, key, w);
}
/* rand 0.8 Standard f64: (next_u64() >> 11) * 2^-53  — used by rng.gen::<f64>() */
static inline double u01_53(uint32_t lo, uint32_t hi) {
  uint64_t u = ((uint64_t)hi << 32) | lo;
  return (double)(u >> 11) * (1.0 / 9007199254740992.0);
}
/* rand 0.8 gen_range(low..high) for f64 = value0_1 * (high-low) + low with value0_1 on a
 * uniform binary grid in [0,1).  The crate uses a 52-bit grid; we spend one 32-bit Philox
 * word per coordinate (grid 2^-32) so one Philox call feeds a whole rejection attempt. */
static inline double range_m1_1(uint32_t w) {
  double value0_1 = (double)w * (1.0 / 4294967296.0);
  double scale = 1.0 - (-1.0);
  return value0_1 * scale + (-1.0);
}
static inline uint32_t rt_child_node(uint32_t node, uint32_t light_j) {
  uint32_t x = node * 0x9E3779B1u + (light_j + 1u) * 0x85EBCA77u;
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15;
  return 0x80000000u | (x & 0x7FFFFFFEu);
}
void rt_oracle_draws(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t node, uint32_t slot,
                     double out_u01[2], double out_range[3]) {
  RtScene s; memset(&s, 0, sizeof s); s.seed = seed;
  Ctx c; memset(&c, 0, sizeof c); c.scene = &s; c.pixel = pixel; c.sample = sample;
  uint32_t w[4]; rng_words(&c, node, slot, w);
  out_u01[0] = u01_53(w[0], w[1]); out_u01[1] = u01_53(w[2], w[3]);
  out_range[0] = range_m1_1(w[0]); out_range[1] = range_m1_1(w[1]); out_range[2] = range_m1_1(w[2]);
}

/* point3d.rs:22-38: Point3D::random(-1,1) until length_squared < 1 */
static P3 random_in_unit_sphere(const Ctx* c, uint32_t node) {
  for (uint32_t attempt = 0;; ++attempt) {
    uint32_t w[4]; rng_words(c, node, 1u + attempt, w);
    P3 p = p3(range_m1_1(w[0]), range_m1_1(w[1]), range_m1_1(w[2]));
    if (p3_length_squared(p) < 1.0) return p;
  }
}

/* ------------------------------------------------------------------ sphere.rs */
typedef struct {
  double t; P3 point, normal; int front_face; uint32_t idx; double u, v;
} HitRecord;                                                                    /* ray.rs:23-31 */

/* sphere.rs:35-43 */
static void u_v_from_sphere_hit_point(P3 hp, double* u, double* v) {
  P3 n = p3_unit(hp);
  *u = (rt_atan2(n.x, n.z) / (2.0 * 3.14159265358979323846264338327950288)) + 0.5;
  *v = n.y * 0.5 + 0.5;
}
/* sphere.rs:46-78 */
static int sphere_hit(P3 center, double radius, Ray ray, double t_min, double t_max, HitRecord* rec) {
  P3 oc = p3_sub(ray.origin, center);
  double a = p3_length_squared(ray.direction);
  double half_b = p3_dot(oc, ray.direction);
  double c = p3_length_squared(oc) - radius * radius;
  double discriminant = (half_b * half_b) - (a * c);
  if (discriminant >= 0.0) {
    double sqrtd = sqrt(discriminant);
    double roots[2] = {((-half_b) - sqrtd) / a, ((-half_b) + sqrtd) / a};
    for (int i = 0; i < 2; ++i) {
      double root = roots[i];
      if (root < t_max && root > t_min) {
        P3 p = ray_at(ray, root);
        P3 normal = p3_divs(p3_sub(p, center), radius);
        int front_face = p3_dot(ray.direction, normal) < 0.0;
        double u, v;
        u_v_from_sphere_hit_point(p3_sub(p, center), &u, &v);
        rec->t = root; rec->point = p;
        rec->normal = front_face ? normal : p3_neg(normal);
        rec->front_face = front_face; rec->u = u; rec->v = v;
        return 1;
      }
    }
  }
  return 0;
}
int rt_oracle_sphere_hit(const double center[3], double radius, const double origin[3],
                         const double dir[3], double t_min, double t_max, double out[10]) {
  Ray r = {p3(origin[0], origin[1], origin[2]), p3(dir[0], dir[1], dir[2])};
  HitRecord h; memset(&h, 0, sizeof h);
  int hit = sphere_hit(p3(center[0], center[1], center[2]), radius, r, t_min, t_max, &h);
  out[0] = h.t; out[1] = h.point.x; out[2] = h.point.y; out[3] = h.point.z;
  out[4] = h.normal.x; out[5] = h.normal.y; out[6] = h.normal.z; out[7] = h.front_face;
  out[8] = h.u; out[9] = h.v;
  return hit;
}

double rt_oracle_atan2(double y, double x) { return rt_atan2(y, x); }
void rt_oracle_atan2_v(const double* y, const double* x, double* out, uint64_t n) {
  for (uint64_t i = 0; i < n; ++i) out[i] = rt_atan2(y[i], x[i]);
}

/* hooks for the reference's vector / ray unit tests (point3d.rs:197-272, ray.rs:38-63) */
int rt_oracle_p3_op(int op, const double a[3], const double b[3], double s, double out[3]) {
  P3 A = p3(a[0], a[1], a[2]), B = b ? p3(b[0], b[1], b[2]) : p3(0, 0, 0), R = p3(0, 0, 0);
  switch (op) {
    case 0: R = p3_add(A, B); break;                                             /* :89-99   */
    case 1: R = p3_sub(A, B); break;                                             /* :101-111 */
    case 2: R = p3_neg(A); break;                                                /* :113-123 */
    case 3: R = p3(A.x * B.x, A.y * B.y, A.z * B.z); break;                      /* :125-135 Mul<Point3D> */
    case 4: R = p3(A.x / B.x, A.y / B.y, A.z / B.z); break;                      /* :149-159 Div<Point3D> */
    case 5: R = p3_muls(A, s); break;                                            /* :137-147 */
    case 6: R = p3_divs(A, s); break;                                            /* :161-171 */
    case 7: R = p3(p3_dot(A, B), 0, 0); break;                                   /* :72-74   */
    case 8: R = p3(p3_length_squared(A), 0, 0); break;                           /* :59-61   */
    case 9: R = p3((double)p3_near_zero(A), 0, 0); break;                        /* :84-86   */
    case 10: R = p3(p3_length(A), 0, 0); break;                                  /* :63-65   */
    case 11: R = p3_unit(A); break;                                              /* :67-70   */
    case 12: R = p3_cross(A, B); break;                                          /* :76-82   */
    default: return -1;
  }
  out[0] = R.x; out[1] = R.y; out[2] = R.z;
  return 0;
}
void rt_oracle_ray_at(const double origin[3], const double dir[3], double t, double out[3]) {  /* ray.rs:18-20 */
  Ray r = {p3(origin[0], origin[1], origin[2]), p3(dir[0], dir[1], dir[2])};
  P3 q = ray_at(r, t);
  out[0] = q.x; out[1] = q.y; out[2] = q.z;
}

/* raytracer.rs:44-59 */
static int hit_world(Ctx* c, Ray r, double t_min, double t_max, HitRecord* best) {
  const RtScene* sc = c->scene;
  double closest_so_far = t_max;
  int any = 0;
  c->segments++;
  if (c->discarding) c->segments_discarded++;
  for (uint32_t i = 0; i < sc->n_spheres; ++i) {
    const RtSphere* s = &sc->spheres[i];
    HitRecord h;
    if (sphere_hit(p3(s->center[0], s->center[1], s->center[2]), s->radius, r, t_min, closest_so_far, &h)) {
      closest_so_far = h.t;
      h.idx = i;
      *best = h;
      any = 1;
    }
  }
  return any;
}

/* ------------------------------------------------------------------ materials.rs */
static P3 reflect(P3 v, P3 n) { return p3_sub(v, p3_muls(n, 2.0 * p3_dot(v, n))); }      /* :111-113 */
static P3 refract(P3 uv, P3 n, double etai_over_etat) {                                  /* :144-149 */
  double cos_theta = fmin(p3_dot(p3_neg(uv), n), 1.0);
  P3 r_out_perp = p3_muls(p3_add(uv, p3_muls(n, cos_theta)), etai_over_etat);
  P3 r_out_parallel = p3_muls(n, -1.0 * sqrt(fabs(1.0 - p3_length_squared(r_out_perp))));
  return p3_add(r_out_perp, r_out_parallel);
}
static double reflectance(double cosine, double ref_idx) {                               /* :151-155 */
  double r0 = (1.0 - ref_idx) / (1.0 + ref_idx);
  r0 = r0 * r0;
  double x = 1.0 - cosine;      /* powi(5) = x * ((x*x)*(x*x)), compiler-rt __powidf2 order */
  double x2 = x * x, x4 = x2 * x2;
  return r0 + (1.0 - r0) * (x * x4);
}
void rt_oracle_reflect(const double v[3], const double n[3], double out[3]) {
  P3 r = reflect(p3(v[0], v[1], v[2]), p3(n[0], n[1], n[2]));
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void rt_oracle_refract(const double uv[3], const double n[3], double e, double out[3]) {
  P3 r = refract(p3(uv[0], uv[1], uv[2]), p3(n[0], n[1], n[2]), e);
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
double rt_oracle_reflectance(double cosine, double ref_idx) { return reflectance(cosine, ref_idx); }

/* Rust `f64 as u64` / `f32 as usize`: saturating, NaN -> 0 */
static inline uint64_t sat_u64(double x) {
  if (!(x > 0.0)) return 0;
  if (x >= 18446744073709551616.0) return UINT64_MAX;
  return (uint64_t)x;
}
/* materials.rs:236-254 Texture::get_albedo.  The reference panics on an out-of-range index;
 * we clamp to the last texel and count it in tex_oob. */
static Rgb texture_albedo(Ctx* c, const RtSphere* s, const RtTexture* tex, double u, double v) {
  double rot = u + s->h_offset;
  if (rot > 1.0) rot = rot - 1.0;
  double uu = rot * (double)s->tex_w;
  double vv = (1.0 - v) * (double)(s->tex_h - 1);
  uint64_t base_pixel = 3 * (sat_u64(floor(vv)) * s->tex_w + sat_u64(floor(uu)));
  if (tex->nbytes < 3 || base_pixel > tex->nbytes - 3) {
    if (c) c->tex_oob++;
    base_pixel = tex->nbytes >= 3 ? (tex->nbytes / 3 - 1) * 3 : 0;
    if (tex->nbytes < 3) return rgb(0.f, 0.f, 0.f);
  }
  uint8_t pr = tex->rgb8[base_pixel], pg = tex->rgb8[base_pixel + 1], pb = tex->rgb8[base_pixel + 2];
  return rgb((float)pr / 255.0f, (float)pg / 255.0f, (float)pb / 255.0f);
}
void rt_oracle_texture_albedo(const RtSphere* s, const RtTexture* tex, double u, double v, float out[3]) {
  Rgb a = texture_albedo(NULL, s, tex, u, v);
  out[0] = a.r; out[1] = a.g; out[2] = a.b;
}

enum { SCATTER_ABSORBED = 0, SCATTER_EMIT = 1, SCATTER_RAY = 2 };
/* materials.rs:44-54 dispatch; returns Option<(Option<Ray>, Srgb)> as a status code */
static int material_scatter(Ctx* c, uint32_t node, Ray ray, const HitRecord* rec, Ray* scattered, Rgb* attenuation) {
  const RtSphere* s = &c->scene->spheres[rec->idx];
  switch (s->kind) {
    case RT_MAT_LIGHT: /* :65-69 */
      *attenuation = rgb(1.0f, 1.0f, 1.0f);
      return SCATTER_EMIT;
    case RT_MAT_LAMBERTIAN: /* :84-95 */
    case RT_MAT_TEXTURE: {  /* :256-267 */
      P3 scatter_direction = p3_add(rec->normal, random_in_unit_sphere(c, node));
      if (p3_near_zero(scatter_direction)) scatter_direction = rec->normal;
      P3 target = p3_add(rec->point, scatter_direction);
      scattered->origin = rec->point;
      scattered->direction = p3_sub(target, rec->point);
      if (s->kind == RT_MAT_TEXTURE)
        *attenuation = texture_albedo(c, s, &c->scene->textures[s->tex_id], rec->u, rec->v);
      else
        *attenuation = rgb(s->albedo[0], s->albedo[1], s->albedo[2]);
      return SCATTER_RAY;
    }
    case RT_MAT_METAL: { /* :115-129; the RNG is consumed even when fuzz == 0 */
      P3 reflected = reflect(ray.direction, rec->normal);
      scattered->origin = rec->point;
      scattered->direction = p3_add(reflected, p3_muls(random_in_unit_sphere(c, node), s->fuzz_or_ior));
      *attenuation = rgb(s->albedo[0], s->albedo[1], s->albedo[2]);
      return p3_dot(scattered->direction, rec->normal) > 0.0 ? SCATTER_RAY : SCATTER_ABSORBED;
    }
    case RT_MAT_GLASS: { /* :176-199 */
      *attenuation = rgb(1.0f, 1.0f, 1.0f);
      double refraction_ratio = rec->front_face ? 1.0 / s->fuzz_or_ior : s->fuzz_or_ior;
      P3 unit_direction = p3_unit(ray.direction);
      double cos_theta = fmin(p3_dot(p3_neg(unit_direction), rec->normal), 1.0);
      double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
      int cannot_refract = refraction_ratio * sin_theta > 1.0;
      int do_reflect = cannot_refract;
      if (!do_reflect) { /* short-circuit `||`: the draw happens only here (:189) */
        uint32_t w[4]; rng_words(c, node, 0, w);
        do_reflect = reflectance(cos_theta, refraction_ratio) > u01_53(w[0], w[1]);
      }
      scattered->origin = rec->point;
      scattered->direction = do_reflect ? reflect(unit_direction, rec->normal)
                                        : refract(unit_direction, rec->normal, refraction_ratio);
      return SCATTER_RAY;
    }
    default:
      *attenuation = rgb(0.f, 0.f, 0.f);
      return SCATTER_ABSORBED;
  }
}

/* ------------------------------------------------------------------ raytracer.rs */
static inline float clampf(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); } /* :61-69 */

/* Rust `f32 as usize`: saturating, NaN -> 0 */
static inline uint64_t sat_usize_f32(float x) {
  if (!(x > 0.0f)) return 0;
  if (x >= 18446744073709551616.0f) return UINT64_MAX;
  return (uint64_t)x;
}

/* raytracer.rs:71-165.  `node` addresses the RNG; `nest` counts nested light-ray levels. */
static Rgb ray_color(Ctx* c, Ray ray, uint32_t max_depth, uint32_t depth, uint32_t node, uint32_t nest) {
  const RtScene* sc = c->scene;
  if (depth == 0) return rgb(0.0f, 0.0f, 0.0f);                                         /* :80-82 */
  HitRecord rec; memset(&rec, 0, sizeof rec);
  if (hit_world(c, ray, 0.001, 1.7976931348623157e308, &rec)) {                         /* :83 */
    Ray sr; Rgb albedo;
    int st = material_scatter(c, node, ray, &rec, &sr, &albedo);                        /* :86 */
    if (st == SCATTER_ABSORBED) return rgb(0.0f, 0.0f, 0.0f);                           /* :127-131 */
    float light_red = 0.0f, light_green = 0.0f, light_blue = 0.0f;
    double prob = sc->spheres[rec.idx].kind == RT_MAT_GLASS ? 0.05 : 0.1;              /* :92-98 */
    /* :99-102.  `depth > max_depth - 2` is usize arithmetic: for max_depth < 2 it wraps in
     * release builds (the README's `cargo run --release`) and the test is false. */
    int depth_ok = max_depth >= 2 && depth > max_depth - 2;
    if (c->n_lights > 0 && depth_ok && nest < RT_MAX_LIGHT_NEST) {
      /* the draw's words: see "RNG addressing" above (a Glass hit's come from its slot-0 call, any other hit's high word
       * is the fourth word of attempt 0's call) */
      uint32_t w[4]; rng_words(c, node, 0, w);
      uint32_t high = w[3];
#ifndef RT_ORACLE_INDEPENDENT_LIGHT_DRAW
      /* (librt_oracle_indep.so is built WITH the macro: every hit's draw is slot 0's words 2 and 3, a Philox block no other
       *  draw of the node touches — round 3's addressing.  Only tests/test_light_draw_statistics.py uses it: an independent
       *  reference for the statistics of the shared-word addressing below; its images differ, by design.) */
      if (sc->spheres[rec.idx].kind != RT_MAT_GLASS) { uint32_t a0[4]; rng_words(c, node, 1, a0); high = a0[3]; }
#endif
      if (u01_53(w[2], high) > (1.0 - (double)c->n_lights * prob)) {
        if (st == SCATTER_EMIT) c->discarding++;  /* (counter only: :124 returns `albedo`, not the light sum) */
        for (uint32_t j = 0; j < c->n_lights; ++j) {                                    /* :103-110 */
          const RtSphere* light = &sc->spheres[c->lights[j]];
          Ray light_ray = {rec.point, p3_sub(p3(light->center[0], light->center[1], light->center[2]), rec.point)};
          Rgb tc = ray_color(c, light_ray, 2, 1, rt_child_node(node, j), nest + 1);
          light_red += albedo.r * tc.r;
          light_green += albedo.g * tc.g;
          light_blue += albedo.b * tc.b;
        }
        light_red /= (float)c->n_lights;                                                /* :111-113 */
        light_green /= (float)c->n_lights;
        light_blue /= (float)c->n_lights;
        if (st == SCATTER_EMIT) c->discarding--;
      }
    }
    if (st == SCATTER_RAY) {                                                            /* :116-123 */
      Rgb tc = ray_color(c, sr, max_depth, depth - 1, node + 1u, nest);
      return rgb(clampf(light_red + albedo.r * tc.r), clampf(light_green + albedo.g * tc.g),
                 clampf(light_blue + albedo.b * tc.b));
    }
    return albedo;                                                                      /* :124 */
  }
  /* miss: sky, :134-163 */
  P3 ud = p3_unit(ray.direction);
  float t = clampf(0.5f * ((float)ud.y + 1.0f));
  float u = clampf(0.5f * ((float)ud.x + 1.0f));
  if (sc->sky_mode == RT_SKY_NONE) return rgb(0.0f, 0.0f, 0.0f);
  if (sc->sky_mode == RT_SKY_GRADIENT)
    return rgb((1.0f - t) * 1.0f + t * 0.5f, (1.0f - t) * 1.0f + t * 0.7f, (1.0f - t) * 1.0f + t * 1.0f);
  uint64_t x = sat_usize_f32(u * (float)(sc->sky_w - 1));
  uint64_t y = sat_usize_f32((1.0f - t) * (float)(sc->sky_h - 1));
  uint64_t base = (y * sc->sky_w + x) * 3;
  if (base + 2 >= sc->sky_w * sc->sky_h * 3) { c->tex_oob++; base = (sc->sky_w * sc->sky_h - 1) * 3; }
  const uint8_t* px = sc->sky_rgb8 + base;
  return rgb(0.7f * (float)px[0] / 255.0f, 0.7f * (float)px[1] / 255.0f, 0.7f * (float)px[2] / 255.0f);
}

/* camera.rs:79-84 */
static Ray camera_get_ray(const RtScene* sc, double u, double v) {
  P3 origin = p3(sc->cam_origin[0], sc->cam_origin[1], sc->cam_origin[2]);
  P3 llc = p3(sc->cam_lower_left[0], sc->cam_lower_left[1], sc->cam_lower_left[2]);
  P3 hor = p3(sc->cam_horizontal[0], sc->cam_horizontal[1], sc->cam_horizontal[2]);
  P3 ver = p3(sc->cam_vertical[0], sc->cam_vertical[1], sc->cam_vertical[2]);
  Ray r = {origin, p3_sub(p3_add(p3_add(llc, p3_muls(hor, u)), p3_muls(ver, v)), origin)};
  return r;
}
void rt_oracle_get_ray(const RtScene* sc, double u, double v, double out[6]) {
  Ray r = camera_get_ray(sc, u, v);
  out[0] = r.origin.x; out[1] = r.origin.y; out[2] = r.origin.z;
  out[3] = r.direction.x; out[4] = r.direction.y; out[5] = r.direction.z;
}
/* camera.rs:45-77 */
void rt_oracle_camera_new(const double lf[3], const double la[3], const double up[3], double vfov,
                          double aspect, double out[13]) {
  double theta = vfov * (3.14159265358979323846264338327950288 / 180.0); /* f64::to_radians */
  double half_height = tan(theta / 2.0);
  double half_width = aspect * half_height;
  P3 look_from = p3(lf[0], lf[1], lf[2]), look_at = p3(la[0], la[1], la[2]), vup = p3(up[0], up[1], up[2]);
  P3 w = p3_unit(p3_sub(look_from, look_at));
  P3 u = p3_unit(p3_cross(vup, w));
  P3 v = p3_cross(w, u);
  P3 origin = look_from;
  P3 llc = p3_sub(p3_sub(p3_sub(origin, p3_muls(u, half_width)), p3_muls(v, half_height)), w);
  P3 horizontal = p3_muls(p3_muls(u, 2.0), half_width);
  P3 vertical = p3_muls(p3_muls(v, 2.0), half_height);
  out[0] = origin.x; out[1] = origin.y; out[2] = origin.z;
  out[3] = llc.x; out[4] = llc.y; out[5] = llc.z;
  out[6] = horizontal.x; out[7] = horizontal.y; out[8] = horizontal.z;
  out[9] = vertical.x; out[10] = vertical.y; out[11] = vertical.z;
  out[12] = p3_length(p3_sub(look_from, look_at));
}

/* raytracer.rs:220-229 */
uint32_t rt_oracle_find_lights(const RtSphere* spheres, uint32_t n, uint32_t* out_idx, uint32_t cap) {
  uint32_t k = 0;
  for (uint32_t i = 0; i < n; ++i)
    if (spheres[i].kind == RT_MAT_LIGHT) { if (out_idx && k < cap) out_idx[k] = i; ++k; }
  return k;
}

/* raytracer.rs:213 `color.into_format().into_raw()`: palette 0.6 FromComponent<f32> for u8, restated from the
 * crate's source as  scaled = (x * 255.0).min(255.0);  (scaled + 2^23).to_bits().saturating_sub(bits(2^23)) as u8
 * i.e. round-to-nearest-even of min(x*255, 255); negatives -> 0 (the subtraction saturates); and NaN -> 255,
 * because Rust's f32::min returns its non-NaN operand.  THIRD-PARTY AND UNPINNED: the crate is not under
 * /root/reference and no reference test exercises it; oracle and kernel share this definition.  NaN pixels
 * only arise for frames one pixel wide or high (raytracer.rs:199-200 divides by width-1 / height-1). */
uint8_t rt_oracle_f32_to_u8(float x) {
  float scaled = x * 255.0f;
  if (scaled != scaled) return 255;
  if (!(scaled > 0.0f)) return 0;
  if (scaled > 255.0f) scaled = 255.0f;
  return (uint8_t)nearbyintf(scaled);
}

void rt_oracle_ray_color(const RtScene* scene, const double o[3], const double d[3], uint32_t max_depth,
                         uint32_t depth, uint32_t pixel, uint32_t sample, float out[3]) {
  uint32_t lights[256];
  Ctx c; memset(&c, 0, sizeof c);
  c.scene = scene; c.pixel = pixel; c.sample = sample;
  c.n_lights = rt_oracle_find_lights(scene->spheres, scene->n_spheres, lights, 256);
  c.lights = lights;
  Ray r = {p3(o[0], o[1], o[2]), p3(d[0], d[1], d[2])};
  Rgb col = ray_color(&c, r, max_depth, depth, max_depth - depth, 0);
  out[0] = col.r; out[1] = col.g; out[2] = col.b;
}

/* raytracer.rs:191-218 for pixels [x0, x1) of scanline y; out_row/out_lin are that row's 3*w bytes / floats */
static void render_line(Ctx* c, uint32_t y, uint32_t x0, uint32_t x1, uint8_t* out_row, float* out_lin) {
  const RtScene* sc = c->scene;
  const uint32_t w = sc->width, h = sc->height;
  for (uint32_t x = x0; x < x1; ++x) {
    float pixel_colors[3] = {0.0f, 0.0f, 0.0f};
    c->pixel = y * w + x;
    for (uint32_t s = 0; s < sc->samples_per_pixel; ++s) {
      c->sample = s;
      uint32_t jw[4]; rng_words(c, NODE_CAMERA, 0, jw);
      double u = ((double)x + u01_53(jw[0], jw[1])) / ((double)w - 1.0);                 /* :199 */
      double v = ((double)h - ((double)y + u01_53(jw[2], jw[3]))) / ((double)h - 1.0);   /* :200 */
      Ray r = camera_get_ray(sc, u, v);
      Rgb col = ray_color(c, r, sc->max_depth, sc->max_depth, 0, 0);
      pixel_colors[0] += col.r; pixel_colors[1] += col.g; pixel_colors[2] += col.b;
    }
    float scale = 1.0f / (float)sc->samples_per_pixel;
    for (int k = 0; k < 3; ++k) {
      float lin = scale * pixel_colors[k];
      if (out_lin) out_lin[3 * x + k] = lin;
      if (out_row) out_row[3 * x + k] = rt_oracle_f32_to_u8(sqrtf(lin));                 /* :207-216 */
    }
  }
}

int rt_oracle_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

static double now_ms(void) {
  struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* raytracer.rs:250-263.  The reference hands whole scanlines to rayon (:255-262); pixels are independent, so the
 * checker cuts a scanline into blocks of RT_ORACLE_XBLOCK pixels to keep every core busy when only a few rows are
 * asked for (the full-size parity tests compare single 4K rows at spp 1024).  Same pixels, same bits. */
#define RT_ORACLE_XBLOCK 32u
static int render_rows(const RtScene* scene, const RtRowTiles* tiles, uint32_t x0, uint32_t x1, uint8_t* rgb8, float* linear,
                       RtStats* stats, int n_threads) {
  if (!scene || scene->abi_version != RT_ABI_VERSION) return RT_ERR_INVALID;
  if (scene->width == 0 || scene->height == 0 || (scene->n_spheres && !scene->spheres)) return RT_ERR_INVALID;
  if (x1 > scene->width) x1 = scene->width;
  if (x0 > x1) x0 = x1;
  uint32_t n_lights = rt_oracle_find_lights(scene->spheres, scene->n_spheres, NULL, 0);
  uint32_t* lights = (uint32_t*)malloc(sizeof(uint32_t) * (n_lights ? n_lights : 1));
  rt_oracle_find_lights(scene->spheres, scene->n_spheres, lights, n_lights);
  const uint32_t rows = rt_tiles_local_rows(scene->height, tiles);
  const size_t row_elems = (size_t)scene->width * 3;
  const uint32_t xblocks = (x1 - x0 + RT_ORACLE_XBLOCK - 1) / RT_ORACLE_XBLOCK;
  const uint64_t n_tasks = (uint64_t)rows * xblocks;
  uint64_t segments = 0, tex_oob = 0, discarded = 0;
#ifdef _OPENMP
  if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
  n_threads = 1;
#endif
  double t0 = now_ms();
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads) reduction(+ : segments, tex_oob, discarded)
  for (uint64_t task = 0; task < n_tasks; ++task) {
    const uint32_t lr = (uint32_t)(task / xblocks), xb = (uint32_t)(task % xblocks);
    Ctx c; memset(&c, 0, sizeof c);
    c.scene = scene; c.lights = lights; c.n_lights = n_lights;
    uint32_t y = rt_tiles_global_row(tiles, lr);
    const uint32_t xa = x0 + xb * RT_ORACLE_XBLOCK, xe = xa + RT_ORACLE_XBLOCK < x1 ? xa + RT_ORACLE_XBLOCK : x1;
    render_line(&c, y, xa, xe, rgb8 ? rgb8 + lr * row_elems : NULL, linear ? linear + lr * row_elems : NULL);
    segments += c.segments; tex_oob += c.tex_oob; discarded += c.segments_discarded;
  }
  double t1 = now_ms();
  free(lights);
  if (stats) {
    memset(stats, 0, sizeof *stats);
    stats->samples = (uint64_t)rows * (x1 - x0) * scene->samples_per_pixel;
    stats->segments = segments;
    stats->sphere_tests = segments * scene->n_spheres;
    stats->exact_tests = segments * scene->n_spheres;
    stats->tex_oob = tex_oob;
    stats->kernel_ms = t1 - t0;
    stats->frame_ms = t1 - t0;
    stats->grid_steps = 0; /* the reference has no acceleration structure (raytracer.rs:52-57) */
    stats->segments_discarded = discarded;
    stats->n_gpus_used = 0;
  }
  return RT_OK;
}
int rt_oracle_render(const RtScene* scene, const RtRowTiles* tiles, uint8_t* rgb8, float* linear,
                     RtStats* stats, int n_threads) {
  return render_rows(scene, tiles, 0, scene ? scene->width : 0, rgb8, linear, stats, n_threads);
}
/* pixels [x0, x1) of the selected rows only; the buffers still hold whole rows (other pixels untouched) */
int rt_oracle_render_window(const RtScene* scene, const RtRowTiles* tiles, uint32_t x0, uint32_t x1, uint8_t* rgb8,
                            float* linear, RtStats* stats, int n_threads) {
  return render_rows(scene, tiles, x0, x1, rgb8, linear, stats, n_threads);
}
