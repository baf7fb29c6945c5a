/*
 * rt_abi_test.h — device probes and diagnostics of the MI355X path tracer: TEST INFRASTRUCTURE, not part of the seam.
 *
 * These entry points exist in librt_hip_probe.so only (the same sources as librt_hip.so compiled with -DRT_TEST_PROBES;
 * rust-raytracer_amd/build.py) — the product library exports none of them.  tests/test_gpu_parity.py and tools/diag.py
 * bind them; a host that replaces raytracer.rs:260-262 needs include/rt_abi.h alone.
 */
#ifndef RT_ABI_TEST_H
#define RT_ABI_TEST_H

#include "rt_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostics (builds with -DRT_PROFILE only; other builds return stale memory): the 32 raw
 * launch counters, then {start, end, time the
 * wave found the tile queue empty, iterations after that | lanes x iterations << 32} of every
 * wave of the last launch, on the chip-wide 100 MHz clock.  Returns the number of waves copied
 * (out holds 32 + 4 x max_waves uint64) or a negative RtStatus. */
int rt_hip_debug_timeline(RtHipScene*, uint64_t* out, uint32_t max_waves);
/* Diagnostics: the deepest camera path per pixel tile that the last MEASURING frame recorded (tile_order 2: the first two
 * frames of a view) — what the queue order of later frames is sorted by.  out[tile], row-major over the launch's tile grid
 * (*tiles_x tiles wide); returns the number of tiles copied (<= cap) or a negative RtStatus. */
int rt_hip_debug_tile_depth(RtHipScene*, uint32_t* out, uint32_t cap, uint32_t* tiles_x);
/* Device self-tests (device pointers; tests/test_gpu_parity.py): correctly rounded f64 sqrt / divide,
 * f32 sqrt and atan2 of n operands; Sphere::hit (sphere.rs:46-58) of n (ray, sphere) pairs through
 * the kernel's own hit test — rays = n x {origin[3], direction[3]}, spheres = n x {center[3], radius},
 * out_t = the accepted root with t_max = f64::MAX, or -1. */
int rt_hip_math_probe(const double* x, const double* y, double* out_sqrt, double* out_div, float* out_sqrtf, double* out_atan2,
                      uint32_t n, void* stream);
int rt_hip_hit_probe(const double* rays, const double* spheres, double* out_t, uint32_t n, void* stream);
/* f64::atan2 (sphere.rs:39) of n (y, x) pairs through the device build of the routine kernel and CPU checker share
 * (csrc/common/rt_atan2.h): tests compare it with a committed fixture of correctly rounded results. */
int rt_hip_atan2_probe(const double* d_y, const double* d_x, double* d_out, uint32_t n, void* stream);
/* The texel of a Texture hit on the device, both ways (materials.rs:236-254 through sphere.rs:35-43): the kernel's fast
 * (u, v) — v_rsq_f64 / v_rcp_f64 + Newton steps, which only the device build takes — beside the exact path, for n hit
 * points (device, 3 doubles each) on the sphere centre_radius (host, 4 doubles).  d_out = n x {fast_ok, fast col, fast
 * row, exact col, exact row} (u64); d_uv (optional) = n x {fast u, fast v, exact u, exact v}, fast u = NaN where the fast
 * path declined.  rt_hip_quot_probe: rt_fast_quot(x, y), rt_fast_rsqrt(x) and (d_div optional) rt_div_inrange(x, y) — the
 * library division without range scaling and fix-up: must equal the IEEE quotient — of n positive normal operands. */
int rt_hip_texel_probe(const double* d_points, const double centre_radius[4], double h_offset, uint64_t tex_w, uint64_t tex_h,
                       uint64_t* d_out, double* d_uv, uint32_t n, void* stream);
int rt_hip_quot_probe(const double* d_x, const double* d_y, double* d_quot, double* d_rsqrt, double* d_div, uint32_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RT_ABI_TEST_H */
