/*
 * rt_abi.h — C ABI of the MI355X path-tracer hot path.
 *
 * The reference (dps/rust-raytracer) has NO FFI of its own: the seam this
 * header defines is cut at the rayon closure in
 *   raytracer/src/raytracer.rs:260-262   bands.into_par_iter().for_each(render_line)
 * Inputs of that closure are the immutable `Config` (raytracer/src/config.rs:66-75)
 * and the light list (`find_lights`, raytracer.rs:220-229); its output is the
 * `pixels: Vec<u8>` framebuffer (raytracer.rs:254), row-major, top row first, RGB8.
 * Everything in this file is plain C: pointers, sizes, POD structs.  No torch /
 * HIP types appear in signatures (streams and device pointers travel as void*).
 *
 * Two product libraries implement it (the CPU checker under oracle/ reuses the structs
 * below but is test infrastructure and is never linked into either):
 *   librt_host.so  (C++, product)   scene JSON/JPEG/PNG/camera — the host plumbing
 *                                   that stays on the CPU (main.rs, config.rs, camera.rs)
 *   librt_hip.so   (HIP,  product)  the gfx950 megakernel = render_line/ray_color/hit_world
 */
#ifndef RT_ABI_H
#define RT_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 5u /* v5: RtGroupInfo.transport_fallback, RtGroupRank + rt_hip_group_ranks, rt_hip_group_fallback_reason; the device probes and
                           * debug calls moved to rt_abi_test.h (librt_hip_probe.so); v4: RtStats.group_us, rt_hip_group_submit / _collect;
                           * v3: RtScene.n_gpus, RtStats.{segments_discarded, n_gpus_used, gather_ms, setup_ms}, rt_abi_sizeof */

/* Nested light-ray recursion (raytracer.rs:103-110 calls ray_color(.., 2, 1), which can
 * itself trigger light sampling again) is unbounded in the reference.  Oracle and kernel
 * both stop spawning new light rays below this nesting level (level 0 = camera path). */
#define RT_MAX_LIGHT_NEST 8u

/* ---- error codes (the reference panics; a C ABI returns codes instead) ---- */
enum {
  RT_OK = 0,
  RT_ERR_INVALID = -1,   /* null pointer / inconsistent scene                      */
  RT_ERR_NO_DEVICE = -2, /* no gfx950 device visible                               */
  RT_ERR_HIP = -3,       /* a HIP runtime call failed (see rt_hip_last_error)      */
  RT_ERR_IO = -4,        /* main.rs:14 "Unable to read config file."               */
  RT_ERR_PARSE = -5,     /* main.rs:15 "Unable to parse config json"               */
  RT_ERR_TEXTURE = -6,   /* materials.rs:214-217 / config.rs:37-40 texture failure */
  RT_ERR_PNG = -7,       /* raytracer.rs:265 "error writing image"                 */
  RT_ERR_UNSUPPORTED = -8
};

/* materials.rs:35-42  enum Material, in declaration order */
enum {
  RT_MAT_LAMBERTIAN = 0,
  RT_MAT_METAL = 1,
  RT_MAT_GLASS = 2,
  RT_MAT_TEXTURE = 3,
  RT_MAT_LIGHT = 4
};

/* config.rs:22-28, 49-64: sky null -> black; {"texture":""} -> gradient; path -> texture */
enum { RT_SKY_NONE = 0, RT_SKY_GRADIENT = 1, RT_SKY_TEXTURE = 2 };

/* sphere.rs:18-23 + the material payloads (materials.rs:71-76, 97-103, 131-134, 201-211).
 * Spheres are kept in JSON object order: closest-hit ties (raytracer.rs:52-57) and the
 * light order (raytracer.rs:220-229) follow it. */
typedef struct RtSphere {
  double center[3];
  double radius;      /* may be negative: hollow glass, test_scene.json:137 */
  double fuzz_or_ior; /* Metal.fuzz | Glass.index_of_refraction             */
  double h_offset;    /* Texture.h_offset                                   */
  uint64_t tex_w;     /* Texture.width / height AS WRITTEN IN THE JSON      */
  uint64_t tex_h;     /*   (materials.rs:206-210), not the decoded size     */
  float albedo[3];    /* Lambertian/Metal albedo; ignored for Texture       */
  uint32_t kind;      /* RT_MAT_*                                           */
  uint32_t tex_id;    /* index into RtScene.textures when kind==TEXTURE     */
  uint32_t reserved;
} RtSphere;

/* decoded texture pixels, RGB8 (materials.rs:213-219) */
typedef struct RtTexture {
  const uint8_t* rgb8;
  uint64_t nbytes;
  uint32_t width, height; /* decoded size, informational */
} RtTexture;

/* config.rs:66-75 Config, with the camera already derived (camera.rs:45-77). */
typedef struct RtScene {
  uint32_t abi_version; /* RT_ABI_VERSION */
  uint32_t width, height;
  uint32_t samples_per_pixel;
  uint32_t max_depth;
  uint32_t sky_mode; /* RT_SKY_* */
  double cam_origin[3];
  double cam_lower_left[3];
  double cam_horizontal[3];
  double cam_vertical[3];
  const uint8_t* sky_rgb8; /* sky_mode==TEXTURE */
  uint64_t sky_w, sky_h;
  const RtSphere* spheres;
  uint32_t n_spheres;
  uint32_t n_textures;
  const RtTexture* textures;
  uint64_t seed; /* Philox key; replaces rand::thread_rng (raytracer.rs:78,192) */
  uint32_t n_gpus;   /* rt_render_rgb8 only: GPUs to shard the frame over by interleaved scanline tiles;
                      * 0 = the RT_GPUS environment variable, else 1.  More than rt_hip_device_count()
                      * is RT_ERR_INVALID.  The frame is bit-identical for every value. */
  uint32_t reserved0;
} RtScene;

/* Which scanline tiles to render.  Tile k covers rows [k*tile_rows, (k+1)*tile_rows).
 * This call renders tiles first_tile, first_tile+tile_stride, ... and packs them back to
 * back in the output (local tile j <-> global tile first_tile + j*tile_stride).
 * NULL, or tile_rows==0, means the whole frame.  Rank r of G GPUs uses {T, r, G}. */
typedef struct RtRowTiles {
  uint32_t tile_rows;
  uint32_t first_tile;
  uint32_t tile_stride;
} RtRowTiles;

typedef struct RtStats {
  uint64_t samples;      /* camera paths traced = pixels * spp                              */
  uint64_t segments;     /* ray_color invocations that ran hit_world (raytracer.rs:83)      */
  uint64_t sphere_tests; /* ALGORITHMIC ray-sphere tests = segments * n_spheres             */
  uint64_t exact_tests;  /* f64 Sphere::hit evaluations actually executed (after culling)   */
  uint64_t tex_oob;      /* texture fetches the reference would have panicked on            */
  double kernel_ms;      /* device time of the render kernel(s)                             */
  double frame_ms;       /* wall time of the whole call                                     */
  uint64_t grid_steps;   /* cells entered by the grid walks of hit_world (0: no grid)       */
  /* wave-level trip counts (diagnostics of SIMT efficiency; 0 from the CPU oracle):
   * [0] iterations of the segment loop, [1] of the cell-step loop, [2] of the exact-test loop,
   * [3] work items.  lane utilisation of hit_world = segments / (64 * wave_iters[0]) */
  uint64_t wave_iters[4];
  /* shader-clock cycles summed over waves per kernel section; only filled by builds with
   * -DRT_PROFILE (tools/ab_bench.py "prof" arm): [0] sample refill, [1] `large` list,
   * [2] lane_shade, [3] grid entry + walk, [4] pixel sums + tile bookkeeping, [5] item fetch, [6] whole wave */
  uint64_t prof_cycles[12]; /* ... [7] longest wave, [8] shortest wave, [10] waves launched */
  /* CPU oracle only (0 from the GPU): segments traced inside the light loop of a hit on a Light sphere,
   * whose sum raytracer.rs:124 throws away (`None => albedo`).  The kernel does not trace them:
   * kernel.segments == oracle.segments - oracle.segments_discarded, exactly. */
  uint64_t segments_discarded;
  uint32_t n_gpus_used; /* rt_render_rgb8: devices the frame was sharded over (1 elsewhere) */
  uint32_t segments_repeated; /* lit scenes whose light records are a pool: segments traced a second time because no record was
                               * free (counted once in `segments`; their exact tests and grid steps are counted as done); saturates */
  double gather_ms; /* rt_hip_group_*: what the frame spent NOT rendering, on device 0's clock: (start of rank 0's kernel -> frame in
                     * scanline order on device 0) minus kernel_ms of the slowest rank = launch skew between the ranks + the gather +
                     * the de-interleave.  (Until ABI v3 it started at the END of rank 0's kernel and so held the load imbalance.) */
  double setup_ms;  /* rt_render_rgb8: HIP context + table build + scene upload, NOT part of frame_ms
                     * (frame_ms is the window the reference times, raytracer.rs:259-263: the parallel loop
                     * until the pixels are in the caller's buffer) */
  /* rt_hip_group_*: host clock, microseconds since the frame's submit was entered (0 elsewhere):
   * [0] the last rank's host thread is running, [1] the last rank's launch (+ its side of the transfer) is enqueued,
   * [2] the submitting thread knows that, [3] the gather is enqueued (ncclGroupEnd returned / the peer copies' events are
   * waited for), [4] submit returns (de-interleave + device-to-host copy enqueued), [5] collect saw the frame assembled on
   * device 0, [6] ... and in the caller's buffer (= frame_ms), [7] the ranks' counters are read: collect returns.
   * A blocking frame's non-kernel time is frame_ms - kernel_ms; [5]/[6] of a pipelined frame include the wait of its collect. */
  double group_us[8];
} RtStats;

/* rows this call renders (its packed RGB8 output is rows*width*3 bytes) */
static inline uint32_t rt_tiles_local_rows(uint32_t height, const RtRowTiles* tiles) {
  if (!tiles || tiles->tile_rows == 0 || tiles->tile_stride == 0) return height;
  uint32_t rows = 0;
  uint64_t n_tiles = ((uint64_t)height + tiles->tile_rows - 1) / tiles->tile_rows;
  for (uint64_t k = tiles->first_tile; k < n_tiles; k += tiles->tile_stride) {
    uint64_t r0 = k * tiles->tile_rows, r1 = r0 + tiles->tile_rows;
    if (r1 > height) r1 = height;
    rows += (uint32_t)(r1 - r0);
  }
  return rows;
}
/* global row of local packed row `lr` (inverse of the packing above) */
static inline uint32_t rt_tiles_global_row(const RtRowTiles* tiles, uint32_t lr) {
  if (!tiles || tiles->tile_rows == 0 || tiles->tile_stride == 0) return lr;
  uint32_t j = lr / tiles->tile_rows, r = lr % tiles->tile_rows;
  return (tiles->first_tile + j * tiles->tile_stride) * tiles->tile_rows + r;
}

/* Where scanline y of the assembled frame sits in a gather buffer of G ranks, each contributing `pad_rows` packed rows
 * (rank r's rows are rt_tiles_global_row({tile_rows, r, G}, 0..)): the inverse of the packing above, used by the
 * de-interleave kernel of rt_hip_group_* and by any host that assembles the ranks' tiles itself. */
static inline uint32_t rt_tiles_stacked_row(uint32_t y, uint32_t n_ranks, uint32_t tile_rows, uint32_t pad_rows) {
  const uint32_t k = y / tile_rows, r = k % n_ranks, j = k / n_ranks;
  return r * pad_rows + j * tile_rows + y % tile_rows;
}

/* ------------------------------------------------------------------------------------
 * librt_host.so — host plumbing (stays on the CPU, like main.rs/config.rs/camera.rs)
 * ---------------------------------------------------------------------------------- */
typedef struct RtSceneFile RtSceneFile; /* owns the RtScene and every buffer it points to */

/* main.rs:14-15: read + parse a scene file; texture paths resolve relative to cwd. */
int rt_scene_load_file(const char* json_path, RtSceneFile** out);
int rt_scene_load_string(const char* json_text, size_t len, RtSceneFile** out);
/* where a load went, milliseconds: out = {reading the file, parsing the JSON text, the longest JPEG decode (the textures are
 * decoded concurrently, beside the parse), the whole call}; the CLI prints them under RT_STATS=1 */
void rt_scene_load_timings(const RtSceneFile*, double out[4]);
const RtScene* rt_scene_get(const RtSceneFile*);
RtScene* rt_scene_get_mut(RtSceneFile*); /* tests override width/height like raytracer.rs:272-273 */
void rt_scene_free(RtSceneFile*);
/* serde_json::to_string(&config) — exact strings of config.rs:101,128 */
int rt_scene_to_json(const RtSceneFile*, char* buf, size_t cap, size_t* needed);
const char* rt_host_last_error(void);

/* camera.rs:45-77 Camera::new: out = origin[3], lower_left[3], horizontal[3], vertical[3], focal_length */
void rt_camera_derive(const double look_from[3], const double look_at[3], const double vup[3],
                      double vfov_deg, double aspect, double out[13]);
/* camera description of a loaded scene file (camera.rs:29-36): look_from[3], look_at[3], vup[3], vfov, aspect */
void rt_scene_camera(const RtSceneFile*, double out[11]);
/* raytracer.rs:220-229 find_lights: writes indices of Light spheres in object order */
uint32_t rt_find_lights(const RtSphere* spheres, uint32_t n, uint32_t* out_idx, uint32_t cap);
/* materials.rs:213-219 load_texture_image: Huffman JPEG (baseline, extended sequential, progressive; 8 bit, 1 or 3
 * components) -> RGB8 (malloc'd, free with rt_free) */
int rt_jpeg_decode_file(const char* path, uint8_t** rgb8, uint32_t* w, uint32_t* h);
int rt_jpeg_decode_mem(const uint8_t* data, size_t len, uint8_t** rgb8, uint32_t* w, uint32_t* h);
const char* rt_jpeg_last_error(void); /* why the last rt_jpeg_decode_* of this thread returned RT_ERR_TEXTURE */
/* raytracer.rs:33-42 write_image: PNG, ColorType::RGB(8) */
int rt_png_write_rgb8(const char* path, const uint8_t* rgb8, uint32_t w, uint32_t h);
void rt_free(void*);

/* ------------------------------------------------------------------------------------
 * librt_hip.so — the hot path on gfx950 (replaces raytracer.rs:260-262)
 * ---------------------------------------------------------------------------------- */
typedef struct RtHipScene RtHipScene; /* scene tables + textures resident in HBM on one GPU */

/* Layout check for foreign-language bindings: sizeof of "RtSphere", "RtTexture",
 * "RtScene", "RtRowTiles", "RtStats" as THIS library was compiled, 0 for an unknown name.  A binding
 * compares them with its own struct sizes once at start-up; rt_abi_version() returns RT_ABI_VERSION. */
size_t rt_abi_sizeof(const char* struct_name);
uint32_t rt_abi_version(void);

int rt_hip_device_count(void);
/* Bring the runtime up on `device` ahead of the first scene: its context, its queues, this library's code object — tens of
 * milliseconds that rt_hip_scene_create otherwise pays for the first scene of a process.  Optional and idempotent; the CLI calls it
 * on a thread while it reads the scene file. */
int rt_hip_device_warm(int device);
const char* rt_hip_last_error(void);
/* Where set-up time went: the stages of the most recent rt_hip_group_create / rt_render_rgb8 of this process (rank 0's scene —
 * table build, texel conversion, uploads, kernel configuration — and the group's own work: streams, frame buffers, transport,
 * pinned staging) as one JSON object {"stage": milliseconds, ...}.  Diagnostics (the CLI prints it under RT_STATS=1); the
 * pointer is valid until the calling thread's next call. */
const char* rt_hip_setup_profile(void);
/* Upload scene tables, textures and sky to HBM of `device`.  The caller may free the
 * RtScene and everything it points to as soon as this returns. */
int rt_hip_scene_create(const RtScene* scene, int device, RtHipScene** out);
void rt_hip_scene_destroy(RtHipScene*);
/* Launch the megakernel for the given row tiles on `stream` (a hipStream_t, NULL = default).
 *   d_rgb8    device buffer, rt_tiles_local_rows()*width*3 bytes, packed, top row first;
 *             4-byte aligned (any hipMalloc'd buffer is)
 *   d_linear  optional device buffer of 3 floats per pixel: mean radiance before the
 *             sqrt gamma (raytracer.rs:207-212) — what the parity tests compare
 * Asynchronous: returns after enqueueing.  Inputs are already in HBM.
 * Limits (all return RT_ERR_* instead of misbehaving):
 *   - NOT re-entrant per scene: an RtHipScene owns ONE tile-queue cursor, counter block and event pair.
 *     Launches of one scene must be ordered on ONE stream at a time (back-to-back launches on the same
 *     stream are fine; rt_hip_wait then reports the last one).  Launching on a second stream while the
 *     first stream still holds unfinished work of this scene is RT_ERR_INVALID (finished = rt_hip_wait()
 *     returned, or the caller drained that stream itself: hipStreamQuery says so).  Use one RtHipScene per concurrent stream
 *     (tables are ~100 KB + textures); distinct scenes and distinct devices are fully independent.
 *   - frames wider than 524 280 pixels or with more than 2^31 pixel tiles are RT_ERR_UNSUPPORTED.
 *   - any sphere count is accepted (the reference scans a Vec<Sphere>, raytracer.rs:52-57); above 65 535 spheres the uniform
 *     grid is built with 32-bit item lists ("wide" tables, rt_hip_scene_query "grid_wide") and stays in L2. */
int rt_hip_render(RtHipScene*, const RtRowTiles* tiles, void* d_rgb8, void* d_linear, void* stream);
/* Block until the last rt_hip_render on this scene finished; fill counters and the HIP-event
 * duration of its kernel (events are recorded on the stream the kernel was launched on). */
int rt_hip_wait(RtHipScene*, RtStats* stats);
/* Options of a resident scene: (key, value) pairs, out-of-range values and unknown keys are RT_ERR_INVALID.  The keys, their
 * ranges and defaults are tabulated in INTEGRATION.md §5 ("samples_per_pixel", "max_depth", "seed" override the scene's
 * values; "variant" 1 = the reference's brute force; the rest shape the work distribution and never the image).
 * rt_hip_group_set_option forwards to every rank's scene and takes "spin_us" itself. */
int rt_hip_set_option(RtHipScene*, const char* key, int64_t value);
/* What a resident scene was built into (diagnostics): "n_spheres", "n_lights", "grid_cells" (padded cell table, 8 B each),
 * "grid_items" (u16 each), "grid_wide" (1: more than 65 535 spheres — 16-byte cells, u32 items), "grid_large" (spheres every ray tests), "table_bytes" (geometry + material cores + cell table +
 * item lists: what a workgroup stages into LDS once per launch), "texel_bytes" (textures + sky as 4-byte texels in HBM);
 * of the last launch: "lds_bytes" (dynamic LDS of a workgroup), "lds_tables" (1: the tables were staged in LDS),
 * "light_pool_slots" / "light_base_slots" (lit scenes: records in the workgroup's pools of light frames / colour-map bases).
 * -1 for an unknown key. */
int64_t rt_hip_scene_query(const RtHipScene*, const char* key);
/* Animation (the reference's `anim/frame_%03d.png` workflow, README.md:43-57, main.rs:17): move the
 * camera of a resident scene — the four vectors of camera.rs:52-63 — without touching its tables,
 * and render whole frames of it into a host buffer (internal device framebuffer, blocking). */
int rt_hip_set_camera(RtHipScene*, const double origin[3], const double lower_left[3], const double horizontal[3],
                      const double vertical[3]);
int rt_hip_render_to_host(RtHipScene*, uint8_t* out_rgb8, RtStats* stats);
/* A frame over the GPUs of one node, scene resident (the parallel loop of raytracer.rs:254-262 spread over devices;
 * animation: README.md:43-57).  n_gpus = 0 takes scene->n_gpus, then RT_GPUS, then 1.  Each rank renders
 * interleaved 2-scanline tiles (RtRowTiles{2, r, G}) on its own host thread and stream; ONE gather per frame
 * (RCCL `ncclGather` over xGMI, or peer copies with RT_GATHER=peer) brings the packed tiles to device 0,
 * a kernel puts the scanlines in order, ONE device-to-host copy delivers them.  Bit-identical for every G.
 * stats: counters summed over ranks, kernel_ms = slowest rank, frame_ms = the whole call, gather_ms. */
typedef struct RtHipGroup RtHipGroup;
/* What a group actually runs on (rt_hip_group_info): bench.py echoes it next to its numbers. */
#define RT_GROUP_INFO_MAX_RANKS 64u
enum { RT_GATHER_NONE = 0, RT_GATHER_RCCL = 1, RT_GATHER_PEER = 2 };
typedef struct RtGroupInfo {
  uint32_t n_ranks;    /* G */
  uint32_t n_devices;  /* distinct HIP device ordinals among the ranks (== n_ranks unless RT_GPUS_EMULATE=1) */
  uint32_t transport;  /* RT_GATHER_*: how the packed tiles reach the first device (NONE: one rank, nothing to gather) */
  uint32_t rccl_comms; /* RCCL communicators created by ncclCommInitAll (n_ranks with RT_GATHER_RCCL, else 0) */
  uint32_t tile_rows;  /* scanlines per interleaved tile (2) */
  uint32_t pad_rows;   /* rows of one rank's slice of the gather buffer */
  uint32_t emulated;   /* 1: ranks share devices (RT_GPUS_EMULATE=1, tests) */
  uint32_t transport_fallback; /* 1: RCCL was wanted but could not be used (library, communicators or its self-test gather failed):
                                * the group runs on peer copies instead; rt_hip_group_fallback_reason() says why */
  int32_t device[RT_GROUP_INFO_MAX_RANKS]; /* device ordinal of rank r; -1 beyond n_ranks */
} RtGroupInfo;
/* One rank of a group (rt_hip_group_ranks): where it runs and what its last frame cost — enough for one bench line of an
 * N-GPU run to explain its own efficiency. */
typedef struct RtGroupRank {
  int32_t device;        /* HIP device ordinal */
  int32_t numa_node;     /* NUMA node of the device's PCI function (sysfs numa_node), -1 unknown */
  int32_t pinned_cpus;   /* CPUs in the affinity mask the rank's host thread was pinned to (that node's), 0: not pinned */
  int32_t peer_to_root;  /* hipDeviceCanAccessPeer(this device -> the first rank's device); 1 for rank 0 and shared devices */
  char pci_bus_id[16];   /* "0000:c1:00.0" */
  double kernel_ms;      /* the frame collected last: this rank's kernel (HIP events on its stream) */
  double t_wake_us;      /* the frame submitted last, host clock since its submit was entered: the rank's host thread is running, */
  double t_enq_us;       /* ... its launch (+ its side of the transfer) is enqueued */
} RtGroupRank;
/* Creation never fails because of the TRANSPORT: if RCCL is wanted (the default with one device per rank) but its library
 * cannot be loaded (RT_RCCL_LIB overrides the path), ncclCommInitAll fails, or the self-test gather run at creation fails,
 * times out (RT_RCCL_TIMEOUT_MS, default 20 000) or delivers wrong bytes, the communicators are torn down and the group
 * runs on peer copies (RtGroupInfo.transport = RT_GATHER_PEER, .transport_fallback = 1); a gather that fails to enqueue in a
 * later frame switches the same way and re-sends that frame's tiles.  Rank threads are pinned to the CPUs of their device's
 * NUMA node unless RT_GROUP_PIN=0. */
int rt_hip_group_create(const RtScene* scene, uint32_t n_gpus, RtHipGroup** out);
int rt_hip_group_info(const RtHipGroup*, RtGroupInfo* info);
uint32_t rt_hip_group_ranks(const RtHipGroup*, RtGroupRank* out, uint32_t cap); /* fills min(cap, n_ranks) entries, returns n_ranks */
const char* rt_hip_group_fallback_reason(const RtHipGroup*);                    /* "" unless RtGroupInfo.transport_fallback */
/* the same frame, left in HBM: scanline order, RGB8, on the group's first device (rt_hip_group_frame returns the device
 * pointer, valid until the group is destroyed, and that device's ordinal) — no device-to-host copy.  Blocking. */
int rt_hip_group_render(RtHipGroup*, RtStats* stats);
const void* rt_hip_group_frame(const RtHipGroup*, int* device_out);
/* The same frame in two halves, for callers that render frame after frame (an animation): submit enqueues everything
 * frame i needs — G launches, the gather, the de-interleave and, if out_rgb8 is not NULL, the copy into that host buffer —
 * and returns; collect blocks until the OLDEST submitted frame is complete and fills its stats.  Two frames may be in flight
 * (a third submit is RT_ERR_INVALID): every buffer exists twice, a rank's tiles travel on a transfer stream of their own,
 * so frame i's gather + copy run while frame i+1 renders and a step costs the slowest rank's kernel.  The camera and
 * options a frame is rendered with are those in force when it is SUBMITTED.  out_rgb8 may be pageable memory: the frame
 * leaves the device into a pinned staging buffer of the group (so submit never blocks on the copy) and collect moves
 * it into out_rgb8, which must stay valid until then; a buffer that is itself pinned (hipHostMalloc / hipHostRegister)
 * is written directly.  rt_hip_group_frame() points at the frame collected last.  rt_hip_group_render / _render_to_host are
 * submit + collect (after collecting whatever was still in flight). */
int rt_hip_group_submit(RtHipGroup*, uint8_t* out_rgb8);
int rt_hip_group_collect(RtHipGroup*, RtStats* stats);
void rt_hip_group_destroy(RtHipGroup*);
uint32_t rt_hip_group_size(const RtHipGroup*);
int rt_hip_group_set_camera(RtHipGroup*, const double origin[3], const double lower_left[3], const double horizontal[3],
                            const double vertical[3]);
int rt_hip_group_set_option(RtHipGroup*, const char* key, int64_t value);
int rt_hip_group_render_to_host(RtHipGroup*, uint8_t* out_rgb8, RtStats* stats);
/* the group's layout arithmetic as the library compiled it (no GPU needed): rt_tiles_stacked_row() with the group's
 * tile height; *tile_rows_out receives that height (2) */
uint32_t rt_hip_group_stacked_row(uint32_t y, uint32_t n_ranks, uint32_t pad_rows, uint32_t* tile_rows_out);
/* Convenience = the drop-in for render()'s parallel loop (raytracer.rs:254-262): host buffers in,
 * host RGB8 out.  Blocking.  Renders on scene->n_gpus devices (see RtScene.n_gpus / RT_GPUS): the scene
 * is replicated, device r renders scanline tiles r, r+G, ... (2 rows each) on its own host thread and
 * stream, the packed tiles meet on device 0 through ONE gather (RCCL send/recv over xGMI, or peer copies
 * with RT_GATHER=peer), are de-interleaved by a small kernel and leave in ONE device-to-host copy. */
int rt_render_rgb8(const RtScene* scene, uint8_t* out_rgb8, RtStats* stats);
const char* rt_strerror(int code);

#ifdef __cplusplus
}
#endif
#endif /* RT_ABI_H */
