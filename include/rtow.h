/*
 * rtow.h — C-ABI of the MI355X path-tracing hot path.
 *
 * This is the drop-in boundary for ONE path of joaotavora/raytracing-one-weekend:
 * the per-pixel sample loop of `rtweekend::render(const Scene&, const Config&)`
 * (reference src/render.h:35, src/render.cpp:135-191).  The reference has no FFI;
 * a maintainer would replace the body of render() below the Scene/Config boundary
 * with: flatten the Scene into `rtow_scene_t`, call `rtow_render()`, and hand the
 * returned per-pixel sums to the unchanged write_color()/PPM loop
 * (src/render.cpp:11-20,182-186).  INTEGRATION.md shows that binding.
 *
 * Conventions
 *   - plain C, plain pointers and sizes, no C++ / torch types;
 *   - every function returns 0 on success or a negative RTOW_E* code and never
 *     throws or aborts across the boundary; `rtow_last_error()` gives the text
 *     (thread-local);
 *   - the caller owns every host pointer for the duration of the call only;
 *   - the context owns all device memory (scene, workspace); one call in flight
 *     per context;
 *   - all geometry and radiance are IEEE binary64, like the reference
 *     (src/vec3.h:6-8: vec3 = glm::dvec3).
 */
#ifndef RTOW_H
#define RTOW_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTOW_ABI_VERSION 7

/* error codes */
#define RTOW_OK 0
#define RTOW_EINVAL (-1)   /* bad argument / inconsistent scene            */
#define RTOW_ENODEV (-2)   /* no usable HIP device                         */
#define RTOW_EHIP (-3)     /* a HIP runtime call failed                    */
#define RTOW_ENOSCENE (-4) /* render called before a scene was uploaded    */
#define RTOW_EEMPTY (-5)   /* scene has no primitives (reference: UB,
                              src/render.cpp:81)                           */
#define RTOW_ENOMEM (-6)   /* host allocation failed                       */

/* material kinds — Lambertian / Metal / Dielectric (src/common-model.h:124-151) */
#define RTOW_MAT_LAMBERTIAN 0
#define RTOW_MAT_METAL 1
#define RTOW_MAT_DIELECTRIC 2

/* primitive kinds — Sphere / MovingSphere / Triangle (src/oo-primitives.h:26-88) */
#define RTOW_PRIM_SPHERE 0
#define RTOW_PRIM_MOVING_SPHERE 1
#define RTOW_PRIM_TRIANGLE 2

/* arithmetic modes of the device path */
#define RTOW_F64_STRICT 0 /* binary64, no FMA contraction: bit-identical to the CPU oracle */
#define RTOW_F64_FAST 1   /* binary64, FMA contraction allowed (default for speed)         */
#define RTOW_F32 2        /* binary32 rays, small-primitive tests and shading on binary32
                             records; large primitives and pixel sums stay binary64.
                             NOT the reference's arithmetic: a faster preview mode whose
                             parity with the binary64 builds is by tolerance (<= 1/255 mean
                             absolute difference per channel at >= 100 spp)             */

/* closest-hit strategies */
#define RTOW_KERNEL_AUTO 0
#define RTOW_KERNEL_BRUTE 1 /* every ray tests every primitive (wave-uniform stream)   */
#define RTOW_KERNEL_BVH 2   /* every lane walks a threaded (stackless) BVH in LDS       */
#define RTOW_KERNEL_GRID 3  /* every lane walks a uniform grid (3D-DDA) in LDS; falls back
                               to BVH when the scene does not suit a grid              */
#define RTOW_KERNEL_BVH4 4  /* triangle meshes: every lane walks a 4-wide BVH, nearest child
                               first, with a per-lane stack in LDS; the image (or the top of
                               its tree) is staged in LDS.  Falls back to BVH for scenes with
                               spheres, for the f32 preview build and for meshes beyond the
                               format's limits (2^18 triangles).  Both builders make its image:
                               the host collapses its SAH tree, the device builder its LBVH   */
#define RTOW_KERNEL_REFTREE 5 /* opt-in exactness mode (RTOW_F64_STRICT only): every lane walks the REFERENCE's own
                               tree — median split of the insertion-ordered primitive array, leaves of 1..6 whose
                               boxes include the origin, float-rounded triangle boxes, signed-radius sphere boxes
                               (src/render.cpp:73-110, src/common-model.cpp:127-134,168-207) — left before right with
                               the reference's f64 Aabb::hit (src/common-model.h:71-84, `t_max <= t_min` rejects).
                               The other kernels give the reference's image wherever its tree finds the closest hit;
                               this one also where it does not (flat leaf boxes, coordinates beyond float precision,
                               negative radii, exact ties).  Never chosen by RTOW_KERNEL_AUTO: it is slow */

/* Camera state: exactly the private members of the reference Camera after its
 * constructor ran (src/common-model.h:104-112, src/common-model.cpp:136-154). */
typedef struct rtow_camera_t {
  double origin[3];
  double u[3], v[3], w[3];
  double horizontal[3];
  double vertical[3];
  double lower_left_corner[3];
  double lens_radius;
  double t0, t1;
} rtow_camera_t;

/* One material record (src/common-model.h:124-151). */
typedef struct rtow_material_t {
  double albedo[3]; /* Lambertian, Metal                       */
  double fuzz;      /* Metal, Dielectric (already clamped 0..1) */
  double ir;        /* Dielectric index of refraction           */
  int32_t kind;     /* RTOW_MAT_*                               */
  int32_t pad_;
} rtow_material_t;

/* Flattened scene: one packed record array per primitive class (structure of
 * arrays across classes; each record is a small aligned block of doubles so a
 * wave-uniform read is one scalar load and a cooperative tile load is fully
 * coalesced).  `prim_kind/prim_index` keep the insertion order of the
 * reference's single primitive array (src/render.h:23), which the reference
 * BVH build depends on (src/render.cpp:73-110). */
typedef struct rtow_scene_t {
  rtow_camera_t camera;

  int32_t n_spheres;
  const double *sphere_geom;   /* [n_spheres][4]  cx cy cz r            */
  const int32_t *sphere_mat;   /* [n_spheres]     material index        */

  int32_t n_moving;
  const double *moving_geom;   /* [n_moving][8]   c0xyz c1xyz r pad     */
  const int32_t *moving_mat;   /* [n_moving]                            */

  int32_t n_triangles;
  const double *triangle_geom; /* [n_triangles][9] ax ay az bx .. cz    */
  const int32_t *triangle_mat; /* [n_triangles]                         */

  int32_t n_materials;
  const rtow_material_t *materials;

  int32_t n_prims;             /* = n_spheres + n_moving + n_triangles  */
  const int32_t *prim_kind;    /* [n_prims] RTOW_PRIM_* in insertion order */
  const int32_t *prim_index;   /* [n_prims] index into that class's arrays */
} rtow_scene_t;

/* Render parameters.  image_height is derived by the caller exactly as the
 * reference does: int(image_width / aspect_ratio) (src/render.cpp:137).
 *
 * `nstreams` has the meaning of the reference's `nthreads`
 * (src/render.cpp:169-185): the samples of a pixel are split into nstreams
 * equal runs of samples_per_pixel / nstreams samples, each run is summed in
 * sample order into its own partial image, and the partial images are added in
 * run order.  Effective spp = samples_per_pixel / nstreams * nstreams.
 *
 * Random numbers are counter-based: request r of sample s of pixel p (global
 * row-major index, top row first) is ONE Philox4x32-7 block keyed by `seed` — request 0 the
 * sample's pixel jitter, shutter time and lens point, request 1 + b everything bounce b
 * draws (the lens point and the scatter's point of the unit ball are sampled directly, with
 * the distributions of the reference's rejection loops, src/random-utils.cpp:23-41); the image
 * therefore does not depend on nranks, tile_rows or on which lane traced what. */
typedef struct rtow_config_t {
  int32_t image_width;
  int32_t image_height;
  int32_t samples_per_pixel;
  int32_t nstreams;
  int32_t max_child_rays;
  int32_t precision;  /* RTOW_F64_*      */
  int32_t kernel;     /* RTOW_KERNEL_*   */
  int32_t rank;       /* this process's share of the image: horizontal strips of */
  int32_t nranks;     /* tile_rows rows dealt round-robin; strip t belongs to    */
  int32_t tile_rows;  /* rank t % nranks.  nranks=1 → whole image.               */
  uint64_t seed;
  /* Stream range (progressive accumulation, sample-split decompositions).
   * stream_count == 0: all nstreams streams.  Otherwise only streams [stream_first,
   * stream_first + stream_count) are traced; sample indices and random numbers are those of the
   * full render.  accumulate != 0: the new partial images are added onto the sums already in the
   * output buffer, in stream order — rendering [0,a) and then [a,n) with accumulate set gives
   * bit for bit the sums of one call over [0,n). */
  int32_t stream_first;
  int32_t stream_count;
  int32_t accumulate;
  int32_t pad_;
} rtow_config_t;

typedef struct rtow_stats_t {
  uint64_t samples;      /* (pixel, sample) pairs traced by this call            */
  uint64_t segments;     /* ray segments traced (calls of ray_color in the ref.) */
  uint64_t prim_tests;   /* primitive hit tests                                  */
  uint64_t node_tests;   /* BVH box tests (0 for the brute-force kernel)         */
  double kernel_ms;      /* device time of the trace kernel (HIP events)         */
  double total_ms;       /* device time of the whole call (all kernels)          */
  int32_t local_rows;    /* rows of the image owned by this rank                 */
  int32_t kernel_used;   /* RTOW_KERNEL_* actually run                           */
} rtow_stats_t;

typedef struct rtow_ctx rtow_ctx;

int rtow_abi_version(void);
const char *rtow_last_error(void);

/* Bind a context to one HIP device (one process per GPU; no global state). */
int rtow_ctx_create(int device_id, rtow_ctx **out);
void rtow_ctx_destroy(rtow_ctx *ctx);

/* Copy the scene into HBM (and build the device BVH).  The scene stays
 * resident until the next upload or ctx destroy.
 * Ordering: the copies and build kernels are QUEUED on the null stream and the call returns without waiting for
 * them.  A render on the null stream or on a blocking stream is ordered behind them by the runtime; a render on a
 * hipStreamNonBlocking stream (every torch.cuda.Stream() side stream) is ordered behind them by the library (it
 * makes the stream wait for an event recorded behind the upload).  build_info.upload_ms is therefore the host time
 * of the call, not the completion time of the copies. */
int rtow_scene_upload(rtow_ctx *ctx, const rtow_scene_t *scene);

/* Who builds the BVH image at rtow_scene_upload (replaces the reference's BVHNode constructor,
 * src/render.cpp:73-110, which runs on the host inside render()):
 *   HOST_SAH    binned surface-area-heuristic build on the host (best tree)
 *   DEVICE_LBVH on the GPU (csrc/rtow_build.hip): bounds, Morton keys and a radix sort, then (round 5) a binned
 *               surface-area-heuristic build, top-down and level by level — the host builder's algorithm (16 bins
 *               per axis, the split that minimises area x count over the three axes, a stable partition by one
 *               scan per level) with every node of a level handled at once; refit; for a triangle mesh the
 *               4-wide image of the BVH4 kernel (greedy collapse level by level, breadth-first nodes, planes
 *               rounded outwards, records in leaf order); the uniform grid of the GRID kernel is built on the
 *               GPU too (csrc/rtow_build_grid.hip, byte-identical to the host-built image).  Nothing of the
 *               build runs on the host.  The tree is as good as the host's (suzanne: 9.0 node tests per segment
 *               with either; 96,800 triangles: 20.5) and is built in 5 ms for 96,800 triangles against the
 *               host's 12-16 on 16 threads.  RTOW_DEVICE_TREE=ploc|radix selects the earlier device trees
 *               (parallel locally-ordered clustering, 7 % slower to walk; Karras' radix tree, 12 %).
 *   AUTO        (default of a new context) per call of rtow_render / rtow_render_rgb8, which know their config:
 *               the device builder for a scene of triangles only with 16,384 of them or more, the host builder
 *               otherwise (a small mesh: the device's two dozen launches cost more than the host's 0.4 ms;
 *               sphere scenes: the grid).  rtow_scene_upload, which knows no config, takes the host builder
 *               under AUTO.
 * Images are bit-identical with either builder (the closest hit is tree-independent).
 * Takes effect at the next upload; the environment variable RTOW_BUILDER=host|device|auto sets the
 * default of new contexts.  rtow_build_info_t::builder says which one built the resident image. */
#define RTOW_BUILDER_HOST_SAH 0
#define RTOW_BUILDER_DEVICE_LBVH 1
#define RTOW_BUILDER_AUTO 2
int rtow_ctx_set_builder(rtow_ctx *ctx, int32_t builder);

typedef struct rtow_build_info_t {
  int32_t builder;          /* builder that produced the resident BVH image */
  int32_t bvh_nodes;        /* node records (without the END record) */
  int32_t bvh_image_bytes;
  int32_t grid_image_bytes; /* 0 = scene not suited to the grid */
  double bvh_build_ms;      /* HOST_SAH: host wall time; DEVICE_LBVH: wall time of the launch sequence
                               including its two small read-backs */
  double grid_build_ms;     /* host wall time */
  double upload_ms;         /* whole rtow_scene_upload call */
  int32_t bvh4_nodes;       /* 4-wide BVH image (triangle meshes, either builder): nodes, 0 = none */
  int32_t bvh4_image_bytes;
  /* RTOW_KERNEL_REFTREE (built at the first render that asks for it; 0 before): nodes of the reference's
   * tree and its "Total BVH stupid volume" diagnostic (src/render.cpp:36-50,148) */
  int32_t ref_tree_nodes;
  int32_t bvh4_node_bytes;  /* 128: binary32 planes (image staged in LDS whole); 64: binary16 planes in the mesh's own
                               frame (bigger meshes, nodes read from L2); 0 = no 4-wide image */
  double ref_tree_stupid_volume;
  double ref_tree_build_ms;
} rtow_build_info_t;
/* Facts about the last rtow_scene_upload of this context. */
int rtow_build_info(rtow_ctx *ctx, rtow_build_info_t *out);

/* Number of image rows owned by cfg->rank, and their global row numbers
 * (ascending) — pure host arithmetic, usable without a GPU. */
int rtow_local_rows(const rtow_config_t *cfg);
int rtow_local_row_list(const rtow_config_t *cfg, int32_t *rows_out, int32_t capacity);

/* Trace this rank's rows.  `d_rgb_sums` is a DEVICE pointer to
 * local_rows*image_width*3 doubles (row-major, this rank's rows in ascending
 * global order); it receives the per-pixel radiance SUMS over the effective spp,
 * i.e. the reference's `global_image` (src/render.cpp:144,176-180) before
 * write_color divides by spp.  `hip_stream` is a hipStream_t (NULL = default
 * stream); the call enqueues work on it and returns without synchronising,
 * unless `stats` is non-NULL, in which case it synchronises the stream and
 * fills `stats`.
 * Samples are never dropped silently: the trace kernel's end-of-launch protocol has a structural trip bound (never
 * observed to fire); lanes that reach it count themselves in a device word, and every entry point that waits for the
 * device — this one with `stats`, rtow_render, rtow_render_rgb8, rtow_multi_render* — returns RTOW_EHIP instead of
 * RTOW_OK when the word is non-zero.  The asynchronous form (stats == NULL) reports it at rtow_profile_collect. */
int rtow_render_device(rtow_ctx *ctx, const rtow_config_t *cfg, void *d_rgb_sums,
                       void *hip_stream, rtow_stats_t *stats);

/* The same with write_color run on the device (src/render.cpp:11-20): `d_rgb8` is a DEVICE pointer to
 * local_rows*image_width*3 BYTES, the values the reference prints for this rank's rows.  The f64 sums stay in the
 * context's workspace (write_color is fused into the reduce kernel when the render is one launch).  cfg->accumulate
 * must be 0.  Asynchronous on `hip_stream` like rtow_render_device.  (What every rank of rtow_multi_render_rgb8 runs.) */
int rtow_render_device_rgb8(rtow_ctx *ctx, const rtow_config_t *cfg, void *d_rgb8, void *hip_stream,
                            rtow_stats_t *stats);

/* write_color on the device (reference src/render.cpp:11-20): for each of the n_values
 * doubles of `d_rgb_sums`, byte = int(256 * clamp(sqrt(sum / spp_effective), 0, 0.999)) into
 * `d_rgb8` (device pointer, n_values bytes).  Enqueued on `hip_stream`, no sync.  The bytes
 * equal the numbers the reference prints in its P3 file. */
int rtow_tonemap_device(rtow_ctx *ctx, const void *d_rgb_sums, int64_t n_values, int32_t spp_effective,
                        void *d_rgb8, void *hip_stream);

/* Every rtow_render_device call brackets its trace-kernel launch with a HIP event
 * pair on the launch stream (no host sync).  This collects the device time of
 * all launches since the previous collect (it waits for them) and resets the
 * ring: `*kernel_ms_sum` = total trace-kernel milliseconds, `*launches` = how
 * many launches that covers (the ring keeps the first 256 per collect). */
int rtow_profile_collect(rtow_ctx *ctx, double *kernel_ms_sum, int32_t *launches);

/* Diagnostic only: copies the 48 device counters of the last launch (see
 * csrc/rtow_trace_body.h; [8..12] are wave-cycle sums per region and [17..22] a histogram of
 * wave end times, [29..33] the finer regions, when the RTOW_STAMPS diagnostic kernel ran). */
int rtow_debug_counters(rtow_ctx *ctx, unsigned long long *out48);

/* Diagnostic only: the LEVELS a render of `cfg` is cut into on this context — pairs (first sample index,
 * sample count), one work item per pixel and level.  RTOW_F64_STRICT: one level per stream (spp / nstreams
 * samples, the reference's threads, src/render.cpp:151-166), summed in stream order like the reference.  Fast
 * builds: the same samples in levels of ONE length that does not depend on nstreams — the divisor of the sample
 * range nearest RTOW_SCHED_CHUNK (10; RTOW_SCHED_CHUNK_MESH = 16 when the resident scene is a triangle mesh) — so
 * that Config::nthreads keeps its arithmetic meaning without setting the size of a work item (csrc/rtow_capi.cpp,
 * level_plan).  A sample range with no divisor within a factor of two of that length (101, 127: primes) is cut into
 * levels of exactly that length with the remainder added to the LAST level (101 = 9 x 10 + 11).  Returns the number of levels (writes at most
 * `capacity_pairs` of them).  `ctx` may be NULL: the table of a new context (pure host arithmetic, usable
 * without a GPU). */
int rtow_debug_schedule(rtow_ctx *ctx, const rtow_config_t *cfg, uint32_t *out_pairs, int32_t capacity_pairs);

/* Diagnostic only: copies a resident scene image to the host (which: 0 BVH image, 1 grid image,
 * 2 / 3 the same of the RTOW_F32 build).  `out` NULL: size query.  The tests compare host-built
 * and device-built images byte for byte with it. */
int rtow_debug_image(rtow_ctx *ctx, int32_t which, void *out, int64_t capacity, int64_t *size_out);

/* Convenience: upload + render + copy this rank's rows to host memory.
 * Lean upload: rtow_render / rtow_render_rgb8 know their config and build only the structures ITS kernel reads
 * (the cover scene through the grid kernel needs no BVH, no 4-wide image, no binary32 images).  The scene they
 * leave resident is therefore partial: a later rtow_render_device with another kernel or precision is refused with
 * RTOW_ENOSCENE until rtow_scene_upload (which builds everything) has run. */
int rtow_render(rtow_ctx *ctx, const rtow_scene_t *scene, const rtow_config_t *cfg,
                double *rgb_sums_host, rtow_stats_t *stats);

/* Convenience: upload + render + write_color on the device + copy this rank's rows as 8-bit
 * RGB (rows*W*3 bytes) to host memory — the payload of a binary P6 PPM. */
int rtow_render_rgb8(rtow_ctx *ctx, const rtow_scene_t *scene, const rtow_config_t *cfg,
                     unsigned char *rgb8_host, rtow_stats_t *stats);

/* One frame over several HIP devices from ONE process (replaces the reference's thread fan-out and
 * in-order sum, src/render.cpp:169-180): one host thread + context per entry of `device_ids`, strips of
 * cfg->tile_rows rows dealt round-robin (cfg->rank / nranks are ignored: rank = position in the list),
 * then with use_rccl != 0 ONE ncclGather (RCCL over xGMI; librccl.so is loaded on demand) of the strip
 * buffers to the first device and ONE device-to-host copy, with use_rccl == 0 one copy per device.
 * `rgb_sums_host`: image_height*image_width*3 doubles, row-major from the top, the radiance sums of
 * the whole frame — identical, bit for bit, to a one-device rtow_render of the same config.  RCCL rejects
 * a device listed twice; use_rccl == 0 accepts it (tests of the partition on a one-GPU machine). */
int rtow_render_multi(int32_t n_devices, const int32_t *device_ids, const rtow_scene_t *scene,
                      const rtow_config_t *cfg, double *rgb_sums_host, rtow_stats_t *stats, int32_t use_rccl);

/* The same as a persistent handle, for more than one frame: rtow_render_multi pays for its contexts,
 * streams, buffers, worker threads and — by far the largest item — the RCCL communicator on every call
 * (it is create + upload + render + destroy).  The handle owns all of them; rtow_multi_upload builds
 * the scene's acceleration structures once per device (concurrently); rtow_multi_render is then one
 * trace launch per device, the one ncclGather enqueued behind them on the same streams, a small kernel on the
 * first device that puts the strips' rows in place, and ONE device-to-host copy straight into the caller's buffer
 * (no host pass over the pixels).  One call in flight per handle.
 * rtow_multi_render_rgb8: the same with write_color run by the rank that owns the pixel (the fused reduce of
 * rtow_render_device_rgb8), so the gather and the copy move 3 bytes per pixel instead of 24; `rgb8_host` receives
 * image_height*image_width*3 bytes, equal to a one-device rtow_render_rgb8 of the same config.
 * If a rank cannot enqueue its side of the gather, every communicator is aborted (ncclCommAbort) before anything
 * is waited for, the call returns RTOW_EHIP and the handle refuses further frames: an error is a code, never a hang.
 * rtow_multi_build_info reports the first device's build (every device builds the same structures).
 * A frame is ONE hand-off to the worker threads (trace launch, a meeting of the workers, each rank's side of the
 * gather), the placement kernel only when there is more than one rank, one asynchronous copy into the caller's
 * buffer and ONE wait, on the first device's stream (it is ordered behind every other rank's queue by events).
 * rtow_multi_frame_breakdown: where the last successful frame's time went, RTOW_MULTI_BREAKDOWN_FIELDS doubles in
 * milliseconds indexed by RTOW_MB_* (host clock around the call's stages; HIP events on the first device's stream
 * for the device side); returns the number of fields written. */
typedef struct rtow_multi rtow_multi;
enum {
  RTOW_MB_TOTAL = 0,           /* entry to return of rtow_multi_render* (host clock) */
  RTOW_MB_HANDOFF_ENQUEUE = 1, /* buffers, the hand-off to the workers, until every rank has queued its frame */
  RTOW_MB_PLACE_ENQUEUE = 2,   /* error collection, stream-wait events, the placement kernel's launch */
  RTOW_MB_WAIT_AND_COPY = 3,   /* queueing the device-to-host copy + the one wait (device work still running included) */
  RTOW_MB_WAIT_ONLY = 4,       /* RTOW_MULTI_D2H=blocking only: the wait before the blocking copy */
  RTOW_MB_DEV_TRACE = 5,       /* first device: trace + reduce (+ write_color), events */
  RTOW_MB_DEV_GATHER = 6,      /* first device: the gather (or the strips' copy), events */
  RTOW_MB_DEV_PLACE_COPY = 7,  /* first device: waits for the other ranks, placement kernel, device-to-host copy, events */
  RTOW_MULTI_BREAKDOWN_FIELDS = 8
};
int rtow_multi_create(int32_t n_devices, const int32_t *device_ids, int32_t use_rccl, rtow_multi **out);
int rtow_multi_set_builder(rtow_multi *m, int32_t builder);
int rtow_multi_upload(rtow_multi *m, const rtow_scene_t *scene);
int rtow_multi_build_info(rtow_multi *m, rtow_build_info_t *out);
int rtow_multi_render(rtow_multi *m, const rtow_config_t *cfg, double *rgb_sums_host, rtow_stats_t *stats);
int rtow_multi_render_rgb8(rtow_multi *m, const rtow_config_t *cfg, unsigned char *rgb8_host, rtow_stats_t *stats);
int rtow_multi_frame_breakdown(rtow_multi *m, double *out_ms, int32_t n_fields);
void rtow_multi_destroy(rtow_multi *m);

/* ---- host-side scene construction (no GPU needed) --------------------------
 * C entry points over the C++ mirror of the reference's scene-build API
 * (host/scene.h ≙ src/common-model.h, src/oo-primitives.h, src/render.h):
 * the two scene scripts of the reference's main.cpp, flattened. */
typedef struct rtow_host_config_t {
  int32_t number_of_balls_sqrt; /* src/render.h:12  */
  double aspect_ratio;          /* src/render.h:13  */
  int32_t moving_spheres;       /* src/render.h:16  */
} rtow_host_config_t;

/* lots_of_balls() (src/main.cpp:23-83): consumes the process-global mt19937
 * stream from its default seed, like the reference. */
int rtow_host_scene_cover(const rtow_host_config_t *cfg, rtow_scene_t **out);
/* foo() (src/main.cpp:85-136): triangles of the first shape of an OBJ file. */
int rtow_host_scene_obj(const rtow_host_config_t *cfg, const char *obj_path, rtow_scene_t **out);
/* The same scripts on each of the reference's scene models (it picks one at compile time): OO primitives
 * (src/oo-primitives.h — what the two entry points above use), variant primitives
 * (src/variant-primitives.h:84-113, RTWEEKEND_USE_VARIANT_PRIMITIVES) and the World of src/vmodel.h:250-253
 * (spheres only: the OBJ script has no World form).  Built by the same calls, the models flatten to the
 * same rtow_scene_t. */
#define RTOW_MODEL_OO 0
#define RTOW_MODEL_VARIANT 1
#define RTOW_MODEL_WORLD 2
int rtow_host_scene_cover_model(const rtow_host_config_t *cfg, int32_t model, rtow_scene_t **out);
int rtow_host_scene_obj_model(const rtow_host_config_t *cfg, const char *obj_path, int32_t model, rtow_scene_t **out);
void rtow_host_scene_free(rtow_scene_t *scene);
/* Reset the host scene-construction RNG to the reference's default seed. */
void rtow_host_rng_reset(void);

/* write_color + PPM (src/render.cpp:11-20,182-186): P3 text into a malloc'ed
 * buffer (`*out_text`, free with rtow_host_free). */
int rtow_host_ppm(const double *rgb_sums, int32_t width, int32_t height, int32_t spp_effective,
                  char **out_text, uint64_t *out_len);
void rtow_host_free(void *p);

/* The reference's own bounding-volume tree (src/render.cpp:73-110) over a flattened scene, as RTOW_KERNEL_REFTREE
 * walks it, built on the host (no GPU): node count, depth and the reference's "Total BVH stupid volume"
 * diagnostic (src/render.cpp:36-50,148). */
int rtow_host_reftree_info(const rtow_scene_t *scene, int32_t *n_nodes, int32_t *depth, double *stupid_volume);

#ifdef __cplusplus
}
#endif
#endif /* RTOW_H */
