// rtow_device.h — device-side data layout shared by the C-ABI shim and the kernels.
//
// Everything the trace kernel reads is laid out once at scene upload:
//   * one packed record array per primitive class ("SoA across classes"); a record
//     is a 32/64/96-byte aligned block of doubles, so the wave-uniform read of
//     primitive i in the streaming kernel is a single scalar load (s_load_dwordx8/16)
//     that lands in SGPRs and feeds the VALU as an operand — no VGPRs, no LDS
//     traffic — and a per-lane read in the BVH kernel is 16-byte vector loads;
//   * ray-independent per-primitive terms the reference recomputes in every hit
//     test (r*r, c1-c0, e1, e2, e1×e2; src/common-model.cpp:73,106-108,
//     src/oo-primitives.h:65) are computed once on the host with the same IEEE
//     operations, so the results are bit-identical.
#pragma once
#include <stdint.h>

namespace rtow {

struct DevMaterial {       // 48 B
  double att[3];           // attenuation: albedo (Lambertian, Metal) or (1,1,1) (Dielectric)
  double fuzz;
  double ir;
  int32_t kind;
  int32_t pad_;
};

struct DevCamera {         // src/common-model.h:104-112
  double origin[3], u[3], v[3];
  double horizontal[3], vertical[3], llc[3];
  double lens_radius, t0, t1;
};

// class-major global primitive id: [0,n_sph) spheres, then moving, then triangles
struct DevScene {
  const double *sph;       // [n_sph][4]  cx cy cz copysign(r*r, r)
  const double *sph_r;     // [n_sph]     r (sign decides front_facing)
  const double *mov;       // [n_mov][8]  c0xyz (c1-c0)xyz copysign(r*r, r) r
  const double *tri;       // [n_tri][12] a e1 e2 n=e1×e2
  const double *tri16;     // the same records at a stride of 16 doubles (128 B: a record is one 64-byte and one 32-byte
                           // aligned scalar load for the STREAM kernel; at 96 B it arrived in five or six pieces); NULL
                           // unless rtow_scene_upload built everything or the render asked for the STREAM kernel
  uint32_t stream_tile_lds; // STREAM kernel: bytes of LDS every wave has for the tiled triangle loop (0: the scalar-load loop)
  const int32_t *prim_mat; // [n_prims]   material index by class-major id
  const DevMaterial *mats;
  int32_t n_sph, n_mov, n_tri, n_mats;
  // scene image for the BVH kernel (kernel RTOW_KERNEL_BVH): one 16-byte aligned blob
  //   [nodes n_nodes x 32 B][prim ids n_prims x 4 B][sph n x 32 B][mov n x 64 B][tri n x 96 B]
  // node = { f32 lo[3], f32 hi[3], u32 skip, u32 leaf }: threaded (stackless) BVH in
  // depth-first order — on a box hit an inner node continues at node+1, otherwise at
  // `skip`; leaf != 0 encodes (first << 3 | count) into the prim-id section.
  // The blob is copied into LDS by every workgroup when it fits (blob_in_lds).
  const unsigned char *blob;
  uint32_t blob_bytes;
  uint32_t off_ids, off_sph, off_mov, off_tri;
  uint32_t off_pmat, off_mats;  // shading data inside the image: prim -> material, material records
  uint32_t off_sph32, off_mov32; // f32 build's image: binary32 sphere records (16 B / 32 B); its off_tri
                                 // points at binary32 triangle records (48 B)
  int32_t n_nodes;
  int32_t leaf_direct;     // BVH image: records and material indices are in leaf order, ids = identity
  // scene image for the GRID kernel (rtow_grid.h): header, cells, ids, records
  const unsigned char *gblob;
  uint32_t gblob_bytes;
  uint32_t g_off_cells, g_off_ids, g_off_sph, g_off_mov, g_off_tri, g_off_pmat, g_off_mats;
  uint32_t g_off_sph32, g_off_mov32;
  // scene image for the BVH4 kernel (rtow_bvh4.h; triangle meshes): 128- or 64-byte nodes in breadth-first
  // order, then triangle records in leaf order, material index per record, materials
  const unsigned char *blob4;
  uint32_t blob4_bytes;
  uint32_t b4_off_tri, b4_off_pmat, b4_off_mats;
  uint32_t b4_lds_limit;   // bytes of the image staged in LDS: all of it, or a node-aligned prefix of the nodes
  // when only the top of the tree is staged: the END of the image from byte b4_aux_src on (the materials, and the
  // material indices if they are small) is staged too, at LDS offset b4_aux_lds — shading reads them after every hit
  uint32_t b4_aux_src, b4_aux_lds;
  uint32_t b4_stack_base;  // LDS byte offset of the traversal stack ([entry][lane of the workgroup], 4 B each)
  uint32_t b4_stack_k;     // entries per lane in LDS; deeper entries go to TraceParams::spill
  // an image that is not staged whole has 64-byte nodes with binary16 planes (b4_half; rtow_bvh4.h) in the mesh's
  // own frame: plane' = (plane - b4_c) * s per axis; the ray takes (o - b4_c) and 1/(d s) so that t comes out
  // unchanged.  b4_is = 1/s.
  double b4_c[3];
  float b4_is[3];
  uint32_t b4_half;
  // the reference's own tree (rtow_reftree.h; kernel RTOW_KERNEL_REFTREE): [nodes x 64 B][ids], in global memory
  const unsigned char *rtree;
  uint32_t rt_off_ids;
};
constexpr uint32_t kRefNodeBytes = 64;
constexpr uint32_t kRefLeafFlag = 0x80000000u;
constexpr int kRefStackDepth = 48;

struct FastDiv {  // unsigned division by a per-launch constant (see fastdiv() in the kernel)
  uint32_t magic, shift;
};

struct TraceParams {
  DevScene sc;
  const DevCamera *cam;    // in device memory: read with scalar loads where rays are generated
  const float *cam32;      // the same 21 values as binary32 (f32 build)
                           // (by value it pinned 42 SGPRs across the whole kernel)
  int32_t W, H;            // full image
  // A LEVEL is a run of `spt` consecutive samples of every pixel, one work item per pixel.  Strict build: a level
  // is one reference "thread" (stream): spt = spp / nstreams (src/render.cpp:151-166, 174).  Fast builds: the
  // launch's sample range is cut into levels by the host (rtow_capi.cpp, level_plan) whatever nstreams is.
  int32_t spt;             // samples per level
  int32_t nstreams;        // levels traced by THIS launch
  int32_t stream_first;    // index of its first level (strict build: sample_base = stream_first * spt)
  uint32_t sample_base;    // sample index of the first sample of this launch's level 0 (level k starts at sample_base + k * spt)
  int32_t spt_last;        // samples of the launch's LAST level (= spt unless the host's schedule ends on a longer one:
                           // a sample count with no divisor near the aimed-at item length, rtow_capi.cpp level_plan)
  uint32_t tail_bound;     // trips a wave may spend in the end-of-launch protocol before it gives up (structural bound)
  double inv_wm1, inv_hm1; // 1 / (W - 1), 1 / (H - 1): the fast builds multiply where the reference divides (src/render.cpp:158-159)
  int32_t max_child_rays;
  int32_t rank, nranks, tile_rows;
  int32_t local_rows;
  uint32_t seed_lo, seed_hi;
  uint32_t n_items;        // local_rows * W * nstreams
  uint32_t n_lanes;        // grid * block (stride of the path stack)
  FastDiv div_npix, div_w, div_tile;  // item -> (stream, row, column, strip)
  FastDiv div_ns;                     // tiled order: (item / 64) -> (tile, stream)
  // pixel order inside this rank's rows: tiles of 2^tile_w_log2 x 2^tile_h_log2 = 64 pixels,
  // tiles row-major (tile_h_log2 == 0 and tile_w_log2 == 0: plain row-major order)
  uint32_t tile_w_log2, tile_h_log2;
  FastDiv div_tpr;         // tiles per row = W >> tile_w_log2
  uint32_t div_tpr_n;      // the divisor itself
  uint32_t n_tile_rows, sky_rows;  // tiled order: tile rows of this rank, and how many of the top ones come last
  double *partials;        // [nstreams][local_rows*W][3]
  uint32_t *stack;         // [max_child_rays][n_lanes] material index per bounce
  uint32_t *spill;         // BVH4 kernel: traversal stack entries beyond the LDS part, [entry][n_lanes]
  uint32_t spec;           // GRID kernel: scene-class specialisation of the code (0 generic; 1 static spheres only with
                           // 48-byte fat cell lists; 2 static + moving spheres with 80-byte fat lists) — see kSpec* below
  uint32_t b4_trips;       // BVH4: 1 = the trip-structured kernel (rtow_trace_body.h, default), 0 = the state machine
  uint32_t fetch_votes;              // trip kernels: lanes that must need a new work item before the fetch block runs
  uint32_t leaf_votes;               // GRID / BVH4 walks: lanes that must hold a queued cell / leaf before a leaf phase runs
  uint32_t walk_cap, walk_max_open;  // GRID / BVH4 walks: resumable walk (rtow_trace_grid.h); cap 0xffffffff = never stop
  uint32_t sm4_restart, sm4_scatter, sm4_leaf;  // state machine: lanes that must wait for a block before it runs
  unsigned long long *counters; // [0] next item, [1] segments, [2] prim tests, [3] node tests
  unsigned long long *dropped;  // sticky (never reset by a launch): lanes that gave up samples at the tail bound; the
                                // host turns a non-zero word into RTOW_EHIP at its next synchronising entry point
  unsigned long long *t_origin; // diagnostic build: earliest wave start (100 MHz clock)
};

// Scene-class specialisations of the GRID trace kernel (round 4).  The generic kernel carries the code of every
// primitive class and list format; a scene of static spheres executes a third of it, but pays for all of it in
// registers (127 VGPRs and a private segment against 118 and none) and code layout: the specialised instantiation is
// 3.5 % faster on the cover scene.  The host picks it from the resident scene (rtow_capi.cpp); the image is the same.
constexpr uint32_t kSpecGeneric = 0u, kSpecStaticSpheres = 1u, kSpecMovingSpheres = 2u;

struct ReduceParams {
  const double *partials;  // [stream][tiled pixel][3]
  double *out;             // [local_rows*W][3], row-major
  uint32_t npix3;          // local_rows*W*3
  int32_t nstreams;
  int32_t accumulate;      // start from the sums already in `out`
  uint32_t W, tile_w_log2, tile_h_log2, tiles_per_row;
  unsigned char *rgb8;     // not NULL: write_color(sum / spp) as bytes instead of the sums (rtow_render_rgb8)
  double spp;
};

// launchers, one pair per arithmetic mode (separate translation units compiled
// with -ffp-contract=off / -ffp-contract=fast)
// `lds_bytes` > 0 selects the variant that stages the scene blob in LDS
int launch_trace_strict(const TraceParams &p, int kernel, int grid, int block, unsigned lds_bytes,
                        void *stream);
int launch_trace_fast(const TraceParams &p, int kernel, int grid, int block, unsigned lds_bytes,
                      void *stream);
int launch_trace_f32(const TraceParams &p, int kernel, int grid, int block, unsigned lds_bytes,
                     void *stream);
int trace_occupancy_strict(int kernel, int block, unsigned lds_bytes);
int trace_occupancy_fast(int kernel, int block, unsigned lds_bytes);
int trace_occupancy_f32(int kernel, int block, unsigned lds_bytes);
int launch_reduce(const ReduceParams &p, void *stream);
int launch_tonemap(const double *sums, unsigned char *rgb8, uint32_t n, double spp, void *stream);
// rtow_reduce.hip: [rank][max_rows][row_bytes] (what one gather delivers) -> [height][row_bytes] in global row order
int launch_place_rows(const void *gathered, void *image, uint32_t n_ranks, uint32_t max_rows, uint32_t row_bytes,
                      uint32_t height, uint32_t tile_rows, void *stream);

}  // namespace rtow
