// rtow_capi.cpp — the C-ABI shim of include/rtow.h over the HIP kernels.
//
// One context = one HIP device.  The context owns the flattened scene in HBM,
// the workspace (per-stream partial images, per-lane path stack, counters) and a
// ring of event pairs that time every trace-kernel launch without a host sync.
// There is no CPU fallback: without a usable HIP device every entry point fails
// with RTOW_ENODEV / RTOW_EHIP.

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rtow.h"
#include "rtow_bvh.h"
#include "rtow_bvh4.h"
#include "rtow_device.h"
#include "rtow_grid.h"
#include "rtow_reftree.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

// No C++ exception crosses the C ABI (include/rtow.h): the entry points that allocate or start threads run
// their bodies through this.
template <class F>
int guarded(const char *what, F &&f) noexcept {
  try {
    return f();
  } catch (const std::bad_alloc &) {
    return fail(RTOW_ENOMEM, "%s: out of host memory", what);
  } catch (const std::exception &e) {
    return fail(RTOW_EINVAL, "%s: %s", what, e.what());
  } catch (...) {
    return fail(RTOW_EINVAL, "%s: unknown C++ exception", what);
  }
}

}  // namespace
namespace rtow {
// for the other translation units of the library (rtow_multi.cpp): same thread-local message
int set_last_error(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
}  // namespace rtow
namespace {

#define HIPCHK(expr)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return fail(RTOW_EHIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

constexpr int kBlock = 256;      // STREAM kernel workgroup
constexpr int kStreamTileMin = 64;  // STREAM kernel: from this many triangles on they are streamed through LDS tiles (below:
                                    //   scalar loads — the kernel's real use, scenes of <= 16 primitives)
constexpr int kBvhBlock = 512;   // BVH kernel workgroup (one LDS scene image per workgroup)
constexpr unsigned kLdsLimit = 160u * 1024u;
constexpr int kEventRing = 256;

struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  int ensure(size_t need) {
    if (need <= bytes) return RTOW_OK;
    if (p) {
      HIPCHK(hipFree(p));
      p = nullptr;
      bytes = 0;
    }
    size_t want = need + need / 8 + 256;
    HIPCHK(hipMalloc(&p, want));
    bytes = want;
    return RTOW_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
};

// magic/shift for fastdiv(): libdivide's branch-free u32 scheme
// q = (((n - t) >> 1) + t) >> shift with t = mulhi(n, magic)
rtow::FastDiv make_fastdiv(uint32_t d) {
  rtow::FastDiv f{0u, 0u};
  if (d <= 1) {  // identity: flagged with shift = 255
    f.shift = 255u;
    return f;
  }
  uint32_t fl = 31u - (uint32_t)__builtin_clz(d);
  if ((d & (d - 1)) == 0) {  // power of two: t = 0 would give n>>1>>(fl-1)
    f.magic = 0u;
    f.shift = fl - 1u;
    return f;
  }
  const uint64_t two = (uint64_t)1 << (32 + fl);
  uint64_t m = two / d;
  const uint64_t rem = two - m * d;
  // round up, doubled (the 33-bit magic's top bit is implicit in the (n-t)>>1 + t step)
  m = m * 2;
  const uint64_t twice_rem = rem * 2;
  if (twice_rem >= d || twice_rem < rem) m += 1;
  f.magic = (uint32_t)(m + 1);
  f.shift = fl;
  return f;
}

// Host-to-device copies of a scene upload go through one pinned staging arena and are queued on the null
// stream without a host wait each (a dozen small synchronous copies were 0.2 ms of a cover-scene upload); the
// arena is rewound at the start of the next upload, behind that upload's device synchronisation.  What does
// not fit (a big mesh's images) is copied synchronously from pageable memory as before.
struct PinnedArena {
  unsigned char *p = nullptr;
  size_t cap = 0, used = 0;
};
thread_local PinnedArena *g_arena = nullptr;  // the arena of the upload in progress on this thread

template <class T>
int upload(DevBuf &b, const std::vector<T> &v) {
  size_t n = v.size() * sizeof(T);
  int rc = b.ensure(n ? n : sizeof(T));
  if (rc) return rc;
  if (!n) return RTOW_OK;
  PinnedArena *a = g_arena;
  const size_t at = a ? (a->used + 255) / 256 * 256 : 0;
  if (a && a->p && at + n <= a->cap) {
    std::memcpy(a->p + at, v.data(), n);
    a->used = at + n;
    HIPCHK(hipMemcpyAsync(b.p, a->p + at, n, hipMemcpyHostToDevice, nullptr));
  } else {
    HIPCHK(hipMemcpy(b.p, v.data(), n, hipMemcpyHostToDevice));
  }
  return RTOW_OK;
}

// A small host block to a device address inside an upload: through the arena like upload() (the source may be a
// stack buffer: an asynchronous copy must not read it after this returns), synchronously when the arena is full.
int h2d_block(void *dst, const void *src, size_t n) {
  PinnedArena *a = g_arena;
  const size_t at = a ? (a->used + 255) / 256 * 256 : 0;
  if (a && a->p && at + n <= a->cap) {
    std::memcpy(a->p + at, src, n);
    a->used = at + n;
    HIPCHK(hipMemcpyAsync(dst, a->p + at, n, hipMemcpyHostToDevice, nullptr));
  } else {
    HIPCHK(hipMemcpy(dst, src, n, hipMemcpyHostToDevice));
  }
  return RTOW_OK;
}

// Scene image of the f32 build, derived from a binary64 image: the same prefix (nodes or
// header+cells, then ids — everything before the sphere records), the binary64 sphere and moving
// records (large-primitive list, BVH kernel, shading), binary32 triangle records (the only
// triangle section), binary32 sphere and moving records (grid cells), material indices, materials.
struct Image32 {
  std::vector<unsigned char> blob;  // prefix left zeroed when `prefix` is NULL (filled on the device)
  uint32_t off_sph = 0, off_mov = 0, off_tri = 0, off_sph32 = 0, off_mov32 = 0, off_pmat = 0, off_mats = 0;
};
void make_image32(const unsigned char *prefix, uint32_t prefix_bytes, const std::vector<double> &sph,
                  const std::vector<double> &mov, const std::vector<double> &tri, const std::vector<int32_t> &pmat,
                  const std::vector<unsigned char> &mats_bytes, Image32 &out) {
  auto up16 = [](size_t v) { return (v + 15) / 16 * 16; };
  const size_t ns = sph.size() / 4, nm = mov.size() / 8, nt = tri.size() / 12;
  out.off_sph = prefix_bytes;
  out.off_mov = (uint32_t)(out.off_sph + ns * 32);
  out.off_tri = (uint32_t)(out.off_mov + nm * 64);
  out.off_sph32 = (uint32_t)up16(out.off_tri + nt * 48);
  out.off_mov32 = (uint32_t)(out.off_sph32 + ns * 16);
  out.off_pmat = (uint32_t)up16(out.off_mov32 + nm * 32);
  out.off_mats = (uint32_t)up16(out.off_pmat + pmat.size() * 4);
  out.blob.assign(up16(out.off_mats + mats_bytes.size()), 0);
  unsigned char *b = out.blob.data();
  if (prefix) std::memcpy(b, prefix, prefix_bytes);
  if (ns) std::memcpy(b + out.off_sph, sph.data(), ns * 32);
  if (nm) std::memcpy(b + out.off_mov, mov.data(), nm * 64);
  float *t32 = reinterpret_cast<float *>(b + out.off_tri);
  for (size_t i = 0; i < nt * 12; ++i) t32[i] = (float)tri[i];  // A e1 e2 n, 12 floats per triangle
  float *s32 = reinterpret_cast<float *>(b + out.off_sph32);
  for (size_t i = 0; i < ns * 4; ++i) s32[i] = (float)sph[i];    // cx cy cz copysign(r*r, r)
  float *m32 = reinterpret_cast<float *>(b + out.off_mov32);
  for (size_t i = 0; i < nm; ++i) {                              // c0xyz dx | dy dz copysign(r*r, r) 0
    for (int k = 0; k < 7; ++k) m32[i * 8 + k] = (float)mov[i * 8 + k];
    m32[i * 8 + 7] = 0.0f;
  }
  if (!pmat.empty()) std::memcpy(b + out.off_pmat, pmat.data(), pmat.size() * 4);
  if (!mats_bytes.empty()) std::memcpy(b + out.off_mats, mats_bytes.data(), mats_bytes.size());
}

double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

}  // namespace

namespace rtow {  // csrc/rtow_build.hip
int lbvh_build(const double *sph, const double *sph_r, const double *mov, const double *tri, int ns, int nm,
               int nt, double time0, double time1, const double cam_origin[3], int leaf_max, void *stream,
               void **handle, int *n_nodes, int ploc_radius);
int lbvh_emit(void *handle, int leaf_max, unsigned char *blob_dev, uint32_t off_ids, int n_nodes, void *stream);
int lbvh_bvh4_collapse(void *handle, int leaf_max, void *stream, int *n_nodes, int *depth, float root_box[6]);
int lbvh_bvh4_emit(void *handle, unsigned char *blob_dev, int half, const double map_c[3], const double map_s[3],
                   uint32_t off_tri, uint32_t off_pmat, const double *tri, const int32_t *prim_mat, void *stream);
void lbvh_release(void *handle);
// csrc/rtow_build_grid.hip
struct GridBuildBounds {
  double gmn[3], gmx[3];
  double scale_prims;
  int32_t n_small, n_large;
};
int grid_build_phase1(const double *sph, const double *sph_r, const double *mov, const double *tri, int ns, int nm,
                      int nt, double time0, double time1, double large_ratio, void *stream, void **handle,
                      GridBuildBounds *out);
int grid_build_phase2(void *handle, const float gminf[3], const float cellf[3], const int32_t n[3], double pad,
                      void *stream, unsigned long long *total_ids, unsigned long long *max_list);
int grid_build_phase3(void *handle, const float gminf[3], const float cellf[3], const int32_t n[3], double pad,
                      unsigned long long total_ids, unsigned char *blob_dev, uint32_t off_cells, uint32_t off_ids,
                      void *stream, const double *sph, uint32_t off_fat, const double *mov, int ns, uint32_t fat_stride);
void grid_build_release(void *handle);
}  // namespace rtow

// Experiment knobs (RTOW_* environment variables), read ONCE at rtow_ctx_create: nothing on the
// render path touches the environment.
struct Knobs {
  int bvh_leaf = 0;               // RTOW_BVH_LEAF: leaf size cap of the BVH builders (0 = default)
  double bvh_ct = -1.0;           // RTOW_BVH_CT: SAH cost of descending one level (< 0: 0, meshes 1.5)
  bool no_leaf_order = false;     // RTOW_NO_LEAF_ORDER
  double grid_cpp = 1.0;          // RTOW_GRID_CPP: grid cells per primitive (C2 0.5 / 0.75 / 1 / 1.25 / 1.5 / 2 / 2.5:
                                  // 11.01 / 11.03 / 11.18 / 10.72 / 10.99 / 10.98 / 10.85 Gsamples/s; moving 9.47 / 9.53 / 9.45 /
                                  // 9.14 / 8.99 / 8.83 / 7.64; 3-D clouds are indifferent, scripts/bench_cloud.py)
  double grid_large = 4.0;        // RTOW_GRID_LARGE: diagonal ratio that makes a primitive "large"
  int grid_max_tris = 8192;       // RTOW_GRID_MAX_TRIS
  unsigned long long partials_cap = 8ull << 30;  // RTOW_PARTIALS_MAX_MB
  int bvh_block = 0;              // RTOW_BVH_BLOCK: 256 / 512 / 1024 (0 = automatic)
  int blocks_per_cu = 0;          // RTOW_BLOCKS_PER_CU (0 = occupancy query)
  bool no_tiles = false;          // RTOW_NO_TILES
  int sky_eighths = 1;            // RTOW_SKY_EIGHTHS
  bool stamps = false;            // RTOW_STAMPS: diagnostic region-stamp build
  bool bvh4_no_aux = false;       // RTOW_BVH4_NO_AUX: materials / material indices stay in L2 when only the top of the tree is staged
  bool no_bvh4 = false;           // RTOW_NO_BVH4: triangle meshes keep the binary threaded walk
  // Scheduling of the trip kernels (measured on one MI355X; 0 / "off" switches a measure off):
  int fetch_votes = 0;            // RTOW_FETCH_VOTES: lanes that must need a work item before the fetch block runs
                                  // (0 = per kernel: 4, BVH4 2 — 2 / 4 / 8: C4 4.01 / 3.99 / 3.89, C5 1.98 / 1.96 / 1.92, C2 10.72 / 10.73 / 10.71)
  int leaf_votes = 0;             // RTOW_LEAF_VOTES: lanes that must hold a queued cell / leaf before a leaf phase
                                  //   (0 = per kernel: GRID 16, BVH4 28)
  int walk_cap = -1, walk_max_open = 0;  // RTOW_WALK_CAP=cap,max_open | off: resumable walk (-1 = per kernel:
                                         //   GRID 3,16; BVH4 4,24 with every node in LDS, 4,32 otherwise)
  bool bvh4_sm = false;           // RTOW_BVH4_SM: the state-machine form of the BVH4 kernel (rtow_trace_sm4.h)
  int sm4_votes[3] = {8, 16, 16};  // RTOW_SM4_VOTES=restart,scatter,leaf: quorum of the state machine's blocks
  int bvh4_stack_k = 0;           // RTOW_BVH4_STACK_K: image staged whole if this many stack entries per lane still fit
                                  //   (default 8), else the entries per lane beside the staged top of the tree (default 24)
  int sched_chunk = 10;           // RTOW_SCHED_CHUNK: fast builds: samples per work item aimed at (0 = one item per stream and
                                  //   pixel, like the strict build).  5 / 10 / 20 / 25: 10.88 / 11.18 / 10.22 / 9.55 Gsamples/s on C2
  int sched_chunk_mesh = 16;      //   ... for a scene of triangles only (its walks are longer and resumable: fewer, longer items;
                                  //   4 / 8 / 16 / 32 samples: 4.60 / 4.64 / 4.69 / 4.68 Gsamples/s on C4, 2.29 / 2.30 / 2.32 / 2.32 on C5)
  int ploc_radius = -1;           // RTOW_PLOC_RADIUS: device builder: search radius of the PLOC pass (csrc/rtow_build.hip, round 5);
                                  //   0 = Karras' radix tree (rounds 1-4); -1 = default: 16 up to 16,384 primitives, 8 above
  int device_tree = 2;            // RTOW_DEVICE_TREE=radix|ploc|sah: how the device builder makes its binary tree (0 / 1 / 2);
                                  //   sah = binned surface-area heuristic, top-down, level by level (pass 3c): the default
  bool stream_scalar = false;     // RTOW_STREAM_SCALAR: the STREAM kernel's triangle loop through scalar loads at every size (A/B
                                  //   against the LDS-tiled loop; rounds 1-5a)
  bool no_spec = false;           // RTOW_NO_SPEC: always the generic GRID kernel (A/B against the scene-class specialisations)
  int tail_bound = 0;             // RTOW_TAIL_BOUND (tests only): trips of the end-of-launch protocol before a wave gives up
                                  //   its samples (0 = the structural bound); a small value forces the RTOW_EHIP path
  void read() {
    auto geti = [](const char *n, int d) { const char *e = std::getenv(n); return e ? std::atoi(e) : d; };
    auto getd = [](const char *n, double d) { const char *e = std::getenv(n); return e ? std::atof(e) : d; };
    bvh_leaf = geti("RTOW_BVH_LEAF", 0);
    bvh_ct = getd("RTOW_BVH_CT", -1.0);
    no_leaf_order = std::getenv("RTOW_NO_LEAF_ORDER") != nullptr;
    grid_cpp = getd("RTOW_GRID_CPP", 1.0);
    grid_large = getd("RTOW_GRID_LARGE", 4.0);
    grid_max_tris = geti("RTOW_GRID_MAX_TRIS", 8192);
    if (const char *e = std::getenv("RTOW_PARTIALS_MAX_MB")) partials_cap = (unsigned long long)std::atoll(e) << 20;
    bvh_block = geti("RTOW_BVH_BLOCK", 0);
    if (bvh_block != 256 && bvh_block != 512 && bvh_block != 1024) bvh_block = 0;
    blocks_per_cu = geti("RTOW_BLOCKS_PER_CU", 0);
    if (blocks_per_cu < 1 || blocks_per_cu > 16) blocks_per_cu = 0;
    no_tiles = std::getenv("RTOW_NO_TILES") != nullptr;
    sky_eighths = std::min(std::max(geti("RTOW_SKY_EIGHTHS", 1), 0), 8);
    stamps = std::getenv("RTOW_STAMPS") != nullptr;
    no_bvh4 = std::getenv("RTOW_NO_BVH4") != nullptr;
    bvh4_no_aux = std::getenv("RTOW_BVH4_NO_AUX") != nullptr;
    fetch_votes = std::min(std::max(geti("RTOW_FETCH_VOTES", 0), 0), 64);
    leaf_votes = std::min(std::max(geti("RTOW_LEAF_VOTES", 0), 0), 64);
    bvh4_sm = std::getenv("RTOW_BVH4_SM") != nullptr;
    if (const char *e = std::getenv("RTOW_WALK_CAP")) {
      int a = 0, b = 0;
      if (std::sscanf(e, "%d,%d", &a, &b) == 2 && a > 0 && b > 0) {
        walk_cap = a;
        walk_max_open = std::min(b, 64);
      } else {
        walk_cap = 0;  // "off"
      }
    }
    if (const char *e = std::getenv("RTOW_SM4_VOTES")) {
      int a = 0, b = 0, d = 0;
      if (std::sscanf(e, "%d,%d,%d", &a, &b, &d) == 3) {
        sm4_votes[0] = std::min(std::max(a, 1), 64);
        sm4_votes[1] = std::min(std::max(b, 1), 64);
        sm4_votes[2] = std::min(std::max(d, 1), 64);
      }
    }
    bvh4_stack_k = std::min(std::max(geti("RTOW_BVH4_STACK_K", 0), 0), 64);
    sched_chunk = std::min(std::max(geti("RTOW_SCHED_CHUNK", 10), 0), 4096);
    sched_chunk_mesh = std::min(std::max(geti("RTOW_SCHED_CHUNK_MESH", 16), 0), 4096);
    tail_bound = std::max(geti("RTOW_TAIL_BOUND", 0), 0);
    no_spec = std::getenv("RTOW_NO_SPEC") != nullptr;
    stream_scalar = std::getenv("RTOW_STREAM_SCALAR") != nullptr;
    if (const char *e = std::getenv("RTOW_PLOC_RADIUS")) ploc_radius = std::min(std::max(std::atoi(e), 0), 64);
    if (const char *e = std::getenv("RTOW_DEVICE_TREE"))
      device_tree = std::strcmp(e, "radix") == 0 ? 0 : (std::strcmp(e, "ploc") == 0 ? 1 : 2);
    if (ploc_radius >= 0 && !std::getenv("RTOW_DEVICE_TREE")) device_tree = ploc_radius == 0 ? 0 : 1;  // (the radius alone selects PLOC / radix)
  }
};

// what a scene upload builds besides the record arrays: rtow_scene_upload builds everything (the scene stays
// resident for any later render); rtow_render / rtow_render_rgb8 know their config and build what its kernel reads
// kNeedBvh: the binary threaded image (BVH kernel); kNeedBvh4: the 4-wide image of a triangle mesh (BVH4 kernel) — a
// render that walks the one does not pay for the other (the 96.8k-triangle mesh: 20 ms of host work and 10 MB of
// upload less per rtow_render call)
constexpr unsigned kNeedBvh = 1u, kNeedGrid = 2u, kNeedF32 = 4u, kNeedBvh4 = 8u, kNeedAll = 15u;

struct rtow_ctx {
  int device = 0;
  Knobs knobs;
  int num_cus = 0;
  bool have_scene = false;
  rtow::DevScene ds{};
  rtow::DevCamera cam{};
  int n_prims = 0;
  // scene buffers
  DevBuf sph, sph_r, mov, tri, tri16, prim_mat, mats, blob, cam_dev, gblob;
  DevBuf blob4;                        // 4-wide BVH image (triangle meshes)
  bool have_bvh4 = false;
  bool have_tri16 = false;             // the STREAM kernel's 128-byte-stride triangle records are resident
  int bvh4_depth = 0;
  DevBuf blob32, gblob32, cam32_dev;  // the f32 build's scene images and camera
  rtow::DevScene ds32{};               // ds with the f32 images' pointers and offsets
  uint32_t gblob_bytes = 0;
  uint32_t grid_fat_stride = 0;  // bytes per fat cell-list entry of the resident grid image (0: plain id lists)
  bool have_grid = false;
  uint32_t blob_bytes = 0;
  long long bvh_nodes = 0;
  int builder = RTOW_BUILDER_HOST_SAH;      // the builder of the upload in progress / of the resident images
  int builder_req = RTOW_BUILDER_AUTO;      // what the caller asked for (rtow_ctx_set_builder, RTOW_BUILDER)
  void *lbvh_scratch = nullptr;  // device builder's buffers, kept across uploads
  void *grid_scratch = nullptr;
  rtow_build_info_t build_info{};
  // RTOW_KERNEL_REFTREE: the reference's own tree is built on first use from a host copy of the scene as uploaded
  struct HostSceneCopy {
    std::vector<double> sg, mg, tg;
    std::vector<int32_t> pk, pi;
    bool have_order = false;
  } host_scene;
  DevBuf rtree;
  bool have_rtree = false;
  PinnedArena arena;     // staging of the scene uploads
  unsigned built = 0;    // kNeed* bits of what the resident scene holds (rtow_render uploads only what its kernel reads)
  // workspace
  DevBuf partials, stack, counters, spill;
  DevBuf out, out8;  // rtow_render / rtow_render_rgb8: device-side output of the host-buffer entry points
  // rtow_render_rgb8: the reduce kernel of the next single-launch render writes bytes here instead of sums
  unsigned char *fuse_rgb8 = nullptr;
  double fuse_spp = 0.0;
  bool fuse_used = false;
  DevBuf counters_init;  // what the 48 counters look like before a launch (one D2D copy instead of three memsets)
  // Samples given up at the end-of-launch bound (rtow_trace_body.h): a device word no launch resets, mirrored into
  // pinned memory behind every render; every entry point that waits for the device checks it (check_dropped)
  DevBuf dropped;
  unsigned long long *h_dropped = nullptr;
  hipEvent_t upload_ev = nullptr;  // recorded on the null stream behind the last copy / build kernel of a scene upload
  // launch shape per [precision][kernel-1]: blocks per CU (0 = not queried yet)
  int occ[3][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}};
  // profiling ring: event pairs around each trace-kernel launch since the last collect
  hipEvent_t ev[kEventRing][2] = {};
  int ev_count = 0;
  hipEvent_t call_ev[2] = {};
  // pinned host mirror of the counters for stats
  unsigned long long *h_counters = nullptr;
};

// BVH4: stack entries per lane (4 B x 1024 lanes each) an image staged whole must leave room for (RTOW_BVH4_STACK_K)
// (round 5: 6, was 8 — suzanne's host tree walks at the same rate with 8, 7 and 6 entries per lane in LDS, and two more
// entry levels are 8 KB = 64 nodes of room for a mesh near the limit: the device-built suzanne tree, 276 nodes against
// the host's 261, is staged whole with 7)
static uint32_t bvh4_min_stack(const rtow_ctx *c) { return c->knobs.bvh4_stack_k > 0 ? (uint32_t)c->knobs.bvh4_stack_k : 6u; }

extern "C" {

int rtow_abi_version(void) { return RTOW_ABI_VERSION; }

const char *rtow_last_error(void) { return g_err.c_str(); }

static int impl_ctx_create(int device_id, rtow_ctx **out) {
  if (!out) return fail(RTOW_EINVAL, "rtow_ctx_create: out is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(RTOW_ENODEV, "no HIP device (%s)", e == hipSuccess ? "count=0" : hipGetErrorString(e));
  if (device_id < 0 || device_id >= n) return fail(RTOW_ENODEV, "device %d out of range [0,%d)", device_id, n);
  HIPCHK(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device_id));
  rtow_ctx *c = new rtow_ctx();
  c->device = device_id;
  c->num_cus = prop.multiProcessorCount;
  c->knobs.read();
  if (const char *e = std::getenv("RTOW_BUILDER"))
    c->builder_req = std::strcmp(e, "device") == 0 ? RTOW_BUILDER_DEVICE_LBVH
                     : std::strcmp(e, "auto") == 0 ? RTOW_BUILDER_AUTO : RTOW_BUILDER_HOST_SAH;
  c->builder = c->builder_req == RTOW_BUILDER_DEVICE_LBVH ? RTOW_BUILDER_DEVICE_LBVH : RTOW_BUILDER_HOST_SAH;
  // every failure below releases what was created so far (rtow_ctx_destroy skips null handles)
  hipError_t he = hipSuccess;
  for (int i = 0; i < kEventRing && he == hipSuccess; ++i)
    for (int k = 0; k < 2 && he == hipSuccess; ++k) he = hipEventCreate(&c->ev[i][k]);
  for (int k = 0; k < 2 && he == hipSuccess; ++k) he = hipEventCreate(&c->call_ev[k]);
  if (he == hipSuccess)
    he = hipHostMalloc((void **)&c->h_counters, 48 * sizeof(unsigned long long), hipHostMallocDefault);
  if (he == hipSuccess) he = hipHostMalloc((void **)&c->h_dropped, sizeof(unsigned long long), hipHostMallocDefault);
  if (he == hipSuccess) {
    *c->h_dropped = 0ull;
    he = hipEventCreateWithFlags(&c->upload_ev, hipEventDisableTiming);
  }
  if (he == hipSuccess && c->dropped.ensure(sizeof(unsigned long long)) != RTOW_OK) he = hipErrorOutOfMemory;
  if (he == hipSuccess) he = hipMemset(c->dropped.p, 0, sizeof(unsigned long long));
  if (he != hipSuccess) {
    rtow_ctx_destroy(c);
    return fail(RTOW_EHIP, "rtow_ctx_create: %s", hipGetErrorString(he));
  }
  *out = c;
  return RTOW_OK;
}

void rtow_ctx_destroy(rtow_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  for (DevBuf *b : {&c->sph, &c->sph_r, &c->mov, &c->tri, &c->tri16, &c->prim_mat, &c->mats, &c->blob, &c->cam_dev, &c->gblob,
                    &c->blob32, &c->gblob32, &c->cam32_dev, &c->blob4,
                    &c->partials, &c->stack, &c->counters, &c->spill, &c->out, &c->out8, &c->rtree, &c->counters_init,
                    &c->dropped})
    b->release();
  if (c->h_dropped) (void)hipHostFree(c->h_dropped);
  if (c->upload_ev) (void)hipEventDestroy(c->upload_ev);
  if (c->arena.p) (void)hipHostFree(c->arena.p);
  for (int i = 0; i < kEventRing; ++i)
    for (int k = 0; k < 2; ++k)
      if (c->ev[i][k]) (void)hipEventDestroy(c->ev[i][k]);
  for (int k = 0; k < 2; ++k)
    if (c->call_ev[k]) (void)hipEventDestroy(c->call_ev[k]);
  if (c->h_counters) (void)hipHostFree(c->h_counters);
  rtow::lbvh_release(c->lbvh_scratch);
  rtow::grid_build_release(c->grid_scratch);
  // nothing of this context is pending in the runtime when the caller goes on (often: to exit())
  (void)hipDeviceSynchronize();
  delete c;
}

static int validate_scene(const rtow_scene_t *s) {
  if (!s) return fail(RTOW_EINVAL, "scene is NULL");
  if (s->n_spheres < 0 || s->n_moving < 0 || s->n_triangles < 0 || s->n_materials < 0)
    return fail(RTOW_EINVAL, "negative count in scene");
  const long long np = (long long)s->n_spheres + s->n_moving + s->n_triangles;
  if (np == 0) return fail(RTOW_EEMPTY, "scene has no primitives");
  if (s->n_prims != np) return fail(RTOW_EINVAL, "n_prims (%d) != sum of classes (%lld)", s->n_prims, np);
  if (s->n_materials == 0 || !s->materials) return fail(RTOW_EINVAL, "scene has no materials");
  if ((s->n_spheres && (!s->sphere_geom || !s->sphere_mat)) ||
      (s->n_moving && (!s->moving_geom || !s->moving_mat)) ||
      (s->n_triangles && (!s->triangle_geom || !s->triangle_mat)))
    return fail(RTOW_EINVAL, "NULL geometry/material array");
  auto chk = [&](const int32_t *m, int n) {
    for (int i = 0; i < n; ++i)
      if (m[i] < 0 || m[i] >= s->n_materials) return false;
    return true;
  };
  if (!chk(s->sphere_mat, s->n_spheres) || !chk(s->moving_mat, s->n_moving) ||
      !chk(s->triangle_mat, s->n_triangles))
    return fail(RTOW_EINVAL, "material index out of range");
  for (int i = 0; i < s->n_materials; ++i)
    if (s->materials[i].kind < 0 || s->materials[i].kind > 2)
      return fail(RTOW_EINVAL, "material %d has unknown kind %d", i, s->materials[i].kind);
  return RTOW_OK;
}

static int scene_upload(rtow_ctx *c, const rtow_scene_t *s, unsigned need);
static int impl_scene_upload(rtow_ctx *c, const rtow_scene_t *s) { return scene_upload(c, s, kNeedAll); }

static int scene_upload(rtow_ctx *c, const rtow_scene_t *s, unsigned need) {
  if (!c) return fail(RTOW_EINVAL, "ctx is NULL");
  int rc = validate_scene(s);
  if (rc) return rc;
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipDeviceSynchronize());
  if (!c->arena.p) {
    if (hipHostMalloc((void **)&c->arena.p, 8u << 20, hipHostMallocDefault) == hipSuccess)
      c->arena.cap = 8u << 20;
    else
      c->arena.p = nullptr;  // (uploads then copy synchronously)
  }
  c->arena.used = 0;
  struct ArenaScope {  // the free function upload() finds the arena of the upload in progress here
    explicit ArenaScope(PinnedArena *a) { g_arena = a; }
    ~ArenaScope() { g_arena = nullptr; }
  } arena_scope(&c->arena);
  if (need & kNeedF32) need |= kNeedBvh | kNeedGrid;  // the binary32 images are derived from the binary64 ones (BVH2, grid)
  c->built = 0;
  c->have_scene = false;
  c->have_rtree = false;
  c->build_info.ref_tree_nodes = 0;
  c->build_info.ref_tree_stupid_volume = 0.0;
  const double t_up0 = now_ms();
  {
    rtow_ctx::HostSceneCopy &h = c->host_scene;
    h.sg.assign(s->sphere_geom, s->sphere_geom + 4 * (size_t)s->n_spheres);
    h.mg.assign(s->moving_geom, s->moving_geom + 8 * (size_t)s->n_moving);
    h.tg.assign(s->triangle_geom, s->triangle_geom + 9 * (size_t)s->n_triangles);
    h.have_order = s->prim_kind && s->prim_index;
    if (h.have_order) {
      for (int i = 0; i < s->n_prims; ++i) {
        const int k = s->prim_kind[i], ix = s->prim_index[i];
        const int cnt = k == RTOW_PRIM_SPHERE ? s->n_spheres : k == RTOW_PRIM_MOVING_SPHERE ? s->n_moving
                        : k == RTOW_PRIM_TRIANGLE ? s->n_triangles : -1;
        if (ix < 0 || ix >= cnt) return fail(RTOW_EINVAL, "prim_kind / prim_index entry %d out of range", i);
      }
      h.pk.assign(s->prim_kind, s->prim_kind + s->n_prims);
      h.pi.assign(s->prim_index, s->prim_index + s->n_prims);
    } else {
      h.pk.clear();
      h.pi.clear();
    }
  }

  // Ray-independent terms, computed with the reference's operations (this file is
  // built with -ffp-contract=off, so each is one IEEE operation, as on the CPU).
  const int ns = s->n_spheres, nm = s->n_moving, nt = s->n_triangles;
  std::vector<double> sph((size_t)ns * 4), sph_r(ns), mov((size_t)nm * 8), tri((size_t)nt * 12);
  for (int i = 0; i < ns; ++i) {
    const double *g = s->sphere_geom + 4 * (size_t)i;
    sph[4 * (size_t)i + 0] = g[0];
    sph[4 * (size_t)i + 1] = g[1];
    sph[4 * (size_t)i + 2] = g[2];
    // radius * radius (src/common-model.cpp:73), carrying the radius' sign: the tests take |.|
    // for free (VOP3 input modifier), the shading step reads the sign (front_facing, :88)
    sph[4 * (size_t)i + 3] = std::copysign(g[3] * g[3], g[3]);
    sph_r[i] = g[3];
  }
  for (int i = 0; i < nm; ++i) {
    const double *g = s->moving_geom + 8 * (size_t)i;
    double *d = &mov[8 * (size_t)i];
    d[0] = g[0];
    d[1] = g[1];
    d[2] = g[2];
    d[3] = g[3] - g[0];  // center1 - center0, src/oo-primitives.h:65
    d[4] = g[4] - g[1];
    d[5] = g[5] - g[2];
    d[6] = std::copysign(g[6] * g[6], g[6]);
    d[7] = g[6];
  }
  for (int i = 0; i < nt; ++i) {
    const double *g = s->triangle_geom + 9 * (size_t)i;
    double *d = &tri[12 * (size_t)i];
    const double e1[3] = {g[3] - g[0], g[4] - g[1], g[5] - g[2]};  // b - a
    const double e2[3] = {g[6] - g[0], g[7] - g[1], g[8] - g[2]};  // c - a
    d[0] = g[0];
    d[1] = g[1];
    d[2] = g[2];
    for (int k = 0; k < 3; ++k) {
      d[3 + k] = e1[k];
      d[6 + k] = e2[k];
    }
    // n = cross(e1, e2), src/common-model.cpp:108
    d[9] = e1[1] * e2[2] - e2[1] * e1[2];
    d[10] = e1[2] * e2[0] - e2[2] * e1[0];
    d[11] = e1[0] * e2[1] - e2[0] * e1[1];
  }
  std::vector<int32_t> pmat((size_t)ns + nm + nt);
  for (int i = 0; i < ns; ++i) pmat[i] = s->sphere_mat[i];
  for (int i = 0; i < nm; ++i) pmat[(size_t)ns + i] = s->moving_mat[i];
  for (int i = 0; i < nt; ++i) pmat[(size_t)ns + nm + i] = s->triangle_mat[i];
  std::vector<rtow::DevMaterial> mats(s->n_materials);
  for (int i = 0; i < s->n_materials; ++i) {
    const rtow_material_t &m = s->materials[i];
    rtow::DevMaterial &d = mats[i];
    std::memset(&d, 0, sizeof d);
    const bool diel = m.kind == RTOW_MAT_DIELECTRIC;  // attenuation {1,1,1}, common-model.cpp:61
    for (int k = 0; k < 3; ++k) d.att[k] = diel ? 1.0 : m.albedo[k];
    d.fuzz = m.kind == RTOW_MAT_LAMBERTIAN ? 0.0 : m.fuzz;
    d.ir = m.ir;
    d.kind = m.kind;
  }

  // BVH over the same records, packed with them into one scene image
  // host SAH stops splitting by cost (mostly 1-2 primitives per leaf, cap 4); the radix tree has
  // no cost model, so its leaves are capped at 2 (measured: 1 and 2 equal, 4 is 8-17 % slower)
  int leaf_max = c->builder == RTOW_BUILDER_DEVICE_LBVH ? 2 : 4;
  // Triangle meshes (the tree the 4-wide image is collapsed from): PAIRS of triangles per leaf — a cost of 1.5
  // primitive tests per level descended makes the SAH stop at two triangles, the cap keeps it from stopping
  // earlier.  Against single-triangle leaves (cost 0, what the builder makes of any cap): suzanne 452 -> 261
  // 4-wide nodes, the whole image now fits LDS with 7 stack entries per lane, 4.06 -> 4.67 Gsamples/s; the
  // 96.8k-triangle mesh 2.00 -> 2.08.  (Cost 1 / cap 4: 4.45 / 2.01; cap 3: 4.49 / 2.01; cost 3 / cap 2: the same
  // tree as 1.5 / 2.)  The binary walk over the same tree loses 3.5 % on suzanne (3.39 -> 3.27): it is the fallback.
  const bool mesh_tree = ns == 0 && nm == 0 && nt > 0 && c->builder != RTOW_BUILDER_DEVICE_LBVH && !c->knobs.no_bvh4;
  if (mesh_tree) leaf_max = 2;
  if (c->knobs.bvh_leaf > 0) leaf_max = std::min(std::max(c->knobs.bvh_leaf, 1), 7);
  rtow::SceneImage img;
  std::vector<unsigned char> mats_bytes(mats.size() * sizeof(rtow::DevMaterial));
  std::memcpy(mats_bytes.data(), mats.data(), mats_bytes.size());
  if ((rc = upload(c->sph, sph)) || (rc = upload(c->sph_r, sph_r)) || (rc = upload(c->mov, mov)) ||
      (rc = upload(c->tri, tri)) || (rc = upload(c->prim_mat, pmat)) || (rc = upload(c->mats, mats)))
    return rc;
  // the STREAM kernel's copy of the triangle records at a 128-byte stride (rtow_device.h): for the uploads that may be
  // followed by a STREAM render of triangles — everything (rtow_scene_upload) or nothing but the records (a render
  // that asked for the STREAM / reference-tree kernels)
  c->have_tri16 = false;
  if (nt > 0 && (need == kNeedAll || need == 0u)) {
    std::vector<double> tri16((size_t)nt * 16, 0.0);
    for (int i = 0; i < nt; ++i) std::memcpy(&tri16[(size_t)i * 16], &tri[(size_t)i * 12], 12 * sizeof(double));
    if ((rc = upload(c->tri16, tri16))) return rc;
    c->have_tri16 = true;
  }
  bool leaf_direct = false;
  std::vector<double> tri_img;    // leaf-ordered copies (triangle meshes, host builder)
  std::vector<int32_t> pmat_img;
  const double t_bvh0 = now_ms();
  c->have_bvh4 = false;
  c->build_info.bvh4_nodes = 0;
  c->build_info.bvh4_image_bytes = 0;
  c->build_info.bvh4_node_bytes = 0;
  const bool want2 = (need & kNeedBvh) != 0, want4 = (need & kNeedBvh4) != 0 && ns == 0 && nm == 0 && nt > 0 && !c->knobs.no_bvh4;
  if (want2 || want4) {
  if (c->builder == RTOW_BUILDER_DEVICE_LBVH) {
    // the tree is built in HBM from the record arrays just uploaded; the host only lays out
    // the image sections around it
    int n_nodes = 0;
    int brc = rtow::lbvh_build((const double *)c->sph.p, (const double *)c->sph_r.p, (const double *)c->mov.p,
                               (const double *)c->tri.p, ns, nm, nt, s->camera.t0, s->camera.t1, s->camera.origin,
                               leaf_max, nullptr, &c->lbvh_scratch, &n_nodes,
                               c->knobs.device_tree == 2   ? -1
                               : c->knobs.device_tree == 0 ? 0
                               : c->knobs.ploc_radius > 0  ? c->knobs.ploc_radius
                                                           : (ns + nm + nt <= 16384 ? 16 : 8));
    if (brc) return fail(RTOW_EHIP, "device BVH build failed (stage %d): %s", brc, hipGetErrorString(hipGetLastError()));
    if (want2) {
    rtow::layout_scene_image(n_nodes, (size_t)(ns + nm + nt), sph, mov, tri, pmat, mats_bytes, img, true);
    if ((rc = c->blob.ensure(img.total_bytes))) return rc;
    // image sections: device-to-device copies of the arrays uploaded above, zeroed node section
    unsigned char *bp = (unsigned char *)c->blob.p;
    HIPCHK(hipMemsetAsync(bp, 0, img.off_sph, nullptr));
    if (ns) HIPCHK(hipMemcpyAsync(bp + img.off_sph, c->sph.p, sph.size() * 8, hipMemcpyDeviceToDevice, nullptr));
    if (nm) HIPCHK(hipMemcpyAsync(bp + img.off_mov, c->mov.p, mov.size() * 8, hipMemcpyDeviceToDevice, nullptr));
    if (nt) HIPCHK(hipMemcpyAsync(bp + img.off_tri, c->tri.p, tri.size() * 8, hipMemcpyDeviceToDevice, nullptr));
    HIPCHK(hipMemcpyAsync(bp + img.off_pmat, c->prim_mat.p, pmat.size() * 4, hipMemcpyDeviceToDevice, nullptr));
    HIPCHK(hipMemcpyAsync(bp + img.off_mats, c->mats.p, mats_bytes.size(), hipMemcpyDeviceToDevice, nullptr));
    brc = rtow::lbvh_emit(c->lbvh_scratch, leaf_max, bp, img.off_ids, n_nodes, nullptr);
    if (brc == 4) return fail(RTOW_EINVAL, "internal error: device-built scene image failed validation (BVH links)");
    if (brc) return fail(RTOW_EHIP, "device BVH emit failed (stage %d): %s", brc, hipGetErrorString(hipGetLastError()));
    }
    // triangle meshes: the 4-wide image the BVH4 kernel walks, collapsed from the same radix tree ON THE DEVICE
    // (csrc/rtow_build.hip) — the reference builds its tree inside render()'s timer (src/render.cpp:73-110,141-188);
    // with the device builder nothing of the build runs on the host
    if (want4 && leaf_max <= 4) {
      int n4 = 0, depth4 = 0;
      float rb[6];
      brc = rtow::lbvh_bvh4_collapse(c->lbvh_scratch, leaf_max, nullptr, &n4, &depth4, rb);
      if (brc == 4) return fail(RTOW_EINVAL, "internal error: device 4-wide collapse failed validation");
      if (brc != 0 && brc != 5)
        return fail(RTOW_EHIP, "device 4-wide collapse failed (stage %d): %s", brc, hipGetErrorString(hipGetLastError()));
      if (brc == 0) {  // (5: beyond the format's limits — the binary walk takes the scene)
        auto up16 = [](size_t v) { return (v + 15) / 16 * 16; };
        auto layout = [&](uint32_t node_bytes, uint32_t &off_tri, uint32_t &off_pmat, uint32_t &off_mats) {
          off_tri = (uint32_t)((size_t)n4 * node_bytes);
          off_pmat = (uint32_t)up16(off_tri + (size_t)nt * 96);
          off_mats = (uint32_t)up16(off_pmat + (size_t)nt * 4);
          return up16(off_mats + mats_bytes.size());
        };
        uint32_t off_tri4 = 0, off_pmat4 = 0, off_mats4 = 0;
        size_t total4 = layout(rtow::kBvh4NodeBytes, off_tri4, off_pmat4, off_mats4);
        // the same rule as the host builder: binary32 planes when LDS holds the whole image beside the stack,
        // 64-byte nodes with binary16 planes in the mesh's own frame otherwise (rtow_bvh4.h)
        const bool half = total4 + bvh4_min_stack(c) * 4u * 1024u > kLdsLimit;
        double map_c[3] = {0, 0, 0}, map_s[3] = {1, 1, 1};
        if (half) {
          total4 = layout(rtow::kBvh4HalfNodeBytes, off_tri4, off_pmat4, off_mats4);
          const double widest = std::max({(double)rb[3] - rb[0], (double)rb[4] - rb[1], (double)rb[5] - rb[2]});
          for (int k = 0; k < 3; ++k) {
            const double lo = rb[k], hi = rb[3 + k];
            const double half_k = std::max({0.5 * (hi - lo), 1e-4 * widest, 1e-30});
            map_c[k] = 0.5 * (lo + hi);
            map_s[k] = 1000.0 / half_k;
          }
        }
        if ((rc = c->blob4.ensure(total4))) return rc;
        unsigned char *b4 = (unsigned char *)c->blob4.p;
        HIPCHK(hipMemsetAsync(b4 + off_pmat4, 0, total4 - off_pmat4, nullptr));  // (alignment gaps)
        HIPCHK(hipMemcpyAsync(b4 + off_mats4, c->mats.p, mats_bytes.size(), hipMemcpyDeviceToDevice, nullptr));
        brc = rtow::lbvh_bvh4_emit(c->lbvh_scratch, b4, half ? 1 : 0, map_c, map_s, off_tri4, off_pmat4,
                                   (const double *)c->tri.p, (const int32_t *)c->prim_mat.p, nullptr);
        if (brc == 4) return fail(RTOW_EINVAL, "internal error: device-built 4-wide image failed validation");
        if (brc) return fail(RTOW_EHIP, "device 4-wide emit failed (stage %d): %s", brc, hipGetErrorString(hipGetLastError()));
        c->have_bvh4 = true;
        c->bvh4_depth = depth4;
        c->ds.blob4 = b4;
        c->ds.blob4_bytes = (uint32_t)total4;
        c->ds.b4_off_tri = off_tri4;
        c->ds.b4_off_pmat = off_pmat4;
        c->ds.b4_off_mats = off_mats4;
        c->ds.b4_half = half ? 1u : 0u;
        for (int k = 0; k < 3; ++k) {
          c->ds.b4_c[k] = map_c[k];
          c->ds.b4_is[k] = (float)(1.0 / map_s[k]);
        }
        c->build_info.bvh4_nodes = n4;
        c->build_info.bvh4_image_bytes = (int32_t)total4;
        c->build_info.bvh4_node_bytes = (int32_t)(half ? rtow::kBvh4HalfNodeBytes : rtow::kBvh4NodeBytes);
      }
    }
  } else {
    rtow::HostBvh bvh;
    const double c_trav = c->knobs.bvh_ct >= 0.0 ? c->knobs.bvh_ct : (mesh_tree ? 1.5 : 0.0);
    rtow::build_bvh(sph, sph_r, mov, tri, bvh, leaf_max, c_trav, s->camera.t0, s->camera.t1);
    std::vector<int32_t> prim_order;  // the tree's primitive order before the leaf-order pass renumbers it (for the 4-wide image)
    if (want4 && leaf_max <= 4) prim_order = bvh.prim;
    if (want2) {
    if (ns == 0 && nm == 0 && !c->knobs.no_leaf_order) {
      // Triangle meshes: the image holds the triangle records and their material indices in LEAF order
      // and the id list is the identity, so a leaf test reads its records directly instead of id ->
      // record (one LDS / L2 round trip less per leaf, and leaf neighbours are memory neighbours:
      // +10 % on the 96.8k-triangle mesh, whose image lives in global memory).  `best.prim` is then a
      // leaf slot; shading reads the same permuted sections.  (Host builder only.)
      tri_img.resize(tri.size());
      pmat_img.resize(pmat.size());
      for (size_t sl = 0; sl < bvh.prim.size(); ++sl) {
        std::memcpy(&tri_img[sl * 12], &tri[(size_t)bvh.prim[sl] * 12], 96);
        pmat_img[sl] = pmat[bvh.prim[sl]];
      }
      for (size_t sl = 0; sl < bvh.prim.size(); ++sl) bvh.prim[sl] = (int32_t)sl;
      rtow::make_scene_image(bvh, sph, mov, tri_img, s->camera.origin, img, pmat_img, mats_bytes);
      leaf_direct = true;
    } else
    rtow::make_scene_image(bvh, sph, mov, tri, s->camera.origin, img, pmat, mats_bytes);
    if (!rtow::validate_scene_image(img, ns + nm + nt))
      return fail(RTOW_EINVAL, "internal error: scene image failed validation (BVH links)");
    if ((rc = upload(c->blob, img.blob))) return rc;
    }
    // triangle meshes: the 4-wide tree collapsed from the same SAH tree (rtow_bvh4.h)
    c->have_bvh4 = false;
    if (want4) {
      rtow::Bvh4Image img4;
      if (!prim_order.empty()) {  // the same SAH tree, collapsed (leaves of at most 4 triangles)
        bvh.prim = prim_order;
        // an image that LDS cannot hold whole is read from L2 below the top of its tree: 64-byte nodes with
        // binary16 planes (4 loads per node instead of 7; rtow_bvh4.h).  The records alone decide it for a big mesh
        // (no point in building the binary32 image first: 20 ms of the 96.8k-triangle mesh's upload)
        const bool surely_half = (size_t)nt * 100u + bvh4_min_stack(c) * 4u * 1024u > kLdsLimit;
        rtow::make_bvh4_image(bvh, tri, pmat, mats_bytes, s->camera.origin, img4, /*half=*/surely_half);
        if (!surely_half && img4.ok && img4.blob.size() + bvh4_min_stack(c) * 4u * 1024u > kLdsLimit)
          rtow::make_bvh4_image(bvh, tri, pmat, mats_bytes, s->camera.origin, img4, /*half=*/true);
      }
      if (img4.ok) {
        if (!rtow::validate_bvh4_image(img4, (size_t)nt))
          return fail(RTOW_EINVAL, "internal error: 4-wide scene image failed validation");
        if ((rc = upload(c->blob4, img4.blob))) return rc;
        c->have_bvh4 = true;
        c->bvh4_depth = img4.depth;
        c->ds.blob4 = (const unsigned char *)c->blob4.p;
        c->ds.blob4_bytes = (uint32_t)img4.blob.size();
        c->ds.b4_off_tri = img4.off_tri;
        c->ds.b4_off_pmat = img4.off_pmat;
        c->ds.b4_off_mats = img4.off_mats;
        c->ds.b4_half = img4.half ? 1u : 0u;
        for (int k = 0; k < 3; ++k) {
          c->ds.b4_c[k] = img4.map_c[k];
          c->ds.b4_is[k] = (float)(1.0 / img4.map_s[k]);
        }
        c->build_info.bvh4_nodes = img4.n_nodes;
        c->build_info.bvh4_image_bytes = (int32_t)img4.blob.size();
        c->build_info.bvh4_node_bytes = (int32_t)img4.node_bytes();
      }
    }
  }
  }  // kNeedBvh
  const double t_bvh1 = now_ms();
  c->blob_bytes = (uint32_t)img.total_bytes;
  c->bvh_nodes = img.n_nodes;
  // uniform grid over the small primitives (when the scene suits it)
  rtow::GridImage gimg;
  const double cpp = c->knobs.grid_cpp;
  const double large_ratio = c->knobs.grid_large;
  // big meshes never take the grid (a triangle spans many cells: 0.4x the BVH):
  // skip the host build, a GRID request then falls back to the BVH
  const int grid_max_tris = c->knobs.grid_max_tris;
  bool grid_on_device = false;
  if (!(need & kNeedGrid)) {
    // (not asked for)
  } else if (nt <= grid_max_tris && c->builder == RTOW_BUILDER_DEVICE_LBVH) {
    // the same grid, built in HBM (csrc/rtow_build_grid.hip); the host does the scalar steps between
    // the phases with the code the host builder uses, so the image is byte-identical
    grid_on_device = true;
    rtow::GridBuildBounds gb;
    int grc = rtow::grid_build_phase1((const double *)c->sph.p, (const double *)c->sph_r.p, (const double *)c->mov.p,
                                      (const double *)c->tri.p, ns, nm, nt, s->camera.t0, s->camera.t1, large_ratio,
                                      nullptr, &c->grid_scratch, &gb);
    if (grc) return fail(RTOW_EHIP, "device grid build failed (phase 1, stage %d): %s", grc, hipGetErrorString(hipGetLastError()));
    if (gb.n_small > 0 && gb.n_large <= 64) {
      rtow::GridHeader hd;
      rtow::grid_header(gb.gmn, gb.gmx, gb.scale_prims, s->camera.origin, (size_t)gb.n_small, cpp, hd);
      unsigned long long total_ids = 0, max_list = 0;
      grc = rtow::grid_build_phase2(c->grid_scratch, hd.gminf, hd.cellf, hd.n, hd.pad, nullptr, &total_ids, &max_list);
      if (grc) return fail(RTOW_EHIP, "device grid build failed (phase 2, stage %d): %s", grc, hipGetErrorString(hipGetLastError()));
      if (max_list <= 255 && total_ids + (unsigned long long)gb.n_large < (1u << 24)) {
        const size_t ncell = (size_t)hd.n[0] * hd.n[1] * hd.n[2];
        for (int k = 0; k < 3; ++k) gimg.n[k] = hd.n[k];
        double scale_small = 0.0;  // (the host builder's rule: rtow_grid.h build_grid_image)
        for (int k = 0; k < 3; ++k)
          scale_small = std::max({scale_small, std::fabs(gb.gmn[k]), std::fabs(gb.gmx[k]), std::fabs(s->camera.origin[k])});
        const uint32_t fat = rtow::grid_wants_fat_lists(nm, nt, ncell, (size_t)total_ids + (size_t)gb.n_large,
                                                        (size_t)gb.n_large, sph, mov, tri, pmat, mats_bytes, (size_t)total_ids,
                                                        scale_small);
        rtow::layout_grid_image(ncell, (size_t)total_ids + (size_t)gb.n_large, (size_t)gb.n_large, sph, mov, tri, pmat,
                                mats_bytes, gimg, true, fat ? (size_t)total_ids : 0, fat ? fat : 48u);
        if ((rc = c->gblob.ensure(gimg.total_bytes))) return rc;
        unsigned char *gp = (unsigned char *)c->gblob.p;
        unsigned char header[64];
        const uint32_t off_large = (uint32_t)(gimg.off_ids + total_ids * 4);
        rtow::write_grid_header(header, hd, gimg.n_large, off_large, gimg.off_fat, gimg.fat_stride);
        HIPCHK(hipMemsetAsync(gp, 0, gimg.total_bytes, nullptr));
        if ((rc = h2d_block(gp, header, 64))) return rc;  // (`header` is on the stack: never the source of an async copy)
        if (ns) HIPCHK(hipMemcpyAsync(gp + gimg.off_sph, c->sph.p, sph.size() * 8, hipMemcpyDeviceToDevice, nullptr));
        if (nm) HIPCHK(hipMemcpyAsync(gp + gimg.off_mov, c->mov.p, mov.size() * 8, hipMemcpyDeviceToDevice, nullptr));
        if (nt) HIPCHK(hipMemcpyAsync(gp + gimg.off_tri, c->tri.p, tri.size() * 8, hipMemcpyDeviceToDevice, nullptr));
        HIPCHK(hipMemcpyAsync(gp + gimg.off_pmat, c->prim_mat.p, pmat.size() * 4, hipMemcpyDeviceToDevice, nullptr));
        HIPCHK(hipMemcpyAsync(gp + gimg.off_mats, c->mats.p, mats_bytes.size(), hipMemcpyDeviceToDevice, nullptr));
        grc = rtow::grid_build_phase3(c->grid_scratch, hd.gminf, hd.cellf, hd.n, hd.pad, total_ids, gp, gimg.off_cells,
                                      gimg.off_ids, nullptr, (const double *)c->sph.p, gimg.off_fat,
                                      (const double *)c->mov.p, ns, gimg.fat_stride);
        if (grc) return fail(RTOW_EHIP, "device grid build failed (phase 3, stage %d): %s", grc, hipGetErrorString(hipGetLastError()));
        gimg.ok = true;
      }
    }
  } else if (nt <= grid_max_tris) {
    rtow::build_grid_image(sph, sph_r, mov, tri, s->camera.origin, gimg, cpp, large_ratio, s->camera.t0,
                           s->camera.t1, pmat, mats_bytes);
  }
  const double t_grid1 = now_ms();
  c->have_grid = gimg.ok;
  c->grid_fat_stride = (gimg.ok && gimg.off_fat) ? gimg.fat_stride : 0u;
  c->gblob_bytes = 0;
  if (gimg.ok) {
    if (!grid_on_device && (rc = upload(c->gblob, gimg.blob))) return rc;
    c->gblob_bytes = (uint32_t)(grid_on_device ? gimg.total_bytes : gimg.blob.size());
  }
  for (auto &o : c->occ) o[0] = o[1] = o[2] = o[3] = o[4] = 0;

  rtow::DevScene &ds = c->ds;
  ds.sph = (const double *)c->sph.p;
  ds.sph_r = (const double *)c->sph_r.p;
  ds.mov = (const double *)c->mov.p;
  ds.tri = (const double *)c->tri.p;
  ds.tri16 = c->have_tri16 ? (const double *)c->tri16.p : nullptr;
  ds.prim_mat = (const int32_t *)c->prim_mat.p;
  ds.mats = (const rtow::DevMaterial *)c->mats.p;
  ds.n_sph = ns;
  ds.n_mov = nm;
  ds.n_tri = nt;
  ds.n_mats = s->n_materials;
  ds.blob = (const unsigned char *)c->blob.p;
  ds.blob_bytes = c->blob_bytes;
  ds.off_ids = img.off_ids;
  ds.off_sph = img.off_sph;
  ds.off_mov = img.off_mov;
  ds.off_tri = img.off_tri;
  ds.off_pmat = img.off_pmat;
  ds.off_mats = img.off_mats;
  ds.n_nodes = img.n_nodes;
  ds.leaf_direct = leaf_direct ? 1 : 0;
  ds.gblob = (const unsigned char *)c->gblob.p;
  ds.gblob_bytes = c->gblob_bytes;
  ds.g_off_cells = gimg.off_cells;
  ds.g_off_ids = gimg.off_ids;
  ds.g_off_sph = gimg.off_sph;
  ds.g_off_mov = gimg.off_mov;
  ds.g_off_tri = gimg.off_tri;
  ds.g_off_pmat = gimg.off_pmat;
  ds.g_off_mats = gimg.off_mats;
  c->n_prims = ns + nm + nt;

  const rtow_camera_t &k = s->camera;
  rtow::DevCamera &dc = c->cam;
  for (int i = 0; i < 3; ++i) {
    dc.origin[i] = k.origin[i];
    dc.u[i] = k.u[i];
    dc.v[i] = k.v[i];
    dc.horizontal[i] = k.horizontal[i];
    dc.vertical[i] = k.vertical[i];
    dc.llc[i] = k.lower_left_corner[i];
  }
  dc.lens_radius = k.lens_radius;
  dc.t0 = k.t0;
  dc.t1 = k.t1;
  {
    std::vector<rtow::DevCamera> one(1, dc);
    if ((rc = upload(c->cam_dev, one))) return rc;
  }
  // ---- the f32 build's images (same trees, binary32 records for the small primitives) ----
  c->ds32 = c->ds;
  if (need & kNeedF32) {
    Image32 b32;
    const bool device_tree = c->builder == RTOW_BUILDER_DEVICE_LBVH;
    make_image32(device_tree ? nullptr : img.blob.data(), img.off_sph, sph, mov, leaf_direct ? tri_img : tri,
                 leaf_direct ? pmat_img : pmat, mats_bytes, b32);
    if ((rc = upload(c->blob32, b32.blob))) return rc;
    if (device_tree)  // nodes + ids exist only in HBM
      HIPCHK(hipMemcpy(c->blob32.p, c->blob.p, img.off_sph, hipMemcpyDeviceToDevice));
    rtow::DevScene &d32 = c->ds32;
    d32 = c->ds;
    d32.blob = (const unsigned char *)c->blob32.p;
    d32.blob_bytes = (uint32_t)b32.blob.size();
    d32.off_sph = b32.off_sph;
    d32.off_mov = b32.off_mov;
    d32.off_tri = b32.off_tri;
    d32.off_sph32 = b32.off_sph32;
    d32.off_mov32 = b32.off_mov32;
    d32.off_pmat = b32.off_pmat;
    d32.off_mats = b32.off_mats;
    d32.gblob = nullptr;
    d32.gblob_bytes = 0;
    if (gimg.ok) {
      Image32 g32;
      make_image32(grid_on_device ? nullptr : gimg.blob.data(), gimg.off_sph, sph, mov, tri, pmat, mats_bytes, g32);
      if ((rc = upload(c->gblob32, g32.blob))) return rc;
      if (grid_on_device)  // header, cells and ids exist only in HBM
        HIPCHK(hipMemcpy(c->gblob32.p, c->gblob.p, gimg.off_sph, hipMemcpyDeviceToDevice));
      d32.gblob = (const unsigned char *)c->gblob32.p;
      d32.gblob_bytes = (uint32_t)g32.blob.size();
      d32.g_off_sph = g32.off_sph;
      d32.g_off_mov = g32.off_mov;
      d32.g_off_tri = g32.off_tri;
      d32.g_off_sph32 = g32.off_sph32;
      d32.g_off_mov32 = g32.off_mov32;
      d32.g_off_pmat = g32.off_pmat;
      d32.g_off_mats = g32.off_mats;
    }
    const double *cd = reinterpret_cast<const double *>(&dc);
    std::vector<float> cam32(21);
    for (int i = 0; i < 21; ++i) cam32[i] = (float)cd[i];
    if ((rc = upload(c->cam32_dev, cam32))) return rc;
  }
  // Ordering contract (include/rtow.h): the copies and build kernels above are queued on the null stream without a
  // host wait; a render on ANOTHER stream (a hipStreamNonBlocking stream is not ordered behind the null stream) waits
  // for this event first (render_levels)
  HIPCHK(hipEventRecord(c->upload_ev, nullptr));
  c->built = need;
  c->have_scene = true;
  rtow_build_info_t &bi = c->build_info;
  bi.builder = c->builder;
  bi.bvh_nodes = (int32_t)img.n_nodes;
  bi.bvh_image_bytes = (int32_t)c->blob_bytes;
  bi.grid_image_bytes = (int32_t)c->gblob_bytes;
  bi.bvh_build_ms = t_bvh1 - t_bvh0;
  bi.grid_build_ms = t_grid1 - t_bvh1;
  bi.upload_ms = now_ms() - t_up0;
  return RTOW_OK;
}

static int validate_cfg(const rtow_config_t *cfg) {
  if (!cfg) return fail(RTOW_EINVAL, "config is NULL");
  if (cfg->image_width <= 0 || cfg->image_height <= 0)
    return fail(RTOW_EINVAL, "image size %dx%d", cfg->image_width, cfg->image_height);
  if (cfg->nstreams <= 0) return fail(RTOW_EINVAL, "nstreams must be >= 1");
  if (cfg->samples_per_pixel < 0 || cfg->max_child_rays < 0)
    return fail(RTOW_EINVAL, "negative samples_per_pixel / max_child_rays");
  if (cfg->nranks <= 0 || cfg->rank < 0 || cfg->rank >= cfg->nranks)
    return fail(RTOW_EINVAL, "rank %d of %d", cfg->rank, cfg->nranks);
  if (cfg->tile_rows <= 0) return fail(RTOW_EINVAL, "tile_rows must be >= 1");
  if (cfg->precision != RTOW_F64_STRICT && cfg->precision != RTOW_F64_FAST && cfg->precision != RTOW_F32)
    return fail(RTOW_EINVAL, "unknown precision %d", cfg->precision);
  if (cfg->kernel < RTOW_KERNEL_AUTO || cfg->kernel > RTOW_KERNEL_REFTREE)
    return fail(RTOW_EINVAL, "unknown kernel %d", cfg->kernel);
  if ((long long)cfg->image_width * cfg->image_height > 0x7fffffffLL)
    return fail(RTOW_EINVAL, "image too large");
  if (cfg->stream_first < 0 || cfg->stream_count < 0 ||
      (cfg->stream_count > 0 && cfg->stream_first + cfg->stream_count > cfg->nstreams) ||
      (cfg->stream_count == 0 && cfg->stream_first != 0))
    return fail(RTOW_EINVAL, "stream range [%d, %d+%d) outside [0, %d)", cfg->stream_first, cfg->stream_first,
                cfg->stream_count, cfg->nstreams);
  return RTOW_OK;
}

int rtow_ctx_set_builder(rtow_ctx *c, int32_t builder) {
  if (!c) return fail(RTOW_EINVAL, "ctx is NULL");
  if (builder != RTOW_BUILDER_HOST_SAH && builder != RTOW_BUILDER_DEVICE_LBVH && builder != RTOW_BUILDER_AUTO)
    return fail(RTOW_EINVAL, "unknown builder %d", builder);
  c->builder_req = builder;
  c->builder = builder == RTOW_BUILDER_DEVICE_LBVH ? RTOW_BUILDER_DEVICE_LBVH : RTOW_BUILDER_HOST_SAH;  // (AUTO without a config: host)
  return RTOW_OK;
}

int rtow_build_info(rtow_ctx *c, rtow_build_info_t *out) {
  if (!c || !out) return fail(RTOW_EINVAL, "NULL argument");
  if (!c->have_scene) return fail(RTOW_EINVAL, "no scene uploaded");
  *out = c->build_info;
  return RTOW_OK;
}

int rtow_local_rows(const rtow_config_t *cfg) {
  int rc = validate_cfg(cfg);
  if (rc) return rc;
  int rows = 0;
  for (int i = 0; i < cfg->image_height; ++i)
    if ((i / cfg->tile_rows) % cfg->nranks == cfg->rank) ++rows;
  return rows;
}

int rtow_local_row_list(const rtow_config_t *cfg, int32_t *rows_out, int32_t capacity) {
  int rc = validate_cfg(cfg);
  if (rc) return rc;
  if (!rows_out && capacity > 0) return fail(RTOW_EINVAL, "rows_out is NULL");
  int n = 0;
  for (int i = 0; i < cfg->image_height; ++i)
    if ((i / cfg->tile_rows) % cfg->nranks == cfg->rank) {
      if (n < capacity) rows_out[n] = i;
      ++n;
    }
  return n;
}

// Levels: what one work item per pixel covers.  Strict build: a level is a stream, spp / nstreams samples summed
// in sample order, the partial images added in stream order — the reference's own decomposition
// (src/render.cpp:169-185), bit for bit.  Fast builds (tolerance builds: FMA contraction already re-associates):
// the SAME samples, cut into levels of a length that does not depend on nstreams — the divisor of the sample
// range nearest RTOW_SCHED_CHUNK (10, the measured optimum of the item length) — so Config::nthreads keeps its
// arithmetic meaning (which samples exist: spp / nthreads * nthreads) without setting the size of a work item: a
// drop-in caller with the reference's default of 4 threads (25 or 125 samples per stream) runs the items the
// bench runs.  The sum of a pixel is then the fixed-order sum of its level sums; the image is a pure function
// of (scene, config, seed) and of nothing else.
// (the item length aimed at depends on the class of the resident scene, not on anything else)
static int sched_chunk_for(const rtow_ctx *c) {
  const bool mesh = c->have_scene && c->ds.n_tri > 0 && c->ds.n_sph == 0 && c->ds.n_mov == 0;
  return mesh ? c->knobs.sched_chunk_mesh : c->knobs.sched_chunk;
}
struct LevelPlan {
  int spt;             // samples per level
  int first;           // index of the first level when the levels are the config's streams (strict build)
  int count;           // levels
  long long base;      // sample index of the first sample of level 0: level k covers [base + k * spt, ...)
  int last;            // samples of the LAST level (= spt, or spt <= last < 2 spt when the schedule is ragged)
  long long samples() const { return count > 0 ? (long long)(count - 1) * spt + last : 0; }
};
static LevelPlan level_plan(const rtow_config_t *cfg, int chunk) {
  const int spt = cfg->samples_per_pixel / cfg->nstreams;  // src/render.cpp:174
  const int streams_now = cfg->stream_count > 0 ? cfg->stream_count : cfg->nstreams;
  LevelPlan p{spt, cfg->stream_first, streams_now, (long long)cfg->stream_first * spt, spt};
  if (cfg->precision == RTOW_F64_STRICT || chunk <= 0 || spt <= 0) return p;
  // a level length d must divide the first sample index and the number of samples: any divisor of their gcd
  const long long s0 = (long long)cfg->stream_first * spt, total = (long long)streams_now * spt;
  long long g = total, r = s0;
  while (r) {
    const long long t = g % r;
    g = r;
    r = t;
  }
  long long best = 1;
  double best_err = 1e300;
  for (long long d = 1; d * d <= g; ++d) {
    if (g % d) continue;
    for (long long e : {d, g / d}) {
      const double err = std::fabs(std::log((double)e / (double)chunk));
      if (err < best_err - 1e-12 || (std::fabs(err - best_err) <= 1e-12 && e < best)) {
        best_err = err;
        best = e;
      }
    }
  }
  if (2 * best < chunk || best > 2LL * chunk) {
    // No divisor within a factor of two of the aimed-at length (a prime sample count: 101 would be 101 levels of
    // one sample — 101 work items and 101 partial images per pixel — or one level of 101).  Ragged schedule
    // instead: levels of `chunk` samples from the first sample on, the remainder added to the LAST level
    // (chunk <= last < 2 chunk; a range shorter than a chunk is one level).  Still a pure function of the config.
    const long long n = std::max<long long>(total / chunk, 1);
    if (n > 0x7fffffffLL) return p;
    p.spt = n > 1 ? chunk : (int)total;
    p.count = (int)n;
    p.last = (int)(total - (n - 1) * (long long)p.spt);
    p.base = s0;
    p.first = 0;
    return p;
  }
  if (total / best > 0x7fffffffLL || best > 0x7fffffffLL) return p;
  p.spt = (int)best;
  p.first = (int)(s0 / best);
  p.count = (int)(total / best);
  p.last = p.spt;
  p.base = s0;
  return p;
}

// (diagnostic / tests) the levels a render of `cfg` is cut into: pairs (first sample, count).  ctx may be NULL.
static int impl_debug_schedule(rtow_ctx *c, const rtow_config_t *cfg, uint32_t *out, int32_t capacity_pairs) {
  int rc = validate_cfg(cfg);
  if (rc) return rc;
  Knobs defaults;  // ctx NULL: the defaults of a new context (pure host arithmetic, usable without a GPU)
  if (!c) defaults.read();
  const LevelPlan p = level_plan(cfg, c ? sched_chunk_for(c) : defaults.sched_chunk);
  for (int i = 0; i < p.count && i < capacity_pairs && out; ++i) {
    out[2 * i] = (uint32_t)(p.base + (long long)i * p.spt);
    out[2 * i + 1] = (uint32_t)(i + 1 == p.count ? p.last : p.spt);
  }
  return p.count;
}

// 64-pixel tiles of the queue order: 8x8, 16x4 or 32x2 — the tallest that divides the strip height, this rank's
// rows and the image width (a tile never straddles two strips); 0, 0 = plain row-major order
static void tile_shape(const rtow_config_t *cfg, int rows, uint32_t &th, uint32_t &tw) {
  th = tw = 0;
  for (uint32_t h : {3u, 2u, 1u}) {
    const uint32_t hh = 1u << h, ww = 64u >> h;
    if (cfg->tile_rows % hh == 0 && rows % (int)hh == 0 && cfg->image_width % (int)ww == 0) {
      th = h;
      tw = 6u - h;
      return;
    }
  }
}

static int render_levels(rtow_ctx *c, const rtow_config_t *cfg, void *d_rgb_sums, void *hip_stream,
                         rtow_stats_t *stats, const LevelPlan &plan, int lvl_first, int lvl_count, int accumulate);

// Samples given up at the end-of-launch bound are an ERROR, never a darker image: the kernels count them in a device
// word no launch resets.  Entry points that wait for the device anyway read it: `mirrored` = the word was copied to
// c->h_dropped by a copy queued behind the render on a stream the caller has since waited for; otherwise it is read
// with a blocking copy (the launches in question must have completed).
static int check_dropped(rtow_ctx *c, bool mirrored) {
  unsigned long long v = 0ull;
  if (mirrored)
    v = *c->h_dropped;
  else
    HIPCHK(hipMemcpy(&v, c->dropped.p, sizeof v, hipMemcpyDeviceToHost));
  if (v == 0ull) return RTOW_OK;
  *c->h_dropped = 0ull;
  HIPCHK(hipMemset(c->dropped.p, 0, sizeof v));
  return fail(RTOW_EHIP, "trace kernel: end-of-launch bound reached, %llu lanes gave up their samples", v);
}
static int mirror_dropped(rtow_ctx *c, hipStream_t st) {
  HIPCHK(hipMemcpyAsync(c->h_dropped, c->dropped.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  return RTOW_OK;
}

static int impl_render_device(rtow_ctx *c, const rtow_config_t *cfg, void *d_rgb_sums, void *hip_stream,
                       rtow_stats_t *stats) {
  if (!c) return fail(RTOW_EINVAL, "ctx is NULL");
  int rc = validate_cfg(cfg);
  if (rc) return rc;
  if (!c->have_scene) return fail(RTOW_ENOSCENE, "no scene uploaded");
  if (!d_rgb_sums) return fail(RTOW_EINVAL, "d_rgb_sums is NULL");
  HIPCHK(hipSetDevice(c->device));

  const int rows = rtow_local_rows(cfg);
  const int spt = cfg->samples_per_pixel / cfg->nstreams;  // src/render.cpp:174
  const unsigned long long npix = (unsigned long long)rows * cfg->image_width;
  const int streams_now = cfg->stream_count > 0 ? cfg->stream_count : cfg->nstreams;
  if ((unsigned long long)spt * (unsigned long long)(cfg->stream_first + streams_now) > 0xffffffffULL)
    return fail(RTOW_EINVAL, "sample index beyond 32 bits");
  const LevelPlan plan = level_plan(cfg, sched_chunk_for(c));
  const int n_levels = plan.count;
  // Large sample counts: the per-level partial sums (24 B per pixel and level) are bounded by tracing the
  // levels in ranges that accumulate onto d_rgb_sums — bit-identical to one launch (the reduce kernel adds
  // level sums in level order either way).
  unsigned long long max_levels = (unsigned long long)n_levels;
  if (npix > 0) {
    max_levels = c->knobs.partials_cap / (npix * 24ull);
    if (0xfffffff0ULL / npix < max_levels) max_levels = 0xfffffff0ULL / npix;  // 32-bit item index
    if (max_levels < 1) max_levels = 1;
  }
  if ((unsigned long long)n_levels <= max_levels || spt == 0 || npix == 0)
    return render_levels(c, cfg, d_rgb_sums, hip_stream, stats, plan, 0, n_levels, cfg->accumulate);
  rtow_stats_t total;
  std::memset(&total, 0, sizeof total);
  for (int done = 0; done < n_levels;) {
    const int now = (int)std::min<unsigned long long>(max_levels, (unsigned long long)(n_levels - done));
    rtow_stats_t st1;
    rc = render_levels(c, cfg, d_rgb_sums, hip_stream, stats ? &st1 : nullptr, plan, done, now,
                       (done > 0 || cfg->accumulate) ? 1 : 0);
    if (rc) return rc;
    if (stats) {
      total.samples += st1.samples;
      total.segments += st1.segments;
      total.prim_tests += st1.prim_tests;
      total.node_tests += st1.node_tests;
      total.kernel_ms += st1.kernel_ms;
      total.total_ms += st1.total_ms;
      total.local_rows = st1.local_rows;
      total.kernel_used = st1.kernel_used;
    }
    done += now;
  }
  if (stats) *stats = total;
  return RTOW_OK;
}

// One launch: levels [lvl_first, lvl_first + lvl_count) of the plan.
static int render_levels(rtow_ctx *c, const rtow_config_t *cfg, void *d_rgb_sums, void *hip_stream,
                         rtow_stats_t *stats, const LevelPlan &plan, int lvl_first, int lvl_count, int accumulate) {
  int rc;
  hipStream_t st = (hipStream_t)hip_stream;
  const int rows = rtow_local_rows(cfg);
  const unsigned long long npix = (unsigned long long)rows * cfg->image_width;
  const int streams_now = lvl_count;  // levels of this launch
  const int spt = plan.spt;
  const int spt_last = lvl_first + lvl_count == plan.count ? plan.last : plan.spt;  // (a ragged schedule ends on a longer level)
  const unsigned long long n_items = npix * (unsigned long long)streams_now;
  if (n_items > 0xfffffff0ULL) return fail(RTOW_EINVAL, "too many work items (%llu)", n_items);
  int kernel = cfg->kernel;
  // a handful of primitives is cheaper to stream than to walk
  // AUTO: a handful of primitives is cheaper to stream than to walk; sphere scenes walk the
  // grid (measured 1.3x the BVH on the cover scene); triangle meshes walk the BVH (a triangle
  // spans many cells and its test is 3.5x a sphere's, so duplicates are expensive: 0.4x)
  const bool strict = cfg->precision == RTOW_F64_STRICT;
  const bool f32 = cfg->precision == RTOW_F32;
  const bool bvh4_ok = c->have_bvh4 && !f32;  // triangle mesh, host builder, binary64 build
  if (kernel == RTOW_KERNEL_AUTO)
    kernel = c->n_prims <= 16 ? RTOW_KERNEL_BRUTE
             : (c->have_grid && c->ds.n_tri == 0) ? RTOW_KERNEL_GRID
             : bvh4_ok ? RTOW_KERNEL_BVH4 : RTOW_KERNEL_BVH;
  if (kernel == RTOW_KERNEL_GRID && !c->have_grid) kernel = RTOW_KERNEL_BVH;  // scene not suited to a grid
  if (kernel == RTOW_KERNEL_BRUTE && c->ds.n_tri > 0 && !c->have_tri16)
    return fail(RTOW_ENOSCENE, "the resident scene was uploaded for another kernel (the STREAM kernel's triangle records are "
                               "missing): call rtow_scene_upload");
  if (kernel == RTOW_KERNEL_BVH4 && !bvh4_ok) kernel = RTOW_KERNEL_BVH;
  if ((kernel == RTOW_KERNEL_BVH && !(c->built & kNeedBvh)) || (f32 && !(c->built & kNeedF32)))
    return fail(RTOW_ENOSCENE, "the resident scene was uploaded by rtow_render for another kernel / precision: "
                               "call rtow_scene_upload before rtow_render_device");
  if (kernel == RTOW_KERNEL_REFTREE) {
    // the reference's own tree and box test: an exactness mode, so it exists in the strict build only
    if (!strict) return fail(RTOW_EINVAL, "RTOW_KERNEL_REFTREE needs precision RTOW_F64_STRICT");
    if (!c->have_rtree) {
      rtow_scene_t hs;
      std::memset(&hs, 0, sizeof hs);
      const rtow_ctx::HostSceneCopy &h = c->host_scene;
      hs.n_spheres = c->ds.n_sph;
      hs.n_moving = c->ds.n_mov;
      hs.n_triangles = c->ds.n_tri;
      hs.n_prims = c->n_prims;
      hs.sphere_geom = h.sg.data();
      hs.moving_geom = h.mg.data();
      hs.triangle_geom = h.tg.data();
      hs.prim_kind = h.have_order ? h.pk.data() : nullptr;
      hs.prim_index = h.have_order ? h.pi.data() : nullptr;
      rtow::RefTree rt;
      const double t0 = now_ms();
      rtow::build_reftree(&hs, rt);
      if (!rt.ok) return fail(RTOW_EINVAL, "reference tree: depth %d exceeds the kernel's stack", rt.depth);
      if ((rc = upload(c->rtree, rt.blob))) return rc;
      c->ds.rtree = (const unsigned char *)c->rtree.p;
      c->ds.rt_off_ids = rt.off_ids;
      c->have_rtree = true;
      c->build_info.ref_tree_nodes = rt.n_nodes;
      c->build_info.ref_tree_stupid_volume = rt.stupid_volume;
      c->build_info.ref_tree_build_ms = now_ms() - t0;
    }
  }
  rtow::DevScene scene = f32 ? c->ds32 : c->ds;
  int block = (kernel >= RTOW_KERNEL_BVH && kernel <= RTOW_KERNEL_BVH4) ? kBvhBlock : kBlock;
  const bool image_kernel = kernel >= RTOW_KERNEL_BVH && kernel <= RTOW_KERNEL_BVH4;
  if (image_kernel && c->knobs.bvh_block) block = c->knobs.bvh_block;  // experiment knob
  // the scene image goes to LDS when one copy per workgroup fits (160 KiB per CU)
  const uint32_t image_bytes = kernel == RTOW_KERNEL_GRID ? scene.gblob_bytes : scene.blob_bytes;
  // an image too big for two workgroups per CU: ONE 1024-lane workgroup (16 waves, the same 4 per SIMD)
  // instead of one 512-lane workgroup (2 per SIMD)
  if (image_kernel && block == kBvhBlock && image_bytes <= kLdsLimit && 2u * image_bytes > kLdsLimit &&
      !c->knobs.bvh_block)
    block = 1024;
  unsigned lds_bytes = (image_kernel && image_bytes <= kLdsLimit) ? image_bytes : 0u;
  scene.stream_tile_lds = 0u;
  if (kernel == RTOW_KERNEL_BRUTE && scene.n_tri >= kStreamTileMin && !c->knobs.stream_scalar) {
    // the tiled triangle loop of the STREAM kernel (csrc/rtow_trace_hit.h): two 3 KB tiles of LDS per wave
    scene.stream_tile_lds = 2u * 32u * 96u;
    lds_bytes = (unsigned)(block / 64) * scene.stream_tile_lds;
  }
  int stack_bound = 0;
  if (kernel == RTOW_KERNEL_BVH4) {
    // One 1024-lane workgroup per CU.  LDS = [image, or the top of its tree][stack: K entries x 4 B per lane].
    // A small mesh is staged whole and the stack takes what is left (at least 8 entries per lane);
    // a big one gets 24 stack entries per lane and as many breadth-first nodes as fit beside them.
    block = 1024;
    const uint32_t per_entry = 4u * (uint32_t)block;
    stack_bound = 3 * c->bvh4_depth + 1;
    uint32_t K, staged, aux_src = scene.blob4_bytes, aux_bytes = 0u;
    const uint32_t min_k = bvh4_min_stack(c);
    const uint32_t node_bytes = scene.b4_half ? rtow::kBvh4HalfNodeBytes : rtow::kBvh4NodeBytes;
    if (!scene.b4_half && scene.blob4_bytes + min_k * per_entry <= kLdsLimit) {
      staged = scene.blob4_bytes;
      K = std::min<uint32_t>((kLdsLimit - staged) / per_entry, 32u);
      if (c->knobs.bvh4_stack_k > 0) K = std::min<uint32_t>(K, (uint32_t)c->knobs.bvh4_stack_k);  // (experiments: fewer)
    } else if (!scene.b4_half) {
      return fail(RTOW_EINVAL, "internal error: a 4-wide image with binary32 nodes must fit LDS whole");
    } else {
      // (stack entries 6 / 8 / 10 / 12 / 16 / 24 on the 96.8k-triangle mesh: 2.07 / 2.12 / 2.12 / 2.11 / 2.10 / 2.10
      // Gsamples/s — what the stack does not need holds 128-byte nodes)
      K = c->knobs.bvh4_stack_k > 0 ? std::min<uint32_t>((uint32_t)c->knobs.bvh4_stack_k, 32u) : 10u;
      // the end of the image goes to LDS as well when it is small: materials + material indices, or the
      // materials alone (shading reads index -> material after every hit: two dependent L2 round trips otherwise)
      const uint32_t aux_max = c->knobs.bvh4_no_aux ? 0u : 16u * 1024u;
      if (scene.blob4_bytes - scene.b4_off_pmat <= aux_max)
        aux_src = scene.b4_off_pmat;
      else if (scene.blob4_bytes - scene.b4_off_mats <= aux_max)
        aux_src = scene.b4_off_mats;
      aux_bytes = scene.blob4_bytes - aux_src;
      staged = std::min<uint32_t>((kLdsLimit - K * per_entry - aux_bytes) / node_bytes * node_bytes, scene.b4_off_tri);
    }
    K = std::min<uint32_t>(K, (uint32_t)stack_bound);
    scene.b4_lds_limit = staged;
    scene.b4_aux_src = aux_src;
    scene.b4_aux_lds = (staged + 15u) / 16u * 16u;
    scene.b4_stack_base = (scene.b4_aux_lds + aux_bytes + 15u) / 16u * 16u;
    scene.b4_stack_k = K;
    lds_bytes = scene.b4_stack_base + K * per_entry;
  }

  if (stats) {
    std::memset(stats, 0, sizeof *stats);
    stats->local_rows = rows;
    stats->kernel_used = kernel;
  }
  if (npix == 0) return RTOW_OK;
  if (spt == 0) {
    // fewer samples than streams: zero effective samples (src/render.cpp:174), black sums
    if (!accumulate) HIPCHK(hipMemsetAsync(d_rgb_sums, 0, (size_t)npix * 3 * sizeof(double), st));
    if (stats) HIPCHK(hipStreamSynchronize(st));
    return RTOW_OK;
  }

  // grid: as many 256-lane blocks as stay resident, but no more than there are items
  int &occ = c->occ[strict ? 0 : (f32 ? 2 : 1)][kernel - 1];
  if (kernel == RTOW_KERNEL_BVH4) occ = 0;  // its LDS footprint depends on the scene: query every time (cheap)
  if (occ <= 0) {
    occ = strict ? rtow::trace_occupancy_strict(kernel, block, lds_bytes)
          : f32  ? rtow::trace_occupancy_f32(kernel, block, lds_bytes)
                 : rtow::trace_occupancy_fast(kernel, block, lds_bytes);
    if (occ <= 0) return fail(RTOW_EHIP, "occupancy query failed (kernel %d, %u B of LDS)", kernel, lds_bytes);
    if (occ > 8) occ = 8;
    if (c->knobs.blocks_per_cu) occ = c->knobs.blocks_per_cu;  // experiment knob
  }
  long long grid = (long long)c->num_cus * occ;
  const long long need_blocks = (long long)((n_items + block - 1) / block);
  if (grid > need_blocks) grid = need_blocks;
  if (grid < 1) grid = 1;
  const unsigned long long n_lanes = (unsigned long long)grid * block;

  const size_t depth_slots = (size_t)(cfg->max_child_rays > 0 ? cfg->max_child_rays : 1);
  if ((rc = c->partials.ensure((size_t)n_items * 3 * sizeof(double))) ||
      (rc = c->stack.ensure(strict ? depth_slots * (size_t)n_lanes * sizeof(uint32_t) : 4)) ||  // strict build only
      (rc = c->counters.ensure(48 * sizeof(unsigned long long))))
    return rc;
  if (kernel == RTOW_KERNEL_BVH4) {
    const int extra = std::max(stack_bound - (int)scene.b4_stack_k, 0);
    if ((rc = c->spill.ensure(std::max<size_t>((size_t)extra * (size_t)n_lanes * sizeof(uint32_t), 16)))) return rc;
  }

#ifdef RTOW_TAILSTAT  // (experiment build, scripts/tailstat.py; sphere scenes only: the BVH4 kernel needs its spill array itself)
  if (kernel == RTOW_KERNEL_BVH4) return fail(RTOW_EINVAL, "RTOW_TAILSTAT builds do not run the BVH4 kernel");
  if ((rc = c->spill.ensure(std::max<size_t>((size_t)n_lanes, 16)))) return rc;  // 8 words of 8 bytes per wave
#endif
  rtow::TraceParams P;
  std::memset(&P, 0, sizeof P);
  P.sc = scene;
  P.cam = (const rtow::DevCamera *)c->cam_dev.p;
  P.cam32 = (const float *)c->cam32_dev.p;
  P.W = cfg->image_width;
  P.H = cfg->image_height;
  P.spt = spt;
  P.nstreams = streams_now;
  P.stream_first = plan.first + lvl_first;
  P.sample_base = (uint32_t)(plan.base + (long long)lvl_first * spt);
  P.spt_last = spt_last;
  {
    // structural bound of the end-of-launch protocol (rtow_trace_body.h): x64 because a stopped-and-resumed walk
    // spreads one segment over several trips
    const unsigned long long b = 64ull * (4096ull + 8ull * (unsigned long long)(cfg->max_child_rays + 2) *
                                                        (unsigned long long)(std::max(spt, spt_last) + 1));
    P.tail_bound = c->knobs.tail_bound > 0 ? (uint32_t)c->knobs.tail_bound : (uint32_t)std::min<unsigned long long>(b, 0xfffffff0ull);
  }
  P.dropped = (unsigned long long *)c->dropped.p;
  P.inv_wm1 = 1.0 / (double)(cfg->image_width - 1);  // (W = 1: inf, and the fast build's u = 0 * inf is NaN like the
  P.inv_hm1 = 1.0 / (double)(cfg->image_height - 1);  //  strict build's 0 / 0 — the reference divides by zero there too)
  P.max_child_rays = cfg->max_child_rays;
  P.rank = cfg->rank;
  P.nranks = cfg->nranks;
  P.tile_rows = cfg->tile_rows;
  P.local_rows = rows;
  P.seed_lo = (uint32_t)cfg->seed;
  P.seed_hi = (uint32_t)(cfg->seed >> 32);
  P.n_items = (uint32_t)n_items;
  P.n_lanes = (uint32_t)n_lanes;
  P.div_npix = make_fastdiv((uint32_t)npix);
  P.div_w = make_fastdiv((uint32_t)cfg->image_width);
  P.div_tile = make_fastdiv((uint32_t)cfg->tile_rows);
  P.div_ns = make_fastdiv((uint32_t)streams_now);
  // tile the pixel order when the geometry allows it (a tile never straddles two strips)
  uint32_t th = 0, tw = 0;
  if (!c->knobs.no_tiles) tile_shape(cfg, rows, th, tw);
  P.tile_h_log2 = th;
  P.tile_w_log2 = tw;
  P.div_tpr_n = th ? (uint32_t)cfg->image_width >> tw : 1u;
  P.div_tpr = make_fastdiv(P.div_tpr_n);
  P.n_tile_rows = th ? (uint32_t)rows >> th : 1u;
  {
    const int eighths = c->knobs.sky_eighths;  // the top eighth of the image is traced last (RTOW_SKY_EIGHTHS: 0..8)
    P.sky_rows = th ? P.n_tile_rows * (uint32_t)eighths / 8u : 0u;
  }
  P.partials = (double *)c->partials.p;
  P.stack = (uint32_t *)c->stack.p;
  P.spill = (uint32_t *)c->spill.p;
  // scene-class specialisation of the GRID kernel (rtow_device.h): static spheres with 48-byte fat lists, or static +
  // moving spheres with 80-byte ones, image in LDS, binary64 builds
  P.spec = rtow::kSpecGeneric;
  if (kernel == RTOW_KERNEL_GRID && lds_bytes > 0 && !f32 && !c->knobs.no_spec && scene.n_tri == 0) {
    if (scene.n_mov == 0 && c->grid_fat_stride == 48u) P.spec = rtow::kSpecStaticSpheres;
    if (scene.n_mov > 0 && c->grid_fat_stride == 80u) P.spec = rtow::kSpecMovingSpheres;
  }
  P.b4_trips = c->knobs.bvh4_sm ? 0u : 1u;
  {
    const bool b4 = kernel == RTOW_KERNEL_BVH4;
    // every node in LDS (a small mesh, whether or not its triangles are staged too): a walk may stop once 24
    // lanes are left in it; nodes read from L2 (big mesh): 32 — the longer step hides more of the latency
    // (suzanne 16 / 20 / 24 / 28 / 32 / 40: 3.65 / 3.82 / 3.88 / 3.86 / 3.79 / 3.63 Gsamples/s; 96.8k mesh
    // 1.64 / 1.81 / 1.89 / 1.94 / 1.97 / 1.95)
    const bool b4_nodes_resident = b4 && scene.b4_lds_limit >= scene.b4_off_tri;
    int cap = c->knobs.walk_cap, open = c->knobs.walk_max_open;
    if (cap < 0) {
      cap = b4 ? 4 : 3;
      open = b4 ? (b4_nodes_resident ? 24 : 32) : 16;
    }
    // GRID: a resumed lane re-enters one cell BEHIND the cell it stopped at (rtow_trace_grid.h), and up to
    // two cell crossings can share one ray parameter (a ray through a cell corner), so fewer than 3 steps
    // per trip would not guarantee progress (cap 1 is a livelock).  BVH4 resumes exactly where it stopped.
    if (cap > 0 && !b4) cap = std::max(cap, 3);
    P.walk_cap = cap > 0 ? (uint32_t)cap : 0xffffffffu;
    P.walk_max_open = (uint32_t)open;
    P.leaf_votes = (uint32_t)(c->knobs.leaf_votes > 0 ? c->knobs.leaf_votes : (b4 ? 28 : 16));
  }
  P.fetch_votes = (uint32_t)(c->knobs.fetch_votes > 0 ? c->knobs.fetch_votes : (kernel == RTOW_KERNEL_BVH4 ? 2 : 4));
  P.sm4_restart = (uint32_t)c->knobs.sm4_votes[0];
  P.sm4_scatter = (uint32_t)c->knobs.sm4_votes[1];
  P.sm4_leaf = (uint32_t)c->knobs.sm4_votes[2];
  P.counters = (unsigned long long *)c->counters.p;
  P.t_origin = P.counters + 16;

  if (st != nullptr) HIPCHK(hipStreamWaitEvent(st, c->upload_ev, 0));  // the scene upload was queued on the null stream
  if (stats) HIPCHK(hipEventRecord(c->call_ev[0], st));
  if (!c->counters_init.p) {  // zeros, except the two minima of the diagnostic build ([5] min end, [16] t_origin)
    std::vector<unsigned long long> init(48, 0ull);
    init[5] = init[16] = ~0ull;
    if ((rc = c->counters_init.ensure(48 * sizeof(unsigned long long)))) return rc;
    HIPCHK(hipMemcpy(c->counters_init.p, init.data(), 48 * sizeof(unsigned long long), hipMemcpyHostToDevice));
  }
  HIPCHK(hipMemcpyAsync(c->counters.p, c->counters_init.p, 48 * sizeof(unsigned long long), hipMemcpyDeviceToDevice, st));
  const int slot = c->ev_count < kEventRing ? c->ev_count : -1;
  if (slot >= 0) HIPCHK(hipEventRecord(c->ev[slot][0], st));
  int launch_kernel = kernel;
  if (image_kernel && lds_bytes > 0 && c->knobs.stamps) launch_kernel = kernel + 16;  // diagnostic
  int lrc = strict ? rtow::launch_trace_strict(P, launch_kernel, (int)grid, block, lds_bytes, st)
            : f32  ? rtow::launch_trace_f32(P, launch_kernel, (int)grid, block, lds_bytes, st)
                   : rtow::launch_trace_fast(P, launch_kernel, (int)grid, block, lds_bytes, st);
#ifdef RTOW_TAILSTAT
  if (const char *path = std::getenv("RTOW_TAILSTAT_OUT")) {
    HIPCHK(hipStreamSynchronize(st));
    std::vector<unsigned long long> ts((size_t)n_lanes / 64 * 8);
    HIPCHK(hipMemcpy(ts.data(), c->spill.p, ts.size() * 8, hipMemcpyDeviceToHost));
    if (FILE *f = std::fopen(path, "wb")) {
      std::fwrite(ts.data(), 8, ts.size(), f);
      std::fclose(f);
    }
  }
#endif
  if (lrc != 0) return fail(RTOW_EHIP, "trace kernel launch failed: %s", hipGetErrorString((hipError_t)lrc));
  if (slot >= 0) {
    HIPCHK(hipEventRecord(c->ev[slot][1], st));
    c->ev_count++;
  }
  rtow::ReduceParams R;
  R.partials = P.partials;
  R.out = (double *)d_rgb_sums;
  R.npix3 = (uint32_t)(npix * 3);
  R.nstreams = streams_now;
  R.accumulate = accumulate;
  R.W = (uint32_t)cfg->image_width;
  R.tile_w_log2 = tw;
  R.tile_h_log2 = th;
  R.tiles_per_row = P.div_tpr_n;
  R.rgb8 = nullptr;
  R.spp = 0.0;
  if (c->fuse_rgb8 && lvl_first == 0 && lvl_count == plan.count && !accumulate) {  // the whole render in this launch
    R.rgb8 = c->fuse_rgb8;
    R.spp = c->fuse_spp;
    c->fuse_used = true;
  }
  lrc = rtow::launch_reduce(R, st);
  if (lrc != 0) return fail(RTOW_EHIP, "reduce kernel launch failed: %s", hipGetErrorString((hipError_t)lrc));

  if (stats) {
    HIPCHK(hipEventRecord(c->call_ev[1], st));
    HIPCHK(hipMemcpyAsync(c->h_counters, c->counters.p, 48 * sizeof(unsigned long long),
                          hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (c->h_counters[47] != 0ull) {  // the structural bound of the end-of-launch protocol fired (never observed)
      (void)check_dropped(c, false);    // (clears the sticky word: this call reports the error)
      return fail(RTOW_EHIP, "trace kernel: end-of-launch bound reached, %llu lanes gave up their samples",
                  c->h_counters[47]);
    }
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, c->call_ev[0], c->call_ev[1]));
    stats->total_ms = ms;
    if (slot >= 0) {
      HIPCHK(hipEventElapsedTime(&ms, c->ev[slot][0], c->ev[slot][1]));
      stats->kernel_ms = ms;
    }
    stats->samples = npix * ((unsigned long long)spt * (unsigned long long)(streams_now - 1) + (unsigned long long)spt_last);
    stats->segments = c->h_counters[1];
    if (kernel == RTOW_KERNEL_BRUTE) {
      stats->prim_tests = stats->segments * (unsigned long long)c->n_prims;
      stats->node_tests = 0;
    } else {
      stats->prim_tests = c->h_counters[2];
      stats->node_tests = c->h_counters[3];
    }
  }
  return RTOW_OK;
}

// Diagnostic: copy a resident scene image to the host (0 BVH, 1 grid, 2 BVH of the f32 build,
// 3 grid of the f32 build).  Used by the tests to compare host- and device-built images.
int rtow_debug_image(rtow_ctx *c, int32_t which, void *out, int64_t capacity, int64_t *size_out) {
  if (!c || !size_out) return fail(RTOW_EINVAL, "NULL argument");
  if (!c->have_scene) return fail(RTOW_ENOSCENE, "no scene uploaded");
  const void *src = nullptr;
  int64_t bytes = 0;
  switch (which) {
    case 0: src = c->blob.p; bytes = c->ds.blob_bytes; break;
    case 1: src = c->gblob.p; bytes = c->ds.gblob_bytes; break;
    case 2: src = c->blob32.p; bytes = c->ds32.blob_bytes; break;
    case 3: src = c->gblob32.p; bytes = c->ds32.gblob_bytes; break;
    default: return fail(RTOW_EINVAL, "unknown image %d", which);
  }
  *size_out = bytes;
  if (!out) return RTOW_OK;  // size query
  if (capacity < bytes) return fail(RTOW_EINVAL, "buffer too small (%lld < %lld)", (long long)capacity, (long long)bytes);
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipDeviceSynchronize());
  if (bytes) HIPCHK(hipMemcpy(out, src, (size_t)bytes, hipMemcpyDeviceToHost));
  return RTOW_OK;
}

// Diagnostic: the 16 device counters of the last launch (work queue, segments, prim
// tests, node tests, ... region cycle sums of the RTOW_STAMPS build at [8..12]).
int rtow_debug_counters(rtow_ctx *c, unsigned long long *out48) {
  if (!c || !out48) return fail(RTOW_EINVAL, "NULL argument");
  if (!c->counters.p) return fail(RTOW_ENOSCENE, "no launch yet");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out48, c->counters.p, 48 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return RTOW_OK;
}

// write_color as a device epilogue: 8-bit RGB from the radiance sums.
int rtow_tonemap_device(rtow_ctx *c, const void *d_rgb_sums, int64_t n_values, int32_t spp_effective,
                        void *d_rgb8, void *hip_stream) {
  if (!c) return fail(RTOW_EINVAL, "ctx is NULL");
  if (!d_rgb_sums || !d_rgb8) return fail(RTOW_EINVAL, "NULL device pointer");
  if (n_values < 0 || n_values > 0xffffffffLL) return fail(RTOW_EINVAL, "n_values out of range");
  if (spp_effective <= 0) return fail(RTOW_EINVAL, "spp_effective must be positive");
  HIPCHK(hipSetDevice(c->device));
  int lrc = rtow::launch_tonemap((const double *)d_rgb_sums, (unsigned char *)d_rgb8, (uint32_t)n_values,
                                 (double)spp_effective, hip_stream);
  if (lrc != 0) return fail(RTOW_EHIP, "tonemap launch failed: %s", hipGetErrorString((hipError_t)lrc));
  return RTOW_OK;
}

int rtow_profile_collect(rtow_ctx *c, double *kernel_ms_sum, int32_t *launches) {
  if (!c) return fail(RTOW_EINVAL, "ctx is NULL");
  HIPCHK(hipSetDevice(c->device));
  double sum = 0.0;
  for (int i = 0; i < c->ev_count; ++i) {
    HIPCHK(hipEventSynchronize(c->ev[i][1]));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, c->ev[i][0], c->ev[i][1]));
    sum += ms;
  }
  if (kernel_ms_sum) *kernel_ms_sum = sum;
  if (launches) *launches = c->ev_count;
  const bool waited = c->ev_count > 0;
  c->ev_count = 0;
  // the asynchronous entry point (rtow_render_device without stats) reports dropped samples here: every launch
  // of the ring has completed
  return waited ? check_dropped(c, false) : RTOW_OK;
}

// Upload for ONE render whose config is known: only what its kernel reads (the cover scene through the grid
// kernel needs no BVH image, no 4-wide image and no binary32 images: 0.48 -> 0.15 ms per call).
static int upload_for(rtow_ctx *c, const rtow_scene_t *scene, const rtow_config_t *cfg) {
  if (!c) return fail(RTOW_EINVAL, "ctx is NULL");
  int rc = validate_scene(scene);
  if (rc) return rc;
  const long long np = (long long)scene->n_spheres + scene->n_moving + scene->n_triangles;
  unsigned need;
  const int k = cfg->kernel;
  const bool mesh = scene->n_spheres == 0 && scene->n_moving == 0 && scene->n_triangles > 0 && !c->knobs.no_bvh4;
  if (cfg->precision == RTOW_F32)
    need = kNeedAll;
  else if (k == RTOW_KERNEL_REFTREE || k == RTOW_KERNEL_BRUTE || (k == RTOW_KERNEL_AUTO && np <= 16))
    need = 0u;
  else if ((k == RTOW_KERNEL_AUTO || k == RTOW_KERNEL_GRID) && scene->n_triangles == 0)
    need = kNeedGrid;
  else if (k == RTOW_KERNEL_GRID)
    need = kNeedGrid | kNeedBvh;
  else if (mesh && (k == RTOW_KERNEL_AUTO || k == RTOW_KERNEL_BVH4))
    need = kNeedBvh4;  // the 4-wide image only (the binary image is another 20 ms and 10 MB for the 96.8k-triangle mesh)
  else
    need = kNeedBvh;
  if (c->builder_req == RTOW_BUILDER_AUTO) {
    // AUTO (include/rtow.h): the device builder where it delivers the frame sooner.  Its tree is the host's (binned SAH,
    // csrc/rtow_build.hip pass 3c: the same node and triangle tests per segment), so what decides is the build: a fixed
    // ~1.7 ms of launches plus ~0.055 us per triangle on the device, ~0.15 us per triangle on the host's 16 threads
    // (96,800 triangles: 7 against 12-16 ms) — the device from about 16,000 triangles on, at any sample count.
    const bool device = mesh && need == kNeedBvh4 && scene->n_triangles >= 16384;
    c->builder = device ? RTOW_BUILDER_DEVICE_LBVH : RTOW_BUILDER_HOST_SAH;
  }
  rc = scene_upload(c, scene, need);
  if (rc == RTOW_OK && (need & kNeedGrid) && !(need & kNeedBvh) && !c->have_grid)
    rc = scene_upload(c, scene, need | kNeedBvh);  // the scene does not suit a grid: the walk falls back to the BVH
  if (rc == RTOW_OK && need == kNeedBvh4 && !c->have_bvh4)
    rc = scene_upload(c, scene, kNeedBvh);  // beyond the 4-wide format's limits: the binary walk takes the mesh
  return rc;
}

static int impl_render(rtow_ctx *c, const rtow_scene_t *scene, const rtow_config_t *cfg, double *rgb_sums_host,
                rtow_stats_t *stats) {
  if (!c) return fail(RTOW_EINVAL, "ctx is NULL");
  if (!rgb_sums_host) return fail(RTOW_EINVAL, "rgb_sums_host is NULL");
  int rc = validate_cfg(cfg);
  if (rc) return rc;
  if ((rc = upload_for(c, scene, cfg))) return rc;
  const int rows = rtow_local_rows(cfg);
  const size_t bytes = (size_t)rows * cfg->image_width * 3 * sizeof(double);
  if (bytes == 0) {
    if (stats) {
      std::memset(stats, 0, sizeof *stats);
    }
    return RTOW_OK;
  }
  if ((rc = c->out.ensure(bytes))) return rc;  // kept across calls
  void *d_out = c->out.p;
  if (cfg->accumulate)  // continue from the caller's sums
    HIPCHK(hipMemcpy(d_out, rgb_sums_host, bytes, hipMemcpyHostToDevice));
  rc = rtow_render_device(c, cfg, d_out, nullptr, stats);  // (no stats: no host wait before the copy below)
  if (rc != RTOW_OK) return rc;
  if ((rc = mirror_dropped(c, nullptr))) return rc;
  HIPCHK(hipMemcpy(rgb_sums_host, d_out, bytes, hipMemcpyDeviceToHost));
  return check_dropped(c, true);  // never RTOW_OK with samples dropped, with or without `stats`
}

// This rank's rows as 8-bit RGB on the DEVICE: render + write_color (fused into the reduce kernel when the render is
// one launch), asynchronous on `hip_stream` like rtow_render_device.  The f64 sums live in the context's workspace.
static int impl_render_device_rgb8(rtow_ctx *c, const rtow_config_t *cfg, void *d_rgb8, void *hip_stream,
                                   rtow_stats_t *stats) {
  if (!c) return fail(RTOW_EINVAL, "ctx is NULL");
  if (!d_rgb8) return fail(RTOW_EINVAL, "d_rgb8 is NULL");
  int rc = validate_cfg(cfg);
  if (rc) return rc;
  const int spp_eff = cfg->samples_per_pixel / cfg->nstreams * cfg->nstreams;  // src/render.cpp:185
  if (spp_eff <= 0) return fail(RTOW_EINVAL, "no effective samples (samples_per_pixel < nstreams)");
  if (cfg->accumulate) return fail(RTOW_EINVAL, "the rgb8 entry points own their sums: accumulate must be 0");
  const int rows = rtow_local_rows(cfg);
  const size_t n = (size_t)rows * cfg->image_width * 3;
  if (n == 0) {
    if (stats) std::memset(stats, 0, sizeof *stats);
    return RTOW_OK;
  }
  HIPCHK(hipSetDevice(c->device));
  if ((rc = c->out.ensure(n * sizeof(double)))) return rc;  // kept across calls
  c->fuse_rgb8 = (unsigned char *)d_rgb8;  // write_color inside the reduce kernel when the render is one launch
  c->fuse_spp = (double)spp_eff;
  c->fuse_used = false;
  rc = impl_render_device(c, cfg, c->out.p, hip_stream, stats);
  c->fuse_rgb8 = nullptr;
  if (rc == RTOW_OK && !c->fuse_used) rc = rtow_tonemap_device(c, c->out.p, (int64_t)n, spp_eff, d_rgb8, hip_stream);
  return rc;
}

// upload + render + device write_color + copy 8-bit RGB to the host: the whole output path of
// the reference's render() with 3 bytes per pixel over PCIe instead of 24.
static int impl_render_rgb8(rtow_ctx *c, const rtow_scene_t *scene, const rtow_config_t *cfg, unsigned char *rgb8_host,
                     rtow_stats_t *stats) {
  if (!c) return fail(RTOW_EINVAL, "ctx is NULL");
  if (!rgb8_host) return fail(RTOW_EINVAL, "rgb8_host is NULL");
  int rc = validate_cfg(cfg);
  if (rc) return rc;
  if (cfg->samples_per_pixel / cfg->nstreams <= 0) return fail(RTOW_EINVAL, "no effective samples (samples_per_pixel < nstreams)");
  if (cfg->accumulate) return fail(RTOW_EINVAL, "rtow_render_rgb8 owns its sums: accumulate must be 0");
  if ((rc = upload_for(c, scene, cfg))) return rc;
  const int rows = rtow_local_rows(cfg);
  const size_t n = (size_t)rows * cfg->image_width * 3;
  if (n == 0) {
    if (stats) std::memset(stats, 0, sizeof *stats);
    return RTOW_OK;
  }
  if ((rc = c->out8.ensure(n))) return rc;  // kept across calls
  if ((rc = impl_render_device_rgb8(c, cfg, c->out8.p, nullptr, stats))) return rc;
  // (a pinned landing buffer + host memcpy measured slower than the runtime's own staged copy for these 2.9 MB:
  // 8.67 vs 8.55 ms per call)
  if ((rc = mirror_dropped(c, nullptr))) return rc;
  HIPCHK(hipMemcpy(rgb8_host, c->out8.p, n, hipMemcpyDeviceToHost));
  return check_dropped(c, true);  // never RTOW_OK with samples dropped, with or without `stats`
}

// ---- the guarded entry points (see guarded() above) ----
int rtow_ctx_create(int device_id, rtow_ctx **out) {
  return guarded("rtow_ctx_create", [&] { return impl_ctx_create(device_id, out); });
}
int rtow_scene_upload(rtow_ctx *c, const rtow_scene_t *s) {
  return guarded("rtow_scene_upload", [&] { return impl_scene_upload(c, s); });
}
int rtow_render_device(rtow_ctx *c, const rtow_config_t *cfg, void *d_rgb_sums, void *hip_stream, rtow_stats_t *stats) {
  return guarded("rtow_render_device", [&] { return impl_render_device(c, cfg, d_rgb_sums, hip_stream, stats); });
}
int rtow_render(rtow_ctx *c, const rtow_scene_t *scene, const rtow_config_t *cfg, double *rgb_sums_host,
                rtow_stats_t *stats) {
  return guarded("rtow_render", [&] { return impl_render(c, scene, cfg, rgb_sums_host, stats); });
}
int rtow_render_rgb8(rtow_ctx *c, const rtow_scene_t *scene, const rtow_config_t *cfg, unsigned char *rgb8_host,
                     rtow_stats_t *stats) {
  return guarded("rtow_render_rgb8", [&] { return impl_render_rgb8(c, scene, cfg, rgb8_host, stats); });
}
int rtow_render_device_rgb8(rtow_ctx *c, const rtow_config_t *cfg, void *d_rgb8, void *hip_stream, rtow_stats_t *stats) {
  return guarded("rtow_render_device_rgb8", [&] { return impl_render_device_rgb8(c, cfg, d_rgb8, hip_stream, stats); });
}
int rtow_debug_schedule(rtow_ctx *c, const rtow_config_t *cfg, uint32_t *out, int32_t capacity_pairs) {
  return guarded("rtow_debug_schedule", [&] { return impl_debug_schedule(c, cfg, out, capacity_pairs); });
}

}  // extern "C"

namespace rtow {
// for rtow_multi.cpp: the sticky dropped-samples word of a context, read after its stream has been waited for
int ctx_check_dropped(rtow_ctx *c) { return c ? check_dropped(c, false) : RTOW_OK; }
int ctx_mirror_dropped(rtow_ctx *c, void *stream) { return c ? mirror_dropped(c, (hipStream_t)stream) : RTOW_OK; }
int ctx_check_dropped_mirrored(rtow_ctx *c) {
  if (!c) return RTOW_OK;
  if (*c->h_dropped != 0ull) HIPCHK(hipSetDevice(c->device));  // (the error path resets the device word)
  return check_dropped(c, true);
}
}  // namespace rtow

