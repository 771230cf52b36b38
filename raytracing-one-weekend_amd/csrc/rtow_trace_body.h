// rtow_trace_body.h — the trace kernels (included by rtow_trace_strict.hip and
// rtow_trace_fast.hip, which differ only in -ffp-contract and in RTOW_SUFFIX).
//
// What it computes: the per-pixel sample loop of the reference's render()
// (src/render.cpp:150-167) with ray_color()'s recursion (src/render.cpp:112-129)
// flattened into an iterative per-lane loop.
//
// Execution model (gfx950, wave64):
//   * persistent lanes.  A work item is (stream k, pixel): spt consecutive samples
//     of one pixel, summed in sample order exactly like one reference "thread"
//     (src/render.cpp:151-166).  A lane that finishes an item takes the next one
//     from a global counter; the fetch is wave-aggregated (one atomic per wave per
//     fetch round: __ballot + popcount ranks), so lanes stay dense until the
//     queue is empty, and the image does not depend on which lane traced what.
//   * one trip of the main loop = item bookkeeping, the NEW-RAY stage (camera rays of new samples and the
//     scattered rays of last trip's hits, made together so that both share the Philox block evaluations), the
//     closest-hit walk, and the end of the paths that escaped (sky colour) or ran out of bounces (black).
//     Every lane with a ray advances by one segment per trip, so the walk runs on a full wave — unless its walk
//     was stopped (most lanes done, a few not: GRID and BVH4 walks are resumable) and continues next trip.
//   * closest hit, four strategies with identical results:
//       STREAM — every lane tests every primitive.  The primitive index is
//         wave-uniform, so each record is fetched with ONE scalar load into SGPRs and
//         used directly as a VALU operand: the scene costs no VGPRs, no LDS traffic.
//       BVH — every lane walks a threaded (stackless, skip-link) BVH whose nodes and
//         primitive records sit in LDS (one scene image per workgroup, staged with
//         coalesced 16-byte loads).  Node boxes are f32 and padded, and the slab test
//         is slackened, so culling is conservative; leaf primitives are tested with
//         the same f64 code as STREAM, so the accepted (t, primitive) is the same.
//         Leaves found during the walk are queued per lane and tested in a separate
//         phase, so the box loop and the primitive loop are each SIMT-dense.
//       GRID — 3D-DDA over a uniform grid of the small primitives + an always-test list of the large
//         ones (rtow_trace_grid.h); BVH4 — 4-wide BVH with a per-lane LDS stack for triangle meshes
//         (rtow_trace_bvh4.h; rtow_trace_sm4.h is its state-machine form).
//   * radiance.  The reference multiplies attenuations on the way back up the
//     recursion, a1*(a2*(...*(an*sky))).  The STRICT build reproduces that order bit for bit:
//     the lane records the material index of every bounce in a per-lane path stack in HBM
//     ([bounce][lane], coalesced) and folds it from the end when the path escapes to the sky.
//     The FAST builds multiply forward — ((a1*a2)*...*an)*sky, a running product in registers,
//     equal to a few ulps — and need neither the stack nor the unwind loop (+6.2 %).  A path
//     that ends black contributes an exact zero.
//   * RNG: Philox4x32-7 in REQUESTS, one block each, counter (request, sample, pixel,
//     0), key = seed.  Request 0 of a sample carries the pixel jitter and the shutter time (21 bits each) and the two
//     32-bit uniforms of the lens point; request 1 + b everything bounce b draws: the five uniforms of its point in the
//     unit ball and the dielectric coin.  EXACTLY one block per request: the reference's rejection loops
//     (src/random-utils.cpp:23-41) are replaced by direct samplers of the same distributions (rtow_trace_rng.h, round 5)
//     — a wave used to run them as long as its unluckiest lane, 2.35 further block evaluations per trip for 7 lanes each.
//
// In the strict build (-ffp-contract=off) every expression below has the operand
// order of the reference expression it restates, f64 sqrt and division are the
// correctly rounded IEEE forms, and the result is bit-identical to oracle/.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "rtow_device.h"

#ifndef RTOW_SUFFIX
#error "define RTOW_SUFFIX"
#endif
#define RTOW_CAT2(a, b) a##b
#define RTOW_CAT(a, b) RTOW_CAT2(a, b)

namespace rtow {
namespace {
#include "rtow_trace_math.h"
#include "rtow_trace_rng.h"
#include "rtow_trace_hit.h"
#include "rtow_trace_stamps.h"
#include "rtow_trace_bvh.h"
#include "rtow_trace_grid.h"
#include "rtow_trace_bvh4.h"
#ifndef RTOW_FAST_MATH
#include "rtow_trace_reftree.h"
#endif

// --------------------------------------------------------------- the kernel ---
// n / d for a divisor fixed per launch: q = (((n - t) >> 1) + t) >> shift, t = mulhi(n, magic)
// (round-up method, exact for every 32-bit n; magic/shift come from the host).
__device__ __forceinline__ uint32_t fastdiv(uint32_t n, FastDiv f) {
  const uint32_t t = __umulhi(n, f.magic);
  return f.shift == 255u ? n : (((n - t) >> 1) + t) >> f.shift;  // shift 255: divisor 1
}

constexpr uint32_t kItemBatch = 64;  // work items fetched per global atomic (per wave)

__device__ __forceinline__ unsigned lane_id() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// cross-lane reads (ds_bpermute: lane i receives the value of lane src_i; all lanes active)
__device__ __forceinline__ uint32_t lane_read(uint32_t src, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)v);
}
__device__ __forceinline__ double lane_read(uint32_t src, double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const uint32_t lo = lane_read(src, (uint32_t)b), hi = lane_read(src, (uint32_t)(b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double wave_bcast_f64(double v, int src_lane) {  // src_lane: wave-uniform (v_readlane)
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, src_lane);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), src_lane);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ uint32_t lanes_below(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// ---- work items ------------------------------------------------------------------------------
// A wave's share of the global item queue: one atomic buys a batch, lanes take items from the
// wave-local pool [next, end) by ballot rank.  All fields are wave-uniform.
struct ItemPool {
  uint32_t next = 0, end = 0;
  unsigned long long seen = 0ull;  // queue head as of this wave's last fetch
};

// Hands one queue position to every lane of `need_mask` (all lanes of the wave call this together).
// Returns the lane's position, >= kp->n_items when the queue is exhausted.
__device__ __forceinline__ unsigned long long take_items(ItemPool &pool, unsigned long long need_mask, unsigned lane,
                                                         uint32_t n_waves, const RTOW_CONST TraceParams *kp,
                                                         unsigned long long *counters) {
  // Wave-local pool [next, end): one global atomic buys kItemBatch items, which the lanes then take by
  // ballot rank with no further traffic (a single hot counter word saturates near 90 dequeues/us on this
  // chip — one atomic per wave trip was the bottleneck).  All of this is wave-uniform except `mine`.
  const uint32_t want = (uint32_t)__popcll(need_mask);
  const uint32_t avail = pool.end - pool.next;
  const uint32_t rank = (uint32_t)__popcll(need_mask & ((1ull << lane) - 1ull));
  unsigned long long mine = (unsigned long long)pool.next + rank;
  if (want > avail) {
    // guided self-scheduling: 64 items per atomic while the queue is long, shrinking to
    // exactly what this wave needs now as it drains (a wave that hoards items at the end
    // of the queue keeps the whole launch waiting: measured ~4 item durations per launch)
    const unsigned long long left =
        (unsigned long long)kp->n_items > pool.seen ? (unsigned long long)kp->n_items - pool.seen : 0ull;
#ifdef RTOW_NO_SHRINK  // (experiment: a batch is always one tile)
    uint32_t batch = kItemBatch;
    (void)left;
#else
    uint32_t batch = (uint32_t)(left / ((unsigned long long)n_waves * 4ull));
    batch = batch > kItemBatch ? kItemBatch : batch;
    batch = batch < want - avail ? want - avail : batch;
#endif
    const int leader = __ffsll((long long)need_mask) - 1;
    unsigned long long base = pool.seen;
    // a wave that has seen the end of the queue stops polling it: at the end of a launch every
    // wave asks every trip, and the one counter word serves ~100 requests/us (measured: the
    // last trips of a launch took 34 us instead of 13)
    if (pool.seen < (unsigned long long)kp->n_items) {
      if ((int)lane == leader) base = atomicAdd(&counters[0], (unsigned long long)batch);
      base = __shfl(base, leader);
    }
    pool.seen = base + batch;  // how far the queue had advanced when this wave last looked
    if (rank >= avail) mine = base + (rank - avail);
    const unsigned long long nn = base + (want - avail), ne = base + batch;
    const unsigned long long cap = (unsigned long long)kp->n_items;
    pool.next = (uint32_t)(nn < cap ? nn : cap);
    pool.end = (uint32_t)(ne < cap ? ne : cap);
  } else {
    pool.next += want;
  }
  return mine;
}

// Queue position -> (partial-sum slot, column, global row, first sample index).
// Queue order.  Tiled mode: tile-major, stream-minor — all streams of a 64-pixel tile are
// adjacent, tiles run top-to-bottom, and the queue is consumed from its far end, so a launch
// ENDS on the top rows of the image for every stream.  In the reference's scenes that is sky
// (the top 8 % of the cover image: one-segment paths), so most waves run out of work together:
// waves finishing > 0.2 ms after they find the queue empty fell from 51 % to 5 % (+1.9 %).
// (Stream-major order ended only the last stream on the sky; ending on the bottom rows —
// near ground, short paths — measures the same.)  The partial-sum slot stays [stream][pixel].
struct ItemPos {
  uint32_t item, j, gi, sample0;
  int32_t count;  // samples of the item (the launch's last level may be longer than the others)
};
__device__ __forceinline__ ItemPos decode_item(const RTOW_CONST TraceParams *kp, uint32_t mine, uint32_t npix_local) {
  ItemPos ip;
  const uint32_t qi = kp->n_items - 1u - mine;
  uint32_t k, lp, lr;
  if (kp->tile_h_log2 == 0u) {  // row-major, stream-major
    k = fastdiv(qi, FastDiv{kp->div_npix.magic, kp->div_npix.shift});
    lp = qi - k * npix_local;
    lr = fastdiv(lp, FastDiv{kp->div_w.magic, kp->div_w.shift});
    ip.j = lp - lr * (uint32_t)kp->W;
  } else {  // 64-pixel tiles: a wave's batch of 64 items is one compact tile of one stream
    const uint32_t g64 = qi >> 6, w = qi & 63u;
    const uint32_t t = fastdiv(g64, FastDiv{kp->div_ns.magic, kp->div_ns.shift});
    k = g64 - t * (uint32_t)kp->nstreams;
    const uint32_t trq = fastdiv(t, FastDiv{kp->div_tpr.magic, kp->div_tpr.shift});  // tile row by queue position
    const uint32_t tc = t - trq * (kp->div_tpr_n);
    // Tile rows are consumed from the highest position down.  Positions >= sky_rows hold the
    // rows below the top band, top-down (the horizon rows of an outdoor scene — its costliest —
    // go first); the top band (sky_rows tile rows, typically one-segment paths) comes last.
    const uint32_t tr = trq >= kp->sky_rows ? kp->sky_rows + (kp->n_tile_rows - 1u - trq) : trq;
    lp = ((tr * kp->div_tpr_n + tc) << 6) | w;
    lr = (tr << kp->tile_h_log2) + (w >> kp->tile_w_log2);
    ip.j = (tc << kp->tile_w_log2) + (w & ((1u << kp->tile_w_log2) - 1u));
  }
  ip.item = k * npix_local + lp;  // partial-sum slot
  // local row -> global row: this rank's q-th strip is global strip q*nranks+rank
  const uint32_t q = fastdiv(lr, FastDiv{kp->div_tile.magic, kp->div_tile.shift});
  const uint32_t rr = lr - q * (uint32_t)kp->tile_rows;
  ip.gi = (q * (uint32_t)kp->nranks + (uint32_t)kp->rank) * (uint32_t)kp->tile_rows + rr;
  ip.sample0 = kp->sample_base + k * (uint32_t)kp->spt;  // first sample index of this level
  ip.count = k + 1u == (uint32_t)kp->nstreams ? kp->spt_last : kp->spt;
  return ip;
}

// Camera::get_ray for sample (g.pixel, g.sample) of pixel (column j, global row gi): pixel jitter,
// lens point, ray (src/render.cpp:158-159, src/common-model.cpp:156-167), all from the sample's one block
// (rtow_trace_rng.h).  The stand-alone form of the new-ray stage's camera half, used by the state-machine kernel.
__device__ __forceinline__ void camera_ray(const TraceParams &P, Rng &g, uint32_t k0, uint32_t k1, uint32_t j,
                                           uint32_t gi, V3 &ro, V3 &rd, real &rtime) {
  g.r = 0u;
  const int from_top_i = P.H - (int)gi - 1;
  uint32_t o0, o1, o2, o3;
  philox4x32(g.r, g.sample, g.pixel, 0u, k0, k1, o0, o1, o2, o3);
  g.r += 1u;
  real ju, jv, jt, px, py;
  jitter_from_block(o0, o1, o2, ju, jv, jt);
  lens_from_block(o0, o1, o2, o3, px, py);
  const real u = fast_div((real)(int)j + ju, (real)(P.W - 1));
  const real v = fast_div((real)from_top_i + jv, (real)(P.H - 1));
  // camera block: wave-uniform scalar loads (origin u v horizontal vertical llc | lens t0 t1)
#ifdef RTOW_REAL_F32
  const RTOW_CONST float *cm = (const RTOW_CONST float *)P.cam32;
#else
  cdptr cm = (cdptr)(const double *)P.cam;
#endif
  const real lens = cm[18], ct0 = cm[19], ct1 = cm[20];
  const real rdx = lens * px, rdy = lens * py;
  const V3 offset = V3{cm[3], cm[4], cm[5]} * rdx + V3{cm[6], cm[7], cm[8]} * rdy;
  const V3 from = V3{cm[0], cm[1], cm[2]} + offset;
  rd = V3{cm[15], cm[16], cm[17]} + u * V3{cm[9], cm[10], cm[11]} + v * V3{cm[12], cm[13], cm[14]} - from;
  ro = from;
  rtime = jt * (ct1 - ct0) + ct0;
}

// Material::scatter (src/common-model.cpp:13-62) for a hit with shading normal `normal`: the new
// direction, or absorbed — point of the unit ball and dielectric coin from the bounce's one block (rtow_trace_rng.h).
// The stand-alone form of the new-ray stage's scatter half, used by the state-machine kernel.
__device__ __forceinline__ bool scatter_dir(Rng &g, uint32_t k0, uint32_t k1, int kind, real m_fuzz, real m_ir, V3 rd,
                                            V3 normal, bool front, V3 &dir) {
  uint32_t o0, o1, o2, o3;
  philox4x32(g.r, g.sample, g.pixel, 0u, k0, k1, o0, o1, o2, o3);
  g.r += 1u;
  V3 dirbase = {0, 0, 0};
  if (kind == 2) {
    const real ir = m_ir;
    const V3 unit = normalize(rd);
    const real cos_theta = dot(-unit, normal);
    const real sin_theta = fast_sqrt(real(1.0) - cos_theta * cos_theta);
    const real ratio = front ? fast_rcp(ir) : ir;
    bool refl = ratio * sin_theta > real(1.0);
    if (!refl) {
      real r0 = fast_div(real(1.0) - ratio, real(1.0) + ratio);
      r0 = r0 * r0;
      const real x = real(1.0) - cos_theta;
      const real x2 = x * x;
      const real R = r0 + (real(1.0) - r0) * (x2 * x2 * x);
      refl = R > coin_from_block(o0, o1, o3);
    }
    dirbase = refl ? reflect(unit, normal) : refract(unit, normal, ratio);
  } else if (kind == 1) {
    dirbase = reflect(rd, normal);
  }
  // random_unit_vector() (random-utils.cpp:31-33): the point of the unit ball's positive octant, un-normalised
  const V3 rnd = ball_from_block(o0, o1, o2, o3);
  bool absorbed = false;
  if (kind == 0) {
    absorbed = rabs(normal.x - rnd.x) < real(1e-8) && rabs(normal.y - rnd.y) < real(1e-8) &&
               rabs(normal.z - rnd.z) < real(1e-8);
    dir = normal + rnd;
  } else {
    dir = dirbase + m_fuzz * rnd;
  }
  return !absorbed;
}

// KERNEL: 1 = STREAM, 2 = BVH, 3 = GRID, 4 = BVH4;  LDS: scene image staged in LDS (2 and 3; the
// BVH4 kernel always uses LDS: the image or its top, and the traversal stack)
// SPEC: scene-class specialisation (GRID kernel only; rtow_device.h kSpec*)
template <int KERNEL, bool LDS, bool STAMPS = false, int SPEC = 0>
__global__ void __launch_bounds__(KERNEL >= 2 && KERNEL <= 4 ? 1024 : 256)
    RTOW_CAT(rtow_trace_, RTOW_SUFFIX)(const TraceParams P) {
  const DevScene &sc = P.sc;
  const uint32_t k0 = P.seed_lo, k1 = P.seed_hi;
  const unsigned lane = lane_id();
  [[maybe_unused]] const uint32_t lane_g = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t npix_local = (uint32_t)P.local_rows * (uint32_t)P.W;

  Image<LDS> im;
  im.g = KERNEL == 3 ? sc.gblob : sc.blob;
  [[maybe_unused]] Bvh4Reader<LDS> im4;  // LDS: the whole image is staged; otherwise the top of the tree
  if constexpr (KERNEL == 4) {
    im4.g = sc.blob4;
    im4.lds_limit = sc.b4_lds_limit;
    im4.aux_src = sc.b4_aux_src;
    im4.aux_lds = sc.b4_aux_lds;
    const uint4 *src = reinterpret_cast<const uint4 *>(sc.blob4);
    uint4 *dst = reinterpret_cast<uint4 *>(rtow_lds);
    const uint32_t n16 = sc.b4_lds_limit / 16u;
    for (uint32_t i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
    if (sc.b4_aux_src < sc.blob4_bytes) {  // the end of the image (materials, material indices): 16-byte aligned sections
      const uint4 *asrc = reinterpret_cast<const uint4 *>(sc.blob4 + sc.b4_aux_src);
      uint4 *adst = reinterpret_cast<uint4 *>(rtow_lds + sc.b4_aux_lds);
      const uint32_t a16 = (sc.blob4_bytes - sc.b4_aux_src) / 16u;
      for (uint32_t i = threadIdx.x; i < a16; i += blockDim.x) adst[i] = asrc[i];
    }
    __syncthreads();
  } else if constexpr (KERNEL >= 2 && LDS) {
    // stage the scene image: coalesced 16-byte loads, 16-byte LDS stores
    const uint4 *src = reinterpret_cast<const uint4 *>(im.g);
    uint4 *dst = reinterpret_cast<uint4 *>(rtow_lds);
    const uint32_t n16 = (KERNEL == 3 ? sc.gblob_bytes : sc.blob_bytes) / 16u;
    for (uint32_t i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
  }

  // per-lane state
  bool done = false;
  bool need_sample = true;
  bool pending_hit = false;   // the last segment ended in a hit that is scattered at the top of the next trip
  int s_left = 0;             // samples left in the current item
  uint32_t item = 0xffffffffu;
  uint32_t j = 0;             // column
  uint32_t gi = 0;            // global row (from the top)
  V3d acc = {0.0, 0.0, 0.0};  // pixel_color of this item (src/render.cpp:156), binary64 in every build
  V3 ro = {0, 0, 0}, rd = {0, 0, 1};
  real rtime = 0;
  int depth = 0;              // remaining child rays
  [[maybe_unused]] int nb = 0;  // bounces recorded on the path stack
#ifdef RTOW_FAST_MATH
  // fast builds: the product of the attenuations so far, multiplied forward — ((a1*a2)*...)*sky
  // instead of the reference's a1*(a2*(...*sky)), equal to a few ulps — so no path stack and no
  // unwind loop.  The strict build keeps the stack: it reproduces the reference's order bit for bit.
  V3 throughput = {1, 1, 1};
#endif
  Rng g = {0, 0, 0};
  uint32_t nseg = 0, nnode = 0, nprim = 0;
  // GRID kernel: a walk the wave stopped early continues in the next trip (rtow_trace_grid.h)
  [[maybe_unused]] float t_resume = 0.0f;
  [[maybe_unused]] uint32_t w_cur = 0x1fffffu, w_sa = 0u;  // BVH4: the same (kRefNone = no walk in progress)
  Closest best;
  best.t = 0;
  best.prim = -1;
  // end-of-launch sample donation (see "tail" below)
  bool helping = false;    // this lane traces a sample donated by another lane of the wave
  bool holding = false;    // ... has finished it and keeps its colour in `acc` until the owner adds it
  uint32_t partners = 0u;  // owner: stack of its helpers' lane ids (6 bits each, most recent lowest);
                           // helper: its owner's lane id
  int n_out = 0;           // owner: donated samples not yet added
  // end-of-launch protocol: trips this wave may still spend in it (structural bound, computed by the host).  A
  // count-down held in a register from the start: read at its point of use — a scalar load from the kernel-argument
  // segment and an lgkmcnt wait in EVERY trip of a draining wave — it cost +0.2 ms on the end of every launch.
  uint32_t tail_left = P.tail_bound;
  asm volatile("" : "+v"(tail_left));
  ItemPool pool;  // wave-uniform: this wave's batch of work items
  const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
#ifdef RTOW_TAILSTAT  // (experiment build: what a wave still holds when it first finds the queue empty, and how long that takes)
  unsigned long long ts_start = __builtin_amdgcn_s_memrealtime(), ts_empty = 0ull;
  uint32_t ts_trips = 0u, ts_trips_after = 0u, ts_live = 0u, ts_left = 0u, ts_pool = 0u;
#endif
  Stamps<STAMPS> stamps;
  stamps.start();
  unsigned long long t_empty = 0ull;  // diagnostic: when this wave first saw the queue empty
  unsigned long long trips_after_empty = 0ull;
  if constexpr (STAMPS) {
    if (lane == 0) atomicMin(P.t_origin, (unsigned long long)__builtin_amdgcn_s_memrealtime());
  }

  for (;;) {
    // ---- item bookkeeping ---------------------------------------------------
    bool need_item = false;
    if (!done && need_sample && s_left <= 0) {
      if (helping) {  // a donated sample is finished: keep its colour for the owner
        helping = false;
        holding = true;
        done = true;
      } else if (n_out == 0) {
        need_item = true;
      }  // else: an owner waiting for donated samples
    }
    unsigned long long need_mask = __ballot(need_item);
    // Lanes that need a new item wait until `fetch_votes` of them do (or nothing else is left to do): with
    // 64 desynchronised lanes some lane finishes an item in almost every trip, and the fetch block — queue
    // atomic, item decode, partial-sum store — would run every trip for two or three lanes.
    if (need_mask != 0ull && (uint32_t)__popcll(need_mask) < P.fetch_votes && __ballot(!done && !need_item) != 0ull) {
      need_mask = 0ull;
      need_item = false;
    }
    if (need_mask != 0ull) {
      // The item-decoding parameters are read here from the kernel-argument segment (scalar loads)
      // instead of living in SGPRs for the whole launch: the kernel is VALU-issue-bound and ran out
      // of SGPRs, so every one of them cost a v_readlane (VALU) per use.
      const RTOW_CONST TraceParams *kp = (const RTOW_CONST TraceParams *)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(kp));  // opaque per trip: keeps the loads from being hoisted out of the loop
      if (need_item && item != 0xffffffffu) {  // the finished item goes to its partial-sum slot
        double *dst = P.partials + (size_t)item * 3;
        dst[0] = acc.x;
        dst[1] = acc.y;
        dst[2] = acc.z;
        item = 0xffffffffu;
      }
      const unsigned long long mine = take_items(pool, need_mask, lane, n_waves, kp, P.counters);
      if (need_item) {
        if (mine >= (unsigned long long)kp->n_items) {
          done = true;
#ifdef RTOW_TAILSTAT
          if (ts_empty == 0ull) ts_empty = 1ull;  // marked; the wave-level snapshot is taken below
#endif
          if constexpr (STAMPS) {
            if (t_empty == 0ull) t_empty = __builtin_amdgcn_s_memrealtime();
          }
        } else {
          const ItemPos ip = decode_item(kp, (uint32_t)mine, npix_local);
          item = ip.item;
          j = ip.j;
          gi = ip.gi;
          g.pixel = gi * (uint32_t)kp->W + j;
          g.sample = ip.sample0;
          s_left = ip.count;
          acc = {0.0, 0.0, 0.0};
        }
      }
    }
#ifdef RTOW_TAILSTAT
    {
      const bool first = __ballot(ts_empty == 1ull) != 0ull && __ballot(ts_empty > 1ull) == 0ull;
      if (first) {  // wave-uniform: the first trip in which some lane found the queue empty
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        ts_live = (uint32_t)__popcll(__ballot(!done));
        uint32_t sl = (!done && s_left > 0) ? (uint32_t)s_left : 0u;
        for (int off = 32; off >= 1; off >>= 1) sl += __shfl_xor(sl, off);
        ts_left = sl;
        ts_pool = pool.end - pool.next;
        ts_empty = now < 2ull ? 2ull : now;
      } else if (__ballot(ts_empty > 1ull) != 0ull) {
        ts_empty = (unsigned long long)__shfl((long long)ts_empty, __ffsll((long long)__ballot(ts_empty > 1ull)) - 1);
      }
      ++ts_trips;
      if (__ballot(ts_empty > 1ull) != 0ull) ++ts_trips_after;
    }
#endif
    // ---- tail: sample donation ------------------------------------------------------------
    // Once the queue is empty a wave is as slow as its slowest lane's item (up to spt samples of
    // up to max_child_rays segments each) while its other lanes idle.  An idle lane therefore
    // takes over the LAST unstarted
    // sample of a lane that still has two or more to go.  The owner adds the donated colours
    // after its own samples, in sample order (most recent donation first), so the pixel sum is
    // the sequential one bit for bit.  Only lanes of the same wave trade (registers + ds_bpermute).
    // Measured: +0.9 % on C2.  What remains of the tail is one PATH: a trip takes ~13.7 us with four
    // waves per SIMD, so a 50-bounce path started just before the queue empties runs ~0.2-0.7 ms.
    if (__ballot(done) != 0ull) {
#ifdef RTOW_DONATE_R2  // (the round-2 form: at most 5 samples out per giver, lane ids on a 6-bit stack)
      // (1) owners that have finished their own samples take the next donated colour, if ready
      const bool ready = !done && need_sample && s_left <= 0 && n_out > 0;
      if (__ballot(ready) != 0ull) {
        const uint32_t h = partners & 63u;
        const uint32_t src = ready ? h : lane;
        // (not `ready && lane_read(..)`: short-circuit evaluation would run the cross-lane read with
        // only the ready lanes active, and an inactive source lane reads as 0)
        const uint32_t partner_holds = lane_read(src, (uint32_t)holding);
        const bool take = ready && partner_holds != 0u;
        const V3d c = {lane_read(src, acc.x), lane_read(src, acc.y), lane_read(src, acc.z)};
        // a holding helper asks its owner whether it was the one taken
        const uint32_t ow = holding ? partners : lane;
        const bool o_take = lane_read(ow, (uint32_t)take) != 0u;
        const uint32_t o_h = lane_read(ow, h);
        if (take) {
          acc = acc + c;
          partners >>= 6;
          --n_out;
        }
        if (holding && o_take && o_h == lane) holding = false;
      }
      const bool can_give = !done && !helping && s_left >= 2 && n_out < 5;
#else
      // (1) owners that have finished their own samples add the donated colours that are ready, in sample
      // order: the owner's next sample index is the lowest donated one; its holder is found by ballot and
      // read with v_readlane (round 3: no stack of helper ids in the owner, so every unstarted sample of an
      // item can be out at once — a costly item met at the end of a launch used to take two rounds of five)
      unsigned long long rm = __ballot(!done && need_sample && s_left <= 0 && n_out > 0);
      while (rm != 0ull) {
        const int o = __ffsll((long long)rm) - 1;  // (wave-uniform)
        rm &= rm - 1ull;
        const uint32_t want_sample = (uint32_t)__builtin_amdgcn_readlane((int)g.sample, o);
        const uint32_t want_pixel = (uint32_t)__builtin_amdgcn_readlane((int)g.pixel, o);
        const unsigned long long held =
            __ballot(holding && partners == (uint32_t)o && g.sample == want_sample + 1u && g.pixel == want_pixel);
        if (held != 0ull) {
          const int h = __ffsll((long long)held) - 1;
          const V3d c = {wave_bcast_f64(acc.x, h), wave_bcast_f64(acc.y, h), wave_bcast_f64(acc.z, h)};
          if ((int)lane == o) {
            acc = acc + c;
            ++g.sample;
            --n_out;
          }
          if ((int)lane == h) holding = false;
          if (__builtin_amdgcn_readlane(n_out, o) > 0) rm |= 1ull << o;  // its next colour may be ready as well
        }
      }
      const bool can_give = !done && !helping && s_left >= 2;
#endif
      // (2) idle lanes take the last unstarted sample of lanes with >= 2 samples to go
      const bool can_help = done && !holding;
      const unsigned long long gm = __ballot(can_give), hm = __ballot(can_help);
      if (gm != 0ull && hm != 0ull) {
        const uint32_t ng = (uint32_t)__popcll(gm), nh = (uint32_t)__popcll(hm);
        const uint32_t cnt = ng < nh ? ng : nh;
        const uint32_t grank = lanes_below(gm), hrank = lanes_below(hm);
        // compaction (a permutation of the lanes): lane k learns the k-th giver / k-th helper
        const uint32_t giver_k = (uint32_t)__builtin_amdgcn_ds_permute(
            (int)((can_give ? grank : ng + (lane - grank)) << 2), (int)lane);
        const bool gives = can_give && grank < cnt;
        const bool helps = can_help && hrank < cnt;
        const uint32_t my_giver = lane_read(helps ? hrank : lane, giver_k);
#ifdef RTOW_DONATE_R2
        const uint32_t helper_k = (uint32_t)__builtin_amdgcn_ds_permute(
            (int)((can_help ? hrank : nh + (lane - hrank)) << 2), (int)lane);
        const uint32_t my_helper = lane_read(gives ? grank : lane, helper_k);
#endif
        const uint32_t gsrc = helps ? my_giver : lane;
        const uint32_t o_j = lane_read(gsrc, j), o_gi = lane_read(gsrc, gi);
        const uint32_t o_pixel = lane_read(gsrc, g.pixel), o_sample = lane_read(gsrc, g.sample);
        const uint32_t o_left = lane_read(gsrc, (uint32_t)s_left);
        if (gives) {
          s_left -= 1;
#ifdef RTOW_DONATE_R2
          partners = (partners << 6) | my_helper;
#endif
          ++n_out;
        }
        if (helps) {
          j = o_j;
          gi = o_gi;
          g.pixel = o_pixel;
          g.sample = o_sample + o_left - 1u;  // the giver's last sample
          s_left = 1;
          need_sample = true;
          acc = {0.0, 0.0, 0.0};
          helping = true;
          done = false;
          partners = my_giver;
        }
      }
      // structural bound on the tail (every wait above ends when a bounded path ends; this makes
      // the exit independent of that argument): give up donating, never hang
      // (x64: a stopped-and-resumed walk spreads one segment over several trips)
      // (the bound — 64 x (4096 + 8 (max_child_rays + 2)(longest item + 1)) trips — comes from the host)
      if (tail_left == 0u) {
        if (!done || holding) {  // samples dropped: the host turns this into an error (the word is sticky across launches)
          const RTOW_CONST TraceParams *kpt = (const RTOW_CONST TraceParams *)__builtin_amdgcn_kernarg_segment_ptr();
          atomicAdd(&P.counters[47], 1ull);
          atomicAdd(kpt->dropped, 1ull);
        }
        done = true;
        helping = false;
        holding = false;
        n_out = 0;
      } else {
        --tail_left;
      }
    }
    if (__ballot(!done || holding) == 0ull) break;
    stamps.mark(RG_FETCH);

    // (a lane whose fresh item has no samples — spt == 0 — goes straight back for the next one)
    const bool live = !done && !(need_sample && s_left <= 0);

    // ---- new rays ------------------------------------------------------------------------------------
    // Two kinds of lanes need a new ray before the walk: those starting a sample (pixel jitter +
    // Camera::get_ray, src/render.cpp:158-159, src/common-model.cpp:156-167) and those whose previous
    // segment ended in a hit (Material::scatter, src/common-model.cpp:13-62).  Both draw exactly one Philox
    // block, and they draw it TOGETHER: one block evaluation serves both kinds of lanes, and so does the one
    // evaluation of what their two samplers share (sine, cosine, square root).  Every lane consumes exactly its own
    // requests in its own order (counter = (request, sample, pixel)), so the image does not depend on who shares a wave
    // with whom.  (Rounds 1-5a ran the reference's rejection loops here: ~6, later 3.4 block evaluations per trip.)
    const bool do_regen = live && need_sample;
    const bool do_scat = live && pending_hit;
    // (read only by the lanes of `do_scat`, here and in the two scatter blocks below: for every other lane the value
    // is unspecified — `anyv` — which spares the dozen v_mov a zero initialiser costs in every trip)
    V3 normal = anyv3();
    bool front = true;
    int mi = 0, kind = 0;
    real m_fuzz = anyv(real(0)), m_ir = anyv(real(0));
#ifdef RTOW_FAST_MATH
    V3 m_att = anyv3();
#endif
    if (do_scat) {
      // rebuild the Hit of the winner (src/common-model.cpp:83-90, :121).  The walking
      // kernels read the winner's record, its material index and the material from the
      // LDS scene image (a chain of three dependent loads: LDS latency, not L2's)
      // (the hit point replaces the ray origin IN PLACE: it is the origin of the scattered ray, and a separate
      // `where` copied into `ro` at the end of the stage cost the compiler a dozen v_mov_b64 per trip around the merge)
      ro = ro + rd * best.t;
      const int pid = best.prim;
      if constexpr (KERNEL == 4) {
        // triangles only: the un-normalised normal e1 x e2 of the record (src/common-model.cpp:121)
        const uint32_t r = sc.b4_off_tri + 96u * (uint32_t)pid;
        const vd2 q4 = im4.t2(r + 64u), q5 = im4.t2(r + 80u);
        normal = {(real)q4.y, (real)q5.x, (real)q5.y};
        mi = (int)im4.u32(sc.b4_off_pmat + 4u * (uint32_t)pid);
        const uint32_t mr = sc.b4_off_mats + 48u * (uint32_t)mi;
        const vd2 m1 = im4.d2(mr + 16u), m2 = im4.d2(mr + 32u);  // {att.z, fuzz}, {ir, kind|pad}
#ifdef RTOW_FAST_MATH
        const vd2 m0 = im4.d2(mr);  // {att.x, att.y}
        m_att = V3{(real)m0.x, (real)m0.y, (real)m1.x};
#endif
        m_fuzz = (real)m1.y;
        m_ir = (real)m2.x;
        kind = (int)(__double_as_longlong(m2.y) & 0xffffffffll);
      } else if constexpr (KERNEL >= 2 && KERNEL <= 4) {
        const uint32_t o_sph = KERNEL == 3 ? sc.g_off_sph : sc.off_sph;
        const uint32_t o_mov = KERNEL == 3 ? sc.g_off_mov : sc.off_mov;
        const uint32_t o_tri = KERNEL == 3 ? sc.g_off_tri : sc.off_tri;
        const uint32_t o_pmat = KERNEL == 3 ? sc.g_off_pmat : sc.off_pmat;
        const uint32_t o_mats = KERNEL == 3 ? sc.g_off_mats : sc.off_mats;
        if constexpr (SPEC == 1) __builtin_assume(pid < sc.n_sph);
        if constexpr (SPEC == 2) __builtin_assume(pid < sc.n_sph + sc.n_mov);
        if (pid < sc.n_sph + sc.n_mov) {
          V3 center;
          bool inward;  // negative radius: only the sign of the signed r*r is used here
          if (pid < sc.n_sph) {
            const double2 p0 = im.d2(o_sph + 32u * (uint32_t)pid), p1 = im.d2(o_sph + 32u * (uint32_t)pid + 16u);
            center = {(real)p0.x, (real)p0.y, (real)p1.x};
            inward = p1.y < 0.0;
          } else {
            const uint32_t r = o_mov + 64u * (uint32_t)(pid - sc.n_sph);
            const double2 p0 = im.d2(r), p1 = im.d2(r + 16u), p2 = im.d2(r + 32u), p3 = im.d2(r + 48u);
#ifdef RTOW_REAL_F32
            center = {(real)p0.x + rtime * (real)p1.y, (real)p0.y + rtime * (real)p2.x, (real)p1.x + rtime * (real)p2.y};
#else
            center = {p0.x + rtime * p1.y, p0.y + rtime * p2.x, p1.x + rtime * p2.y};
#endif
            inward = p3.x < 0.0;
          }
          normal = normalize(ro - center);
          front = (dot(rd, normal) < real(0.0)) ^ inward;
          normal = front ? normal : -normal;
        } else {
#ifdef RTOW_REAL_F32
          const uint32_t r = o_tri + 48u * (uint32_t)(pid - sc.n_sph - sc.n_mov);
          const float4 q2 = im.f4(r + 32u);
          normal = {q2.y, q2.z, q2.w};
#else
          const uint32_t r = o_tri + 96u * (uint32_t)(pid - sc.n_sph - sc.n_mov);
          const double2 q4 = im.d2(r + 64u), q5 = im.d2(r + 80u);
          normal = {q4.y, q5.x, q5.y};
#endif
        }
        mi = (int)im.u32(o_pmat + 4u * (uint32_t)pid);
        const uint32_t mr = o_mats + 48u * (uint32_t)mi;
        const double2 m1 = im.d2(mr + 16u), m2 = im.d2(mr + 32u);  // {att.z, fuzz}, {ir, kind|pad}
#ifdef RTOW_FAST_MATH
        const double2 m0 = im.d2(mr);  // {att.x, att.y}
        m_att = V3{(real)m0.x, (real)m0.y, (real)m1.x};
#endif
        m_fuzz = (real)m1.y;
        m_ir = (real)m2.x;
        kind = (int)(__double_as_longlong(m2.y) & 0xffffffffll);
      } else {
        if (pid < sc.n_sph + sc.n_mov) {
          V3 center;
          bool inward;
          if (pid < sc.n_sph) {
            const double *q = sc.sph + 4 * (size_t)pid;
            center = {(real)q[0], (real)q[1], (real)q[2]};
            inward = sc.sph_r[pid] < 0.0;
          } else {
            const double *q = sc.mov + 8 * (size_t)(pid - sc.n_sph);
#ifdef RTOW_REAL_F32
            center = {(real)q[0] + rtime * (real)q[3], (real)q[1] + rtime * (real)q[4], (real)q[2] + rtime * (real)q[5]};
#else
            center = {q[0] + rtime * q[3], q[1] + rtime * q[4], q[2] + rtime * q[5]};
#endif
            inward = q[7] < 0.0;
          }
          normal = normalize(ro - center);
          front = (dot(rd, normal) < real(0.0)) ^ inward;
          normal = front ? normal : -normal;
        } else {
          const double *q = sc.tri + 12 * (size_t)(pid - sc.n_sph - sc.n_mov);
          normal = {(real)q[9], (real)q[10], (real)q[11]};
        }
        mi = sc.prim_mat[pid];
        const DevMaterial *m = sc.mats + mi;
#ifdef RTOW_FAST_MATH
        m_att = V3{(real)m->att[0], (real)m->att[1], (real)m->att[2]};
#endif
        kind = m->kind;
        m_fuzz = (real)m->fuzz;
        m_ir = (real)m->ir;
      }

    }
    // The random-number window — the request's one Philox block and what is decoded from it at once — is dense integer
    // and f64 arithmetic with no memory access: it runs at the LOWEST issue priority, so that of the four waves of a SIMD
    // those in a latency-bound stage (cell lists, the DDA's cell words, hit records) issue first and the arithmetic of
    // this window fills the gaps they leave (stage_prio, rtow_trace_math.h; measured in DESIGN.md §4.7 d13).
    stage_prio<kPrioRng>();
    // THE block of the request: request 0 of a new sample (jitter, shutter time, lens point) or request 1 + bounce of
    // a scattered ray (point of the unit ball, dielectric coin).  One evaluation serves both kinds of lanes, and no
    // request draws a second block (rtow_trace_rng.h: the reference's rejection loops are direct samplers here).
    uint32_t o0 = anyv(0u), o1 = anyv(0u), o2 = anyv(0u), o3 = anyv(0u);
    if (do_regen) g.r = 0u;
    if (do_regen || do_scat) {
      philox4x32(g.r, g.sample, g.pixel, 0u, k0, k1, o0, o1, o2, o3);
      g.r += 1u;
    }
    if constexpr (STAMPS) {  // wave-level count of block evaluations: the first active lane reports
      if (__ballot(do_regen || do_scat) != 0ull) stamps.blocks += 1;
    }
    V3 dirbase = anyv3();  // bounce: direction before the fuzz term
    // Both samplers need the sine and cosine of an angle of the first quadrant and one square root (rtow_trace_rng.h):
    // evaluated ONCE for the camera and the scatter lanes together — the same operations on the same values as in
    // lens_from_block / ball_from_block (which the state-machine kernel's camera_ray / scatter_dir call as they are), at
    // twice the lane density: +0.7 % on the cover scene at 500 spp, +1.3 % on the moving cover, +0.6 % on suzanne
    real s_ang = anyv(real(0)), s_arg = anyv(real(0)), s_z = anyv(real(0));
    uint32_t s_b = anyv(0u);
    if (do_regen) {
      s_b = (o0 & 0x7ffu) | ((o1 & 0x7ffu) << 11) | ((o2 & 0x3ffu) << 22);
      s_ang = (real)(s_b & 0x3fffffffu) * real(0x1p-30 * kHalfPi);
      s_arg = (real)o3 * real(0x1p-32);
    }
    if (do_scat) {
      s_z = (real)(o0 >> 8) * real(0x1p-24);
      s_ang = (real)(o1 >> 8) * real(0x1p-24 * kHalfPi);
      s_arg = real(1.0) - s_z * s_z;
    }
    real s_sn = anyv(real(0)), s_cs = anyv(real(0)), s_sq = anyv(real(0));
    if (do_regen || do_scat) {
      opaque(s_ang);  // (rtow_trace_rng.h: the angle as a rounded product, here as in the helpers)
      s_sn = sin_quarter(s_ang);
      s_cs = sin_quarter(real(kHalfPi) - s_ang);
      s_sq = fast_sqrt(s_arg);
    }
    if (do_scat) {
      if (kind == 2) {
        const real ir = m_ir;
        const V3 unit = normalize(rd);
        const real cos_theta = dot(-unit, normal);
        const real sin_theta = fast_sqrt(real(1.0) - cos_theta * cos_theta);
        const real ratio = front ? fast_rcp(ir) : ir;
        bool refl = ratio * sin_theta > real(1.0);
        if (!refl) {  // src/common-model.cpp:53-54: the coin is drawn only if refraction is possible
          real r0 = fast_div(real(1.0) - ratio, real(1.0) + ratio);
          r0 = r0 * r0;
          const real x = real(1.0) - cos_theta;
          const real x2 = x * x;
          const real R = r0 + (real(1.0) - r0) * (x2 * x2 * x);
          refl = R > coin_from_block(o0, o1, o3);
        }
        dirbase = refl ? reflect(unit, normal) : refract(unit, normal, ratio);
      } else if (kind == 1) {
        dirbase = reflect(rd, normal);
      }
    }
    // (the window stays open through the camera ray's construction and the scatter's finish below — arithmetic on what
    // the block delivered, no memory access either — and closes in front of the walk: with the rejection loops gone the
    // other waves' latencies need this arithmetic as their filler, +0.7 % on the cover scene, +1.6 % on the 96.8k mesh)
    if (do_regen) {
      // src/render.cpp:158-159, src/common-model.cpp:156-167
      const int from_top_i = P.H - (int)gi - 1;
      real ju, jv, jt;  // pixel jitter, shutter time
      jitter_from_block(o0, o1, o2, ju, jv, jt);
#ifdef RTOW_FAST_MATH
      // the divisors are constants of the launch: their reciprocals come from the host (two scalar loads from the
      // kernel-argument segment beside the camera block's) instead of two v_rcp_f64 + refinement per new sample
      const RTOW_CONST TraceParams *kq = (const RTOW_CONST TraceParams *)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(kq));
      const real u = ((real)(int)j + ju) * (real)kq->inv_wm1;
      const real v = ((real)from_top_i + jv) * (real)kq->inv_hm1;
#else
      const real u = fast_div((real)(int)j + ju, (real)(P.W - 1));
      const real v = fast_div((real)from_top_i + jv, (real)(P.H - 1));
#endif
      // camera block: wave-uniform scalar loads (origin u v horizontal vertical llc | lens t0 t1)
#ifdef RTOW_REAL_F32
      const RTOW_CONST float *cm = (const RTOW_CONST float *)P.cam32;
#else
      cdptr cm = (cdptr)(const double *)P.cam;
#endif
      const real lens = cm[18], ct0 = cm[19], ct1 = cm[20];
      real px, py;  // random_in_unit_disk (src/common-model.cpp:157)
      {  // (lens_from_block's last step: the quadrant)
        const uint32_t q = s_b >> 30;
        const real cx = (q & 1u) ? s_sn : s_cs, sy = (q & 1u) ? s_cs : s_sn;
        px = s_sq * ((q == 1u || q == 2u) ? -cx : cx);
        py = s_sq * (q >= 2u ? -sy : sy);
      }
      const real rdx = lens * px, rdy = lens * py;
      const V3 offset = V3{cm[3], cm[4], cm[5]} * rdx + V3{cm[6], cm[7], cm[8]} * rdy;
      const V3 from = V3{cm[0], cm[1], cm[2]} + offset;
      rd = V3{cm[15], cm[16], cm[17]} + u * V3{cm[9], cm[10], cm[11]} + v * V3{cm[12], cm[13], cm[14]} - from;
      ro = from;
      rtime = jt * (ct1 - ct0) + ct0;
      depth = P.max_child_rays;
      nb = 0;
#ifdef RTOW_FAST_MATH
      throughput = {1, 1, 1};
#endif
      need_sample = false;
    }
    if (do_scat) {
      pending_hit = false;
      bool absorbed = false;
      // (the new direction is written over the old one in both branches; an absorbed path — black, src/render.cpp:120 —
      // starts a new sample in the next trip and never reads it)
      // random_unit_vector() (src/common-model.cpp:16,26,58): ball_from_block's last step — radius times direction
      const real s_r = (real)max(max(o2 & 0xffffu, o2 >> 16), o3 & 0xffffu) * real(0x1p-16);
      const real s_rs = s_r * s_sq;
      const V3 rnd = V3{s_rs * s_cs, s_rs * s_sn, s_r * s_z};
      if (kind == 0) {
        absorbed = rabs(normal.x - rnd.x) < real(1e-8) && rabs(normal.y - rnd.y) < real(1e-8) &&
                   rabs(normal.z - rnd.z) < real(1e-8);
        rd = normal + rnd;
      } else {
        rd = dirbase + m_fuzz * rnd;
      }
      if (absorbed) {
        need_sample = true;  // src/render.cpp:120: black (the lane starts its next sample in the next trip)
        --s_left;
        ++g.sample;
      } else {
#ifdef RTOW_FAST_MATH
        throughput = throughput * m_att;
#else
        P.stack[(size_t)nb * P.n_lanes + lane_g] = (uint32_t)mi;
#endif
        ++nb;
        --depth;
      }
    }
    stage_prio<kPrioStage>();
    const bool tracing = live && !need_sample;  // has a ray to advance in this trip

    stamps.mark(RG_REGEN, __ballot(do_regen || do_scat));
    if constexpr (STAMPS) {
      stamps.trips += 1;
      if (t_empty != 0ull) trips_after_empty += 1;
      stamps.tail = t_empty != 0ull;
    }
    // ---- one ray segment: closest hit --------------------------------------------
    if constexpr (KERNEL != 3 && KERNEL != 4) {
      best.t = 0;
      best.prim = -1;
    }
    if constexpr (KERNEL == 4) {
      best = closest_hit_bvh4<LDS, STAMPS>(im4, sc, P, ro, rd, rtime, tracing, lane_g, nnode, nprim, stamps, best, w_cur, w_sa,
                                           P.walk_cap, P.walk_max_open);
    } else if constexpr (KERNEL == 3) {
      if constexpr (STAMPS) stamps.primary = __ballot(tracing && depth == P.max_child_rays);
      best = closest_hit_grid<LDS, STAMPS, SPEC>(im, sc, ro, rd, rtime, tracing, nnode, nprim, stamps, best, t_resume, P.walk_cap,
                                           P.walk_max_open, P.leaf_votes);
    } else if constexpr (KERNEL == 2) {
      // the walk uses wave votes, so every lane of the wave enters it
      best = closest_hit_bvh<LDS, STAMPS>(im, sc, ro, rd, rtime, tracing, nnode, nprim, stamps);
    } else if constexpr (KERNEL == 5) {
#ifndef RTOW_FAST_MATH
      if (tracing) best = closest_hit_reftree(sc, to_f64(ro), to_f64(rd), (double)rtime, nnode, nprim);
#endif
    } else {
      best = closest_hit_stream(sc, to_f64(ro), to_f64(rd), (double)rtime, tracing);  // (every lane: the tiled loop stages with all 64)
    }

    stamps.mark(RG_WALK, __ballot(tracing));
    bool arrived = tracing;  // the segment's closest hit is known (a stopped GRID / BVH4 walk continues next trip)
    if constexpr (KERNEL == 3) arrived = tracing && !(t_resume > 0.0f);
    if constexpr (KERNEL == 4) arrived = tracing && w_cur == 0x1fffffu;
    if (arrived) {
      ++nseg;
      if (best.prim >= 0) {
        if (depth <= 0)
          need_sample = true;  // src/render.cpp:115: black
        else
          pending_hit = true;  // scattered at the top of the next trip, together with the new camera rays
      } else {
        // ---- background + unwind of the recursion (src/render.cpp:119,122-128) --
        const V3 unit = normalize(rd);
        const real t = real(0.5) * (unit.y + real(+1.0));
        V3 c = (real(1.0) - t) * V3{1, 1, 1} + t * V3{real(0.5), real(0.7), real(1.0)};
#ifdef RTOW_FAST_MATH
        c = throughput * c;
#else
        for (int q = nb - 1; q >= 0; --q) {
          const uint32_t smi = P.stack[(size_t)q * P.n_lanes + lane_g];
          if constexpr (KERNEL == 4) {
            const uint32_t mr = sc.b4_off_mats + 48u * smi;
            const vd2 a0 = im4.d2(mr), a1 = im4.d2(mr + 16u);
            c = V3{(real)a0.x, (real)a0.y, (real)a1.x} * c;
          } else if constexpr (KERNEL >= 2 && KERNEL <= 4) {
            const uint32_t mr = (KERNEL == 3 ? sc.g_off_mats : sc.off_mats) + 48u * smi;
            const double2 a0 = im.d2(mr), a1 = im.d2(mr + 16u);
            c = V3{(real)a0.x, (real)a0.y, (real)a1.x} * c;
          } else {
            const DevMaterial *m = sc.mats + smi;
            c = V3{(real)m->att[0], (real)m->att[1], (real)m->att[2]} * c;
          }
        }
#endif
        {
          // never fused with the multiply that produced `c`: a donated sample (end-of-launch tail)
          // reaches its owner as a rounded colour, so a fused multiply-add here would make the
          // pixel sum depend on which lane traced the sample (the empty asm hides `c` from the
          // contraction pass)
          V3d cd = to_f64(c);
          asm volatile("" : "+v"(cd.x), "+v"(cd.y), "+v"(cd.z));
          acc = acc + cd;  // pixel_color += ray_color(...)
        }
        need_sample = true;
      }
      if (need_sample) {
        --s_left;
        ++g.sample;
      }
    }
    stamps.mark(RG_SHADE, __ballot(arrived));
  }
  if constexpr (STAMPS) {
    if (lane == 0)
    {
      // wave lifetime on the 100 MHz constant clock: [4] sum of end times, [5] min, [6] max,
      // [7] sum of the times at which the wave first found the queue empty (relative to the
      // earliest start, kept in [8+RG_COUNT+3] as a min)
      const unsigned long long tend = __builtin_amdgcn_s_memrealtime();
      atomicAdd(&P.counters[4], tend - P.t_origin[0]);
      atomicMin(&P.counters[5], tend - P.t_origin[0]);
      atomicMax(&P.counters[6], tend - P.t_origin[0]);
      atomicAdd(&P.counters[7], t_empty - P.t_origin[0]);
      // [17..22]: histogram of (end - this wave's own queue-empty time) in 0.2 ms bins (last: >= 1 ms)
      {
        const unsigned long long after = tend - t_empty;  // 100 MHz ticks
        unsigned bin = (unsigned)(after / 20000ull);
        bin = bin > 5u ? 5u : bin;
        atomicAdd(&P.counters[17 + bin], 1ull);
        atomicMax(&P.counters[40], trips_after_empty);
        if (bin >= 2u) {  // stragglers: trips and time after the queue emptied
          atomicAdd(&P.counters[41], trips_after_empty);
          atomicAdd(&P.counters[42], after);
          atomicAdd(&P.counters[43], 1ull);
        }
      }
      for (int r = 0; r < RG_COUNT; ++r) atomicAdd(&P.counters[8 + r], stamps.t[r]);
      for (int r = 0; r < RG_COUNT; ++r) atomicAdd(&P.counters[29 + r], stamps.tt[r]);  // after queue-empty only
      atomicAdd(&P.counters[37], trips_after_empty);
      atomicAdd(&P.counters[38], tend - t_empty);
      for (int r = 0; r < RG_COUNT; ++r) atomicAdd(&P.counters[23 + r], stamps.lt[r]);
      atomicAdd(&P.counters[34], stamps.step_lanes);
      atomicAdd(&P.counters[44], stamps.blocks);
      atomicAdd(&P.counters[36], stamps.block_lanes);
      atomicAdd(&P.counters[35], stamps.leaf_lanes);
      atomicAdd(&P.counters[45], stamps.iters_cam);
      atomicAdd(&P.counters[46], stamps.phases_cam);
      atomicAdd(&P.counters[13], stamps.iters);
      atomicAdd(&P.counters[14], stamps.trips);
      atomicAdd(&P.counters[15], stamps.phases);
    }
  }

#ifdef RTOW_TAILSTAT
  if (lane == 0 && P.spill != nullptr) {  // (the experiment borrows the BVH4 spill pointer for its table: 8 words per wave)
    unsigned long long *ts = (unsigned long long *)P.spill + (size_t)((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 8;
    ts[0] = ts_start;
    ts[1] = ts_empty;
    ts[2] = __builtin_amdgcn_s_memrealtime();
    ts[3] = ts_trips;
    ts[4] = ts_trips_after;
    ts[5] = ts_live;
    ts[6] = ts_left;
    ts[7] = ts_pool;
  }
#endif
  // stats: one atomic per wave and counter
  unsigned long long t0 = nseg, t1 = nprim, t2 = nnode;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    t0 += __shfl_down(t0, off);
    t1 += __shfl_down(t1, off);
    t2 += __shfl_down(t2, off);
  }
  if (lane == 0) {
    atomicAdd(&P.counters[1], t0);
    if (KERNEL >= 2) {
      atomicAdd(&P.counters[2], t1);
      atomicAdd(&P.counters[3], t2);
    }
  }
}

#include "rtow_trace_sm4.h"

}  // namespace

// The trace kernels address the dynamic LDS block from 0 (lds_read / lds_write, rtow_trace_bvh4.h): an instantiation
// that had static LDS of its own would read the wrong bytes.  Checked once per instantiation, on the kernel that is
// actually launched (the occupancy query below looks at one representative only).
template <class Kern>
static int no_static_lds(Kern k) {
  hipFuncAttributes fa;
  const hipError_t e = hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k));
  if (e != hipSuccess) return (int)e;
  return fa.sharedSizeBytes == 0 ? 0 : (int)hipErrorInvalidValue;
}

template <bool L, bool S>
static int launch_sm4(const TraceParams &p, int grid, int block, unsigned lds_bytes, hipStream_t st) {
  auto k = RTOW_CAT(rtow_trace4_, RTOW_SUFFIX)<L, S>;
  static const int lds_ok = no_static_lds(k);
  if (lds_ok != 0) return lds_ok;
  if (lds_bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds_bytes, st, p);
  return (int)hipGetLastError();
}

// kernel: 1 STREAM, 2 BVH, 3 GRID, 4 BVH4; +16 = diagnostic region stamps (LDS variants only)
template <int K, bool L, bool S, int SPEC = 0>
static int launch_one(const TraceParams &p, int grid, int block, unsigned lds_bytes, hipStream_t st) {
  auto k = RTOW_CAT(rtow_trace_, RTOW_SUFFIX)<K, L, S, SPEC>;
  static const int lds_ok = no_static_lds(k);  // (one static per instantiation)
  if (lds_ok != 0) return lds_ok;
  if (lds_bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds_bytes, st, p);
  return (int)hipGetLastError();
}

int RTOW_CAT(launch_trace_, RTOW_SUFFIX)(const TraceParams &p, int kernel, int grid, int block,
                                         unsigned lds_bytes, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  const bool lds = lds_bytes > 0;
  switch (kernel) {
    case 1: return launch_one<1, false, false>(p, grid, block, lds_bytes, st);  // (LDS: the tiled triangle loop's per-wave tiles)
    case 2: return lds ? launch_one<2, true, false>(p, grid, block, lds_bytes, st)
                       : launch_one<2, false, false>(p, grid, block, 0, st);
    case 3:
      if (lds && p.spec == kSpecStaticSpheres) return launch_one<3, true, false, 1>(p, grid, block, lds_bytes, st);
      if (lds && p.spec == kSpecMovingSpheres) return launch_one<3, true, false, 2>(p, grid, block, lds_bytes, st);
      return lds ? launch_one<3, true, false>(p, grid, block, lds_bytes, st)
                 : launch_one<3, false, false>(p, grid, block, 0, st);
    case 4:
    case 4 + 16: {
      const bool full = p.sc.b4_half == 0u, stamps = kernel == 4 + 16;  // (binary32 nodes <=> staged whole)
      if (p.b4_trips)  // the trip-structured form (default); RTOW_BVH4_SM selects the state machine
        return full ? (stamps ? launch_one<4, true, true>(p, grid, block, lds_bytes, st)
                              : launch_one<4, true, false>(p, grid, block, lds_bytes, st))
                    : (stamps ? launch_one<4, false, true>(p, grid, block, lds_bytes, st)
                              : launch_one<4, false, false>(p, grid, block, lds_bytes, st));
      return full ? (stamps ? launch_sm4<true, true>(p, grid, block, lds_bytes, st)
                            : launch_sm4<true, false>(p, grid, block, lds_bytes, st))
                  : (stamps ? launch_sm4<false, true>(p, grid, block, lds_bytes, st)
                            : launch_sm4<false, false>(p, grid, block, lds_bytes, st));
    }
#ifndef RTOW_FAST_MATH
    case 5: return launch_one<5, false, false>(p, grid, block, 0, st);  // the reference's tree: strict build only
#endif
    case 2 + 16: return launch_one<2, true, true>(p, grid, block, lds_bytes, st);
    case 3 + 16: return launch_one<3, true, true>(p, grid, block, lds_bytes, st);
    default: return (int)hipErrorInvalidValue;
  }
}

// Workgroups per CU that stay resident: min over the register file (512 VGPRs per
// SIMD lane, allocated in granules of 8), the 32-wave CU limit and the 160 KiB of LDS.
// (The runtime's occupancy query ignores LDS above 64 KiB per CU on this stack; a grid
// that turns out larger than resident only queues the surplus workgroups, which then
// find the work queue empty — there is no inter-workgroup dependency.)
int RTOW_CAT(trace_occupancy_, RTOW_SUFFIX)(int kernel, int block, unsigned lds_bytes) {
  const void *fn;
  const bool lds = lds_bytes > 0;
#ifndef RTOW_FAST_MATH
  if (kernel == 5)
    fn = reinterpret_cast<const void *>(RTOW_CAT(rtow_trace_, RTOW_SUFFIX)<5, false, false>);
  else
#endif
  if (kernel == 4)  // (both variants have the same launch bounds; the full-LDS one stands for both)
    fn = reinterpret_cast<const void *>(RTOW_CAT(rtow_trace_, RTOW_SUFFIX)<4, true, false>);
  else if (kernel == 3)
    fn = lds ? reinterpret_cast<const void *>(RTOW_CAT(rtow_trace_, RTOW_SUFFIX)<3, true, false>)
             : reinterpret_cast<const void *>(RTOW_CAT(rtow_trace_, RTOW_SUFFIX)<3, false, false>);
  else if (kernel == 2)
    fn = lds ? reinterpret_cast<const void *>(RTOW_CAT(rtow_trace_, RTOW_SUFFIX)<2, true, false>)
             : reinterpret_cast<const void *>(RTOW_CAT(rtow_trace_, RTOW_SUFFIX)<2, false, false>);
  else
    fn = reinterpret_cast<const void *>(RTOW_CAT(rtow_trace_, RTOW_SUFFIX)<1, false, false>);
  if (lds_bytes > 48 * 1024)
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  hipFuncAttributes fa;
  if (hipFuncGetAttributes(&fa, fn) != hipSuccess) return -1;
  if (fa.sharedSizeBytes != 0) return -1;  // (the 4-wide walk addresses the dynamic LDS block from 0, rtow_trace_bvh4.h lds_read)
  const int regs = fa.numRegs > 0 ? fa.numRegs : 128;
  const int alloc = ((regs + 7) / 8) * 8;
  int waves_per_simd = 512 / alloc;
  if (waves_per_simd > 8) waves_per_simd = 8;
  if (waves_per_simd < 1) waves_per_simd = 1;
  const int waves_per_block = block / 64;
  int nb = (waves_per_simd * 4) / waves_per_block;
  if (lds_bytes > 0) {
    const int by_lds = (int)((160u * 1024u) / lds_bytes);
    if (by_lds < nb) nb = by_lds;
  }
  if (nb < 1) nb = 1;
  return nb;
}

}  // namespace rtow

