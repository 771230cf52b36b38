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
//     0), key = seed.  The first block of a sample carries the pixel jitter and the shutter
//     time (21 bits each) AND the first lens-disk candidate (32 bits per coordinate); a
//     further lens-disk block carries two candidates.  Unit-ball candidates have 21 bits per
//     coordinate, one candidate per pair of words: the first block of a bounce carries one
//     candidate and the dielectric coin (32 bits), every further block two candidates.
//     Whole blocks per request keep the rejection loops free of per-lane parity divergence;
//     Philox is a large share of the kernel's VALU time, so blocks are not wasted (packing the
//     lens candidates removed ~1.9 of the ~4 blocks a wave spends per trip on new camera rays:
//     +3.1 %; two unit-ball candidates per block cut the scatter loop from ~5.9 to ~3.8 blocks
//     per trip: +2.0 %).
//
// In the strict build (-ffp-contract=off) every expression below has the operand
// order of the reference expression it restates, f64 sqrt and division are the
// correctly rounded IEEE forms, and the result is bit-identical to oracle/.

#include <hip/hip_runtime.h>
#include <stdint.h>

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

constexpr uint32_t kItemBatch = 64;  // work items fetched per global atomic (per wave): one 64-pixel tile of one level

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
__device__ __forceinline__ unsigned long long wave_bcast(unsigned long long v, int src_lane) {  // src_lane: wave-uniform
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src_lane);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src_lane);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t lanes_below(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// ---- work items ------------------------------------------------------------------------------
// A wave's share of the global item queue: one atomic buys a batch of 64 items — one 64-pixel tile of one
// level — and the lanes take items from the wave-local pool [next, end) by ballot rank.  All fields are
// wave-uniform.  (One atomic per wave TRIP saturated the counter word at ~100 dequeues/us.  Until round 3 the
// batch shrank as the queue drained — "guided" batches against hoarding — which brought that storm back for
// the last tenth of every launch: ~40 us per trip.  The end of a launch is balanced by splitting ITEMS now,
// see "exported samples" below, and a batch is always a tile.)
struct ItemPool {
  uint32_t next = 0, end = 0;
  bool dry = false;  // this wave has seen the end of the queue
#ifdef RTOW_GUIDED
  unsigned long long seen = 0ull;
#endif
};

// Hands one queue position to every lane of `need_mask` (all lanes of the wave call this together).
// Returns the lane's position, >= kp->n_items when the queue is exhausted (pool.dry is then set).
__device__ __forceinline__ unsigned long long take_items(ItemPool &pool, unsigned long long need_mask, unsigned lane,
                                                         const RTOW_CONST TraceParams *kp, unsigned long long *counters) {
  const uint32_t want = (uint32_t)__popcll(need_mask);
  const uint32_t avail = pool.end - pool.next;
  const uint32_t rank = (uint32_t)__popcll(need_mask & ((1ull << lane) - 1ull));
  unsigned long long mine = (unsigned long long)pool.next + rank;
  if (want > avail) {
#ifdef RTOW_GUIDED
    // (experiment: the round-2 guided batches — shrinking to what the wave needs as the queue drains)
    const unsigned long long left_g = (unsigned long long)kp->n_items > pool.seen ? (unsigned long long)kp->n_items - pool.seen : 0ull;
    uint32_t batch = (uint32_t)(left_g / ((unsigned long long)((gridDim.x * blockDim.x) >> 6) * 4ull));
    batch = batch > kItemBatch ? kItemBatch : batch;
    batch = batch < want - avail ? want - avail : batch;
#else
    const uint32_t batch = kItemBatch;  // (want - avail <= 64)
#endif
    const int leader = __ffsll((long long)need_mask) - 1;
    unsigned long long base = (unsigned long long)kp->n_items;
    // a wave that has seen the end of the queue stops polling it
    if (!pool.dry) {
      if ((int)lane == leader) base = atomicAdd(&counters[0], (unsigned long long)batch);
      base = wave_bcast(base, leader);
    }
    const unsigned long long cap = (unsigned long long)kp->n_items;
#ifdef RTOW_GUIDED
    pool.seen = base + batch;
#endif
    if (base + batch >= cap) pool.dry = true;
    if (rank >= avail) mine = base + (rank - avail);
    const unsigned long long nn = base + (want - avail), ne = base + batch;
    pool.next = (uint32_t)(nn < cap ? nn : cap);
    pool.end = (uint32_t)(ne < cap ? ne : cap);
  } else {
    pool.next += want;
  }
  return mine;
}

// ---- exported samples ---------------------------------------------------------------------------
// Per-pixel cost is heavy-tailed: the reference never absorbs a path (no Russian roulette), so in the
// crevices between a sphere and the ground paths run to the bounce limit, and an item of 16 samples there
// costs hundreds of trips — taken late, it IS the end of the launch (0.5 ms of a 8.3 ms frame in round 2,
// when only the lanes of its own wave could take samples off it, five at a time, once the wave was idle).
// Now every item has a SEGMENT BUDGET set when it is fetched: about the time the queue still lasts at that
// point (long at the start of a launch, zero at its end).  A lane that starts a sample with its budget spent
// EXPORTS the samples after it: their descriptors go to the ring of its workgroup (counters in LDS, entries in
// HBM), and the lanes of the workgroup take ring entries before queue items.  An exported sample is traced
// like any other; its colour goes to the ring entry's slot instead of a pixel sum, and the reduce kernel adds
// the exported colours behind the item's own sum in sample order — so the pixel sum is the sequential one bit
// for bit whoever traced what, in the strict build too.  The split point depends only on the item's own path
// lengths, and where a full ring refuses an export the owner simply traces on: the image never depends on it.
__shared__ uint32_t rtow_wg_head;  // ring entries reserved so far (may run past the capacity: clamp)
__shared__ uint32_t rtow_wg_tail;  // ring entries claimed so far
__shared__ uint32_t rtow_wg_busy;  // waves of the workgroup that still hold work (and so may still export)

constexpr uint32_t kNoItem = 0xffffffffu;
constexpr uint32_t kExported = 0x80000000u;  // `item` of a lane tracing an exported sample: flag | ring slot

__device__ __forceinline__ void item_pools_init() {
  if (threadIdx.x == 0) {
    rtow_wg_head = 0u;
    rtow_wg_tail = 0u;
    rtow_wg_busy = blockDim.x >> 6;
  }
}

// Queue position -> (partial-sum slot, column, global row, sample range).
// A work item is (level, pixel): the level's run of consecutive samples of one pixel, summed in sample order.
// Strict build: a level is one reference "thread" (stream): spp / nstreams samples (src/render.cpp:151-166).
// Fast builds: the levels are a SCHEDULE over the launch's sample range, independent of nstreams — long chunks
// first, then chunks that shrink geometrically down to single samples (rtow_capi.cpp, make_schedule) — and the
// queue is level-major, so the launch ends on items of one sample and the end-of-launch tail is one path, not
// one item.  The sum of a pixel is then the fixed-order sum of its level sums (tolerance build: re-association
// of the reference's sum, like the contraction it already allows); the image stays a pure function of
// (scene, config, seed).
// Queue order.  Uniform levels: tile-major, level-minor — all streams of a 64-pixel tile are
// adjacent, tiles run top-to-bottom, and the queue is consumed from its far end, so a launch
// ENDS on the top rows of the image for every stream.  In the reference's scenes that is sky
// (the top 8 % of the cover image: one-segment paths), so most waves run out of work together:
// waves finishing > 0.2 ms after they find the queue empty fell from 51 % to 5 % (+1.9 %).
// (Stream-major order ended only the last stream on the sky; ending on the bottom rows —
// near ground, short paths — measures the same.)  The partial-sum slot is [level][pixel] in both orders.
struct ItemPos {
  uint32_t item, j, gi, sample0, count;
};
__device__ __forceinline__ ItemPos decode_item(const RTOW_CONST TraceParams *kp, uint32_t mine, uint32_t npix_local) {
  ItemPos ip;
  const uint32_t qi = kp->n_items - 1u - mine;
  uint32_t k, lp, lr;
  if (kp->tile_h_log2 == 0u) {  // row-major, level-major
    k = fastdiv(qi, FastDiv{kp->div_npix.magic, kp->div_npix.shift});
    lp = qi - k * npix_local;
    if (kp->level_major != 0u) k = (uint32_t)kp->nstreams - 1u - k;  // table order: the shortest levels last
    lr = fastdiv(lp, FastDiv{kp->div_w.magic, kp->div_w.shift});
    ip.j = lp - lr * (uint32_t)kp->W;
  } else {  // 64-pixel tiles: a wave's batch of 64 items is one compact tile of one level
    const uint32_t g64 = qi >> 6, w = qi & 63u;
    uint32_t t;
    if (kp->level_major != 0u) {  // every level covers the image once, levels in table order
      const uint32_t lq = fastdiv(g64, FastDiv{kp->div_ntiles.magic, kp->div_ntiles.shift});
      t = g64 - lq * kp->n_tiles;
      k = (uint32_t)kp->nstreams - 1u - lq;
    } else {  // all levels of a tile adjacent
      t = fastdiv(g64, FastDiv{kp->div_ns.magic, kp->div_ns.shift});
      k = g64 - t * (uint32_t)kp->nstreams;
    }
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
  // The level's sample range, from the table in device memory.  The lanes of a fetch hold consecutive queue
  // positions, i.e. one level or two: each distinct level costs one scalar load (a loop over the distinct values).
  ip.sample0 = 0u;
  ip.count = 0u;
  const RTOW_CONST uint32_t *lvl = (const RTOW_CONST uint32_t *)kp->lvl;
  for (bool todo = true; todo;) {
    const uint32_t ku = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
    const uint32_t f = lvl[2u * ku], n = lvl[2u * ku + 1u];
    if (k == ku) {
      ip.sample0 = f;
      ip.count = n;
      todo = false;
    }
  }
  return ip;
}

// Camera::get_ray for sample (g.pixel, g.sample) of pixel (column j, global row gi): pixel jitter,
// lens disk, ray (src/render.cpp:158-159, src/common-model.cpp:156-167; disk sample: y draws first,
// random-utils.cpp:36).  The first lens candidate comes with the jitter block; every further block
// carries two.
__device__ __forceinline__ void camera_ray(const TraceParams &P, Rng &g, uint32_t k0, uint32_t k1, uint32_t j,
                                           uint32_t gi, V3 &ro, V3 &rd, real &rtime) {
  g.r = 0u;
  const int from_top_i = P.H - (int)gi - 1;
  real ju, jv, jt, c0, c1;
  rng_jitter(g, k0, k1, ju, jv, jt, c0, c1);
  const real u = fast_div((real)(int)j + ju, (real)(P.W - 1));
  const real v = fast_div((real)from_top_i + jv, (real)(P.H - 1));
  real py = c0 * (real(1.0) - real(-1.0)) + real(-1.0);
  real px = c1 * (real(1.0) - real(-1.0)) + real(-1.0);
  while (px * px + py * py + real(0.0) * real(0.0) >= real(1.0)) {
    real a0, b0, a1, b1;
    rng_disk2(g, k0, k1, a0, b0, a1, b1);
    py = a0 * (real(1.0) - real(-1.0)) + real(-1.0);
    px = b0 * (real(1.0) - real(-1.0)) + real(-1.0);
    if (px * px + py * py + real(0.0) * real(0.0) >= real(1.0)) {
      py = a1 * (real(1.0) - real(-1.0)) + real(-1.0);
      px = b1 * (real(1.0) - real(-1.0)) + real(-1.0);
    }
  }
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
// direction, or absorbed.  The first unit-ball candidate of the bounce comes with the dielectric coin.
__device__ __forceinline__ bool scatter_dir(Rng &g, uint32_t k0, uint32_t k1, int kind, real m_fuzz, real m_ir, V3 rd,
                                            V3 normal, bool front, V3 &dir) {
  real coin;
  V3 rnd = rng_scatter(g, k0, k1, coin);
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
      refl = R > coin;
    }
    dirbase = refl ? reflect(unit, normal) : refract(unit, normal, ratio);
  } else if (kind == 1) {
    dirbase = reflect(rd, normal);
  }
  // random_unit_vector(): reject candidates outside the unit ball (random-utils.cpp:23-33)
  while (dot(rnd, rnd) >= real(1.0)) {  // two candidates per further block
    V3 ca, cb;
    rng_scatter2(g, k0, k1, ca, cb);
    rnd = dot(ca, ca) >= real(1.0) ? cb : ca;
  }
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
template <int KERNEL, bool LDS, bool STAMPS = false>
__global__ void __launch_bounds__(KERNEL >= 2 && KERNEL <= 4 ? 1024 : 256)
    RTOW_CAT(rtow_trace_, RTOW_SUFFIX)(const TraceParams P) {
  const DevScene &sc = P.sc;
  const uint32_t k0 = P.seed_lo, k1 = P.seed_hi;
  const unsigned lane = lane_id();
  [[maybe_unused]] const uint32_t lane_g = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t npix_local = (uint32_t)P.local_rows * (uint32_t)P.W;

  item_pools_init();
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
  } else {
    __syncthreads();  // the workgroup's item pool
  }

  // per-lane state
  bool need_sample = true;
  bool pending_hit = false;   // the last segment ended in a hit that is scattered at the top of the next trip
  int s_left = 0;             // samples left in the current item
  uint32_t item = kNoItem;    // partial-sum slot of the current item, or kExported | ring slot of an exported sample
  uint32_t seg_limit = 0xffffffffu;  // value of nseg from which the item is over its segment budget
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
  ItemPool pool;             // wave-uniform: this wave's batch of work items
  bool counted_busy = true;  // wave-uniform: this wave is counted in rtow_wg_busy
  Stamps<STAMPS> stamps;
  stamps.start();
  unsigned long long t_empty = 0ull;  // diagnostic: when this wave first saw the queue empty
  unsigned long long trips_after_empty = 0ull;
  if constexpr (STAMPS) {
    if (lane == 0) atomicMin(P.t_origin, (unsigned long long)__builtin_amdgcn_s_memrealtime());
  }

  for (;;) {
    // ---- item bookkeeping ---------------------------------------------------
    bool idle = need_sample && s_left <= 0;  // finished its item (or has none yet)
    unsigned long long need_mask = __ballot(idle);
    // Lanes that need a new item wait until `fetch_votes` of them do (or nothing else is left to do): with
    // 64 desynchronised lanes some lane finishes an item in almost every trip, and the fetch block — queue
    // atomic, item decode, partial-sum store — would run every trip for two or three lanes.  (Not once the
    // queue is dry: what is left then is exported samples, and nobody should wait for company.)
    if (need_mask != 0ull && !pool.dry && (uint32_t)__popcll(need_mask) < P.fetch_votes && ~need_mask != 0ull)
      need_mask = 0ull;
    if (need_mask != 0ull) {
      // The item-decoding parameters are read here from the kernel-argument segment (scalar loads)
      // instead of living in SGPRs for the whole launch: the kernel is VALU-issue-bound and ran out
      // of SGPRs, so every one of them cost a v_readlane (VALU) per use.
      const RTOW_CONST TraceParams *kp = (const RTOW_CONST TraceParams *)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(kp));  // opaque per trip: keeps the loads from being hoisted out of the loop
      const int leader = __ffsll((long long)need_mask) - 1;
      if (idle && item != kNoItem) {  // the finished item goes to its partial-sum slot, an exported sample to its ring slot
        double *dst = (item & kExported) != 0u ? kp->ovf_color + (size_t)(item & ~kExported) * 3
                                               : kp->partials + (size_t)item * 4;
        dst[0] = acc.x;
        dst[1] = acc.y;
        dst[2] = acc.z;
        item = kNoItem;
      }
      // (1) samples exported by the lanes of this workgroup come before new items
#ifndef RTOW_NO_RING
      {
        uint32_t h = 0u, t = 0u;
        if ((int)lane == leader) {
          h = atomicAdd(&rtow_wg_head, 0u);
          t = atomicAdd(&rtow_wg_tail, 0u);
        }
        h = (uint32_t)__builtin_amdgcn_readlane((int)h, leader);
        t = (uint32_t)__builtin_amdgcn_readlane((int)t, leader);
        const uint32_t cap = kp->ovf_cap;
        h = h < cap ? h : cap;
        if (t < h) {
          const uint32_t want = (uint32_t)__popcll(need_mask);
          const uint32_t n = want < h - t ? want : h - t;
          uint32_t ok = 0u;
          if ((int)lane == leader) ok = atomicCAS(&rtow_wg_tail, t, t + n) == t ? 1u : 0u;
          ok = (uint32_t)__builtin_amdgcn_readlane((int)ok, leader);
          const uint32_t rank = (uint32_t)__popcll(need_mask & ((1ull << lane) - 1ull));
          if (ok != 0u && idle && rank < n) {
            const uint32_t slot = blockIdx.x * cap + t + rank;
            // the entry was reserved before it was written: wait for its tag (the writer is a lane of this
            // workgroup a few instructions behind its reservation — or of this wave, a trip ago)
            const uint32_t *dp = kp->ovf_desc + (size_t)slot * 4;
            uint32_t d_pixel = 0u, d_sample = 0u, spins = 0u;
            for (;;) {
              const uint32_t tag = __hip_atomic_load(dp + 2, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
              if (tag == kp->ovf_tag) {
                d_pixel = __hip_atomic_load(dp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                d_sample = __hip_atomic_load(dp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                break;
              }
              if (++spins > (1u << 22)) {  // (structural bound; the host turns the flag into an error)
                atomicAdd(&P.counters[47], 1ull);
                break;
              }
              __builtin_amdgcn_s_sleep(1);
            }
            if (spins <= (1u << 22)) {
              gi = fastdiv(d_pixel, FastDiv{kp->div_w.magic, kp->div_w.shift});
              j = d_pixel - gi * (uint32_t)kp->W;
              g.pixel = d_pixel;
              g.sample = d_sample;
              s_left = 1;
              acc = {0.0, 0.0, 0.0};
              item = kExported | slot;
              seg_limit = 0xffffffffu;
              idle = false;
            }
          }
        }
      }
#endif
      // (2) the queue
      const unsigned long long rest_mask = __ballot(idle);
      if (rest_mask != 0ull && !(pool.dry && pool.next >= pool.end)) {
        const unsigned long long mine = take_items(pool, rest_mask, lane, kp, P.counters);
        if (idle && mine < (unsigned long long)kp->n_items) {
          const ItemPos ip = decode_item(kp, (uint32_t)mine, npix_local);
          item = ip.item;
          j = ip.j;
          gi = ip.gi;
          g.pixel = gi * (uint32_t)kp->W + j;
          g.sample = ip.sample0;
          s_left = (int)ip.count;
          acc = {0.0, 0.0, 0.0};
          reinterpret_cast<unsigned long long *>(kp->partials)[(size_t)item * 4 + 3] = 0ull;  // no exported samples (yet)
          // segment budget: about the trips the queue still lasts from here on
          const float bud = (float)(kp->n_items - (uint32_t)mine) * (float)ip.count * kp->budget_k;
          seg_limit = nseg + (bud < 1e9f ? (uint32_t)bud : 0x3fffffffu);
          idle = false;
        }
        if constexpr (STAMPS) {
          if (pool.dry && t_empty == 0ull) t_empty = __builtin_amdgcn_s_memrealtime();
        }
      }
    }
    // ---- end of the launch -------------------------------------------------------------------
    // A wave without work stays while other waves of its workgroup hold items: they may still export samples.
    if (__ballot(!(need_sample && s_left <= 0)) == 0ull) {
#ifdef RTOW_NO_RING
      if (pool.dry && pool.next >= pool.end) break;
#else
      if (pool.dry && pool.next >= pool.end) {
        uint32_t busy = 0u, h = 0u, t = 0u;
        if (lane == 0u) {
          if (counted_busy) (void)atomicSub(&rtow_wg_busy, 1u);
          busy = atomicAdd(&rtow_wg_busy, 0u);
          h = atomicAdd(&rtow_wg_head, 0u);
          t = atomicAdd(&rtow_wg_tail, 0u);
        }
        counted_busy = false;
        busy = (uint32_t)__builtin_amdgcn_readfirstlane((int)busy);
        h = (uint32_t)__builtin_amdgcn_readfirstlane((int)h);
        t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
        h = h < P.ovf_cap ? h : P.ovf_cap;
        if (busy == 0u && t >= h) break;
        if (t >= h) __builtin_amdgcn_s_sleep(16);
      }
#endif
      continue;
    }
#ifndef RTOW_NO_RING
    if (!counted_busy) {  // (took exported samples after having run out of work)
      if (lane == 0u) (void)atomicAdd(&rtow_wg_busy, 1u);
      counted_busy = true;
    }
#endif
    stamps.mark(RG_FETCH);

    const bool live = !(need_sample && s_left <= 0);

    // ---- new rays ------------------------------------------------------------------------------------
    // Two kinds of lanes need a new ray before the walk: those starting a sample (pixel jitter +
    // Camera::get_ray, src/render.cpp:158-159, src/common-model.cpp:156-167) and those whose previous
    // segment ended in a hit (Material::scatter, src/common-model.cpp:13-62).  Both draw one Philox
    // block and then, while their candidate lies outside the unit disk / unit ball, further blocks.
    // They draw TOGETHER: one block evaluation serves both kinds of lanes, and the rejection loop runs
    // max(lens, ball) trips instead of lens + ball — a wave used to spend ~6 block evaluations per trip on
    // ~1.4 needed per lane.  Every lane still consumes exactly its own requests in its own order
    // (counter = (request, sample, pixel)), so nothing changes in the image.
    const bool do_regen = live && need_sample;
    const bool do_scat = live && pending_hit;
    // ---- export: a lane about to start a sample with its item's segment budget spent hands the samples
    // AFTER this one to its workgroup (see "exported samples" above)
#ifndef RTOW_NO_RING
    {
      const bool exp = do_regen && (item & kExported) == 0u && s_left >= 2 && nseg >= seg_limit;
      if (__ballot(exp) != 0ull) {
        if (exp) {
          const uint32_t r = (uint32_t)s_left - 1u;
          const uint32_t idx = atomicAdd(&rtow_wg_head, r);
          const uint32_t cap = P.ovf_cap;
          const uint32_t n = idx >= cap ? 0u : (cap - idx < r ? cap - idx : r);  // (a full ring: the owner traces on)
          const uint32_t base = blockIdx.x * cap + idx;
          const uint32_t first = g.sample + (uint32_t)s_left - n;  // the LAST n samples of the item
          for (uint32_t i = 0; i < n; ++i) {
            uint32_t *dp = P.ovf_desc + (size_t)(base + i) * 4;
            dp[0] = g.pixel;
            dp[1] = first + i;
            __hip_atomic_store(dp + 2, P.ovf_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          if (n != 0u) {
            reinterpret_cast<unsigned long long *>(P.partials)[(size_t)item * 4 + 3] = ((unsigned long long)base << 32) | n;
            s_left -= (int)n;
          }
          seg_limit = 0xffffffffu;  // once per item
        }
        // (producer and consumers of a ring are waves of ONE workgroup, i.e. of one CU and its L1: workgroup
        // scope.  Agent scope costs an L2 write-back / invalidate per export and per claim on this chip — its
        // XCDs do not share an L2 — and measured 3x the whole launch.)
        __threadfence_block();
      }
    }
#endif
    V3 where = {0, 0, 0}, normal = {0, 0, 0};
    bool front = true;
    int mi = 0, kind = 0;
    real m_fuzz = 0, m_ir = 0;
#ifdef RTOW_FAST_MATH
    V3 m_att = {1, 1, 1};
#endif
    if (do_scat) {
      // rebuild the Hit of the winner (src/common-model.cpp:83-90, :121).  The walking
      // kernels read the winner's record, its material index and the material from the
      // LDS scene image (a chain of three dependent loads: LDS latency, not L2's)
      where = ro + rd * best.t;
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
          normal = normalize(where - center);
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
          normal = normalize(where - center);
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
    // first block of the sample / of the bounce
    uint32_t o0 = 0u, o1 = 0u, o2 = 0u, o3 = 0u;
    if (do_regen) g.r = 0u;
    if (do_regen || do_scat) {
      philox4x32(g.r, g.sample, g.pixel, 0u, k0, k1, o0, o1, o2, o3);
      g.r += 1u;
    }
    if constexpr (STAMPS) {  // wave-level count of block evaluations: the first active lane reports
      if (__ballot(do_regen || do_scat) != 0ull) stamps.blocks += 1;
    }
    real ju = 0, jv = 0, jt = 0, px = 0, py = 0;  // new sample: jitter, shutter time, lens-disk candidate
    V3 rnd = {0, 0, 0}, dirbase = {0, 0, 0};      // bounce: unit-ball candidate, direction before the fuzz term
    bool rej = false;
    if (do_regen) {
      // disk sample: y draws first (random-utils.cpp:36).  The first candidate comes with the jitter block.
      real c0, c1;
      jitter_from_block(o0, o1, o2, o3, ju, jv, jt, c0, c1);
      py = c0 * (real(1.0) - real(-1.0)) + real(-1.0);
      px = c1 * (real(1.0) - real(-1.0)) + real(-1.0);
      rej = px * px + py * py + real(0.0) * real(0.0) >= real(1.0);
    }
    if (do_scat) {
      // the first unit-ball candidate of the bounce comes with the dielectric coin (word 2)
      const real coin = (real)o2 * real(0x1p-32);
      rnd = ball_from_pair(o0, o1);
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
          refl = R > coin;
        }
        dirbase = refl ? reflect(unit, normal) : refract(unit, normal, ratio);
      } else if (kind == 1) {
        dirbase = reflect(rd, normal);
      }
      rej = dot(rnd, rnd) >= real(1.0);
    }
    // rejection sampling (random-utils.cpp:23-41): every further block carries two candidates
    uint32_t rej_trips = 0u;  // (diagnostic build only)
    while (rej) {
      if constexpr (STAMPS) ++rej_trips;
      philox4x32(g.r, g.sample, g.pixel, 0u, k0, k1, o0, o1, o2, o3);
      g.r += 1u;
      if (do_regen) {
        const real s32 = real(0x1p-32);
        py = ((real)o0 * s32) * (real(1.0) - real(-1.0)) + real(-1.0);
        px = ((real)o1 * s32) * (real(1.0) - real(-1.0)) + real(-1.0);
        if (px * px + py * py + real(0.0) * real(0.0) >= real(1.0)) {
          py = ((real)o2 * s32) * (real(1.0) - real(-1.0)) + real(-1.0);
          px = ((real)o3 * s32) * (real(1.0) - real(-1.0)) + real(-1.0);
        }
        rej = px * px + py * py + real(0.0) * real(0.0) >= real(1.0);
      } else {
        const V3 ca = ball_from_pair(o0, o1), cb = ball_from_pair(o2, o3);
        rnd = dot(ca, ca) >= real(1.0) ? cb : ca;
        rej = dot(rnd, rnd) >= real(1.0);
      }
    }
    if constexpr (STAMPS) {  // what the wave ran: as many trips as its unluckiest lane needed
      uint32_t mx = rej_trips, sum = rej_trips;
      for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t m2 = (uint32_t)__shfl_xor((int)mx, off), s2 = (uint32_t)__shfl_xor((int)sum, off);
        mx = mx > m2 ? mx : m2;
        sum += s2;
      }
      stamps.blocks += mx;
      stamps.block_lanes += sum;
    }
    if (do_regen) {
      // src/render.cpp:158-159, src/common-model.cpp:156-167
      const int from_top_i = P.H - (int)gi - 1;
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
      V3 dir;
      if (kind == 0) {
        absorbed = rabs(normal.x - rnd.x) < real(1e-8) && rabs(normal.y - rnd.y) < real(1e-8) &&
                   rabs(normal.z - rnd.z) < real(1e-8);
        dir = normal + rnd;
      } else {
        dir = dirbase + m_fuzz * rnd;
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
        ro = where;
        rd = dir;
      }
    }
    const bool tracing = live && !need_sample;  // has a ray to advance in this trip

    stamps.mark(RG_REGEN, __ballot(do_regen || do_scat));
    if constexpr (STAMPS) {
      stamps.trips += 1;
      if (t_empty != 0ull) trips_after_empty += 1;
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
      best = closest_hit_grid<LDS, STAMPS>(im, sc, ro, rd, rtime, tracing, nnode, nprim, stamps, best, t_resume, P.walk_cap,
                                           P.walk_max_open, P.leaf_votes);
    } else if constexpr (KERNEL == 2) {
      // the walk uses wave votes, so every lane of the wave enters it
      best = closest_hit_bvh<LDS, STAMPS>(im, sc, ro, rd, rtime, tracing, nnode, nprim, stamps);
    } else if constexpr (KERNEL == 5) {
#ifndef RTOW_FAST_MATH
      if (tracing) best = closest_hit_reftree(sc, to_f64(ro), to_f64(rd), (double)rtime, nnode, nprim);
#endif
    } else {
      if (tracing) best = closest_hit_stream(sc, to_f64(ro), to_f64(rd), (double)rtime);
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
      for (int r = 0; r < RG_COUNT; ++r) atomicAdd(&P.counters[23 + r], stamps.lt[r]);
      atomicAdd(&P.counters[34], stamps.step_lanes);
      atomicAdd(&P.counters[44], stamps.blocks);
      atomicAdd(&P.counters[36], stamps.block_lanes);
      atomicAdd(&P.counters[35], stamps.leaf_lanes);
      atomicAdd(&P.counters[13], stamps.iters);
      atomicAdd(&P.counters[14], stamps.trips);
      atomicAdd(&P.counters[15], stamps.phases);
    }
  }

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

template <bool L, bool S>
static int launch_sm4(const TraceParams &p, int grid, int block, unsigned lds_bytes, hipStream_t st) {
  auto k = RTOW_CAT(rtow_trace4_, RTOW_SUFFIX)<L, S>;
  if (lds_bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds_bytes, st, p);
  return (int)hipGetLastError();
}

// kernel: 1 STREAM, 2 BVH, 3 GRID, 4 BVH4; +16 = diagnostic region stamps (LDS variants only)
template <int K, bool L, bool S>
static int launch_one(const TraceParams &p, int grid, int block, unsigned lds_bytes, hipStream_t st) {
  auto k = RTOW_CAT(rtow_trace_, RTOW_SUFFIX)<K, L, S>;
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
    case 1: return launch_one<1, false, false>(p, grid, block, 0, st);
    case 2: return lds ? launch_one<2, true, false>(p, grid, block, lds_bytes, st)
                       : launch_one<2, false, false>(p, grid, block, 0, st);
    case 3: return lds ? launch_one<3, true, false>(p, grid, block, lds_bytes, st)
                       : launch_one<3, false, false>(p, grid, block, 0, st);
    case 4:
    case 4 + 16: {
      const bool full = p.sc.b4_lds_limit == p.sc.blob4_bytes, stamps = kernel == 4 + 16;
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

