// rtow_build.hip — device-side build of the BVH kernel's scene image.
//
// Replaces, for the device, what the reference does on the host in BVHNode's constructor
// (src/render.cpp:73-110: recursive median split, one std::sort per level).  The device tree
// is a linear BVH: Morton codes of the primitive centroids, one radix sort, Karras' binary
// radix tree (every inner node found independently), a bottom-up pass that unions boxes and
// counts subtree sizes, and an emit pass that writes the SAME threaded depth-first node
// records the host builder (rtow_bvh.h) produces.  The closest hit does not depend on the
// tree (rtow_bvh.h header), so images are bit-identical with either builder; what changes is
// build time (one launch sequence of ~0.1 ms instead of a host SAH sweep) and tree quality.
//
// All kernels are index-checked against `n`; the emitted links are validated on the device
// before the trace kernel may use the image (same rules as validate_scene_image()).
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <hip/hip_fp16.h>
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <utility>

#include "rtow_device.h"

namespace rtow {

namespace {

constexpr int kLeafBit = (int)0x80000000u;

struct Scratch {
  int n = 0;         // primitives of the current build
  int capacity = 0;  // primitives the buffers can hold
  float *pbox = nullptr;               // [n][6] padded f32 primitive boxes
  double *pbox64 = nullptr;            // [n][6] exact f64 bounds
  unsigned long long *keys_a = nullptr, *keys_b = nullptr;
  uint32_t *vals_a = nullptr, *vals_b = nullptr;
  int32_t *child_l = nullptr, *child_r = nullptr, *first = nullptr, *last = nullptr;
  int32_t *parent_int = nullptr, *parent_leaf = nullptr;
  float *ibox = nullptr;               // [n-1][6]
  int32_t *size = nullptr;             // [n-1] emitted records in the subtree
  uint32_t *flags = nullptr;           // [n-1] arrival counters; later: near-child-first bit
  uint32_t *glob = nullptr;            // [8]: 0..2 centroid min, 3..5 centroid max (ordered u32), 6 scale bits, 7 error flag
  void *sort_tmp = nullptr;
  size_t sort_tmp_bytes = 0;
  // 4-wide collapse (triangle meshes): per 4-wide node, breadth-first
  int32_t *b4_src = nullptr;    // [capacity] radix-tree node each 4-wide node expands
  int32_t *b4_child = nullptr;  // [capacity][4] the children as radix-tree refs (kB4Empty = none)
  uint32_t *b4_cw = nullptr;    // [capacity][4] child words of the image (rtow_bvh4.h: ref21)
  uint32_t *b4_cnt = nullptr, *b4_pos = nullptr;  // [capacity] inner children per node of a level, their exclusive scan
  uint32_t *b4_seen = nullptr;  // [2 * capacity] validation: references per triangle record and per node
  void *scan_tmp = nullptr;
  size_t scan_tmp_bytes = 0;
  int b4_nodes = 0, b4_depth = 0;
  // PLOC (round 5): the cluster array in two copies (refs, boxes), nearest neighbours, the packed keep / merge flags
  // and their scan, subtree leaf counts, leaf positions in depth-first order
  int32_t *pl_ref[2] = {nullptr, nullptr};
  float *pl_box[2] = {nullptr, nullptr};
  int32_t *pl_nn = nullptr;
  unsigned long long *pl_flag = nullptr, *pl_scan = nullptr;
  int32_t *pl_cnt = nullptr, *pl_pos = nullptr, *pl_parent_leaf = nullptr;
  uint32_t *pl_state = nullptr;  // [0] clusters in the array, [1] nodes created so far
  void *pl_scan_tmp = nullptr;
  size_t pl_scan_tmp_bytes = 0;
  // binned SAH, top-down on the device (round 5, pass 3c): the primitives in two copies (the order is partitioned level by
  // level), the node each position belongs to, per-node ranks in the level's work list, the work lists, per-rank
  // centroid bounds / bins / split decisions, the partition's flags and their scan
  uint32_t *sh_item[2] = {nullptr, nullptr};
  int32_t *sh_node[2] = {nullptr, nullptr};
  int32_t *sh_rank = nullptr, *sh_list[2] = {nullptr, nullptr};
  uint32_t *sh_cnt = nullptr;       // [kSahMaxLevels + 2] nodes per level
  uint32_t *sh_cb = nullptr;        // [ranks][6] centroid bounds (ordered u32)
  uint32_t *sh_bin = nullptr;       // [ranks][48][7]: count, box min[3], box max[3] (ordered u32) per axis and bin
  int32_t *sh_split = nullptr;      // [ranks][4]: axis, last bin of the left side, primitives on the left, mode
  uint32_t *sh_flag = nullptr, *sh_scan = nullptr;
  void *sh_scan_tmp = nullptr;
  size_t sh_scan_tmp_bytes = 0;
  size_t sh_ranks = 0;
};
constexpr int kSahMaxLevels = 160;   // hard bound on the depth of the split; from kSahMedianFrom on every split halves its range
constexpr int kSahMedianFrom = 64;
constexpr int kSahBins = 16;

__device__ __forceinline__ uint32_t ordered(float f) {
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float unordered(uint32_t o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// ---- pass 1: exact primitive bounds (f64), scene scale, centroid bounds -----------------
__global__ void k_bounds(const double *sph, const double *sph_r, const double *mov, const double *tri, int ns,
                         int nm, int nt, double time0, double time1, double *pbox64, uint32_t *glob) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = ns + nm + nt;
  // scene-wide minima/maxima: LDS atomics per workgroup, then 7 global atomics per workgroup
  __shared__ uint32_t sh[7];
  if (threadIdx.x < 7) sh[threadIdx.x] = threadIdx.x < 3 ? 0xffffffffu : 0u;
  __syncthreads();
  if (i < n) {
  double mn[3], mx[3];
  if (i < ns) {
    const double r = fabs(sph_r[i]);
    for (int k = 0; k < 3; ++k) {
      mn[k] = sph[(size_t)i * 4 + k] - r;
      mx[k] = sph[(size_t)i * 4 + k] + r;
    }
  } else if (i < ns + nm) {
    // centre(time) = c0 + time*delta over the shutter interval, widened a little (rtow_bvh.h)
    const double *m = mov + (size_t)(i - ns) * 8;
    const double r = fabs(m[7]);
    const double w = 1e-6 * (1.0 + fabs(time0) + fabs(time1));
    const double ta = fmin(time0, time1) - w, tb = fmax(time0, time1) + w;
    for (int k = 0; k < 3; ++k) {
      const double a0 = m[k] + ta * m[3 + k], a1 = m[k] + tb * m[3 + k];
      mn[k] = fmin(a0, a1) - r;
      mx[k] = fmax(a0, a1) + r;
    }
  } else {
    const double *t = tri + (size_t)(i - ns - nm) * 12;
    for (int k = 0; k < 3; ++k) {
      const double a = t[k], b = t[k] + t[3 + k], c = t[k] + t[6 + k];
      mn[k] = fmin(a, fmin(b, c));
      mx[k] = fmax(a, fmax(b, c));
    }
  }
  float amax = 0.0f;
  for (int k = 0; k < 3; ++k) {
    // slack ≫ any rounding in a f64 slab test (pad_box in rtow_bvh.h)
    const double ext = fmax(fabs(mn[k]), fabs(mx[k]));
    const double pad = 1e-9 * (1.0 + ext);
    mn[k] -= pad;
    mx[k] += pad;
    pbox64[(size_t)i * 6 + k] = mn[k];
    pbox64[(size_t)i * 6 + 3 + k] = mx[k];
    amax = fmaxf(amax, __double2float_ru(fmax(fabs(mn[k]), fabs(mx[k]))));
    const double c = 0.5 * (mn[k] + mx[k]);
    atomicMin(&sh[k], ordered(__double2float_rd(c)));
    atomicMax(&sh[3 + k], ordered(__double2float_ru(c)));
  }
  atomicMax(&sh[6], __float_as_uint(amax));  // non-negative floats order like their bits
  }
  __syncthreads();
  if (threadIdx.x < 3) atomicMin(&glob[threadIdx.x], sh[threadIdx.x]);
  else if (threadIdx.x < 7) atomicMax(&glob[threadIdx.x], sh[threadIdx.x]);
}

__device__ __forceinline__ unsigned long long spread21(unsigned long long v) {  // 21 bits -> every third bit
  v &= 0x1fffffull;
  v = (v | (v << 32)) & 0x1f00000000ffffull;
  v = (v | (v << 16)) & 0x1f0000ff0000ffull;
  v = (v | (v << 8)) & 0x100f00f00f00f00full;
  v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}

// ---- pass 2: padded f32 boxes (conservative for the f32 slab test) + 63-bit Morton keys ----
__global__ void k_morton(const double *pbox64, int n, double cam_scale, const uint32_t *glob, float *pbox,
                         unsigned long long *keys, uint32_t *vals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // the f32 slab test sees the ray origin and the planes rounded to f32: pad every box by more
  // than that rounding can move a plane or an origin anywhere in the scene (make_scene_image)
  const double scale = fmax(fmax((double)__uint_as_float(glob[6]), cam_scale), 1.0);
  unsigned long long key = 0;
  for (int k = 0; k < 3; ++k) {
    const double lo = pbox64[(size_t)i * 6 + k], hi = pbox64[(size_t)i * 6 + 3 + k];
    const double pad = 2e-6 * scale + 2e-6 * fmax(fabs(lo), fabs(hi));
    pbox[(size_t)i * 6 + k] = nextafterf(__double2float_rd(lo - pad), -INFINITY);
    pbox[(size_t)i * 6 + 3 + k] = nextafterf(__double2float_ru(hi + pad), INFINITY);
    const double cmin = (double)unordered(glob[k]), cmax = (double)unordered(glob[3 + k]);
    const double ext = cmax - cmin;
    const double c = 0.5 * (lo + hi);
    double u = ext > 0.0 ? (c - cmin) / ext : 0.0;
    u = fmin(fmax(u, 0.0), 1.0);
    const unsigned long long q = (unsigned long long)fmin(u * 2097152.0, 2097151.0);
    key |= spread21(q) << (2 - k);
  }
  keys[i] = key;
  vals[i] = (uint32_t)i;
}

// ---- pass 3: Karras' binary radix tree over the sorted keys -------------------------------
__device__ __forceinline__ int delta(const unsigned long long *keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const unsigned long long a = keys[i], b = keys[j];
  if (a == b) return 64 + __clz((unsigned)(i ^ j));  // equal keys: the index breaks the tie
  return __clzll((long long)(a ^ b));
}

__global__ void k_radix_tree(const unsigned long long *keys, int n, int32_t *child_l, int32_t *child_r,
                             int32_t *first, int32_t *last, int32_t *parent_int, int32_t *parent_leaf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  const int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
  const int dmin = delta(keys, n, i, i - d);
  int lmax = 2;
  while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;  // i + lmax*d leaves [0,n) after <= log2(n)+1 doublings
  int l = 0;
  for (int t = lmax / 2; t >= 1; t /= 2)
    if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = delta(keys, n, i, j);
  int s = 0, t = l;
  do {
    t = (t + 1) / 2;
    if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
  } while (t > 1);
  const int gamma = i + s * d + (d < 0 ? -1 : 0);
  const int lo = i < j ? i : j, hi = i < j ? j : i;
  const bool leaf_l = lo == gamma, leaf_r = hi == gamma + 1;
  child_l[i] = leaf_l ? (gamma | kLeafBit) : gamma;
  child_r[i] = leaf_r ? ((gamma + 1) | kLeafBit) : gamma + 1;
  first[i] = lo;
  last[i] = hi;
  if (leaf_l) parent_leaf[gamma] = i; else parent_int[gamma] = i;
  if (leaf_r) parent_leaf[gamma + 1] = i; else parent_int[gamma + 1] = i;
  if (i == 0) parent_int[0] = -1;
}

// ---- pass 3b (round 5): the same arrays by parallel locally-ordered clustering --------------------------------
// Karras' tree splits a range where the Morton keys' common prefix ends — the middle of an octree cell, wherever the
// triangles are; the 4-wide tree collapsed from it walks 12-20 % slower than the host's SAH tree (17-18 % more
// expected node visits by the surface-area metric, scripts/experiments/ploc_proto.cpp).  PLOC (Meister & Bittner,
// "Parallel Locally-Ordered Clustering for Bounding Volume Hierarchy Construction", 2018) builds the tree bottom-up
// over the same Morton order: every cluster of the current array looks `radius` places to either side for the
// neighbour whose union with it has the smallest surface area; two clusters that choose each other merge into a new
// node, which takes the lower one's place; the array is compacted in order; repeat until one cluster is left.  On
// the project's meshes the metric comes within 1 % (suzanne) and 9 % (96.8k triangles) of the host tree's.
// One iteration = nearest-neighbour kernel, flag kernel, one scan (survivors and merges in one 64-bit word), compact
// kernel; the cluster count stays on the device and is read back every fourth iteration to shrink the launches.
// Determinism: ties go to the lower index, node ids come from the scan (the k-th merge of the build is node
// n - 2 - k, so the last one — the root — is node 0, as the rest of the pipeline expects), nothing depends on timing.
// A PLOC node covers clusters that were neighbours in the array, not a contiguous range of the Morton order, and
// the image's leaves are ranges of the triangle records: the leaves are therefore renumbered in depth-first order
// afterwards (k_ploc_count, k_ploc_place, k_ploc_relabel) and `vals` becomes that order.
__device__ __forceinline__ float union_area(const float *a, const float *b) {
  const float dx = fmaxf(a[3], b[3]) - fminf(a[0], b[0]), dy = fmaxf(a[4], b[4]) - fminf(a[1], b[1]),
              dz = fmaxf(a[5], b[5]) - fminf(a[2], b[2]);
  return dx * dy + dy * dz + dz * dx;
}
__global__ void k_ploc_init(int n, const uint32_t *vals, const float *pbox, int32_t *ref, float *box, uint32_t *state) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j == 0) {
    state[0] = (uint32_t)n;
    state[1] = 0u;
  }
  if (j >= n) return;
  ref[j] = j | kLeafBit;
  for (int k = 0; k < 6; ++k) box[(size_t)j * 6 + k] = pbox[(size_t)vals[j] * 6 + k];
}
__global__ void k_ploc_nn(const uint32_t *state, int radius, const float *box, int32_t *nn) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int m = (int)state[0];
  if (i >= m || m < 2) return;  // (the host's bound on the array may lag behind: one cluster left = nothing to do)
  float me[6];
  for (int k = 0; k < 6; ++k) me[k] = box[(size_t)i * 6 + k];
  float best = INFINITY;
  int bj = i > 0 ? i - 1 : i + 1;  // (boxes with NaN or inf areas: still a neighbour; m >= 2 here)
  const int j0 = max(i - radius, 0), j1 = min(i + radius, m - 1);
  for (int j = j0; j <= j1; ++j) {
    if (j == i) continue;
    const float a = union_area(me, box + (size_t)j * 6);
    if (a < best) {  // ties: the lower index
      best = a;
      bj = j;
    }
  }
  nn[i] = bj;
}
// low word: this slot survives (everything but the higher cluster of a merging pair); high word: this slot is the
// lower cluster of a merging pair (a node is created in its place).  Slots beyond the array are zero for the scan.
__global__ void k_ploc_flags(int bound, const uint32_t *state, const int32_t *nn, unsigned long long *flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= bound) return;
  const int m = (int)state[0];
  unsigned long long f = 0ull;
  if (m < 2) {  // the tree is complete (the host's bound lags behind the device's count): the cluster stays, no merge
    f = i < m ? 1ull : 0ull;
  } else if (i < m) {
    const int j = nn[i];
    const bool mutual = nn[j] == i;
    f = (mutual && i > j) ? 0ull : 1ull;
    if (mutual && i < j) f |= 1ull << 32;
  }
  flag[i] = f;
}
__global__ void k_ploc_compact(int n, int bound, uint32_t *state_next, const uint32_t *state, const int32_t *nn,
                               const unsigned long long *flag, const unsigned long long *scan, const int32_t *ref,
                               const float *box, int32_t *ref_out, float *box_out, int32_t *child_l, int32_t *child_r,
                               int32_t *parent_int, int32_t *parent_leaf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int m = (int)state[0];
  if (i >= bound || i >= m) return;
  const unsigned long long f = flag[i], sc = scan[i];
  const uint32_t pos = (uint32_t)sc, rank = (uint32_t)(sc >> 32);
  if (i == m - 1) {  // the totals: what the next iteration works on
    state_next[0] = pos + (uint32_t)(f & 1ull);
    state_next[1] = state[1] + rank + (uint32_t)(f >> 32);
  }
  if (!(f & 1ull)) return;
  float b[6];
  for (int k = 0; k < 6; ++k) b[k] = box[(size_t)i * 6 + k];
  int32_t r = ref[i];
  if (f >> 32) {
    const int j = nn[i];
    const int32_t id = (n - 2) - (int32_t)(state[1] + rank);
    const int32_t rl = r, rr = ref[j];
    child_l[id] = rl;
    child_r[id] = rr;
    if (rl < 0) parent_leaf[rl & 0x7fffffff] = id; else parent_int[rl] = id;
    if (rr < 0) parent_leaf[rr & 0x7fffffff] = id; else parent_int[rr] = id;
    const float *o = box + (size_t)j * 6;
    for (int k = 0; k < 3; ++k) {
      b[k] = fminf(b[k], o[k]);
      b[3 + k] = fmaxf(b[3 + k], o[3 + k]);
    }
    r = id;
  }
  ref_out[pos] = r;
  for (int k = 0; k < 6; ++k) box_out[(size_t)pos * 6 + k] = b[k];
}
// leaves under every node (bottom-up, the second arrival at a node finishes it: the pattern of k_refit)
__global__ void k_ploc_count(int n, const int32_t *child_l, const int32_t *child_r, const int32_t *parent_int,
                             const int32_t *parent_leaf, int32_t *cnt, uint32_t *flags) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  int node = parent_leaf[p];
  while (node >= 0) {
    __threadfence();
    const uint32_t old = atomicAdd(&flags[node], 1u);
    if (old == 0u) return;
    __threadfence();
    int c = 0;
    for (const int32_t ch : {child_l[node], child_r[node]})
      c += ch < 0 ? 1 : __hip_atomic_load(cnt + ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(cnt + node, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    node = parent_int[node];
  }
}
// depth-first position of the first leaf under `ref` (left child first): the leaves to the left of the path to the root
__device__ __forceinline__ int ploc_first(int32_t ref, int32_t parent, const int32_t *child_l, const int32_t *parent_int,
                                          const int32_t *cnt) {
  int pos = 0;
  int32_t cur = ref, par = parent;
  while (par >= 0) {
    const int32_t l = child_l[par];
    if (l != cur) pos += l < 0 ? 1 : cnt[l];
    cur = par;
    par = parent_int[par];
  }
  return pos;
}
__global__ void k_ploc_place(int n, const int32_t *child_l, const int32_t *parent_int, const int32_t *parent_leaf,
                             const int32_t *cnt, const uint32_t *vals, uint32_t *vals_out, int32_t *pos, int32_t *first,
                             int32_t *last) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * n - 1) return;
  if (t < n) {  // leaf at Morton position t
    const int q = ploc_first(t | kLeafBit, parent_leaf[t], child_l, parent_int, cnt);
    pos[t] = q;
    vals_out[q] = vals[t];
  } else {  // inner node
    const int id = t - n;
    const int q = ploc_first(id, parent_int[id], child_l, parent_int, cnt);
    first[id] = q;
    last[id] = q + cnt[id] - 1;
  }
}
// leaf references and the leaves' parents in the new numbering
__global__ void k_ploc_relabel(int n, const int32_t *pos, const int32_t *parent_leaf_old, int32_t *child_l, int32_t *child_r,
                               int32_t *parent_leaf) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * n - 1) return;
  if (t < n) {
    parent_leaf[pos[t]] = parent_leaf_old[t];
  } else {
    const int id = t - n;
    const int32_t l = child_l[id], r = child_r[id];
    if (l < 0) child_l[id] = pos[l & 0x7fffffff] | kLeafBit;
    if (r < 0) child_r[id] = pos[r & 0x7fffffff] | kLeafBit;
  }
}

// ---- pass 3c (round 5): the same arrays by a binned surface-area-heuristic build, top-down, on the device -------------
// What the host builder does (rtow_bvh.h: 16 bins per axis over the node's centroid bounds, the split that minimises
// area(L) n(L) + area(R) n(R) over the three axes, a partition of the node's primitives), level by level for all nodes at
// once.  The prototype (scripts/experiments/ploc_proto.cpp) says where the device trees of this round fell short: PLOC
// followed by an exact SAH over its last 2,048 / 8,192 / 32,768 clusters closes 27 / 44 / 82 % of its gap to the host
// tree — the loss is in the top-down partition, not in the leaves; local restructuring (tree rotations) recovers nothing.
// A node owns a RANGE [lo, hi] of the primitive order; a level partitions every active range in place (stable: one
// exclusive scan over "goes right" flags), so ranges nest and are final from the moment they exist — Karras' numbering
// applies (root 0; a left child is numbered by the last position of its range, a right child by the first), `first` /
// `last` are the range, the leaves of a subtree are contiguous in the final order and no renumbering pass is needed.
// One level = centroid bounds per node (atomics on order-preserving integers, one per wave when the wave lies inside
// one node), bins per node (21 atomics per primitive), one thread per node choosing the split and creating the children,
// the side of every primitive, ONE rocprim scan, the scatter.  Deterministic: the tree depends on no atomic's order
// (sums of integers, minima and maxima), ties go to the lower axis and bin.  A node whose centroids coincide, and
// every node from level kSahMedianFrom on, is split in the middle of its range.
__device__ __forceinline__ float sah_centroid(const float *b, int k) { return 0.5f * (b[k] + b[3 + k]); }
__device__ __forceinline__ int sah_bin_of(float c, float cmin, float cmax) {
  const float scale = (float)kSahBins / (cmax - cmin);  // (cmax > cmin where this is called)
  const int b = (int)((c - cmin) * scale);
  return b < 0 ? 0 : (b > kSahBins - 1 ? kSahBins - 1 : b);
}
__global__ void k_sah_init(int n, const uint32_t *vals, uint32_t *item, int32_t *node, int32_t *first, int32_t *last,
                           int32_t *parent_int, int32_t *rank, int32_t *list, uint32_t *cnt, uint32_t *cb, uint32_t *bin,
                           size_t ranks) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j == 0) {
    first[0] = 0;
    last[0] = n - 1;
    parent_int[0] = -1;
    rank[0] = 0;
    list[0] = 0;
    cnt[0] = 1u;
  }
  if (j < (size_t)n) {
    item[j] = vals[j];
    node[j] = 0;
  }
  if (j < ranks * 6) cb[j] = (j % 6) < 3 ? 0xffffffffu : 0u;
  for (size_t k = j; k < ranks * 48 * 7; k += (size_t)gridDim.x * blockDim.x) {
    const size_t f = k % 7;
    bin[k] = f == 0 ? 0u : (f < 4 ? 0xffffffffu : 0u);
  }
}
__global__ void k_sah_cb(int n, const uint32_t *item, const int32_t *node, const int32_t *rank, const float *pbox, uint32_t *cb) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int nd = j < n ? node[j] : -1;
  float c[3] = {0.f, 0.f, 0.f};
  if (nd >= 0) {
    const float *b = pbox + (size_t)item[j] * 6;
    for (int k = 0; k < 3; ++k) c[k] = sah_centroid(b, k);
  }
  // a wave that lies inside one node (every wave of the top levels) reduces first: six atomics per wave
  const int nd0 = __shfl(nd, 0);
  const bool uniform = __all(nd == nd0);
  if (uniform) {
    if (nd0 < 0) return;
    float mn[3] = {c[0], c[1], c[2]}, mx[3] = {c[0], c[1], c[2]};
    for (int off = 32; off >= 1; off >>= 1)
      for (int k = 0; k < 3; ++k) {
        mn[k] = fminf(mn[k], __shfl_xor(mn[k], off));
        mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], off));
      }
    if ((threadIdx.x & 63u) == 0u) {
      uint32_t *d = cb + (size_t)rank[nd0] * 6;
      for (int k = 0; k < 3; ++k) {
        atomicMin(&d[k], ordered(mn[k]));
        atomicMax(&d[3 + k], ordered(mx[k]));
      }
    }
  } else if (nd >= 0) {
    uint32_t *d = cb + (size_t)rank[nd] * 6;
    for (int k = 0; k < 3; ++k) {
      atomicMin(&d[k], ordered(c[k]));
      atomicMax(&d[3 + k], ordered(c[k]));
    }
  }
}
// Bins.  A WAVE whose 64 positions all lie inside one node (nodes are ranges, so: first and last position in the same
// node — every wave of the first ten levels or so, where the primitives of a node would otherwise hammer the same 336
// words) accumulates in its own slice of LDS and adds its non-empty bins to the node's: the 96,800-triangle mesh's first
// levels took 2 ms each with one global atomic per primitive, axis and field.  Other waves (deep levels: many small
// nodes, no contention) use one global atomic each.
constexpr int kSahBinBlock = 1024;
__global__ void __launch_bounds__(kSahBinBlock)
    k_sah_bin(int n, const uint32_t *item, const int32_t *node, const int32_t *rank, const float *pbox, const uint32_t *cb,
              uint32_t *bin) {
  __shared__ uint32_t sh_all[(kSahBinBlock / 64) * 48 * 7];
  const int lane = (int)(threadIdx.x & 63u);
  uint32_t *sh = sh_all + (threadIdx.x >> 6) * (48 * 7);
  const int j = blockIdx.x * kSahBinBlock + (int)threadIdx.x;
  const int w0 = j - lane, w1 = min(w0 + 63, n - 1);
  const int nd_first = w0 < n ? node[w0] : -1, nd_last = w0 < n ? node[w1] : -1;
  // (wave-uniform.  Two finished positions at the ends of a wave say nothing about its middle.)
  const bool one_node = nd_first >= 0 && nd_first == nd_last;
  if (one_node)
    for (int k = lane; k < 48 * 7; k += 64) sh[k] = (k % 7) == 0 ? 0u : ((k % 7) < 4 ? 0xffffffffu : 0u);
  __syncthreads();
  const int nd = j < n ? node[j] : -1;
  if (nd >= 0) {
    const size_t r = (size_t)rank[nd];
    const float *b = pbox + (size_t)item[j] * 6;
    for (int a = 0; a < 3; ++a) {
      const float cmin = unordered(cb[r * 6 + a]), cmax = unordered(cb[r * 6 + 3 + a]);
      if (!(cmax > cmin)) continue;  // (no split along an axis on which the centroids coincide)
      const int slot = (a * kSahBins + sah_bin_of(sah_centroid(b, a), cmin, cmax)) * 7;
      uint32_t *d = one_node ? sh + slot : bin + r * 48 * 7 + slot;
      atomicAdd(&d[0], 1u);
      for (int k = 0; k < 3; ++k) {
        atomicMin(&d[1 + k], ordered(b[k]));
        atomicMax(&d[4 + k], ordered(b[3 + k]));
      }
    }
  }
  __syncthreads();
  if (one_node) {
    uint32_t *g = bin + (size_t)rank[nd_first] * 48 * 7;
    for (int k = lane; k < 48 * 7; k += 64) {
      const int f = k % 7;
      const uint32_t v = sh[k];
      if (f == 0) {
        if (v != 0u) atomicAdd(&g[k], v);
      } else if (sh[k - f] != 0u) {  // (a bin that got primitives)
        if (f < 4) atomicMin(&g[k], v); else atomicMax(&g[k], v);
      }
    }
  }
}
// One WAVE per node of the level: the split (lane = axis x bin: the 45 candidate splits are evaluated side by side, the
// sums over the bins to the left and to the right of a split by scans inside each axis' sixteen lanes), then — lane 0 —
// the children and the next level's work list; resets the node's accumulators.  (One THREAD per node looped over 3 x 16
// bins of 7 words twice and reset 336 words: 56 us per level whatever the level's size, 1.35 ms of the 96,800-triangle
// mesh's 5.3 ms.)  The same minima, maxima, integer sums and products as the serial loop, ties to the lower axis and
// bin as there (= the lower lane): the same tree.
__global__ void k_sah_split(int level, int force_median, const int32_t *list, int32_t *list_next, uint32_t *cnt, int32_t *rank,
                            uint32_t *cb, uint32_t *bin, int32_t *split, int32_t *child_l, int32_t *child_r, int32_t *first,
                            int32_t *last, int32_t *parent_int, int32_t *parent_leaf) {
  const uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = (int)(threadIdx.x & 63u);
  if (t >= cnt[level]) return;  // (wave-uniform)
  const int nd = list[t];
  const int lo = first[nd], hi = last[nd], size = hi - lo + 1;
  int best_axis = -1, best_bin = 0, best_nl = size / 2;
  if (!force_median) {
    const int a = lane >> 4, b = lane & 15;
    const bool on = lane < 48;
    float cmin = 0.f, cmax = 0.f;
    if (on) cmin = unordered(cb[(size_t)t * 6 + a]), cmax = unordered(cb[(size_t)t * 6 + 3 + a]);
    // this lane's bin: count and box (identities when the bin is empty, as the serial loop skipped it)
    uint32_t c = 0u;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (on) {
      const uint32_t *B = bin + (((size_t)t * 3 + a) * kSahBins + b) * 7;
      c = B[0];
      if (c != 0u)
        for (int k = 0; k < 3; ++k) mn[k] = unordered(B[1 + k]), mx[k] = unordered(B[4 + k]);
    }
    // inclusive scans over the axis' sixteen lanes: bins 0 .. b (left of the split after bin b) and bins b .. 15
    uint32_t cl = c, cr = c;
    float lmn[3] = {mn[0], mn[1], mn[2]}, lmx[3] = {mx[0], mx[1], mx[2]};
    float rmn[3] = {mn[0], mn[1], mn[2]}, rmx[3] = {mx[0], mx[1], mx[2]};
    for (int off = 1; off < 16; off <<= 1) {
      const uint32_t ocl = __shfl_up(cl, off, 16), ocr = __shfl_down(cr, off, 16);
      float omn[3], omx[3], pmn[3], pmx[3];
      for (int k = 0; k < 3; ++k) {
        omn[k] = __shfl_up(lmn[k], off, 16), omx[k] = __shfl_up(lmx[k], off, 16);
        pmn[k] = __shfl_down(rmn[k], off, 16), pmx[k] = __shfl_down(rmx[k], off, 16);
      }
      if (b >= off) {
        cl += ocl;
        for (int k = 0; k < 3; ++k) lmn[k] = fminf(lmn[k], omn[k]), lmx[k] = fmaxf(lmx[k], omx[k]);
      }
      if (b + off < 16) {
        cr += ocr;
        for (int k = 0; k < 3; ++k) rmn[k] = fminf(rmn[k], pmn[k]), rmx[k] = fmaxf(rmx[k], pmx[k]);
      }
    }
    // the right side of the split after bin b is bins b + 1 .. 15: the inclusive suffix of the next lane
    const uint32_t crx = __shfl_down(cr, 1, 16);
    float xmn[3], xmx[3];
    for (int k = 0; k < 3; ++k) xmn[k] = __shfl_down(rmn[k], 1, 16), xmx[k] = __shfl_down(rmx[k], 1, 16);
    float cost = INFINITY;
    if (on && cmax > cmin && b < kSahBins - 1 && cl != 0u && crx != 0u) {
      const float dx = lmx[0] - lmn[0], dy = lmx[1] - lmn[1], dz = lmx[2] - lmn[2];
      const float ex = xmx[0] - xmn[0], ey = xmx[1] - xmn[1], ez = xmx[2] - xmn[2];
      const float r_area = ex * ey + ey * ez + ez * ex;
      cost = (dx * dy + dy * dz + dz * dx) * (float)cl + r_area * (float)crx;
    }
    // the cheapest split; among equals the lower axis, then the lower bin: the lower lane
    float bc = cost;
    int bl = lane;
    for (int off = 32; off >= 1; off >>= 1) {
      const float oc = __shfl_xor(bc, off);
      const int ol = __shfl_xor(bl, off);
      if (oc < bc || (oc == bc && ol < bl)) bc = oc, bl = ol;
    }
    const uint32_t nl_best = __shfl(cl, bl);
    if (bc < INFINITY) best_axis = bl >> 4, best_bin = bl & 15, best_nl = (int)nl_best;
    // the node's accumulators back to their identities (k_sah_bin of the next level adds into them)
    uint32_t *B0 = bin + (size_t)t * 48 * 7;
    for (int k = lane; k < 48 * 7; k += 64) {
      const int f = k % 7;
      B0[k] = f == 0 ? 0u : (f < 4 ? 0xffffffffu : 0u);
    }
  }
  if (lane != 0) return;
  int32_t *sp = split + (size_t)t * 4;
  sp[0] = best_axis;
  sp[1] = best_bin;
  sp[2] = best_nl;
  sp[3] = best_axis < 0 ? 1 : 0;  // mode 1: by position (the first best_nl of the range go left)
  // children: [lo, lo + nl - 1] and [lo + nl, hi]
  const int nl = best_nl, g = lo + nl - 1;
  const bool leaf_l = nl == 1, leaf_r = size - nl == 1;
  child_l[nd] = leaf_l ? (g | kLeafBit) : g;            // a left child is numbered by the last position of its range
  child_r[nd] = leaf_r ? ((g + 1) | kLeafBit) : g + 1;  // a right child by the first
  uint32_t k = 0;
  if (!leaf_l || !leaf_r) k = atomicAdd(&cnt[level + 1], (leaf_l ? 0u : 1u) + (leaf_r ? 0u : 1u));
  if (leaf_l) {
    parent_leaf[g] = nd;
  } else {
    parent_int[g] = nd;
    first[g] = lo;
    last[g] = g;
    rank[g] = (int32_t)k;
    list_next[k++] = g;
  }
  if (leaf_r) {
    parent_leaf[g + 1] = nd;
  } else {
    parent_int[g + 1] = nd;
    first[g + 1] = g + 1;
    last[g + 1] = hi;
    rank[g + 1] = (int32_t)k;
    list_next[k] = g + 1;
  }
}
// (the accumulators of the ranks used by this level back to their identities, after the side kernel has read them)
__global__ void k_sah_reset(int level, const uint32_t *cnt, uint32_t *cb, uint32_t *bin) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t used = cnt[level];
  if (i < used * 6) cb[i] = (i % 6) < 3 ? 0xffffffffu : 0u;
  if (i < used * 48 * 7) {
    const size_t f = i % 7;
    bin[i] = f == 0 ? 0u : (f < 4 ? 0xffffffffu : 0u);
  }
}
__global__ void k_sah_side(int n, const uint32_t *item, const int32_t *node, const int32_t *rank, const float *pbox,
                           const uint32_t *cb, const int32_t *split, const int32_t *first, uint32_t *flag) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int nd = node[j];
  uint32_t right = 0u;
  if (nd >= 0) {
    const size_t r = (size_t)rank[nd];
    const int32_t *sp = split + r * 4;
    if (sp[3] != 0) {
      right = (j - first[nd]) >= sp[2] ? 1u : 0u;
    } else {
      const int a = sp[0];
      const float cmin = unordered(cb[r * 6 + a]), cmax = unordered(cb[r * 6 + 3 + a]);
      right = sah_bin_of(sah_centroid(pbox + (size_t)item[j] * 6, a), cmin, cmax) > sp[1] ? 1u : 0u;
    }
  }
  flag[j] = right;
}
__global__ void k_sah_scatter(int n, const uint32_t *item, const int32_t *node, const int32_t *rank, const int32_t *split,
                              const int32_t *first, const int32_t *last, const uint32_t *flag, const uint32_t *scan,
                              uint32_t *item_out, int32_t *node_out, uint32_t *cb) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int nd = node[j];
  if (nd < 0) {  // a finished position keeps its primitive
    item_out[j] = item[j];
    node_out[j] = -1;
    return;
  }
  const int lo = first[nd], hi = last[nd];
  const int nl = split[(size_t)rank[nd] * 4 + 2], nr = hi - lo + 1 - nl;
  if (j == lo) {  // (k_sah_side, the last reader of this node's centroid bounds, has run)
    uint32_t *d = cb + (size_t)rank[nd] * 6;
    for (int k = 0; k < 3; ++k) d[k] = 0xffffffffu, d[3 + k] = 0u;
  }
  const int rights_before = (int)(scan[j] - scan[lo]);
  const bool right = flag[j] != 0u;
  const int pos = right ? lo + nl + rights_before : lo + (j - lo) - rights_before;
  item_out[pos] = item[j];
  node_out[pos] = right ? (nr >= 2 ? lo + nl : -1) : (nl >= 2 ? lo + nl - 1 : -1);
}
__global__ void k_sah_finish(int n, const uint32_t *item, uint32_t *vals) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) vals[j] = item[j];
}

// ---- pass 4: bottom-up boxes, emitted-subtree sizes, near-child-first bit ------------------
__global__ void k_refit(int n, int leaf_max, const uint32_t *vals, const float *pbox, const int32_t *child_l,
                        const int32_t *child_r, const int32_t *first, const int32_t *last,
                        const int32_t *parent_int, const int32_t *parent_leaf, float cx, float cy, float cz,
                        float *ibox, int32_t *size, uint32_t *flags) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  int node = parent_leaf[p];
  while (node >= 0) {
    __threadfence();
    const uint32_t old = atomicAdd(&flags[node], 1u);
    if ((old & 3u) == 0u) return;  // first arrival: the sibling's thread finishes this node
    __threadfence();
    float b[2][6];
    int sz[2];
    const int32_t ch[2] = {child_l[node], child_r[node]};
    for (int c = 0; c < 2; ++c) {
      const int idx = ch[c] & 0x7fffffff;
      const bool leaf = ch[c] < 0;
      const float *src = leaf ? pbox + (size_t)vals[idx] * 6 : ibox + (size_t)idx * 6;
      // written by another workgroup just before its arrival: read past this CU's L1
      for (int k = 0; k < 6; ++k)
        b[c][k] = __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t *>(src + k), __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT));
      sz[c] = leaf ? 1 : __hip_atomic_load(size + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    float u[6];
    for (int k = 0; k < 3; ++k) {
      u[k] = fminf(b[0][k], b[1][k]);
      u[3 + k] = fmaxf(b[0][k + 3], b[1][k + 3]);
    }
    for (int k = 0; k < 6; ++k) ibox[(size_t)node * 6 + k] = u[k];
    const int count = last[node] - first[node] + 1;
    size[node] = count <= leaf_max ? 1 : 1 + sz[0] + sz[1];
    // the child whose box centre is nearer to the camera is numbered (and walked) first
    float d2[2];
    for (int c = 0; c < 2; ++c) {
      const float mx = 0.5f * (b[c][0] + b[c][3]) - cx, my = 0.5f * (b[c][1] + b[c][4]) - cy,
                  mz = 0.5f * (b[c][2] + b[c][5]) - cz;
      d2[c] = mx * mx + my * my + mz * mz;
    }
    if (!(d2[0] <= d2[1])) atomicOr(&flags[node], 4u);  // bit 2: right child first
    node = parent_int[node];
  }
}

// ---- pass 5: emit the threaded depth-first node records + the id section -------------------
// A record is emitted for: inner nodes covering > leaf_max primitives; the topmost nodes
// covering <= leaf_max (as leaf records); single primitives whose parent is an inner record.
__device__ __forceinline__ void write_node(unsigned char *blob, uint32_t idx, const float *box, uint32_t skip,
                                           uint32_t leaf) {
  float *dst = reinterpret_cast<float *>(blob + (size_t)idx * 32);
  for (int k = 0; k < 6; ++k) dst[k] = box[k];
  reinterpret_cast<uint32_t *>(dst)[6] = skip;
  reinterpret_cast<uint32_t *>(dst)[7] = leaf;
}

__global__ void k_emit(int n, int leaf_max, const uint32_t *vals, const float *pbox, const int32_t *child_l,
                       const int32_t *child_r, const int32_t *first, const int32_t *last,
                       const int32_t *parent_int, const int32_t *parent_leaf, const float *ibox,
                       const int32_t *size, const uint32_t *flags, unsigned char *blob, uint32_t off_ids,
                       uint32_t n_nodes) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= 2 * n - 1) return;
  const bool is_leaf = tid >= n - 1;
  const int me = is_leaf ? tid - (n - 1) : tid;
  if (is_leaf) reinterpret_cast<uint32_t *>(blob + off_ids)[me] = vals[me];
  if (n == 1) {  // a single primitive: one leaf record
    if (tid == 0) write_node(blob, 0u, pbox + (size_t)vals[0] * 6, 1u, (0u << 3) | 1u);
  } else {
    const int parent = is_leaf ? parent_leaf[me] : parent_int[me];
    const int count = is_leaf ? 1 : last[me] - first[me] + 1;
    const int pcount = parent < 0 ? 0x7fffffff : last[parent] - first[parent] + 1;
    if (pcount > leaf_max) {  // otherwise this node sits inside a collapsed leaf
      // preorder index: walk to the root, adding 1 per level + the first child's subtree where
      // this path goes through the second child
      uint32_t idx = 0;
      int cur = is_leaf ? (me | kLeafBit) : me;
      int par = parent;
      while (par >= 0) {
        const bool right_first = (flags[par] & 4u) != 0u;
        const int32_t fc = right_first ? child_r[par] : child_l[par];
        idx += 1u;
        if (fc != cur) idx += fc < 0 ? 1u : (uint32_t)size[fc];
        cur = par;
        par = parent_int[par];
      }
      const float *box = is_leaf ? pbox + (size_t)vals[me] * 6 : ibox + (size_t)me * 6;
      if (count > leaf_max) {
        write_node(blob, idx, box, idx + (uint32_t)size[me], 0u);
      } else {
        const uint32_t f = is_leaf ? (uint32_t)me : (uint32_t)first[me];
        write_node(blob, idx, box, idx + 1u, (f << 3) | (uint32_t)count);
      }
    }
  }
  if (tid == 0) {  // sentinel END record: skip link to itself, no leaf (rtow_bvh.h)
    const float e[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    write_node(blob, n_nodes, e, n_nodes, 0u);
  }
}

// ---- pass 6: the trace kernel's termination argument, checked on the device ----------------
__global__ void k_validate(const unsigned char *blob, uint32_t n_nodes, uint32_t off_ids, int n_prims,
                           uint32_t *err) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n_nodes) return;
  const uint32_t *rec = reinterpret_cast<const uint32_t *>(blob + (size_t)i * 32);
  const uint32_t skip = rec[6], leaf = rec[7];
  bool ok = true;
  if (i == n_nodes) {
    ok = skip == n_nodes && leaf == 0u;
  } else {
    ok = skip > i && skip <= n_nodes;
    if (leaf != 0u) {
      const uint32_t f = leaf >> 3, c = leaf & 7u;
      ok = ok && c != 0u && f + c <= (uint32_t)n_prims && skip == i + 1u;
      if (ok)
        for (uint32_t k = 0; k < c; ++k) {
          const uint32_t id = reinterpret_cast<const uint32_t *>(blob + off_ids)[f + k];
          ok = ok && id < (uint32_t)n_prims;
        }
    } else {
      ok = ok && i + 1u < n_nodes;  // an inner node needs a child at i+1
    }
  }
  if (!ok) atomicOr(err, 1u);
}

// ---- 4-wide image for the BVH4 kernel (rtow_bvh4.h), collapsed from the radix tree on the device ----------------
// The host builder collapses its SAH tree; this is the same greedy collapse over the LBVH: a 4-wide node starts from
// the two children of a radix-tree node and, while it has fewer than four, replaces the inner child of largest
// surface area by that child's two children.  "Inner" = covers more than leaf_max triangles; anything smaller is a
// leaf of the 4-wide tree, and because a radix-tree node covers a contiguous range of the SORTED triangles, the
// triangle records are simply laid out in sorted order: a leaf is (first, count) of its range — no allocation.
// One level of the tree per launch pair (breadth-first order, which the kernel's LDS staging of the top of the tree
// needs): k_b4_expand picks every node's children and counts the inner ones, an exclusive scan turns the counts
// into the positions of the next level, k_b4_link writes the child words and the next level's work list.
constexpr int32_t kB4Empty = 0x7fffffff;
constexpr uint32_t kB4RefNone = 0x1fffffu, kB4RefLeaf = 1u << 20;  // = kRefNone, kRefLeaf of rtow_bvh4.h
constexpr uint32_t kB4MaxNodes = 1u << 20, kB4MaxTris = (1u << 18) - 4u;

struct B4Tree {
  const int32_t *child_l, *child_r, *first, *last;
  const float *ibox, *pbox;
  const uint32_t *vals;
  int leaf_max;
  __device__ int count(int32_t ref) const { return ref < 0 ? 1 : last[ref] - first[ref] + 1; }
  __device__ bool inner(int32_t ref) const { return ref >= 0 && ref != kB4Empty && count(ref) > leaf_max; }
  __device__ const float *box(int32_t ref) const {
    return ref < 0 ? pbox + (size_t)vals[ref & 0x7fffffff] * 6 : ibox + (size_t)ref * 6;
  }
  __device__ float area(int32_t ref) const {
    const float *b = box(ref);
    const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return dx * dy + dy * dz + dz * dx;
  }
};

__global__ void k_b4_expand(B4Tree t, int level_n, int base, int root_is_leaf, int32_t root_ref, const int32_t *src,
                            int32_t *child, uint32_t *cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= level_n) return;
  int32_t ch[4] = {kB4Empty, kB4Empty, kB4Empty, kB4Empty};
  int nc = 0;
  if (root_is_leaf) {  // a mesh of at most leaf_max triangles: one node, one leaf
    ch[nc++] = root_ref;
  } else {
    const int32_t me = src[base + i];
    ch[nc++] = t.child_l[me];
    ch[nc++] = t.child_r[me];
    while (nc < 4) {
      int pick = -1;
      float best = -1.0f;
      for (int c = 0; c < nc; ++c)
        if (t.inner(ch[c])) {
          const float a = t.area(ch[c]);
          if (a > best) {
            best = a;
            pick = c;
          }
        }
      if (pick < 0) break;
      const int32_t x = ch[pick];
      ch[pick] = t.child_l[x];
      ch[nc++] = t.child_r[x];
    }
  }
  uint32_t inner = 0;
  for (int c = 0; c < 4; ++c) {
    child[(size_t)(base + i) * 4 + c] = ch[c];
    inner += t.inner(ch[c]) ? 1u : 0u;
  }
  cnt[i] = inner;
}

__global__ void k_b4_link(B4Tree t, int level_n, int base, int next_base, const int32_t *child, const uint32_t *pos,
                          int32_t *src, uint32_t *cw, uint32_t *err) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= level_n) return;
  uint32_t k = 0;
  for (int c = 0; c < 4; ++c) {
    const int32_t ref = child[(size_t)(base + i) * 4 + c];
    uint32_t w;
    if (ref == kB4Empty) {
      w = kB4RefNone;
    } else if (t.inner(ref)) {
      const uint32_t slot = (uint32_t)next_base + pos[i] + k++;
      if (slot >= kB4MaxNodes) atomicOr(err, 2u);
      src[slot < kB4MaxNodes ? slot : 0u] = ref;
      w = slot;
    } else {
      const uint32_t f = ref < 0 ? (uint32_t)(ref & 0x7fffffff) : (uint32_t)t.first[ref];
      const uint32_t n = (uint32_t)t.count(ref);
      if (n < 1u || n > 4u || f + n > kB4MaxTris) atomicOr(err, 4u);
      w = kB4RefLeaf | (f << 2) | ((n - 1u) & 3u);
    }
    cw[(size_t)(base + i) * 4 + c] = w;
  }
}

// planes of one node in the image's format: binary32 (128-byte node) or binary16 in the mesh's frame (64-byte node)
__global__ void k_b4_nodes(B4Tree t, int n4, const int32_t *child, const uint32_t *cw, unsigned char *blob, int half,
                           double c0, double c1, double c2, double s0, double s1, double s2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const uint32_t node_bytes = half ? 64u : 128u, child_off = half ? 48u : 96u;
  float *f = reinterpret_cast<float *>(blob + (size_t)i * node_bytes);
  __half *h = reinterpret_cast<__half *>(blob + (size_t)i * node_bytes);
  uint32_t *w = reinterpret_cast<uint32_t *>(blob + (size_t)i * node_bytes + child_off);
  const double mc[3] = {c0, c1, c2}, ms[3] = {s0, s1, s2};
  for (int c = 0; c < 4; ++c) {
    const int32_t ref = child[(size_t)i * 4 + c];
    for (int k = 0; k < 3; ++k) {
      float lo = INFINITY, hi = -INFINITY;  // empty slot: inverted box, never hit
      if (ref != kB4Empty) {
        const float *b = t.box(ref);  // padded binary32 planes, conservative for the f32 slab test (k_morton, k_refit)
        lo = b[k];
        hi = b[3 + k];
      }
      if (half) {
        // outwards in both steps: binary64 -> binary32 and binary32 -> binary16 with the same directed rounding
        h[k * 8 + c] = __float2half_rd(__double2float_rd(((double)lo - mc[k]) * ms[k]));
        h[k * 8 + 4 + c] = __float2half_ru(__double2float_ru(((double)hi - mc[k]) * ms[k]));
      } else {
        f[k * 8 + c] = lo;
        f[k * 8 + 4 + c] = hi;
      }
    }
    w[c] = cw[(size_t)i * 4 + c];
  }
  if (!half) {  // the 16 unused bytes of a 128-byte node
    uint32_t *pad = reinterpret_cast<uint32_t *>(blob + (size_t)i * node_bytes + 112u);
    pad[0] = pad[1] = pad[2] = pad[3] = 0u;
  }
}

// triangle records and their material indices in SORTED order (= leaf order)
__global__ void k_b4_records(int nt, const uint32_t *vals, const double *tri, const int32_t *prim_mat,
                             unsigned char *blob, uint32_t off_tri, uint32_t off_pmat) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nt) return;
  const uint32_t p = vals[j];
  const double2 *srcp = reinterpret_cast<const double2 *>(tri + (size_t)p * 12);
  double2 *dst = reinterpret_cast<double2 *>(blob + off_tri + (size_t)j * 96);
  for (int k = 0; k < 6; ++k) dst[k] = srcp[k];
  reinterpret_cast<int32_t *>(blob + off_pmat)[j] = prim_mat[p];
}

// what the walk's termination and addressing rest on (validate_bvh4_image of rtow_bvh4.h): child links point to
// LATER nodes, leaves stay inside the record section
__global__ void k_b4_validate(const unsigned char *blob, int n4, int half, int nt, uint32_t *seen, uint32_t *err) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const uint32_t *w = reinterpret_cast<const uint32_t *>(blob + (size_t)i * (half ? 64u : 128u) + (half ? 48u : 96u));
  for (int c = 0; c < 4; ++c) {
    const uint32_t r = w[c];
    if (r == kB4RefNone) continue;
    bool ok = (r & ~0x1fffffu) == 0u;
    if (ok && (r & kB4RefLeaf)) {
      const uint32_t f = (r & (kB4RefLeaf - 1u)) >> 2, n = (r & 3u) + 1u;
      ok = f + n <= (uint32_t)nt;
      if (ok)
        for (uint32_t k = 0; k < n; ++k) atomicAdd(&seen[f + k], 1u);  // (checked by k_b4_validate_seen)
    } else if (ok) {
      ok = (int)r > i && (int)r < n4;
      if (ok) atomicAdd(&seen[nt + r], 1u);  // every node but the root is the child of exactly one node
    }
    if (!ok) atomicOr(err, 8u);
  }
}
// ... and every triangle record belongs to exactly one leaf, every node but the root to exactly one parent: a collapse
// that dropped or doubled a subtree would otherwise render a mesh with holes (or test triangles twice) without an error
__global__ void k_b4_validate_seen(const uint32_t *seen, int nt, int n4, uint32_t *err) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nt + n4) return;
  const uint32_t want = i == nt ? 0u : 1u;  // (seen[nt + 0] is the root)
  if (seen[i] != want) atomicOr(err, 8u);
}

void release(Scratch *s) {
  if (!s) return;
  void *ptrs[] = {s->pbox,     s->pbox64,  s->keys_a,     s->keys_b,      s->vals_a, s->vals_b, s->child_l,
                  s->child_r,  s->first,   s->last,       s->parent_int,  s->parent_leaf,
                  s->ibox,     s->size,    s->flags,      s->glob,        s->sort_tmp,
                  s->b4_src,   s->b4_child, s->b4_cw,     s->b4_cnt,      s->b4_pos, s->scan_tmp, s->b4_seen,
                  s->pl_ref[0], s->pl_ref[1], s->pl_box[0], s->pl_box[1], s->pl_nn, s->pl_flag, s->pl_scan,
                  s->pl_cnt,   s->pl_pos,  s->pl_parent_leaf, s->pl_state, s->pl_scan_tmp,
                  s->sh_item[0], s->sh_item[1], s->sh_node[0], s->sh_node[1], s->sh_rank, s->sh_list[0], s->sh_list[1],
                  s->sh_cnt,   s->sh_cb,   s->sh_bin, s->sh_split, s->sh_flag, s->sh_scan, s->sh_scan_tmp};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  delete s;
}

template <class T>
bool dev_alloc(T *&p, size_t count) {
  return hipMalloc(reinterpret_cast<void **>(&p), (count ? count : 1) * sizeof(T)) == hipSuccess;
}

}  // namespace

// Phase 1: builds the tree in scratch memory.  Returns 0 and the number of node records the
// image will hold (without the END record).  `*handle` is the scratch: NULL on the first call,
// reused (and grown when needed) by later builds, released with lbvh_release.
// `ploc_radius`: > 0 = PLOC with that search radius (pass 3b), 0 = Karras' radix tree (pass 3), < 0 = the binned SAH
// build (pass 3c).
int lbvh_build(const double *sph, const double *sph_r, const double *mov, const double *tri, int ns, int nm,
               int nt, double time0, double time1, const double cam_origin[3], int leaf_max, void *stream,
               void **handle, int *n_nodes, int ploc_radius) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int n = ns + nm + nt;
  if (n <= 0 || leaf_max < 1 || leaf_max > 7) return 1;
  Scratch *s = static_cast<Scratch *>(*handle);
  const size_t ni = n > 1 ? (size_t)n - 1 : 1;
  if (!s || s->capacity < n) {
    release(s);
    *handle = nullptr;
    s = new Scratch;
    const int cap = n + n / 8 + 64;
    const size_t ci = (size_t)cap;
    bool ok = dev_alloc(s->pbox, ci * 6) && dev_alloc(s->pbox64, ci * 6) && dev_alloc(s->keys_a, ci) &&
              dev_alloc(s->keys_b, ci) && dev_alloc(s->vals_a, ci) && dev_alloc(s->vals_b, ci) &&
              dev_alloc(s->child_l, ci) && dev_alloc(s->child_r, ci) && dev_alloc(s->first, ci) &&
              dev_alloc(s->last, ci) && dev_alloc(s->parent_int, ci) && dev_alloc(s->parent_leaf, ci) &&
              dev_alloc(s->ibox, ci * 6) && dev_alloc(s->size, ci) && dev_alloc(s->flags, ci) &&
              dev_alloc(s->glob, 8);
    if (ok) {
      ok = rocprim::radix_sort_pairs(nullptr, s->sort_tmp_bytes, s->keys_a, s->keys_b, s->vals_a, s->vals_b, (size_t)cap, 0u, 63u, st) == hipSuccess &&
           hipMalloc(&s->sort_tmp, s->sort_tmp_bytes ? s->sort_tmp_bytes : 16) == hipSuccess;
    }
    if (!ok) {
      release(s);
      return 2;
    }
    s->capacity = cap;
    *handle = s;
  }
  s->n = n;
  const uint32_t glob0[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};
  const int B = 256, G = (n + B - 1) / B;
  double cam_scale = 0.0;
  for (int k = 0; k < 3; ++k) cam_scale = fmax(cam_scale, fabs(cam_origin[k]));
  bool good = hipMemcpyAsync(s->glob, glob0, sizeof glob0, hipMemcpyHostToDevice, st) == hipSuccess &&
              hipMemsetAsync(s->flags, 0, ni * sizeof(uint32_t), st) == hipSuccess;
  if (good) {
    hipLaunchKernelGGL(k_bounds, dim3(G), dim3(B), 0, st, sph, sph_r, mov, tri, ns, nm, nt, time0, time1,
                       s->pbox64, s->glob);
    hipLaunchKernelGGL(k_morton, dim3(G), dim3(B), 0, st, s->pbox64, n, cam_scale, s->glob, s->pbox, s->keys_a,
                       s->vals_a);
    size_t tmp_bytes = s->sort_tmp_bytes;
    good = rocprim::radix_sort_pairs(s->sort_tmp, tmp_bytes, s->keys_a, s->keys_b, s->vals_a, s->vals_b, (size_t)n, 0u, 63u, st) == hipSuccess;
  }
  int32_t root_size = 1;
  if (good && n > 2 && ploc_radius < 0) {
    const size_t cap = (size_t)s->capacity, ranks = cap / 2 + 2;
    if (!s->sh_cnt) {
      bool ok = dev_alloc(s->sh_item[0], cap) && dev_alloc(s->sh_item[1], cap) && dev_alloc(s->sh_node[0], cap) &&
                dev_alloc(s->sh_node[1], cap) && dev_alloc(s->sh_rank, cap) && dev_alloc(s->sh_list[0], cap) &&
                dev_alloc(s->sh_list[1], cap) && dev_alloc(s->sh_cnt, (size_t)kSahMaxLevels + 2) &&
                dev_alloc(s->sh_cb, ranks * 6) && dev_alloc(s->sh_bin, ranks * 48 * 7) && dev_alloc(s->sh_split, ranks * 4) &&
                dev_alloc(s->sh_flag, cap) && dev_alloc(s->sh_scan, cap);
      if (ok)
        ok = rocprim::exclusive_scan(nullptr, s->sh_scan_tmp_bytes, s->sh_flag, s->sh_scan, 0u, cap, rocprim::plus<uint32_t>(), st) == hipSuccess &&
             hipMalloc(&s->sh_scan_tmp, s->sh_scan_tmp_bytes ? s->sh_scan_tmp_bytes : 16) == hipSuccess;
      if (!ok) return 2;
      s->sh_ranks = ranks;
    }
    good = hipMemsetAsync(s->sh_cnt, 0, ((size_t)kSahMaxLevels + 2) * sizeof(uint32_t), st) == hipSuccess;
    {
      const size_t init_threads = std::max<size_t>((size_t)n, s->sh_ranks * 48 * 7 / 8);
      hipLaunchKernelGGL(k_sah_init, dim3((unsigned)((init_threads + B - 1) / B)), dim3(B), 0, st, n, s->vals_b, s->sh_item[0],
                         s->sh_node[0], s->first, s->last, s->parent_int, s->sh_rank, s->sh_list[0], s->sh_cnt, s->sh_cb, s->sh_bin,
                         s->sh_ranks);
    }
    bool done = false;
    int level = 0;
    for (; good && !done && level < kSahMaxLevels; ++level) {
      const int in = level & 1, out = in ^ 1;
      // a level holds at most min(2^level, n / 2) nodes of two or more primitives
      const long long most = level < 30 ? std::min<long long>(1ll << level, n / 2 + 1) : (long long)(n / 2 + 1);
      const unsigned Gn = (unsigned)((most * 64 + B - 1) / B);  // (k_sah_split: one wave per node)
      const int force_median = level >= kSahMedianFrom ? 1 : 0;
      if (!force_median) {
        hipLaunchKernelGGL(k_sah_cb, dim3(G), dim3(B), 0, st, n, s->sh_item[in], s->sh_node[in], s->sh_rank, s->pbox, s->sh_cb);
        hipLaunchKernelGGL(k_sah_bin, dim3((n + kSahBinBlock - 1) / kSahBinBlock), dim3(kSahBinBlock), 0, st, n, s->sh_item[in],
                           s->sh_node[in], s->sh_rank, s->pbox, s->sh_cb, s->sh_bin);
      }
      hipLaunchKernelGGL(k_sah_split, dim3(Gn), dim3(B), 0, st, level, force_median, s->sh_list[in], s->sh_list[out], s->sh_cnt,
                         s->sh_rank, s->sh_cb, s->sh_bin, s->sh_split, s->child_l, s->child_r, s->first, s->last, s->parent_int,
                         s->parent_leaf);
      hipLaunchKernelGGL(k_sah_side, dim3(G), dim3(B), 0, st, n, s->sh_item[in], s->sh_node[in], s->sh_rank, s->pbox, s->sh_cb,
                         s->sh_split, s->first, s->sh_flag);
      size_t tb = s->sh_scan_tmp_bytes;
      good = rocprim::exclusive_scan(s->sh_scan_tmp, tb, s->sh_flag, s->sh_scan, 0u, (size_t)n, rocprim::plus<uint32_t>(), st) == hipSuccess;
      if (!good) break;
      hipLaunchKernelGGL(k_sah_scatter, dim3(G), dim3(B), 0, st, n, s->sh_item[in], s->sh_node[in], s->sh_rank, s->sh_split,
                         s->first, s->last, s->sh_flag, s->sh_scan, s->sh_item[out], s->sh_node[out], s->sh_cb);
      // (the node's accumulators go back to their identities inside the level: the bins by the thread that read them in
      // k_sah_split, the centroid bounds by the node's first position in k_sah_scatter, behind their last reader)
      if ((level & 7) == 7 || level + 1 == kSahMaxLevels) {
        uint32_t next_n = 1;
        good = hipMemcpyAsync(&next_n, s->sh_cnt + level + 1, 4, hipMemcpyDeviceToHost, st) == hipSuccess &&
               hipStreamSynchronize(st) == hipSuccess && hipGetLastError() == hipSuccess;
        done = next_n == 0u;
      }
    }
    if (good && !done) return 3;  // (cannot happen: from level kSahMedianFrom on every split halves its range)
    if (good) {
      hipLaunchKernelGGL(k_sah_finish, dim3(G), dim3(B), 0, st, n, s->sh_item[level & 1], s->vals_b);
    }
  } else if (good && n > 2 && ploc_radius > 0) {
    const size_t cap = (size_t)s->capacity;
    if (!s->pl_state) {
      bool ok = dev_alloc(s->pl_ref[0], cap) && dev_alloc(s->pl_ref[1], cap) && dev_alloc(s->pl_box[0], cap * 6) &&
                dev_alloc(s->pl_box[1], cap * 6) && dev_alloc(s->pl_nn, cap) && dev_alloc(s->pl_flag, cap) &&
                dev_alloc(s->pl_scan, cap) && dev_alloc(s->pl_cnt, cap) && dev_alloc(s->pl_pos, cap) &&
                dev_alloc(s->pl_parent_leaf, cap) && dev_alloc(s->pl_state, 4);
      if (ok)
        ok = rocprim::exclusive_scan(nullptr, s->pl_scan_tmp_bytes, s->pl_flag, s->pl_scan, 0ull, cap,
                                     rocprim::plus<unsigned long long>(), st) == hipSuccess &&
             hipMalloc(&s->pl_scan_tmp, s->pl_scan_tmp_bytes ? s->pl_scan_tmp_bytes : 16) == hipSuccess;
      if (!ok) return 2;
    }
    hipLaunchKernelGGL(k_ploc_init, dim3(G), dim3(B), 0, st, n, s->vals_b, s->pbox, s->pl_ref[0], s->pl_box[0], s->pl_state);
    good = hipMemsetAsync(s->parent_int, 0xff, sizeof(int32_t), st) == hipSuccess;  // the root (node 0) has no parent
    int bound = n, cur = 0, it = 0;  // bound: clusters the array can still hold (>= the device's count)
    // state[0..1] = (clusters, nodes created) of the array being read; the compact kernel writes the next pair to
    // state[2..3]; the two halves swap roles every iteration
    while (good && bound > 1) {
      uint32_t *st_in = s->pl_state + 2 * (it & 1), *st_out = s->pl_state + 2 * ((it + 1) & 1);
      const int Gb = (bound + B - 1) / B;
      hipLaunchKernelGGL(k_ploc_nn, dim3(Gb), dim3(B), 0, st, st_in, ploc_radius, s->pl_box[cur], s->pl_nn);
      hipLaunchKernelGGL(k_ploc_flags, dim3(Gb), dim3(B), 0, st, bound, st_in, s->pl_nn, s->pl_flag);
      size_t tb = s->pl_scan_tmp_bytes;
      good = rocprim::exclusive_scan(s->pl_scan_tmp, tb, s->pl_flag, s->pl_scan, 0ull, (size_t)bound,
                                     rocprim::plus<unsigned long long>(), st) == hipSuccess;
      if (!good) break;
      hipLaunchKernelGGL(k_ploc_compact, dim3(Gb), dim3(B), 0, st, n, bound, st_out, st_in, s->pl_nn, s->pl_flag, s->pl_scan,
                         s->pl_ref[cur], s->pl_box[cur], s->pl_ref[cur ^ 1], s->pl_box[cur ^ 1], s->child_l, s->child_r,
                         s->parent_int, s->pl_parent_leaf);
      cur ^= 1;
      ++it;
      // every iteration merges at least the closest pair of the array
      bound -= 1;
      if ((it & 3) == 0 || bound <= 1) {
        uint32_t m_now = 0;
        good = hipMemcpyAsync(&m_now, st_out, 4, hipMemcpyDeviceToHost, st) == hipSuccess &&
               hipStreamSynchronize(st) == hipSuccess && hipGetLastError() == hipSuccess;
        if (good && (m_now < 1u || (int)m_now > bound + 1)) return 3;
        bound = (int)m_now;
      }
      if (it > 4 * n) return 3;  // (cannot happen)
    }
    if (good) {  // one cluster left, n - 1 nodes created: anything else is an incomplete tree, never handed on
      uint32_t fin[2] = {0u, 0u};
      good = hipMemcpyAsync(fin, s->pl_state + 2 * (it & 1), 8, hipMemcpyDeviceToHost, st) == hipSuccess &&
             hipStreamSynchronize(st) == hipSuccess && hipGetLastError() == hipSuccess;
      if (good && (fin[0] != 1u || fin[1] != (uint32_t)(n - 1))) return 3;
    }
    if (good) {
      // leaves in depth-first order: counts, positions, the new `vals`, leaf references
      const int G2 = (2 * n - 1 + B - 1) / B;
      hipLaunchKernelGGL(k_ploc_count, dim3(G), dim3(B), 0, st, n, s->child_l, s->child_r, s->parent_int, s->pl_parent_leaf,
                         s->pl_cnt, s->flags);
      hipLaunchKernelGGL(k_ploc_place, dim3(G2), dim3(B), 0, st, n, s->child_l, s->parent_int, s->pl_parent_leaf, s->pl_cnt,
                         s->vals_b, s->vals_a, s->pl_pos, s->first, s->last);
      hipLaunchKernelGGL(k_ploc_relabel, dim3(G2), dim3(B), 0, st, n, s->pl_pos, s->pl_parent_leaf, s->child_l, s->child_r,
                         s->parent_leaf);
      std::swap(s->vals_a, s->vals_b);  // vals_b is what everything downstream reads: now the depth-first order
      good = hipMemsetAsync(s->flags, 0, ni * sizeof(uint32_t), st) == hipSuccess;  // (k_refit's arrival counters)
    }
  } else if (good && n > 1) {
    hipLaunchKernelGGL(k_radix_tree, dim3(G), dim3(B), 0, st, s->keys_b, n, s->child_l, s->child_r, s->first,
                       s->last, s->parent_int, s->parent_leaf);
  }
  if (good && n > 1) {
    hipLaunchKernelGGL(k_refit, dim3(G), dim3(B), 0, st, n, leaf_max, s->vals_b, s->pbox, s->child_l, s->child_r,
                       s->first, s->last, s->parent_int, s->parent_leaf, (float)cam_origin[0],
                       (float)cam_origin[1], (float)cam_origin[2], s->ibox, s->size, s->flags);
    good = hipMemcpyAsync(&root_size, s->size, sizeof root_size, hipMemcpyDeviceToHost, st) == hipSuccess;
  }
  good = good && hipStreamSynchronize(st) == hipSuccess && hipGetLastError() == hipSuccess;
  if (!good || root_size < 1 || root_size > 2 * n) return 3;
  *n_nodes = root_size;
  return 0;
}

// Phase 2: writes node records, END record and the id section into the device image (whose
// record sections the caller fills), validates the links.
int lbvh_emit(void *handle, int leaf_max, unsigned char *blob_dev, uint32_t off_ids, int n_nodes, void *stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  Scratch *s = static_cast<Scratch *>(handle);
  if (!s) return 1;
  const int n = s->n;
  const int B = 256;
  hipLaunchKernelGGL(k_emit, dim3((2 * n - 1 + B - 1) / B), dim3(B), 0, st, n, leaf_max, s->vals_b, s->pbox,
                     s->child_l, s->child_r, s->first, s->last, s->parent_int, s->parent_leaf, s->ibox, s->size,
                     s->flags, blob_dev, off_ids, (uint32_t)n_nodes);
  hipLaunchKernelGGL(k_validate, dim3((n_nodes + 1 + B - 1) / B), dim3(B), 0, st, blob_dev, (uint32_t)n_nodes,
                     off_ids, n, s->glob + 7);
  uint32_t err = 1;
  const bool good = hipMemcpyAsync(&err, s->glob + 7, sizeof err, hipMemcpyDeviceToHost, st) == hipSuccess &&
                    hipStreamSynchronize(st) == hipSuccess && hipGetLastError() == hipSuccess;
  if (!good) return 3;
  return err ? 4 : 0;
}

// 4-wide collapse of the tree lbvh_build left in scratch memory (triangle meshes; leaf_max <= 4).  Level by level:
// two launches, one scan and one 8-byte read-back per level of the 4-wide tree.  Returns the number of nodes, the
// depth (levels of nodes, root = 1) and the root's padded box (the frame of the binary16 format).
int lbvh_bvh4_collapse(void *handle, int leaf_max, void *stream, int *n_nodes, int *depth, float root_box[6]) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  Scratch *s = static_cast<Scratch *>(handle);
  if (!s || leaf_max < 1 || leaf_max > 4) return 1;
  const int n = s->n;
  if ((uint32_t)n > kB4MaxTris) return 5;  // (the caller falls back to the binary walk)
  const size_t cap = (size_t)s->capacity;
  if (!s->b4_src) {
    size_t tb = 0;
    bool ok = dev_alloc(s->b4_src, cap) && dev_alloc(s->b4_child, cap * 4) && dev_alloc(s->b4_cw, cap * 4) &&
              dev_alloc(s->b4_cnt, cap) && dev_alloc(s->b4_pos, cap) && dev_alloc(s->b4_seen, cap * 2) &&
              rocprim::exclusive_scan(nullptr, tb, s->b4_cnt, s->b4_pos, 0u, cap, rocprim::plus<uint32_t>(), st) == hipSuccess &&
              hipMalloc(&s->scan_tmp, tb ? tb : 16) == hipSuccess;
    if (!ok) return 2;
    s->scan_tmp_bytes = tb;
  }
  B4Tree t{s->child_l, s->child_r, s->first, s->last, s->ibox, s->pbox, s->vals_b, leaf_max};
  const bool root_is_leaf = n <= leaf_max;
  const int32_t root_ref = n == 1 ? (int32_t)(0u | 0x80000000u) : 0;
  const int32_t zero = 0;
  bool good = hipMemcpyAsync(s->b4_src, &zero, sizeof zero, hipMemcpyHostToDevice, st) == hipSuccess &&
              hipMemsetAsync(s->glob + 7, 0, sizeof(uint32_t), st) == hipSuccess;
  int base = 0, level_n = 1, levels = 0;
  const int B = 256;
  while (good && level_n > 0) {
    ++levels;
    const int G = (level_n + B - 1) / B;
    if ((size_t)base + (size_t)level_n > cap) return 5;
    hipLaunchKernelGGL(k_b4_expand, dim3(G), dim3(B), 0, st, t, level_n, base, root_is_leaf ? 1 : 0, root_ref, s->b4_src,
                       s->b4_child, s->b4_cnt);
    size_t tb = s->scan_tmp_bytes;
    good = rocprim::exclusive_scan(s->scan_tmp, tb, s->b4_cnt, s->b4_pos, 0u, (size_t)level_n, rocprim::plus<uint32_t>(), st) == hipSuccess;
    if (!good) break;
    const int next_base = base + level_n;
    hipLaunchKernelGGL(k_b4_link, dim3(G), dim3(B), 0, st, t, level_n, base, next_base, s->b4_child, s->b4_pos, s->b4_src,
                       s->b4_cw, s->glob + 7);
    uint32_t tail[2] = {0u, 0u};  // inner children of the level = its last position + its last count
    good = hipMemcpyAsync(&tail[0], s->b4_pos + (level_n - 1), 4, hipMemcpyDeviceToHost, st) == hipSuccess &&
           hipMemcpyAsync(&tail[1], s->b4_cnt + (level_n - 1), 4, hipMemcpyDeviceToHost, st) == hipSuccess &&
           hipStreamSynchronize(st) == hipSuccess && hipGetLastError() == hipSuccess;
    base = next_base;
    level_n = (int)(tail[0] + tail[1]);
    if ((uint32_t)base + (uint32_t)level_n > kB4MaxNodes) return 5;
  }
  uint32_t err = 1;
  float rb[6] = {0, 0, 0, 0, 0, 0};
  good = good && hipMemcpyAsync(&err, s->glob + 7, sizeof err, hipMemcpyDeviceToHost, st) == hipSuccess &&
         hipMemcpyAsync(rb, n > 1 ? s->ibox : s->pbox, sizeof rb, hipMemcpyDeviceToHost, st) == hipSuccess &&
         hipStreamSynchronize(st) == hipSuccess;
  if (!good) return 3;
  if (err) return (err & 6u) ? 5 : 4;
  s->b4_nodes = base;
  s->b4_depth = levels;
  *n_nodes = base;
  *depth = levels;
  for (int k = 0; k < 6; ++k) root_box[k] = rb[k];
  return 0;
}

// Writes the 4-wide image: nodes (binary32 planes, or binary16 planes in the frame map_c / map_s), triangle records
// and material indices in sorted order.  The caller copies the material records to off_mats.  Validates the links.
int lbvh_bvh4_emit(void *handle, unsigned char *blob_dev, int half, const double map_c[3], const double map_s[3],
                   uint32_t off_tri, uint32_t off_pmat, const double *tri, const int32_t *prim_mat, void *stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  Scratch *s = static_cast<Scratch *>(handle);
  if (!s || s->b4_nodes <= 0) return 1;
  const int n = s->n, n4 = s->b4_nodes, B = 256;
  B4Tree t{s->child_l, s->child_r, s->first, s->last, s->ibox, s->pbox, s->vals_b, 0};
  hipLaunchKernelGGL(k_b4_nodes, dim3((n4 + B - 1) / B), dim3(B), 0, st, t, n4, s->b4_child, s->b4_cw, blob_dev, half,
                     map_c[0], map_c[1], map_c[2], map_s[0], map_s[1], map_s[2]);
  hipLaunchKernelGGL(k_b4_records, dim3((n + B - 1) / B), dim3(B), 0, st, n, s->vals_b, tri, prim_mat, blob_dev, off_tri,
                     off_pmat);
  // (the arrival counters of the collapse are free again: [0, n) triangles, [n, n + n4) nodes; n + n4 <= 2 * capacity)
  uint32_t *seen = s->b4_seen;
  if (hipMemsetAsync(seen, 0, ((size_t)n + (size_t)n4) * sizeof(uint32_t), st) != hipSuccess) return 3;
  hipLaunchKernelGGL(k_b4_validate, dim3((n4 + B - 1) / B), dim3(B), 0, st, blob_dev, n4, half, n, seen, s->glob + 7);
  hipLaunchKernelGGL(k_b4_validate_seen, dim3((n + n4 + B - 1) / B), dim3(B), 0, st, seen, n, n4, s->glob + 7);
  uint32_t err = 1;
  const bool good = hipMemcpyAsync(&err, s->glob + 7, sizeof err, hipMemcpyDeviceToHost, st) == hipSuccess &&
                    hipStreamSynchronize(st) == hipSuccess && hipGetLastError() == hipSuccess;
  if (!good) return 3;
  return err ? 4 : 0;
}

void lbvh_release(void *handle) { release(static_cast<Scratch *>(handle)); }

}  // namespace rtow
