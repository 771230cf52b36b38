// rtow_trace_bvh4.h — part of the trace kernels (included by rtow_trace_body.h inside namespace rtow::{anonymous};
// see that file for the execution model).  The 4-wide BVH walk for triangle meshes (image: rtow_bvh4.h).
#pragma once
// --------------------------------------------------------- closest hit: BVH4 ---
// Per-lane ordered traversal with a per-lane stack in LDS.
//   * One step tests the FOUR child boxes of a node.  The node stores its children's planes axis by
//     axis (lo.x[4] hi.x[4] ...), so a lane reads the near planes of all four children with one
//     16-byte load at an address chosen by the sign of its ray direction, and the far planes at
//     that address ^ 16: the slab test is 6 fma + max3 + min3 per child, no min/max per plane.
//     (Half nodes: 8-byte reads from LDS the same way; from L2 one 16-byte load per axis — both halves —
//     and four selects, because there each load instruction costs the address unit 16 cycles whatever its size.)
//   * The nearest hit child is visited next, the other hit children go to the stack with their entry
//     distance (11 bits of it, rounded down), so that a popped entry beyond the closest hit so far is
//     dropped without touching its node.  Stack entry: [tnear >> 20 : 11][ref21 : 21] (rtow_bvh4.h).
//   * The stack lives in LDS, [entry][lane of the workgroup] (bank = lane: conflict-free), K entries
//     per lane; deeper entries spill to a global array [entry][lane of the grid] (coalesced; rare).
//     The host sizes the spill for 3*depth entries, the bound for a 4-wide tree of that depth.
//   * Leaves met on the way are queued (two per lane) and tested in a separate phase so that the box
//     loop and the triangle loop each run SIMT-dense, as in the binary walk (rtow_trace_bvh.h).
//   * The image is read from LDS below `lds_limit` and from global memory above it: a small mesh
//     is staged whole, a big one has the top of its tree (breadth-first node order) in LDS.
// Termination: child links only point to later nodes (validated at upload), so every node is entered
// at most once per ray; every loop trip either pops, enters a node, queues a leaf or runs the leaf phase.
// Node format (rtow_bvh4.h): an image staged whole (FULL) has 128-byte nodes with binary32 planes; a bigger one,
// whose nodes are read from L2 below the staged top, has 64-byte nodes with binary16 planes in the mesh's own frame.
constexpr uint32_t kRefNone = 0x1fffffu;   // rtow_bvh4.h
constexpr uint32_t kRefLeaf = 1u << 20;
constexpr uint32_t kRefPop = 0x1ffffeu;    // traversal state only: take the next entry from the stack

// (typed LDS / global accessors — lds_read, lds_write, glb_read, the native vector types: rtow_trace_math.h)
// FULL: the whole image is staged in LDS (small mesh).  Otherwise LDS holds the first `lds_limit`
// bytes of the node section (the top of the tree); deeper nodes, the triangle records and the
// materials are read from global memory (L2).
template <bool FULL>
struct Bvh4Reader {
  const unsigned char *g;  // the image in global memory
  uint32_t lds_limit;      // bytes of it staged at the start of LDS
  uint32_t aux_src, aux_lds;  // the image's end [aux_src, ...) staged at LDS offset aux_lds (wave-uniform)
  template <class T>
  __device__ __forceinline__ T tri(uint32_t off) const {  // triangle records (the leaf loop: no residency test)
    if constexpr (FULL)
      return lds_read<T>(off);
    else
      return glb_read<T>(g, off);
  }
  template <class T>
  __device__ __forceinline__ T rec(uint32_t off) const {  // material indices, materials (shading)
    if constexpr (FULL) {
      return lds_read<T>(off);
    } else {
      // (material indices and materials are staged as a block or not at all, so the side taken is the same for
      // every lane at a given call site)
      if (off >= aux_src) return lds_read<T>(aux_lds + (off - aux_src));
      return glb_read<T>(g, off);
    }
  }
  __device__ __forceinline__ vd2 t2(uint32_t off) const { return tri<vd2>(off); }
  __device__ __forceinline__ vd2 d2(uint32_t off) const { return rec<vd2>(off); }
  __device__ __forceinline__ uint32_t u32(uint32_t off) const { return rec<uint32_t>(off); }
};

// ---- the pieces of the walk, shared by the trip-structured kernel below and the state machine (rtow_trace_sm4.h) ----
// The f32 forms of a ray the slab tests need.
struct Bvh4Ray {
  float ix, iy, iz, oix, oiy, oiz;  // 1/d (huge if d is 0) and o/d: t(plane) = plane * ix - oix
  // the same for the FAR planes with the interval's relative slack folded in: t_far * slack = plane * (ix * slack) -
  // oix * slack — six multiplications per segment instead of four per node visit
  float ixs, iys, izs, oixs, oiys, oizs;
  uint32_t nxo, nyo, nzo;           // where this ray finds the NEAR planes of a node (the far planes: address ^ 16)
};
constexpr float kBvh4Slack = 1.00002f;  // relative slack on the far side of the slab interval
template <bool FULL>
__device__ __forceinline__ Bvh4Ray bvh4_ray(const DevScene &sc, V3 o, V3 d) {
  Bvh4Ray r;
  r.ix = safe_inv((float)d.x), r.iy = safe_inv((float)d.y), r.iz = safe_inv((float)d.z);
  if constexpr (!FULL) {
    // half nodes: planes are stored as (plane - c) * s, so t = plane' * (1 / (d s)) - (o - c) / d
    r.oix = (float)(o.x - (real)sc.b4_c[0]) * r.ix, r.oiy = (float)(o.y - (real)sc.b4_c[1]) * r.iy,
    r.oiz = (float)(o.z - (real)sc.b4_c[2]) * r.iz;
    r.nxo = r.ix < 0.0f ? 8u : 0u, r.nyo = r.iy < 0.0f ? 24u : 16u, r.nzo = r.iz < 0.0f ? 40u : 32u;
    r.ix *= sc.b4_is[0], r.iy *= sc.b4_is[1], r.iz *= sc.b4_is[2];
  } else {
    r.oix = (float)o.x * r.ix, r.oiy = (float)o.y * r.iy, r.oiz = (float)o.z * r.iz;
    r.nxo = r.ix < 0.0f ? 16u : 0u, r.nyo = r.iy < 0.0f ? 48u : 32u, r.nzo = r.iz < 0.0f ? 80u : 64u;
  }
  r.ixs = r.ix * kBvh4Slack, r.iys = r.iy * kBvh4Slack, r.izs = r.iz * kBvh4Slack;
  r.oixs = r.oix * kBvh4Slack, r.oiys = r.oiy * kBvh4Slack, r.oizs = r.oiz * kBvh4Slack;
  return r;
}
// This lane's traversal stack: LDS slot s at lds + s * kBvh4StackStride (workgroups are 1024 lanes); slots at or
// beyond `end` live in the global spill array [entry][lane of the grid].
constexpr uint32_t kBvh4StackStrideLog2 = 12u, kBvh4StackStride = 1u << kBvh4StackStrideLog2;
struct Bvh4Stack {
  uint32_t lds, end;
};
__device__ __forceinline__ Bvh4Stack bvh4_stack(const DevScene &sc) {
  Bvh4Stack st;
  st.lds = sc.b4_stack_base + 4u * threadIdx.x;
  st.end = st.lds + (sc.b4_stack_k << kBvh4StackStrideLog2);
  return st;
}

// One step of one lane.  `cur` is the node or leaf in hand (kRefPop: take the next stack entry; kRefNone: nothing
// left but the queued leaves), `sa` the address of the next free stack slot, (q0, q1) the queued leaves, oldest
// first, `tmax32` the closest hit so far rounded up to f32.
// FOLD: the far planes use the slack-folded coefficients of the ray (the trip kernel, whose ray lives for one walk);
// the state machine holds its ray across its whole loop and is at the register limit, so it multiplies per node.
template <bool FULL, bool FOLD = true>
__device__ __forceinline__ void bvh4_step(const Bvh4Reader<FULL> &im, const TraceParams &P, const Bvh4Ray &r, float tmax32,
                                          Bvh4Stack st, uint32_t lane_g, uint32_t &cur, uint32_t &sa, uint32_t &q0,
                                          uint32_t &q1, uint32_t &nnode) {
  const float tmin32 = 0.0009f;  // < RTOW_TMIN
  const float tmaxs = FOLD ? tmax32 * kBvh4Slack : tmax32;  // (FOLD: the far side of every interval below carries the slack)
  const float fix = FOLD ? r.ixs : r.ix, fiy = FOLD ? r.iys : r.iy, fiz = FOLD ? r.izs : r.iz;
  const float foix = FOLD ? r.oixs : r.oix, foiy = FOLD ? r.oiys : r.oiy, foiz = FOLD ? r.oizs : r.oiz;
  // (1) a leaf reached by the walk waits in the queue for the next leaf phase
  if ((cur & kRefLeaf) != 0u && cur < kRefPop && q1 == kRefNone) {
    if (q0 == kRefNone)
      q0 = cur;
    else
      q1 = cur;
    cur = kRefPop;
  }
  // (2) next entry from the stack; an entry that starts beyond the closest hit so far is dropped
  if (cur == kRefPop) {
    if (sa == st.lds) {
      cur = kRefNone;
    } else {
      sa -= kBvh4StackStride;
      uint32_t e;
      if (sa < st.end)
        e = lds_read<uint32_t>(sa);
      else
        e = P.spill[(size_t)((sa - st.end) >> kBvh4StackStrideLog2) * P.n_lanes + lane_g];
      cur = (e >> 21) > (__float_as_uint(tmax32) >> 20) ? kRefPop : (e & 0x1fffffu);
    }
  }
  // (3) one node: four slab tests, nearest hit child next, the others to the stack
  if (cur < kRefLeaf) {
    // binary32 planes (FULL), or binary16 planes whose near / far halves of an axis are 8 bytes apart; the fma
    // below takes either as it is (v_fma_mix_f32: binary16 operand, binary32 arithmetic)
    typedef typename std::conditional<FULL, vf4, vh4>::type planes_t;
    constexpr uint32_t kFar = FULL ? 16u : 8u, kChild = FULL ? 96u : 48u;
    const uint32_t nb = cur * (FULL ? 128u : 64u);
    planes_t nx, fx, ny, fy, nz, fz;
    vu4 cw;
    if (FULL || nb < im.lds_limit) {
      nx = lds_read<planes_t>(nb + r.nxo), fx = lds_read<planes_t>((nb + r.nxo) ^ kFar);
      ny = lds_read<planes_t>(nb + r.nyo), fy = lds_read<planes_t>((nb + r.nyo) ^ kFar);
      nz = lds_read<planes_t>(nb + r.nzo), fz = lds_read<planes_t>((nb + r.nzo) ^ kFar);
      cw = lds_read<vu4>(nb + kChild);
    } else if constexpr (!FULL) {
      // four 16-byte loads per node (the address path of the vector memory unit is what a big mesh waits for):
      // both halves of an axis in one load, near / far chosen by selects
      const vu4 ax = glb_read<vu4>(im.g, nb), ay = glb_read<vu4>(im.g, nb + 16u), az = glb_read<vu4>(im.g, nb + 32u);
      cw = glb_read<vu4>(im.g, nb + kChild);
      const bool sx = (r.nxo & 8u) != 0u, sy = (r.nyo & 8u) != 0u, sz = (r.nzo & 8u) != 0u;
      typedef uint32_t vu2 __attribute__((ext_vector_type(2)));
      const vu2 nxu = {sx ? ax.z : ax.x, sx ? ax.w : ax.y}, fxu = {sx ? ax.x : ax.z, sx ? ax.y : ax.w};
      const vu2 nyu = {sy ? ay.z : ay.x, sy ? ay.w : ay.y}, fyu = {sy ? ay.x : ay.z, sy ? ay.y : ay.w};
      const vu2 nzu = {sz ? az.z : az.x, sz ? az.w : az.y}, fzu = {sz ? az.x : az.z, sz ? az.y : az.w};
      nx = __builtin_bit_cast(vh4, nxu), fx = __builtin_bit_cast(vh4, fxu);
      ny = __builtin_bit_cast(vh4, nyu), fy = __builtin_bit_cast(vh4, fyu);
      nz = __builtin_bit_cast(vh4, nzu), fz = __builtin_bit_cast(vh4, fzu);
    }
    ++nnode;
#define RTOW_SLAB(c, slot)                                                                                             \
  const float tn##slot = fmaxf(fmaxf(fmaf((float)nx.c, r.ix, -r.oix), fmaf((float)ny.c, r.iy, -r.oiy)),                \
                               fmaxf(fmaf((float)nz.c, r.iz, -r.oiz), tmin32));                                        \
  const float tf##slot = fminf(fminf(fmaf((float)fx.c, fix, -foix), fmaf((float)fy.c, fiy, -foiy)),                    \
                               fminf(fmaf((float)fz.c, fiz, -foiz), tmaxs));                                           \
  const bool h##slot = FOLD ? tn##slot <= tf##slot : tn##slot <= tf##slot * kBvh4Slack;                                \
  const uint32_t k##slot = h##slot ? ((__float_as_uint(tn##slot) & ~3u) | slot##u) : 0xffffffffu;
    RTOW_SLAB(x, 0)
    RTOW_SLAB(y, 1)
    RTOW_SLAB(z, 2)
    RTOW_SLAB(w, 3)
#undef RTOW_SLAB
    const uint32_t kmin = min(min(k0, k1), min(k2, k3));
    const uint32_t s = kmin & 3u;  // (3 when nothing was hit: h3 is false then, nothing is pushed)
    const uint32_t next = s == 0u ? cw.x : (s == 1u ? cw.y : (s == 2u ? cw.z : cw.w));
    cur = kmin == 0xffffffffu ? kRefPop : next;
#define RTOW_PUSH(slot, child)                                                                       \
  if (h##slot && s != slot##u) {                                                                     \
    const uint32_t e = ((__float_as_uint(tn##slot) >> 20) << 21) | child;                            \
    if (sa < st.end)                                                                                 \
      lds_write<uint32_t>(sa, e);                                                                    \
    else                                                                                             \
      P.spill[(size_t)((sa - st.end) >> kBvh4StackStrideLog2) * P.n_lanes + lane_g] = e;             \
    sa += kBvh4StackStride;                                                                          \
  }
    RTOW_PUSH(0, cw.x)
    RTOW_PUSH(1, cw.y)
    RTOW_PUSH(2, cw.z)
    RTOW_PUSH(3, cw.w)
#undef RTOW_PUSH
  }
}

// The triangles of one queued leaf, with the f64 test every kernel uses (so the accepted (t, primitive) is the same).
template <bool FULL>
__device__ __forceinline__ void bvh4_leaf(const Bvh4Reader<FULL> &im, const DevScene &sc, uint32_t leaf, V3d o64, V3d d64,
                                          Closest &best, uint32_t &nprim) {
  const uint32_t first = (leaf & (kRefLeaf - 1u)) >> 2, count = (leaf & 3u) + 1u;
  for (uint32_t k = 0; k < count; ++k) {
    const uint32_t r = sc.b4_off_tri + 96u * (first + k);
    const vd2 t0 = im.t2(r), t1 = im.t2(r + 16u), t2 = im.t2(r + 32u), t3 = im.t2(r + 48u), t4 = im.t2(r + 64u),
              t5 = im.t2(r + 80u);
    ++nprim;
    triangle_test<double>(o64, d64, V3d{t0.x, t0.y, t1.x}, V3d{t1.y, t2.x, t2.y}, V3d{t3.x, t3.y, t4.x}, V3d{t4.y, t5.x, t5.y},
                          (int)(first + k), RTOW_TMIN, best);
  }
}
// a lane that can take a step (it holds a node, a stack to pop, or a leaf that still fits its queue)
__device__ __forceinline__ bool bvh4_busy(uint32_t cur, uint32_t q1) {
  return cur != kRefNone && !((cur & kRefLeaf) != 0u && cur < kRefPop && q1 != kRefNone);
}

// Resumable like the grid walk (rtow_trace_grid.h): after `cap` loop trips with at most `max_open`
// lanes still walking, the queued leaves are tested and the unfinished lanes keep their place — the
// node in hand (`w_cur`, kRefNone = no walk in progress), the stack (it lives in LDS / the spill
// array; `w_sa` is its top) and the closest hit so far — for the caller's next trip.
template <bool FULL, bool ST>
__device__ __forceinline__ Closest closest_hit_bvh4(const Bvh4Reader<FULL> &im, const DevScene &sc, const TraceParams &P,
                                                    V3 o, V3 d, real time, bool active, uint32_t lane_g, uint32_t &nnode,
                                                    uint32_t &nprim, Stamps<ST> &stamps, Closest best, uint32_t &w_cur,
                                                    uint32_t &w_sa, uint32_t cap, uint32_t max_open) {
  (void)time;  // triangles do not move
  const bool resumed = w_cur != kRefNone;
  if (!resumed) {
    best.t = (real)__builtin_huge_val();
    best.prim = -1;
  }
  const V3d o64 = to_f64(o), d64 = to_f64(d);
  const Bvh4Ray ray = bvh4_ray<FULL>(sc, o, d);
  const Bvh4Stack st = bvh4_stack(sc);
  float tmax32 = round_up_f32(best.t);  // closest hit so far, rounded up
  uint32_t sa = resumed ? w_sa : st.lds;
  uint32_t cur = !active ? kRefNone : (resumed ? w_cur : 0u);  // node 0 = root
  uint32_t q0 = kRefNone, q1 = kRefNone;  // queued leaves, oldest first
  uint32_t trips = 0u;                    // wave-uniform

  // An image staged whole: every node step is a chain of LDS round trips (stack, planes, child words) — the walk issues at
  // the cell lists' priority, ahead of the other waves' new-ray arithmetic (round 5: suzanne +1.2 %; a mesh whose nodes
  // come from L2 loses 1 % with it and stays at the stage priority)
  if constexpr (FULL) stage_prio<kPrioLeaf>();
  for (;;) {
    if constexpr (ST) {
      stamps.iters += 1;
      stamps.step_lanes += (unsigned long long)__popcll(__ballot(bvh4_busy(cur, q1)));
    }
    bvh4_step<FULL>(im, P, ray, tmax32, st, lane_g, cur, sa, q0, q1, nnode);
    const unsigned long long m_walking = __ballot(cur != kRefNone);
    const bool any_walking = m_walking != 0ull;
    ++trips;
    const bool suspend = any_walking && trips >= cap && (uint32_t)__popcll(m_walking) <= max_open;
    // leaf phase: when enough lanes hold a queued leaf, or when no lane can take a step, or at the end
    const unsigned long long m_pending = __ballot(q0 != kRefNone);
    if ((m_pending != 0ull && ((uint32_t)__popcll(m_pending) >= P.leaf_votes || __ballot(bvh4_busy(cur, q1)) == 0ull)) ||
        !any_walking || suspend) {
      stamps.mark(RG_WALK, __ballot(active));
      if constexpr (ST) {
        stamps.phases += 1;
        stamps.leaf_lanes += (unsigned long long)__popcll(m_pending);
      }
      // every lane tests the triangles of the OLDEST leaf it queued
      if (q0 != kRefNone) bvh4_leaf<FULL>(im, sc, q0, o64, d64, best, nprim);
      q0 = q1;
      q1 = kRefNone;
      if (suspend && q0 != kRefNone) {  // both queued leaves before stopping
        bvh4_leaf<FULL>(im, sc, q0, o64, d64, best, nprim);
        q0 = kRefNone;
      }
      tmax32 = round_up_f32(best.t);  // rounded up: never below the f64 value
      stamps.mark(RG_LEAF, m_pending);
      if (suspend) {
        w_cur = cur;  // kRefNone for the lanes that are done
        w_sa = sa;
        if constexpr (FULL) stage_prio<kPrioStage>();
        return best;
      }
      if (!any_walking && !__any(q0 != kRefNone)) break;
    }
  }
  w_cur = kRefNone;
  if constexpr (FULL) stage_prio<kPrioStage>();
  return best;
}
