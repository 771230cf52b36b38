// rtow_trace_bvh4.h — part of the trace kernels (included by rtow_trace_body.h inside namespace rtow::{anonymous};
// see that file for the execution model).  The 4-wide BVH walk for triangle meshes (image: rtow_bvh4.h).
#pragma once
// --------------------------------------------------------- closest hit: BVH4 ---
// Per-lane ordered traversal with a per-lane stack in LDS.
//   * One step tests the FOUR child boxes of a node.  The node stores its children's planes axis by
//     axis (lo.x[4] hi.x[4] ...), so a lane reads the near planes of all four children with one
//     16-byte load at an address chosen by the sign of its ray direction, and the far planes at
//     that address ^ 16: the slab test is 6 fma + max3 + min3 per child, no min/max per plane.
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
#ifndef RTOW_BVH4_NODE_BYTES
#define RTOW_BVH4_NODE_BYTES 128
#endif
constexpr uint32_t kBvh4NodeBytes = RTOW_BVH4_NODE_BYTES;  // rtow_bvh4.h
constexpr uint32_t kRefNone = 0x1fffffu;   // rtow_bvh4.h
constexpr uint32_t kRefLeaf = 1u << 20;
constexpr uint32_t kRefPop = 0x1ffffeu;    // traversal state only: take the next entry from the stack

// LDS is addressed with 32-bit offsets through address-space-3 pointers, global memory through
// address-space-1 pointers: with generic pointers the compiler merges the two sides of an
// "in LDS or in global memory" choice into flat loads (and 64-bit address arithmetic).
typedef float vf4 __attribute__((ext_vector_type(4)));      // native vectors: HIP's float4 class cannot be
typedef uint32_t vu4 __attribute__((ext_vector_type(4)));   // read through an address-space pointer
typedef double vd2 __attribute__((ext_vector_type(2)));
#define RTOW_AS_LDS __attribute__((address_space(3)))
#define RTOW_AS_GLB __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ T lds_read(uint32_t off) {
  return *(const RTOW_AS_LDS T *)((const RTOW_AS_LDS unsigned char *)rtow_lds + off);
}
template <class T>
__device__ __forceinline__ void lds_write(uint32_t off, T v) {
  *(RTOW_AS_LDS T *)((RTOW_AS_LDS unsigned char *)rtow_lds + off) = v;
}
template <class T>
__device__ __forceinline__ T glb_read(const unsigned char *base, uint32_t off) {
  return *(const RTOW_AS_GLB T *)((const RTOW_AS_GLB unsigned char *)base + off);
}

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

// Resumable like the grid walk (rtow_trace_grid.h): after `cap` loop trips with at most `max_open`
// lanes still walking, the queued leaves are tested and the unfinished lanes keep their place — the
// node in hand (`w_cur`, kRefNone = no walk in progress), the stack (it lives in LDS / the spill
// array; `w_sa` is its top) and the closest hit so far — for the caller's next trip.
template <bool FULL, bool ST>
__device__ __forceinline__ Closest closest_hit_bvh4(const Bvh4Reader<FULL> &im, const DevScene &sc, const TraceParams &P,
                                                    V3 o, V3 d, real time, bool active, uint32_t lane_g, uint32_t &nnode,
                                                    uint32_t &nprim, Stamps<ST> &stamps, Closest best, uint32_t &w_cur,
                                                    uint32_t &w_sa, uint32_t cap, uint32_t max_open) {
  const bool resumed = w_cur != kRefNone;
  if (!resumed) {
    best.t = (real)__builtin_huge_val();
    best.prim = -1;
  }
  const RayForms ray = make_ray_forms(o, d, time);
  const float tmin32 = 0.0009f;   // < RTOW_TMIN
  const float slack = 1.00002f;   // relative slack on the far side of the interval, folded into the far-plane terms
  const float ix = safe_inv((float)d.x), iy = safe_inv((float)d.y), iz = safe_inv((float)d.z);
  const float oix = (float)o.x * ix, oiy = (float)o.y * iy, oiz = (float)o.z * iz;
  const float jx = ix * slack, jy = iy * slack, jz = iz * slack;
  const float ojx = oix * slack, ojy = oiy * slack, ojz = oiz * slack;
  float tmax32 = round_up_f32(best.t);  // closest hit so far, rounded up
  float tfm = tmax32 * slack;           // ... times slack
  // where this ray finds the near planes of a node (the far planes are at the same address ^ 16)
  const uint32_t nxo = ix < 0.0f ? 16u : 0u, nyo = iy < 0.0f ? 48u : 32u, nzo = iz < 0.0f ? 80u : 64u;
  // the stack: LDS slot s of this lane at stack_lds + s * kStride (workgroups are 1024 lanes); `sa` is the
  // address of the next free slot, slots at or beyond `sa_end` live in the global spill array
  constexpr uint32_t kStrideLog2 = 12u, kStride = 1u << kStrideLog2;
  const uint32_t stack_lds = sc.b4_stack_base + 4u * threadIdx.x;
  const uint32_t sa_end = stack_lds + (sc.b4_stack_k << kStrideLog2);
  uint32_t sa = resumed ? w_sa : stack_lds;
  uint32_t cur = !active ? kRefNone : (resumed ? w_cur : 0u);  // node 0 = root
  uint32_t q0 = kRefNone, q1 = kRefNone;  // queued leaves, oldest first
  uint32_t trips = 0u;                    // wave-uniform

  for (;;) {
    if constexpr (ST) stamps.iters += 1;
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
      if (sa == stack_lds) {
        cur = kRefNone;
      } else {
        sa -= kStride;
        uint32_t e;
        if (sa < sa_end)
          e = lds_read<uint32_t>(sa);
        else
          e = P.spill[(size_t)((sa - sa_end) >> kStrideLog2) * P.n_lanes + lane_g];
        cur = (e >> 21) > (__float_as_uint(tmax32) >> 20) ? kRefPop : (e & 0x1fffffu);
      }
    }
    // (3) one node: four slab tests, nearest hit child next, the others to the stack
    if (cur < kRefLeaf) {
      const uint32_t nb = cur * kBvh4NodeBytes;
      vf4 nx, fx, ny, fy, nz, fz;
      vu4 cw;
      if (FULL || nb < im.lds_limit) {
        nx = lds_read<vf4>(nb + nxo), fx = lds_read<vf4>((nb + nxo) ^ 16u);
        ny = lds_read<vf4>(nb + nyo), fy = lds_read<vf4>((nb + nyo) ^ 16u);
        nz = lds_read<vf4>(nb + nzo), fz = lds_read<vf4>((nb + nzo) ^ 16u);
        cw = lds_read<vu4>(nb + 96u);
      } else {
        nx = glb_read<vf4>(im.g, nb + nxo), fx = glb_read<vf4>(im.g, (nb + nxo) ^ 16u);
        ny = glb_read<vf4>(im.g, nb + nyo), fy = glb_read<vf4>(im.g, (nb + nyo) ^ 16u);
        nz = glb_read<vf4>(im.g, nb + nzo), fz = glb_read<vf4>(im.g, (nb + nzo) ^ 16u);
        cw = glb_read<vu4>(im.g, nb + 96u);
      }
      ++nnode;
#define RTOW_SLAB(c, slot)                                                                                   \
  const float tn##slot = fmaxf(fmaxf(fmaf(nx.c, ix, -oix), fmaf(ny.c, iy, -oiy)), fmaxf(fmaf(nz.c, iz, -oiz), tmin32)); \
  const float tf##slot = fminf(fminf(fmaf(fx.c, jx, -ojx), fmaf(fy.c, jy, -ojy)), fminf(fmaf(fz.c, jz, -ojz), tfm));    \
  const bool h##slot = tn##slot <= tf##slot;                                                                 \
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
#define RTOW_PUSH(slot, child)                                                                         \
  if (h##slot && s != slot##u) {                                                                       \
    const uint32_t e = ((__float_as_uint(tn##slot) >> 20) << 21) | child;                              \
    if (sa < sa_end)                                                                                   \
      lds_write<uint32_t>(sa, e);                                                                      \
    else                                                                                               \
      P.spill[(size_t)((sa - sa_end) >> kStrideLog2) * P.n_lanes + lane_g] = e;                        \
    sa += kStride;                                                                                     \
  }
      RTOW_PUSH(0, cw.x)
      RTOW_PUSH(1, cw.y)
      RTOW_PUSH(2, cw.z)
      RTOW_PUSH(3, cw.w)
#undef RTOW_PUSH
    }
    const unsigned long long m_walking = __ballot(cur != kRefNone);
    const bool any_walking = m_walking != 0ull;
    ++trips;
    const bool suspend = any_walking && trips >= cap && (uint32_t)__popcll(m_walking) <= max_open;
    // leaf phase: when enough lanes hold a queued leaf, or when no lane can take a node step (every
    // walking lane holds a leaf it cannot queue), or at the end
    const bool busy = cur != kRefNone && !((cur & kRefLeaf) != 0u && cur < kRefPop && q1 != kRefNone);
    const unsigned long long m_pending = __ballot(q0 != kRefNone);
    if ((m_pending != 0ull && ((uint32_t)__popcll(m_pending) >= P.leaf_votes || __ballot(busy) == 0ull)) || !any_walking ||
        suspend) {
      stamps.mark(RG_WALK);
      if constexpr (ST) stamps.phases += 1;
      // leaf phase: every lane tests the triangles of the OLDEST leaf it queued
      if (q0 != kRefNone) {
        const uint32_t first = (q0 & (kRefLeaf - 1u)) >> 2, count = (q0 & 3u) + 1u;
        for (uint32_t k = 0; k < count; ++k) {
          const uint32_t r = sc.b4_off_tri + 96u * (first + k);
          const vd2 t0 = im.t2(r), t1 = im.t2(r + 16u), t2 = im.t2(r + 32u), t3 = im.t2(r + 48u), t4 = im.t2(r + 64u),
                        t5 = im.t2(r + 80u);
          ++nprim;
          triangle_test<double>(ray.o64, ray.d64, V3d{t0.x, t0.y, t1.x}, V3d{t1.y, t2.x, t2.y}, V3d{t3.x, t3.y, t4.x},
                                V3d{t4.y, t5.x, t5.y}, (int)(first + k), RTOW_TMIN, best);
        }
      }
      q0 = q1;
      q1 = kRefNone;
      if (suspend && q0 != kRefNone) {  // both queued leaves before stopping
        const uint32_t first = (q0 & (kRefLeaf - 1u)) >> 2, count = (q0 & 3u) + 1u;
        for (uint32_t k = 0; k < count; ++k) {
          const uint32_t r = sc.b4_off_tri + 96u * (first + k);
          const vd2 t0 = im.t2(r), t1 = im.t2(r + 16u), t2 = im.t2(r + 32u), t3 = im.t2(r + 48u), t4 = im.t2(r + 64u),
                    t5 = im.t2(r + 80u);
          ++nprim;
          triangle_test<double>(ray.o64, ray.d64, V3d{t0.x, t0.y, t1.x}, V3d{t1.y, t2.x, t2.y}, V3d{t3.x, t3.y, t4.x},
                                V3d{t4.y, t5.x, t5.y}, (int)(first + k), RTOW_TMIN, best);
        }
        q0 = kRefNone;
      }
      tmax32 = round_up_f32(best.t);  // rounded up: never below the f64 value
      tfm = tmax32 * slack;
      stamps.mark(RG_LEAF);
      if (suspend) {
        w_cur = cur;  // kRefNone for the lanes that are done
        w_sa = sa;
        return best;
      }
      if (!any_walking && !__any(q0 != kRefNone)) break;
    }
  }
  w_cur = kRefNone;
  return best;
}
