// rtow_trace_sm4.h — part of the trace kernels (included by rtow_trace_body.h inside namespace rtow::{anonymous}).
// The BVH4 kernel for triangle meshes as a per-lane STATE MACHINE.
//
// The trip-structured kernels of rtow_trace_body.h advance every lane by exactly one ray segment per trip:
// the closest-hit walk of a trip lasts as long as its slowest lane (28 wave-level node steps on suzanne for
// 11 useful ones per lane), and lanes whose walk is short — rays that miss everything, 45 % of all segments —
// idle for most of it.  Here the walk loop IS the main loop, and the other stages of a path run inside it
// whenever enough lanes wait for them:
//
//   one trip of the loop =  [restart block]  lanes whose ray escaped: sky colour, pixel accumulate, next
//                                            sample (or next work item) and its camera ray
//                           [scatter block]  lanes whose ray hit: hit record, material, scattered ray
//                           [walk start]     f32 ray forms, root node
//                           [node step]      one 4-wide node for every lane with a node in hand
//                           [leaf phase]     triangles of queued leaves
//
// A block runs when at least sm4_restart / sm4_scatter / sm4_leaf lanes of the wave wait for it (launch
// parameters), or when no lane can walk (so the loop always makes progress).  A lane that escapes after a few nodes is back on a new
// camera ray a few trips later instead of waiting for the longest walk of the wave.  The expensive
// blocks still run at useful occupancy because they wait for their quorum.
//
// The image does not depend on any of this: a lane owns its work item, traces its samples in order and
// adds their colours in order (src/render.cpp:151-166); every random number is a function of
// (seed, pixel, sample, request).  The strict build keeps the reference's attenuation order with the
// per-lane path stack in HBM, exactly like rtow_trace_body.h.  (No end-of-launch sample donation here:
// the mesh launches are long — hundreds of trips per lane — and a trip is short.)
#pragma once

template <bool FULL, bool STAMPS>
__global__ void __launch_bounds__(1024) RTOW_CAT(rtow_trace4_, RTOW_SUFFIX)(const TraceParams P) {
  const DevScene &sc = P.sc;
  const uint32_t k0 = P.seed_lo, k1 = P.seed_hi;
  const unsigned lane = lane_id();
  const uint32_t lane_g = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t npix_local = (uint32_t)P.local_rows * (uint32_t)P.W;
  const uint32_t kRestartVotes = P.sm4_restart, kScatterVotes = P.sm4_scatter, kLeafVotes = P.sm4_leaf;  // (wave-uniform)

  Bvh4Reader<FULL> im;
  im.g = sc.blob4;
  im.lds_limit = sc.b4_lds_limit;
  im.aux_src = sc.b4_aux_src;
  im.aux_lds = sc.b4_aux_lds;
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(sc.blob4);
    uint4 *dst = reinterpret_cast<uint4 *>(rtow_lds);
    const uint32_t n16 = sc.b4_lds_limit / 16u;
    for (uint32_t i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
    if (sc.b4_aux_src < sc.blob4_bytes) {  // the end of the image (materials, material indices)
      const uint4 *asrc = reinterpret_cast<const uint4 *>(sc.blob4 + sc.b4_aux_src);
      uint4 *adst = reinterpret_cast<uint4 *>(rtow_lds + sc.b4_aux_lds);
      const uint32_t a16 = (sc.blob4_bytes - sc.b4_aux_src) / 16u;
      for (uint32_t i = threadIdx.x; i < a16; i += blockDim.x) adst[i] = asrc[i];
    }
    __syncthreads();
  }

  // ---- per-lane state ---------------------------------------------------------------------
  enum : uint32_t { PH_ITEM = 0, PH_SAMPLE, PH_START, PH_WALK, PH_HIT, PH_MISS, PH_DEAD };
  uint32_t phase = PH_ITEM;
  int s_left = 0;             // samples left in the current item
  uint32_t item = 0xffffffffu;
  uint32_t j = 0, gi = 0;     // column, global row (from the top)
  V3d acc = {0.0, 0.0, 0.0};  // pixel_color of this item (src/render.cpp:156)
  V3 ro = {0, 0, 0}, rd = {0, 0, 1};
  int depth = 0;              // remaining child rays
  [[maybe_unused]] int nb = 0;  // bounces recorded on the path stack (strict build)
#ifdef RTOW_FAST_MATH
  V3 throughput = {1, 1, 1};  // see rtow_trace_body.h: the fast builds multiply forward
#endif
  Rng g = {0, 0, 0};
  uint32_t nseg = 0, nnode = 0, nprim = 0;
  ItemPool pool;
  const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
  Stamps<STAMPS> stamps;
  stamps.start();
  if constexpr (STAMPS) {
    if (lane == 0) atomicMin(P.t_origin, (unsigned long long)__builtin_amdgcn_s_memrealtime());
  }

  // ---- walk state (live for the lanes in PH_WALK; rtow_trace_bvh4.h) ------------------------------
  const Bvh4Stack stk = bvh4_stack(sc);
  Bvh4Ray ray = bvh4_ray<FULL>(sc, ro, rd);
  float tmax32 = 0;
  uint32_t sa = stk.lds;
  uint32_t cur = kRefNone, q0 = kRefNone, q1 = kRefNone;
  Closest best;
  best.t = 0;
  best.prim = -1;

  for (;;) {
    if constexpr (STAMPS) stamps.trips += 1;
    const unsigned long long m_walk = __ballot(phase == PH_WALK || phase == PH_START);
    // ---- restart block -----------------------------------------------------------------------
    {
      const unsigned long long m = __ballot(phase == PH_MISS || phase == PH_ITEM || phase == PH_SAMPLE);
      if (m != 0ull && ((uint32_t)__popcll(m) >= kRestartVotes || m_walk == 0ull)) {
        if (phase == PH_MISS) {
          // background + unwind of the recursion (src/render.cpp:119,122-128)
          const V3 unit = normalize(rd);
          const real t = real(0.5) * (unit.y + real(+1.0));
          V3 c = (real(1.0) - t) * V3{1, 1, 1} + t * V3{real(0.5), real(0.7), real(1.0)};
#ifdef RTOW_FAST_MATH
          c = throughput * c;
#else
          for (int q = nb - 1; q >= 0; --q) {
            const uint32_t smi = P.stack[(size_t)q * P.n_lanes + lane_g];
            const uint32_t mr = sc.b4_off_mats + 48u * smi;
            const vd2 a0 = im.d2(mr), a1 = im.d2(mr + 16u);
            c = V3{(real)a0.x, (real)a0.y, (real)a1.x} * c;
          }
#endif
          V3d cd = to_f64(c);
          asm volatile("" : "+v"(cd.x), "+v"(cd.y), "+v"(cd.z));  // never fused into the multiply above
          acc = acc + cd;  // pixel_color += ray_color(...)
          --s_left;
          ++g.sample;
          phase = s_left > 0 ? PH_SAMPLE : PH_ITEM;
        }
        // finished items go to their partial-sum slot, new ones come from the queue
        const bool need_item = phase == PH_ITEM;
        const unsigned long long need_mask = __ballot(need_item);
        if (need_mask != 0ull) {
          const RTOW_CONST TraceParams *kp = (const RTOW_CONST TraceParams *)__builtin_amdgcn_kernarg_segment_ptr();
          asm volatile("" : "+s"(kp));  // opaque per trip: keeps the loads from being hoisted out of the loop
          if (need_item && item != 0xffffffffu) {
            double *dst = P.partials + (size_t)item * 3;
            dst[0] = acc.x;
            dst[1] = acc.y;
            dst[2] = acc.z;
          }
          const unsigned long long mine = take_items(pool, need_mask, lane, n_waves, kp, P.counters);
          if (need_item) {
            if (mine >= (unsigned long long)kp->n_items) {
              phase = PH_DEAD;
              item = 0xffffffffu;
            } else {
              const ItemPos ip = decode_item(kp, (uint32_t)mine, npix_local);
              item = ip.item;
              j = ip.j;
              gi = ip.gi;
              g.pixel = gi * (uint32_t)kp->W + j;
              g.sample = ip.sample0;
              s_left = ip.count;
              acc = {0.0, 0.0, 0.0};
              phase = s_left > 0 ? PH_SAMPLE : PH_ITEM;
            }
          }
        }
        if (phase == PH_SAMPLE) {
          real rtime;
          camera_ray(P, g, k0, k1, j, gi, ro, rd, rtime);
          (void)rtime;  // triangles do not move
          depth = P.max_child_rays;
          nb = 0;
#ifdef RTOW_FAST_MATH
          throughput = {1, 1, 1};
#endif
          phase = PH_START;
        }
      }
    }
    stamps.mark(RG_REGEN);
    // ---- scatter block -----------------------------------------------------------------------
    {
      const unsigned long long m = __ballot(phase == PH_HIT);
      if (m != 0ull && ((uint32_t)__popcll(m) >= kScatterVotes || __ballot(phase == PH_WALK || phase == PH_START) == 0ull)) {
        if (phase == PH_HIT) {
          bool ended = true;  // src/render.cpp:115,120: black
          if (depth > 0) {
            // the Hit of the winner (src/common-model.cpp:121): un-normalised normal e1 x e2, front facing
            const int pid = best.prim;
            const V3 where = ro + rd * best.t;
            const uint32_t r = sc.b4_off_tri + 96u * (uint32_t)pid;
            const vd2 q4 = im.t2(r + 64u), q5 = im.t2(r + 80u);
            const V3 normal = {(real)q4.y, (real)q5.x, (real)q5.y};
            const int mi = (int)im.u32(sc.b4_off_pmat + 4u * (uint32_t)pid);
            const uint32_t mr = sc.b4_off_mats + 48u * (uint32_t)mi;
            const vd2 m1 = im.d2(mr + 16u), m2 = im.d2(mr + 32u);  // {att.z, fuzz}, {ir, kind|pad}
            const int kind = (int)(__double_as_longlong(m2.y) & 0xffffffffll);
            V3 dir;
            if (scatter_dir(g, k0, k1, kind, (real)m1.y, (real)m2.x, rd, normal, true, dir)) {
#ifdef RTOW_FAST_MATH
              const vd2 m0 = im.d2(mr);  // {att.x, att.y}
              throughput = throughput * V3{(real)m0.x, (real)m0.y, (real)m1.x};
#else
              P.stack[(size_t)nb * P.n_lanes + lane_g] = (uint32_t)mi;
#endif
              ++nb;
              --depth;
              ro = where;
              rd = dir;
              ended = false;
            }
          }
          if (ended) {
            --s_left;
            ++g.sample;
            phase = s_left > 0 ? PH_SAMPLE : PH_ITEM;
          } else {
            phase = PH_START;
          }
        }
      }
    }
    stamps.mark(RG_SHADE);
    // ---- walk start --------------------------------------------------------------------------
    if (phase == PH_START) {
      ray = bvh4_ray<FULL>(sc, ro, rd);
      tmax32 = __builtin_huge_valf();
      sa = stk.lds;
      cur = 0u;  // the root
      q0 = q1 = kRefNone;
      best.t = (real)__builtin_huge_val();
      best.prim = -1;
      ++nseg;
      phase = PH_WALK;
    }
    // ---- one step of the walk (rtow_trace_bvh4.h) --------------------------------------------------
    const bool walking = phase == PH_WALK;
    if (__ballot(walking) != 0ull) {
      if constexpr (STAMPS) stamps.iters += 1;
      if (walking) bvh4_step<FULL, false>(im, P, ray, tmax32, stk, lane_g, cur, sa, q0, q1, nnode);
      stamps.mark(RG_WALK);
      // Leaf phase: when enough lanes hold a queued leaf, or when no lane can take a step (every
      // walking lane either has nothing in hand or holds a leaf it cannot queue).
      const bool pending = walking && q0 != kRefNone;
      const unsigned long long m_pending = __ballot(pending);
      if (m_pending != 0ull && ((uint32_t)__popcll(m_pending) >= kLeafVotes || __ballot(walking && bvh4_busy(cur, q1)) == 0ull)) {
        if constexpr (STAMPS) stamps.phases += 1;
        if (pending) {
          bvh4_leaf<FULL>(im, sc, q0, to_f64(ro), to_f64(rd), best, nprim);
          q0 = q1;
          q1 = kRefNone;
          tmax32 = round_up_f32(best.t);  // rounded up: never below the f64 value
        }
        stamps.mark(RG_LEAF);
      }
      // a walk is over when nothing is in hand, on the stack or in the queue
      if (walking && cur == kRefNone && q0 == kRefNone) phase = best.prim >= 0 ? PH_HIT : PH_MISS;
    }
    if (__ballot(phase != PH_DEAD) == 0ull) break;
  }

  if constexpr (STAMPS) {
    if (lane == 0) {
      const unsigned long long tend = __builtin_amdgcn_s_memrealtime();
      atomicAdd(&P.counters[4], tend - P.t_origin[0]);
      atomicMin(&P.counters[5], tend - P.t_origin[0]);
      atomicMax(&P.counters[6], tend - P.t_origin[0]);
      for (int r = 0; r < RG_COUNT; ++r) atomicAdd(&P.counters[8 + r], stamps.t[r]);
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
    atomicAdd(&P.counters[2], t1);
    atomicAdd(&P.counters[3], t2);
  }
}
