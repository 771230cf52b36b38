// rtow_trace_reftree.h — part of the trace kernels (included by rtow_trace_body.h inside namespace rtow::{anonymous};
// see that file for the execution model).  The walk of the REFERENCE's tree (rtow_reftree.h), strict build only.
#pragma once
// ------------------------------------------------------ closest hit: REFTREE ---
// BVHNode::hit (src/render.cpp:52-71) over the tree of src/render.cpp:73-110 with Aabb::hit
// (src/common-model.h:71-84) as the box test, all in binary64 with the reference's operations:
//   inner node: left with tmax, then right with the left hit's t (or tmax); the right hit wins if there is one;
//   leaf: its primitives in array order, each accepted with tmin <= t <= upper bound, which shrinks.
// The recursion hands a subtree the closest t found so far in the subtrees before it and lets a later equal t
// win — which is what a preorder walk with one shrinking `best.t` and `<=` acceptance does, so the walk below
// is that recursion flattened (explicit stack of right children; depth bounded by the host: rtow_reftree.h).
// The box test keeps the reference's selects (not fmin / fmax) and its `t_max <= t_min` rejection, so NaNs
// from 0 * inf and zero-thickness boxes behave as they do there.
__device__ __forceinline__ Closest closest_hit_reftree(const DevScene &sc, V3d o, V3d d, double time, uint32_t &nnode,
                                                      uint32_t &nprim) {
  Closest best;
  best.t = (real)__builtin_huge_val();  // tmax = +inf, src/render.cpp:34
  best.prim = -1;
  const double tmin = RTOW_TMIN;
  const double a = dot(d, d);
  const double inv_a = fast_rcp(a);  // (strict build: unused by sphere_resolve)
  // 1.0F / r.direction()[a] (src/common-model.h:73): the same quotient for every box of the walk
  const double inv[3] = {1.0 / d.x, 1.0 / d.y, 1.0 / d.z};
  const double org[3] = {o.x, o.y, o.z};
  const unsigned char *tree = sc.rtree;
  const int32_t *ids = reinterpret_cast<const int32_t *>(sc.rtree + sc.rt_off_ids);
  uint32_t stack[kRefStackDepth];
  int sp = 0;
  uint32_t node = 0u;
  for (;;) {
    const double *bx = reinterpret_cast<const double *>(tree + (size_t)node * kRefNodeBytes);
    const uint2 ab = *reinterpret_cast<const uint2 *>(tree + (size_t)node * kRefNodeBytes + 48u);
    ++nnode;
    double t_min = tmin, t_max = (double)best.t;
    bool hit = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      double t0 = (bx[k] - org[k]) * inv[k];
      double t1 = (bx[3 + k] - org[k]) * inv[k];
      if (inv[k] < 0.0) {
        const double s = t0;
        t0 = t1;
        t1 = s;
      }
      t_min = t0 > t_min ? t0 : t_min;
      t_max = t1 < t_max ? t1 : t_max;
      if (t_max <= t_min) {
        hit = false;
        break;
      }
    }
    if (hit) {
      if ((ab.y & kRefLeafFlag) == 0u) {  // inner: left now, right afterwards
        if (sp < kRefStackDepth) stack[sp++] = ab.y;
        node = ab.x;
        continue;
      }
      const uint32_t first = ab.x, count = ab.y & ~kRefLeafFlag;
      for (uint32_t q = 0; q < count; ++q) {
        const int pid = ids[first + q];
        ++nprim;
        if (pid < sc.n_sph) {
          const double *g = sc.sph + 4 * (size_t)pid;
          sphere_test<double>(o, d, a, inv_a, g[0], g[1], g[2], g[3], pid, tmin, best);
        } else if (pid < sc.n_sph + sc.n_mov) {
          const double *g = sc.mov + 8 * (size_t)(pid - sc.n_sph);
          // center(time) = c0 + time*(c1-c0), src/oo-primitives.h:64-66 with t0=0, t1=1
          const double cx = g[0] + time * g[3], cy = g[1] + time * g[4], cz = g[2] + time * g[5];
          sphere_test<double>(o, d, a, inv_a, cx, cy, cz, g[6], pid, tmin, best);
        } else {
          const double *g = sc.tri + 12 * (size_t)(pid - sc.n_sph - sc.n_mov);
          triangle_test<double>(o, d, V3d{g[0], g[1], g[2]}, V3d{g[3], g[4], g[5]}, V3d{g[6], g[7], g[8]},
                                V3d{g[9], g[10], g[11]}, pid, tmin, best);
        }
      }
    }
    if (sp == 0) break;
    node = stack[--sp];
  }
  return best;
}
