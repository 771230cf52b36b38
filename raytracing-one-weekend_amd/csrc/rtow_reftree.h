// rtow_reftree.h — the reference's OWN bounding-volume tree, built on the host for the opt-in kernel
// RTOW_KERNEL_REFTREE.
//
// The default kernels find closest hits through acceleration structures of their own (SAH BVH, 4-wide BVH,
// uniform grid) over exact, padded bounds: the closest hit does not depend on the tree, so they give the
// reference's image on every scene the reference renders correctly.  The reference's tree has quirks of its
// own, though, and where they bite it renders something else:
//   * a leaf's box starts from the default box — two points at the origin — and grows by the leaf's
//     primitives (src/render.cpp:74-79), so it always contains the origin;
//   * Aabb::hit rejects a box the ray crosses in zero parameter length, `t_max <= t_min`
//     (src/common-model.h:80): a flat box is never entered;
//   * triangle boxes pass through glm::vec3 = float (src/common-model.cpp:127-134);
//   * sphere boxes are centre -/+ radius with the SIGNED radius (src/common-model.cpp:168-171,197-207);
//   * inner nodes visit left then right, whatever the ray direction, and a later equal `t` wins.
// This file restates that build — median split of the insertion-ordered primitive array by std::sort on
// bbox.min[axis], axis from the first and last primitive's boxes, leaves of 1..6 (src/render.cpp:73-110) —
// so that the kernel can walk the very same boxes in the very same order with the reference's f64 box test.
// std::sort is unstable; the permutation it leaves depends only on the comparison results, so sorting ids
// with libstdc++'s std::sort reproduces the reference's order for equal keys as well.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/rtow.h"
#include "rtow_device.h"

namespace rtow {

// node record (rtow_device.h): kRefNodeBytes = 64: f64 min[3], max[3]; u32 a, b; 8 B of padding.  Inner: a, b =
// left, right node; leaf: b = kRefLeafFlag | count, a = first slot of the id list.  kRefStackDepth = 48 bounds
// the kernel's traversal stack (the tree is balanced: depth <= log2 n + 1).

struct RefTree {
  std::vector<unsigned char> blob;  // [nodes x 64 B][class-major primitive ids, in the reference's leaf order]
  uint32_t off_ids = 0;
  int32_t n_nodes = 0;
  int32_t depth = 0;
  double stupid_volume = 0.0;  // the reference's diagnostic, src/render.cpp:36-50,148
  bool ok = false;
};

namespace reftree_detail {

struct Box {
  double mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};  // default-constructed Aabb: two zero points (src/common-model.h:86-87)
  double volume() const { return (mx[0] - mn[0]) * (mx[1] - mn[1]) * (mx[2] - mn[2]); }
};

// src/common-model.cpp:185-195
inline Box surrounding(const Box &p, const Box &q) {
  Box r;
  for (int k = 0; k < 3; ++k) {
    r.mn[k] = std::fmin(p.mn[k], q.mn[k]);
    r.mx[k] = std::fmax(p.mx[k], q.mx[k]);
  }
  return r;
}

struct Builder {
  std::vector<Box> pbox;       // per primitive, by class-major id
  std::vector<int32_t> order;  // the reference's primitive array (ids), sorted in place range by range
  struct Node {
    Box box;
    int32_t left = -1, right = -1, lo = 0, hi = 0;
  };
  std::vector<Node> nodes;
  int depth = 0;

  int build(int lo, int hi, int level) {
    const int me = (int)nodes.size();
    nodes.emplace_back();
    depth = std::max(depth, level);
    const int n = hi - lo;
    if (n >= 1 && n <= 6) {  // src/render.cpp:74-79
      Box b;
      for (int i = lo; i < hi; ++i) b = surrounding(b, pbox[(size_t)order[(size_t)i]]);
      nodes[(size_t)me].box = b;
      nodes[(size_t)me].lo = lo;
      nodes[(size_t)me].hi = hi;
      return me;
    }
    // src/render.cpp:80-93: the axis along which the first and the last primitive's boxes lie furthest apart
    const Box &first = pbox[(size_t)order[(size_t)lo]], &last = pbox[(size_t)order[(size_t)hi - 1]];
    const double dx = std::fabs(last.mn[0] - first.mn[0]), dy = std::fabs(last.mn[1] - first.mn[1]),
                 dz = std::fabs(last.mn[2] - first.mn[2]);
    const int axis = dx > dy ? (dx > dz ? 0 : 2) : (dy > dz ? 1 : 2);
    std::sort(order.begin() + lo, order.begin() + hi,  // src/render.cpp:94-101
              [&](int32_t p, int32_t q) { return pbox[(size_t)p].mn[axis] < pbox[(size_t)q].mn[axis]; });
    const int leftn = n / 2;  // src/render.cpp:103-106
    const int l = build(lo, lo + leftn, level + 1);
    const int r = build(lo + leftn, hi, level + 1);
    nodes[(size_t)me].left = l;
    nodes[(size_t)me].right = r;
    nodes[(size_t)me].box = surrounding(nodes[(size_t)l].box, nodes[(size_t)r].box);
    return me;
  }

  double stupid(int ni) const {  // src/render.cpp:36-50
    const Node &nd = nodes[(size_t)ni];
    double myown = nd.box.volume(), childrens = 0;
    if (nd.left >= 0 && nd.right >= 0) {
      myown -= nodes[(size_t)nd.left].box.volume();
      myown -= nodes[(size_t)nd.right].box.volume();
      childrens += stupid(nd.left);
      childrens += stupid(nd.right);
    } else {
      myown = 0;
    }
    myown = myown < 0 ? -myown : 0;
    return myown + childrens;
  }
};

}  // namespace reftree_detail

// `s`: the flattened scene as the caller handed it over (geometry as constructed, insertion order in
// prim_kind / prim_index; without them the classes follow each other).  Compile with -ffp-contract=off.
inline void build_reftree(const rtow_scene_t *s, RefTree &out) {
  using namespace reftree_detail;
  const int ns = s->n_spheres, nm = s->n_moving, nt = s->n_triangles, n = ns + nm + nt;
  Builder B;
  B.pbox.resize((size_t)n);
  for (int i = 0; i < ns; ++i) {  // src/common-model.cpp:168-171
    const double *g = s->sphere_geom + 4 * (size_t)i;
    Box &b = B.pbox[(size_t)i];
    for (int k = 0; k < 3; ++k) {
      b.mn[k] = g[k] - g[3];
      b.mx[k] = g[k] + g[3];
    }
  }
  for (int i = 0; i < nm; ++i) {  // src/common-model.cpp:197-207, centre(time) src/oo-primitives.h:64-66 (t0 = 0, t1 = 1)
    const double *g = s->moving_geom + 8 * (size_t)i;
    Box b0, b1;
    const double f0 = (0.0 - 0.0) / (1.0 - 0.0), f1 = (1.0 - 0.0) / (1.0 - 0.0);
    for (int k = 0; k < 3; ++k) {
      const double c0 = g[k] + f0 * (g[3 + k] - g[k]), c1 = g[k] + f1 * (g[3 + k] - g[k]);
      b0.mn[k] = c0 - g[6];
      b0.mx[k] = c0 + g[6];
      b1.mn[k] = c1 - g[6];
      b1.mx[k] = c1 + g[6];
    }
    B.pbox[(size_t)ns + i] = surrounding(b0, b1);
  }
  for (int i = 0; i < nt; ++i) {  // src/common-model.cpp:127-134: the corners pass through glm::vec3 = float
    const double *g = s->triangle_geom + 9 * (size_t)i;
    Box &b = B.pbox[(size_t)ns + nm + i];
    for (int k = 0; k < 3; ++k) {
      b.mn[k] = (double)(float)std::min({g[k], g[3 + k], g[6 + k]});
      b.mx[k] = (double)(float)std::max({g[k], g[3 + k], g[6 + k]});
    }
  }
  B.order.resize((size_t)n);
  if (s->prim_kind && s->prim_index) {
    const int base[3] = {0, ns, ns + nm};
    for (int i = 0; i < n; ++i) B.order[(size_t)i] = base[s->prim_kind[i]] + s->prim_index[i];
  } else {
    for (int i = 0; i < n; ++i) B.order[(size_t)i] = i;
  }
  B.nodes.reserve((size_t)std::max(n / 2, 1));
  B.build(0, n, 0);
  out.n_nodes = (int32_t)B.nodes.size();
  out.depth = B.depth;
  out.stupid_volume = B.stupid(0);
  out.off_ids = (uint32_t)B.nodes.size() * kRefNodeBytes;
  out.blob.assign((size_t)out.off_ids + (((size_t)n * 4 + 15) / 16) * 16, 0);
  for (size_t i = 0; i < B.nodes.size(); ++i) {
    const Builder::Node &nd = B.nodes[i];
    unsigned char *p = out.blob.data() + i * kRefNodeBytes;
    std::memcpy(p, nd.box.mn, 24);
    std::memcpy(p + 24, nd.box.mx, 24);
    uint32_t ab[2];
    if (nd.left >= 0) {
      ab[0] = (uint32_t)nd.left;
      ab[1] = (uint32_t)nd.right;
    } else {
      ab[0] = (uint32_t)nd.lo;
      ab[1] = kRefLeafFlag | (uint32_t)(nd.hi - nd.lo);
    }
    std::memcpy(p + 48, ab, 8);
  }
  std::memcpy(out.blob.data() + out.off_ids, B.order.data(), (size_t)n * 4);
  out.ok = n > 0 && out.depth + 1 < kRefStackDepth;
}

}  // namespace rtow
