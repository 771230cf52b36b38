// rtow_bvh.h — host build of the device BVH (binned SAH, BVH2) and of the scene
// image the BVH kernel walks (threaded depth-first layout, padded f32 boxes).
//
// This is NOT the reference's tree (src/render.cpp:73-110 splits at the median
// of one heuristic axis and unions every leaf box with the origin, so it culls
// poorly — SURVEY.md §3.2).  The closest hit does not depend on the tree: any
// conservative BVH returns the same (t, primitive) as testing every primitive,
// so the device is free to use a better one.  Boxes are padded so that rounding
// in the slab test can never cull a primitive the exact hit test would accept.
//
// Build output (HostBvh): node 0 is the root; the two children of an inner node are
// adjacent (left, left+1).
//   box [n][6]  : min xyz, max xyz (f64, exact primitive bounds)
//   link[n][4]  : inner: {left child, 0, parent, 0}; leaf: {first, count, parent, 0}
//   prim[]      : class-major primitive ids, leaf ranges contiguous
// make_scene_image() then renumbers the nodes depth-first (the child nearer to the
// camera first) and emits the 32-byte threaded node records of rtow_device.h.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <future>
#include <system_error>
#include <vector>

namespace rtow {

struct HostBvh {
  std::vector<double> box;
  std::vector<int32_t> link;
  std::vector<int32_t> prim;
  int depth = 0;
};

namespace bvh_detail {

struct Box {
  double mn[3], mx[3];
  void reset() {
    for (int k = 0; k < 3; ++k) {
      mn[k] = INFINITY;
      mx[k] = -INFINITY;
    }
  }
  void grow(const Box &b) {
    for (int k = 0; k < 3; ++k) {
      mn[k] = std::min(mn[k], b.mn[k]);
      mx[k] = std::max(mx[k], b.mx[k]);
    }
  }
  double half_area() const {
    double dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    if (!(dx >= 0) || !(dy >= 0) || !(dz >= 0)) return 0.0;
    return dx * dy + dy * dz + dz * dx;
  }
};

struct Builder {
  std::vector<Box> pb;        // primitive bounds
  std::vector<double> cen;    // centroids [n][3]
  std::vector<int32_t> order; // permutation being partitioned
  HostBvh *out;
  // Big subtrees are built concurrently (the two children of a node are independent: they partition disjoint
  // ranges of `order`, and a leaf's primitives are its range of `order`, so nothing is appended to shared vectors).
  // Node numbers come from the primitive RANGE, not from a counter: the subtree over n primitives owns a contiguous
  // region of 2n - 2 slots below its root — the children's pair first, then the left child's region, then the
  // right's — so that concurrent subtrees write to disjoint stretches of the arrays (round 4: with pairs handed out
  // by one atomic counter the threads' nodes interleaved, every cache line of the node arrays was shared, and 16
  // threads were 1.5x faster than one; now 3.5x), a subtree's nodes are memory neighbours for the image builders
  // that walk it afterwards, and neither the tree nor its numbering depends on the timing.  Slots a subtree does not
  // need (its leaves hold more than one primitive) stay zero and are never linked.
  std::atomic<int> max_depth{0};
  int parallel_min = 2048;                   // primitives below which a subtree is built by its parent's thread
  static constexpr int kParallelDepth = 6;   // up to 64 concurrent subtrees (the 96.8k-triangle mesh on the 16-core share of the GPU box: 12.3 ms at 4, 9.3 at 6)
  // tests: 1 = pretend no thread can be started (the std::system_error path below)
  int fail_thread_start = 0;
#ifndef RTOW_SAH_BINS
#define RTOW_SAH_BINS 16
#endif
  static constexpr int kBins = RTOW_SAH_BINS;
  int leaf_max = 4;                      // <= 7 (3 bits in the leaf word)
  double c_trav = 0.0;                   // cost of descending one level, in primitive tests
  static constexpr int kSahDepth = 48;   // below this depth fall back to median splits

  void set_node(int idx, const Box &b, int a, int c, int parent) {
    for (int k = 0; k < 3; ++k) {
      out->box[(size_t)idx * 6 + k] = b.mn[k];
      out->box[(size_t)idx * 6 + 3 + k] = b.mx[k];
    }
    out->link[(size_t)idx * 4 + 0] = a;
    out->link[(size_t)idx * 4 + 1] = c;
    out->link[(size_t)idx * 4 + 2] = parent;
    out->link[(size_t)idx * 4 + 3] = 0;
  }

  // `region`: first slot of the 2 (hi - lo) - 2 slots for the nodes below `idx`
  void build(int idx, int lo, int hi, int parent, int depth, int region) {
    for (int d = max_depth.load(std::memory_order_relaxed); depth > d && !max_depth.compare_exchange_weak(d, depth);) {
    }
    Box b, cb;
    b.reset();
    cb.reset();
    for (int i = lo; i < hi; ++i) {
      const int p = order[i];
      b.grow(pb[p]);
      for (int k = 0; k < 3; ++k) {
        cb.mn[k] = std::min(cb.mn[k], cen[(size_t)p * 3 + k]);
        cb.mx[k] = std::max(cb.mx[k], cen[(size_t)p * 3 + k]);
      }
    }
    const int n = hi - lo;
    auto make_leaf = [&]() { set_node(idx, b, lo, n, parent); };  // its primitives: order[lo, hi)
    if (n <= 1) {
      make_leaf();
      return;
    }
    // binned SAH over the three axes: ONE pass over the primitives fills the bins of all three (round 4: the three
    // separate passes were a third of the 96.8k-triangle mesh's build; same arithmetic, same tree)
    int best_axis = -1, best_split = -1;
    double best_cost = INFINITY;
    double c0s[3], scales[3];
    bool use[3];
    for (int ax = 0; ax < 3; ++ax) {
      c0s[ax] = cb.mn[ax];
      use[ax] = cb.mx[ax] > cb.mn[ax];
      scales[ax] = use[ax] ? kBins / (cb.mx[ax] - cb.mn[ax]) : 0.0;
    }
    auto bin_of = [&](int p, int ax) {
      int bi = (int)((cen[(size_t)p * 3 + ax] - c0s[ax]) * scales[ax]);
      return std::min(std::max(bi, 0), kBins - 1);
    };
    constexpr int kSmall = 16;
    if (n <= kSmall) {
      // Few primitives (most nodes of a tree are down here): the same splits evaluated without the 48 bins — the
      // primitives are ordered by their bin along the axis and a split is tried after every bin that holds something
      // (a split after an EMPTY bin has the cost of the one before it and never wins `<`).  Unions of boxes are
      // min / max, so the areas — and the tree — are the binned sweep's, bit for bit; resetting and sweeping 3 x 16
      // bins per node was a third of the 96.8k-triangle mesh's build.
      for (int ax = 0; ax < 3; ++ax) {
        if (!use[ax]) continue;
        int bi[kSmall], ord[kSmall];
        for (int i = 0; i < n; ++i) {
          bi[i] = bin_of(order[lo + i], ax);
          int j = i;
          for (; j > 0 && bi[ord[j - 1]] > bi[i]; --j) ord[j] = ord[j - 1];
          ord[j] = i;
        }
        double right_area[kSmall];  // union of the primitives at sorted positions >= j
        Box acc;
        acc.reset();
        for (int j = n - 1; j >= 1; --j) {
          acc.grow(pb[order[lo + ord[j]]]);
          right_area[j] = acc.half_area();
        }
        acc.reset();
        for (int j = 0; j < n - 1; ++j) {
          acc.grow(pb[order[lo + ord[j]]]);
          const int k = bi[ord[j]];
          if (bi[ord[j + 1]] == k || k >= kBins - 1) continue;  // the bin is not complete yet / no split after the last bin
          const double cost = acc.half_area() * (j + 1) + right_area[j + 1] * (n - 1 - j);
          if (cost < best_cost) {
            best_cost = cost;
            best_axis = ax;
            best_split = k;
          }
        }
      }
    } else {
    Box bins[3][kBins];
    int cnt[3][kBins];
    for (int ax = 0; ax < 3; ++ax)
      for (int k = 0; k < kBins; ++k) {
        bins[ax][k].reset();
        cnt[ax][k] = 0;
      }
    for (int i = lo; i < hi; ++i) {
      const int p = order[i];
      for (int ax = 0; ax < 3; ++ax) {
        if (!use[ax]) continue;
        const int bi = bin_of(p, ax);
        bins[ax][bi].grow(pb[p]);
        cnt[ax][bi]++;
      }
    }
    for (int ax = 0; ax < 3; ++ax) {
      if (!use[ax]) continue;
      double right_area[kBins];
      int right_cnt[kBins];
      Box acc;
      acc.reset();
      int c = 0;
      for (int k = kBins - 1; k >= 1; --k) {
        acc.grow(bins[ax][k]);
        c += cnt[ax][k];
        right_area[k] = acc.half_area();
        right_cnt[k] = c;
      }
      acc.reset();
      c = 0;
      for (int k = 0; k < kBins - 1; ++k) {
        acc.grow(bins[ax][k]);
        c += cnt[ax][k];
        if (c == 0 || right_cnt[k + 1] == 0) continue;
        const double cost = acc.half_area() * c + right_area[k + 1] * right_cnt[k + 1];
        if (cost < best_cost) {
          best_cost = cost;
          best_axis = ax;
          best_split = k;
        }
      }
    }
    }
    int mid = -1;
    if (best_axis >= 0) {
      // SAH: split only if  C_trav + (A_L N_L + A_R N_R) / A_P  <  N   (costs in primitive tests)
      const double leaf_cost = b.half_area() * n;
      if (n <= leaf_max && best_cost + c_trav * b.half_area() >= leaf_cost) {
        make_leaf();
        return;
      }
      const double c0 = cb.mn[best_axis], c1 = cb.mx[best_axis];
      const double scale = kBins / (c1 - c0);
      auto it = std::partition(order.begin() + lo, order.begin() + hi, [&](int p) {
        int bi = (int)((cen[(size_t)p * 3 + best_axis] - c0) * scale);
        bi = std::min(std::max(bi, 0), kBins - 1);
        return bi <= best_split;
      });
      mid = (int)(it - order.begin());
    }
    if (mid <= lo || mid >= hi || depth >= kSahDepth) {
      // degenerate (coincident centroids) or too deep: median split on the widest axis
      if (best_axis < 0 && n <= leaf_max) {
        make_leaf();
        return;
      }
      int ax = 0;
      for (int k = 1; k < 3; ++k)
        if (cb.mx[k] - cb.mn[k] > cb.mx[ax] - cb.mn[ax]) ax = k;
      mid = lo + n / 2;
      std::nth_element(order.begin() + lo, order.begin() + mid, order.begin() + hi, [&](int p, int q) {
        return cen[(size_t)p * 3 + ax] < cen[(size_t)q * 3 + ax];
      });
    }
    const int left = region, region_l = region + 2, region_r = region + 2 * (mid - lo);
    set_node(idx, b, left, 0, parent);
    if (n >= parallel_min && depth < kParallelDepth) {
      // A thread that cannot be started (process limit of the machine: std::async throws std::system_error)
      // is not an error: its subtree is built here.  The tree is the same either way.
      std::future<void> other;
      bool started = false;
      try {
        if (fail_thread_start) throw std::system_error(std::make_error_code(std::errc::resource_unavailable_try_again));
        other = std::async(std::launch::async, [this, left, lo, mid, idx, depth, region_l] { build(left, lo, mid, idx, depth + 1, region_l); });
        started = true;
      } catch (const std::system_error &) {
      }
      if (!started) build(left, lo, mid, idx, depth + 1, region_l);
      build(left + 1, mid, hi, idx, depth + 1, region_r);
      if (started) other.get();
    } else {
      build(left, lo, mid, idx, depth + 1, region_l);
      build(left + 1, mid, hi, idx, depth + 1, region_r);
    }
  }
};

// fn(i0, i1) over [0, n) in up to 16 concurrent stretches of at least `min_stretch` items (one stretch: the caller's
// thread alone).  Like the builder's subtrees: a thread that cannot be started is not an error, its stretch runs here.
template <class F>
inline void for_stretches(size_t n, size_t min_stretch, F &&fn) {
  const size_t parts = std::min<size_t>(16, std::max<size_t>(1, n / std::max<size_t>(min_stretch, 1)));
  if (parts <= 1) {
    fn((size_t)0, n);
    return;
  }
  std::vector<std::future<void>> started;
  started.reserve(parts);
  for (size_t p = 1; p < parts; ++p) {
    const size_t i0 = n * p / parts, i1 = n * (p + 1) / parts;
    try {
      started.push_back(std::async(std::launch::async, [&fn, i0, i1] { fn(i0, i1); }));
    } catch (const std::system_error &) {
      fn(i0, i1);
    }
  }
  fn((size_t)0, n / parts);
  for (auto &f : started) f.get();
}

inline void pad_box(Box &b) {
  // slack ≫ any rounding in the f64 slab test, ≪ any visible geometry
  for (int k = 0; k < 3; ++k) {
    const double ext = std::max(std::fabs(b.mn[k]), std::fabs(b.mx[k]));
    const double pad = 1e-9 * (1.0 + ext);
    b.mn[k] -= pad;
    b.mx[k] += pad;
  }
}

}  // namespace bvh_detail

// sph [n][4] cx cy cz r2, sph_r [n]; mov [n][8] c0 delta r2 r; tri [n][12] a e1 e2 n
inline void build_bvh(const std::vector<double> &sph, const std::vector<double> &sph_r,
                      const std::vector<double> &mov, const std::vector<double> &tri, HostBvh &out,
                      int leaf_max = 4, double c_trav = 0.0, double time0 = 0.0, double time1 = 1.0,
                      int parallel_min = 2048, int fail_thread_start = 0) {
  using namespace bvh_detail;
  const int ns = (int)sph_r.size(), nm = (int)(mov.size() / 8), nt = (int)(tri.size() / 12);
  const int n = ns + nm + nt;
  Builder B;
  B.out = &out;
  B.leaf_max = std::min(std::max(leaf_max, 1), 7);
  B.c_trav = c_trav;
  B.parallel_min = parallel_min;
  B.fail_thread_start = fail_thread_start;
  B.pb.resize(n);
  B.cen.resize((size_t)n * 3);
  for (int i = 0; i < ns; ++i) {
    Box &b = B.pb[i];
    const double r = std::fabs(sph_r[i]);
    for (int k = 0; k < 3; ++k) {
      b.mn[k] = sph[(size_t)i * 4 + k] - r;
      b.mx[k] = sph[(size_t)i * 4 + k] + r;
    }
  }
  for (int i = 0; i < nm; ++i) {
    // centre(time) = c0 + time*delta for time in the camera's shutter interval [time0, time1]
    // (ray times are drawn from it, src/common-model.cpp:165), widened a little
    Box &b = B.pb[ns + i];
    const double *m = &mov[(size_t)i * 8];
    const double r = std::fabs(m[7]);
    const double w = 1e-6 * (1.0 + std::fabs(time0) + std::fabs(time1));
    const double ta = std::min(time0, time1) - w, tb = std::max(time0, time1) + w;
    for (int k = 0; k < 3; ++k) {
      const double a0 = m[k] + ta * m[3 + k], a1 = m[k] + tb * m[3 + k];
      b.mn[k] = std::min(a0, a1) - r;
      b.mx[k] = std::max(a0, a1) + r;
    }
  }
  // (triangles, padding, centroids and the identity order in concurrent stretches: 3 of the 96.8k-triangle mesh's 9 ms)
  B.order.resize(n);
  for_stretches((size_t)nt, 8192, [&](size_t i0, size_t i1) {
    for (size_t i = i0; i < i1; ++i) {
      Box &b = B.pb[(size_t)ns + nm + i];
      const double *t = &tri[i * 12];
      for (int k = 0; k < 3; ++k) {
        const double a = t[k], bb = t[k] + t[3 + k], c = t[k] + t[6 + k];
        b.mn[k] = std::min(a, std::min(bb, c));
        b.mx[k] = std::max(a, std::max(bb, c));
      }
    }
  });
  for_stretches((size_t)n, 8192, [&](size_t i0, size_t i1) {
    for (size_t i = i0; i < i1; ++i) {
      pad_box(B.pb[i]);
      for (int k = 0; k < 3; ++k) B.cen[i * 3 + k] = 0.5 * (B.pb[i].mn[k] + B.pb[i].mx[k]);
      B.order[i] = (int32_t)i;
    }
  });
  // at most 2n - 1 nodes; the arrays are cut to the nodes used afterwards
  out.box.assign((size_t)std::max(2 * n, 1) * 6, 0.0);
  out.link.assign((size_t)std::max(2 * n, 1) * 4, 0);
  out.prim.clear();
  out.depth = 0;
  if (n > 0) B.build(0, 0, n, -1, 0, 1);
  const int used = std::max(2 * n - 1, 1);  // slots, not nodes: see Builder
  out.box.resize((size_t)used * 6);
  out.link.resize((size_t)used * 4);
  out.prim.assign(B.order.begin(), B.order.end());
  out.depth = B.max_depth.load();
}

// ---- scene image for the BVH kernel (layout: rtow_device.h, DevScene::blob) ----------
struct SceneImage {
  std::vector<unsigned char> blob;
  uint32_t off_ids = 0, off_sph = 0, off_mov = 0, off_tri = 0;
  uint32_t off_pmat = 0, off_mats = 0;  // material index per primitive, material records
  int32_t n_nodes = 0;
  size_t total_bytes = 0;
};

// Section offsets and everything but the node records and the id list: zeroed node section for
// n_nodes records + the END record, primitive records, material index per primitive, materials.
// (The device builder, rtow_build.hip, fills nodes and ids in HBM.)
inline void layout_scene_image(int n_nodes, size_t n_ids, const std::vector<double> &sph,
                               const std::vector<double> &mov, const std::vector<double> &tri,
                               const std::vector<int32_t> &prim_mat,
                               const std::vector<unsigned char> &mats_bytes, SceneImage &img,
                               bool offsets_only = false) {
  const size_t nodes_bytes = (size_t)(n_nodes + 1) * 32;  // + the sentinel END node
  const size_t ids_bytes = ((n_ids * 4 + 15) / 16) * 16;
  img.off_ids = (uint32_t)nodes_bytes;
  img.off_sph = (uint32_t)(nodes_bytes + ids_bytes);
  img.off_mov = img.off_sph + (uint32_t)(sph.size() * 8);
  img.off_tri = img.off_mov + (uint32_t)(mov.size() * 8);
  img.off_pmat = (uint32_t)((((size_t)img.off_tri + tri.size() * 8) + 15) / 16 * 16);
  img.off_mats = (uint32_t)((((size_t)img.off_pmat + prim_mat.size() * 4) + 15) / 16 * 16);
  const size_t total = (size_t)img.off_mats + mats_bytes.size();
  img.total_bytes = ((total + 15) / 16) * 16;
  img.n_nodes = n_nodes;
  if (offsets_only) return;  // the device builder assembles the image in HBM
  img.blob.assign(img.total_bytes, 0);
  if (!prim_mat.empty()) std::memcpy(img.blob.data() + img.off_pmat, prim_mat.data(), prim_mat.size() * 4);
  if (!mats_bytes.empty()) std::memcpy(img.blob.data() + img.off_mats, mats_bytes.data(), mats_bytes.size());
  if (!sph.empty()) std::memcpy(img.blob.data() + img.off_sph, sph.data(), sph.size() * 8);
  if (!mov.empty()) std::memcpy(img.blob.data() + img.off_mov, mov.data(), mov.size() * 8);
  if (!tri.empty()) std::memcpy(img.blob.data() + img.off_tri, tri.data(), tri.size() * 8);
  img.n_nodes = n_nodes;
}

inline void make_scene_image(const HostBvh &bvh, const std::vector<double> &sph,
                             const std::vector<double> &mov, const std::vector<double> &tri,
                             const double cam_origin[3], SceneImage &img,
                             const std::vector<int32_t> &prim_mat = {},
                             const std::vector<unsigned char> &mats_bytes = {}) {
  int n = (int)(bvh.link.size() / 4);
  // the f32 slab test sees the ray origin and the planes rounded to f32: pad every box by
  // more than that rounding can move a plane or an origin anywhere in the scene
  double scale = 1.0;
  for (int k = 0; k < 3; ++k) {
    scale = std::max(scale, std::fabs(bvh.box[k]));
    scale = std::max(scale, std::fabs(bvh.box[3 + k]));
    scale = std::max(scale, std::fabs(cam_origin[k]));
  }
  // depth-first renumbering, nearer child (to the camera) first
  std::vector<int> order;          // new index -> old index
  std::vector<uint32_t> skip_new;  // new index -> skip link
  order.reserve(n);
  struct Frame { int node; uint32_t skip_slot; };
  std::vector<int> stack;
  std::vector<int> subtree_end(n, 0);
  // iterative preorder
  stack.push_back(0);
  std::vector<int> newidx(n, -1);
  while (!stack.empty()) {
    const int nd = stack.back();
    stack.pop_back();
    newidx[nd] = (int)order.size();
    order.push_back(nd);
    if (bvh.link[(size_t)nd * 4 + 1] == 0) {
      const int l = bvh.link[(size_t)nd * 4 + 0], r = l + 1;
      auto dist2 = [&](int c) {
        double s = 0;
        for (int k = 0; k < 3; ++k) {
          const double m = 0.5 * (bvh.box[(size_t)c * 6 + k] + bvh.box[(size_t)c * 6 + 3 + k]);
          s += (m - cam_origin[k]) * (m - cam_origin[k]);
        }
        return s;
      };
      const bool left_first = dist2(l) <= dist2(r);
      // push the second child first so the first child is numbered next
      stack.push_back(left_first ? r : l);
      stack.push_back(left_first ? l : r);
    }
  }
  n = (int)order.size();  // nodes, from here on (the arrays hold unlinked slots too: Builder)
  // subtree sizes (postorder over the new numbering: children have larger indices)
  std::vector<int> size(n, 1);
  for (int i = n - 1; i >= 0; --i) {
    const int nd = order[i];
    if (bvh.link[(size_t)nd * 4 + 1] == 0) {
      const int l = bvh.link[(size_t)nd * 4 + 0];
      size[i] = 1 + size[newidx[l]] + size[newidx[l + 1]];
    }
  }
  layout_scene_image(n, bvh.prim.size(), sph, mov, tri, prim_mat, mats_bytes, img);
  for (int i = 0; i < n; ++i) {
    const int nd = order[i];
    float rec[6];
    for (int k = 0; k < 3; ++k) {
      const double lo = bvh.box[(size_t)nd * 6 + k], hi = bvh.box[(size_t)nd * 6 + 3 + k];
      const double pad = 2e-6 * scale + 2e-6 * std::max(std::fabs(lo), std::fabs(hi));
      rec[k] = std::nextafterf((float)(lo - pad), -INFINITY);
      rec[3 + k] = std::nextafterf((float)(hi + pad), INFINITY);
    }
    const uint32_t nxt = (uint32_t)(i + size[i]);
    const uint32_t skip = nxt >= (uint32_t)n ? (uint32_t)n : nxt;  // past the end = sentinel
    const int cnt = bvh.link[(size_t)nd * 4 + 1];
    const uint32_t leaf = cnt > 0 ? ((uint32_t)bvh.link[(size_t)nd * 4 + 0] << 3) | (uint32_t)cnt : 0u;
    unsigned char *dst = img.blob.data() + (size_t)i * 32;
    std::memcpy(dst, rec, 24);
    std::memcpy(dst + 24, &skip, 4);
    std::memcpy(dst + 28, &leaf, 4);
  }
  {  // sentinel END node: an empty box (never hit) whose skip link is itself, no leaf
    const float inf = INFINITY;
    const float rec[6] = {inf, inf, inf, -inf, -inf, -inf};
    const uint32_t self = (uint32_t)n, leaf = 0u;
    unsigned char *dst = img.blob.data() + (size_t)n * 32;
    std::memcpy(dst, rec, 24);
    std::memcpy(dst + 24, &self, 4);
    std::memcpy(dst + 28, &leaf, 4);
  }
  if (!bvh.prim.empty()) std::memcpy(img.blob.data() + img.off_ids, bvh.prim.data(), bvh.prim.size() * 4);
}

// The kernel's termination argument rests on these: every skip link points forward and
// at most to the end marker, leaves reference in-range primitive ids, ids are in range.
inline bool validate_scene_image(const SceneImage &img, int n_prims) {
  const uint32_t n = (uint32_t)img.n_nodes;
  if (n == 0 || img.blob.size() < (size_t)(n + 1) * 32) return false;
  for (uint32_t i = 0; i <= n; ++i) {
    uint32_t skip, leaf;
    std::memcpy(&skip, img.blob.data() + (size_t)i * 32 + 24, 4);
    std::memcpy(&leaf, img.blob.data() + (size_t)i * 32 + 28, 4);
    if (i == n) return skip == n && leaf == 0u;  // the END record
    if (!(skip > i && skip <= n)) return false;
    if (leaf != 0u) {
      const uint32_t first = leaf >> 3, count = leaf & 7u;
      if (count == 0u || first + count > (uint32_t)n_prims) return false;
      for (uint32_t k = 0; k < count; ++k) {
        int32_t id;
        std::memcpy(&id, img.blob.data() + img.off_ids + 4 * (size_t)(first + k), 4);
        if (id < 0 || id >= n_prims) return false;
      }
    } else if (i + 1 >= n) {
      return false;  // an inner node needs a child at i+1
    }
  }
  return true;
}

}  // namespace rtow
