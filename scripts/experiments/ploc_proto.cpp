// ploc_proto.cpp — host prototype behind round 5's device builder (csrc/rtow_build.hip, PLOC pass): tree quality of
//   host   : the product's host SAH builder (csrc/rtow_bvh.h, pairs of triangles per leaf)
//   radix  : Karras' radix tree over 63-bit Morton keys (the device builder of rounds 1-4)
//   sahz   : SAH splits over the Morton-sorted sequence (round 5, first attempt: measured worse on the big mesh)
//   ploc R : parallel locally-ordered clustering (Meister & Bittner 2018) with search radius R, iteration-parallel
//            semantics exactly as the GPU kernels have them (nearest neighbour in the current cluster array, mutual
//            pairs merge, array compacted in order)
// Every binary tree is collapsed 4-wide with the product's greedy rule (largest-area inner child expanded first,
// subtrees of <= 2 triangles are leaves) and scored with the surface-area metric of the 4-wide tree:
//   nodes = sum over 4-wide nodes of area(node) / area(root)      (expected node visits of a random line)
//   tris  = sum over leaves of area(leaf) / area(root) * count    (expected triangle tests)
// build: g++ -O2 -std=c++20 -pthread scripts/experiments/ploc_proto.cpp -o /tmp/ploc_proto ; run: /tmp/ploc_proto mesh.obj
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <numeric>
#include <sstream>
#include <string>
#include <vector>

#include "../../raytracing-one-weekend_amd/csrc/rtow_bvh.h"

struct B {
  float mn[3], mx[3];
  void grow(const B &o) {
    for (int k = 0; k < 3; ++k) mn[k] = std::min(mn[k], o.mn[k]), mx[k] = std::max(mx[k], o.mx[k]);
  }
  float area() const {
    const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    return dx * dy + dy * dz + dz * dx;
  }
};
static B uni(B a, const B &b) {
  a.grow(b);
  return a;
}

// binary tree over `n` leaves in a leaf order: node i < n - 1 inner; child >= 0 inner node, < 0: ~leaf position
struct Tree {
  std::vector<int> l, r, cnt;
  std::vector<B> box;
  int root = 0;
};

static std::vector<double> load_tris(const char *path) {
  std::ifstream in(path);
  std::vector<double> v, tri;
  std::string line;
  while (std::getline(in, line)) {
    std::istringstream ss(line);
    std::string tag;
    ss >> tag;
    if (tag == "v") {
      double x, y, z;
      ss >> x >> y >> z;
      v.insert(v.end(), {x, y, z});
    } else if (tag == "f") {
      long idx[3];
      for (int k = 0; k < 3; ++k) {
        std::string t;
        ss >> t;
        idx[k] = std::stol(t) - 1;
      }
      const double *a = &v[idx[0] * 3], *b = &v[idx[1] * 3], *c = &v[idx[2] * 3];
      const double e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
      tri.insert(tri.end(), {a[0], a[1], a[2], e1[0], e1[1], e1[2], e2[0], e2[1], e2[2], 0, 0, 0});
    }
  }
  return tri;
}

struct Score {
  double nodes = 0, tris = 0;
  int n4 = 0, depth = 0;
};
// greedy 4-wide collapse + surface-area score; `leafbox(pos)` box of leaf position, tree leaves are positions
static Score score(const Tree &t, const std::vector<B> &leaf, int n) {
  Score s;
  if (n <= 2) return s;
  auto cnt = [&](int ref) { return ref < 0 ? 1 : t.cnt[ref]; };
  auto box = [&](int ref) -> const B & { return ref < 0 ? leaf[~ref] : t.box[ref]; };
  auto inner = [&](int ref) { return ref >= 0 && t.cnt[ref] > 2; };
  const double ar = t.box[t.root].area();
  std::vector<std::pair<int, int>> level{{t.root, 1}};
  std::vector<std::pair<int, int>> stack{{t.root, 1}};
  while (!stack.empty()) {
    auto [me, d] = stack.back();
    stack.pop_back();
    ++s.n4;
    s.depth = std::max(s.depth, d);
    s.nodes += t.box[me].area() / ar;
    int ch[4] = {t.l[me], t.r[me], 0, 0}, nc = 2;
    while (nc < 4) {
      int pick = -1;
      float best = -1;
      for (int c = 0; c < nc; ++c)
        if (inner(ch[c]) && box(ch[c]).area() > best) best = box(ch[c]).area(), pick = c;
      if (pick < 0) break;
      const int x = ch[pick];
      ch[pick] = t.l[x];
      ch[nc++] = t.r[x];
    }
    for (int c = 0; c < nc; ++c) {
      if (inner(ch[c]))
        stack.push_back({ch[c], d + 1});
      else
        s.tris += box(ch[c]).area() / ar * cnt(ch[c]);
    }
  }
  return s;
}


// Tree rotations (Kensler 2008): for an inner node with children L, R, swap a child with a grandchild on the other
// side when that lowers the surface area of the child that is rebuilt.  Passes over all nodes until nothing improves
// (or `max_pass`).  Boxes and counts are kept up to date locally; ancestors' boxes do not change (same leaf sets).
static int rotate_pass(Tree &t, const std::vector<B> &leaf, const std::vector<int> &order) {
  auto box = [&](int ref) -> B { return ref < 0 ? leaf[~ref] : t.box[ref]; };
  auto cnt = [&](int ref) { return ref < 0 ? 1 : t.cnt[ref]; };
  int done = 0;
  for (int id : order) {
    int &L = t.l[id], &R = t.r[id];
    // candidates: swap R with a grandchild under L (L is rebuilt), or L with a grandchild under R
    float best = 0.f;
    int which = -1;
    for (int side = 0; side < 2; ++side) {
      const int child = side == 0 ? L : R, other = side == 0 ? R : L;
      if (child < 0) continue;
      const float a0 = t.box[child].area();
      const int g0 = t.l[child], g1 = t.r[child];
      const float a_swap0 = uni(box(other), box(g1)).area();  // other replaces g0
      const float a_swap1 = uni(box(g0), box(other)).area();  // other replaces g1
      if (a0 - a_swap0 > best) best = a0 - a_swap0, which = side * 2;
      if (a0 - a_swap1 > best) best = a0 - a_swap1, which = side * 2 + 1;
    }
    if (which < 0) continue;
    const int side = which >> 1, gi = which & 1;
    int &child = side == 0 ? L : R;
    int &other = side == 0 ? R : L;
    int &g = gi == 0 ? t.l[child] : t.r[child];
    std::swap(other, g);
    t.box[child] = uni(box(t.l[child]), box(t.r[child]));
    t.cnt[child] = cnt(t.l[child]) + cnt(t.r[child]);
    ++done;
  }
  return done;
}

static uint64_t spread21(uint64_t v) {
  v &= 0x1fffffull;
  v = (v | (v << 32)) & 0x1f00000000ffffull;
  v = (v | (v << 16)) & 0x1f0000ff0000ffull;
  v = (v | (v << 8)) & 0x100f00f00f00f00full;
  v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  std::vector<double> tri = load_tris(argv[1]);
  const int n = (int)(tri.size() / 12);
  std::printf("%d triangles\n", n);
  std::vector<B> pb(n);
  B cb;
  for (int k = 0; k < 3; ++k) cb.mn[k] = INFINITY, cb.mx[k] = -INFINITY;
  std::vector<double> cen((size_t)n * 3);
  for (int i = 0; i < n; ++i) {
    const double *t = &tri[(size_t)i * 12];
    for (int k = 0; k < 3; ++k) {
      const double a = t[k], b = t[k] + t[3 + k], c = t[k] + t[6 + k];
      pb[i].mn[k] = (float)std::min({a, b, c});
      pb[i].mx[k] = (float)std::max({a, b, c});
      cen[(size_t)i * 3 + k] = 0.5 * ((double)pb[i].mn[k] + pb[i].mx[k]);
      cb.mn[k] = std::min(cb.mn[k], (float)cen[(size_t)i * 3 + k]);
      cb.mx[k] = std::max(cb.mx[k], (float)cen[(size_t)i * 3 + k]);
    }
  }
  // ---- host SAH
  {
    std::vector<double> none;
    rtow::HostBvh bvh;
    rtow::build_bvh(none, none, none, tri, bvh, 2, 1.5, 0.0, 1.0);
    // convert: leaves of the host tree (<= 2 prims) become pairs of leaf positions in its prim order
    const int n2 = (int)(bvh.link.size() / 4);
    std::vector<B> leaf(n);
    for (int p = 0; p < n; ++p) leaf[p] = pb[bvh.prim[p]];
    Tree t;
    t.l.assign(2 * n, 0), t.r.assign(2 * n, 0), t.cnt.assign(2 * n, 0), t.box.resize(2 * n);
    int next = 0;
    std::function<int(int)> conv = [&](int nd) -> int {  // returns ref
      const int a = bvh.link[(size_t)nd * 4], c = bvh.link[(size_t)nd * 4 + 1];
      if (c > 0) {  // leaf: first a, count c
        if (c == 1) return ~a;
        int ref = ~a;
        for (int k = 1; k < c; ++k) {
          const int id = next++;
          t.l[id] = ref, t.r[id] = ~(a + k);
          t.cnt[id] = k + 1;
          B b = ref < 0 ? leaf[~ref] : t.box[ref];
          b.grow(leaf[a + k]);
          t.box[id] = b;
          ref = id;
        }
        return ref;
      }
      const int id = next++;
      const int L = conv(a), R = conv(a + 1);
      t.l[id] = L, t.r[id] = R;
      t.cnt[id] = (L < 0 ? 1 : t.cnt[L]) + (R < 0 ? 1 : t.cnt[R]);
      t.box[id] = uni(L < 0 ? leaf[~L] : t.box[L], R < 0 ? leaf[~R] : t.box[R]);
      return id;
    };
    t.root = conv(0);
    (void)n2;
    const Score s = score(t, leaf, n);
    std::printf("host    : 4-wide nodes %6d depth %2d  metric nodes %.3f tris %.3f\n", s.n4, s.depth, s.nodes, s.tris);
  }
  // ---- Morton order
  std::vector<uint64_t> key(n);
  std::vector<int> ord(n);
  for (int i = 0; i < n; ++i) {
    uint64_t k64 = 0;
    for (int k = 0; k < 3; ++k) {
      const double ext = (double)cb.mx[k] - cb.mn[k];
      double u = ext > 0 ? (cen[(size_t)i * 3 + k] - cb.mn[k]) / ext : 0.0;
      u = std::min(std::max(u, 0.0), 1.0);
      k64 |= spread21((uint64_t)std::min(u * 2097152.0, 2097151.0)) << (2 - k);
    }
    key[i] = k64;
    ord[i] = i;
  }
  std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return key[a] < key[b]; });
  std::vector<B> leaf(n);
  std::vector<uint64_t> sk(n);
  for (int j = 0; j < n; ++j) leaf[j] = pb[ord[j]], sk[j] = key[ord[j]];
  auto finish = [&](Tree &t) {  // boxes + counts bottom-up
    std::function<void(int)> rec = [&](int id) {
      int c = 0;
      B b;
      bool have = false;
      for (int ref : {t.l[id], t.r[id]}) {
        if (ref >= 0) rec(ref);
        const B &cbx = ref < 0 ? leaf[~ref] : t.box[ref];
        c += ref < 0 ? 1 : t.cnt[ref];
        if (!have) b = cbx, have = true; else b.grow(cbx);
      }
      t.box[id] = b;
      t.cnt[id] = c;
    };
    rec(t.root);
  };
  // ---- radix tree (top-down by highest differing bit; equal keys: middle)
  {
    Tree t;
    t.l.assign(n, 0), t.r.assign(n, 0), t.cnt.assign(n, 0), t.box.resize(n);
    int next = 0;
    std::function<int(int, int)> build = [&](int lo, int hi) -> int {
      if (lo == hi) return ~lo;
      const int id = next++;
      int split;
      if (sk[lo] == sk[hi]) {
        split = (lo + hi) / 2;
      } else {
        const int pre = __builtin_clzll(sk[lo] ^ sk[hi]);  // first differing bit: 0 on the left part, 1 on the right
        const uint64_t bit = 1ull << (63 - pre);
        int a = lo, b = hi;  // sk[a] has the bit clear, sk[b] has it set
        while (a + 1 < b) {
          const int m = (a + b) / 2;
          if (sk[m] & bit) b = m; else a = m;
        }
        split = a;
      }
      const int L = build(lo, split), R = build(split + 1, hi);
      t.l[id] = L, t.r[id] = R;
      return id;
    };
    t.root = build(0, n - 1);
    finish(t);
    const Score s = score(t, leaf, n);
    std::printf("radix   : 4-wide nodes %6d depth %2d  metric nodes %.3f tris %.3f\n", s.n4, s.depth, s.nodes, s.tris);
  }
  // ---- PLOC, optionally stopped at `stop` clusters and finished by an exact sweep-SAH over the clusters
  for (int stop : {1, 2048, 8192, 32768, 1000000})
  for (int R : {8, 16}) {
    Tree t;
    t.l.assign(n, 0), t.r.assign(n, 0), t.cnt.assign(n, 0), t.box.resize(n);
    std::vector<int> ref(n);  // cluster array: refs (leaf ~pos or inner id)
    std::vector<B> cbx(n);
    std::vector<int> ccnt(n, 1);
    for (int j = 0; j < n; ++j) ref[j] = ~j, cbx[j] = leaf[j];
    int m = n, next = 0, iters = 0;
    std::vector<int> nn(n), nref(n), ncnt(n);
    std::vector<B> nbx(n);
    while (m > std::max(stop, 1)) {
      ++iters;
      for (int i = 0; i < m; ++i) {
        float best = INFINITY;
        int bj = -1;
        for (int j = std::max(0, i - R); j <= std::min(m - 1, i + R); ++j) {
          if (j == i) continue;
          const float a = uni(cbx[i], cbx[j]).area();
          if (a < best) best = a, bj = j;  // ties: the lower index
        }
        nn[i] = bj;
      }
      int w = 0;
      for (int i = 0; i < m; ++i) {
        const int j = nn[i];
        if (nn[j] == i) {
          if (i < j) {
            const int id = next++;
            t.l[id] = ref[i], t.r[id] = ref[j];
            nref[w] = id;
            nbx[w] = uni(cbx[i], cbx[j]);
            ncnt[w] = ccnt[i] + ccnt[j];
            ++w;
          }  // the higher one of the pair disappears
        } else {
          nref[w] = ref[i];
          nbx[w] = cbx[i];
          ncnt[w] = ccnt[i];
          ++w;
        }
      }
      m = w;
      std::swap(ref, nref);
      std::swap(cbx, nbx);
      std::swap(ccnt, ncnt);
    }
    if (m > 1) {  // exact sweep SAH over the m clusters (weights = triangle counts)
      std::vector<int> idx(m);
      std::iota(idx.begin(), idx.end(), 0);
      std::function<int(int, int)> top = [&](int lo, int hi) -> int {  // idx[lo, hi)
        if (hi - lo == 1) return ref[idx[lo]];
        float bestc = INFINITY;
        int bax = 0, bsp = lo + (hi - lo) / 2;
        for (int ax = 0; ax < 3; ++ax) {
          std::sort(idx.begin() + lo, idx.begin() + hi, [&](int a, int b) {
            return cbx[a].mn[ax] + cbx[a].mx[ax] < cbx[b].mn[ax] + cbx[b].mx[ax];
          });
          std::vector<float> ra(hi - lo);
          std::vector<int> rc(hi - lo);
          B acc = cbx[idx[hi - 1]];
          int c = 0;
          for (int i = hi - 1; i > lo; --i) {
            if (i < hi - 1) acc.grow(cbx[idx[i]]);
            c += ccnt[idx[i]];
            ra[i - lo] = acc.area();
            rc[i - lo] = c;
          }
          acc = cbx[idx[lo]];
          c = 0;
          for (int i = lo; i < hi - 1; ++i) {
            if (i > lo) acc.grow(cbx[idx[i]]);
            c += ccnt[idx[i]];
            const float cost = acc.area() * c + ra[i + 1 - lo] * rc[i + 1 - lo];
            if (cost < bestc) bestc = cost, bax = ax, bsp = i + 1;
          }
        }
        std::sort(idx.begin() + lo, idx.begin() + hi, [&](int a, int b) {
          return cbx[a].mn[bax] + cbx[a].mx[bax] < cbx[b].mn[bax] + cbx[b].mx[bax];
        });
        const int id = next++;
        const int L = top(lo, bsp), Rr = top(bsp, hi);
        t.l[id] = L, t.r[id] = Rr;
        return id;
      };
      t.root = top(0, m);
    } else {
      t.root = ref[0];
    }
    finish(t);
    if (stop == 1) {
      // top-down order of the inner nodes
      std::vector<int> order;
      std::vector<int> st2{t.root};
      while (!st2.empty()) {
        const int id = st2.back();
        st2.pop_back();
        order.push_back(id);
        if (t.l[id] >= 0) st2.push_back(t.l[id]);
        if (t.r[id] >= 0) st2.push_back(t.r[id]);
      }
      const Score s0 = score(t, leaf, n);
      int passes = 0, total = 0;
      for (; passes < 16; ++passes) {
        const int k = rotate_pass(t, leaf, order);
        total += k;
        if (k == 0) break;
        order.clear();
        st2.assign(1, t.root);
        while (!st2.empty()) {
          const int id = st2.back();
          st2.pop_back();
          order.push_back(id);
          if (t.l[id] >= 0) st2.push_back(t.l[id]);
          if (t.r[id] >= 0) st2.push_back(t.r[id]);
        }
      }
      finish(t);
      const Score s1 = score(t, leaf, n);
      std::printf("ploc %2d + rotations (%d passes, %d rotations): nodes %.3f -> %.3f tris %.3f -> %.3f, 4-wide nodes %d\n", R, passes, total,
                  s0.nodes, s1.nodes, s0.tris, s1.tris, s1.n4);
    }
    const Score s = score(t, leaf, n);
    std::printf("ploc %2d stop %5d: 4-wide nodes %6d depth %2d  metric nodes %.3f tris %.3f  (%d iterations)\n", R, stop, s.n4, s.depth,
                s.nodes, s.tris, iters);
  }
  return 0;
}
