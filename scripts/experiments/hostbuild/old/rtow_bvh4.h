// rtow_bvh4.h — host build of the 4-wide BVH image the BVH4 kernel walks (triangle meshes).
//
// Like rtow_bvh.h this is NOT the reference's tree (src/render.cpp:73-110): closest hits do not
// depend on the tree (exact ties aside), so the device uses the one that suits it.  The 4-wide tree
// is the SAH BVH2 of rtow_bvh.h collapsed: an inner node adopts the children of its child with the
// largest surface area until it has four children (or only leaves are left).
//
// Image ("blob4"), every section 16-byte aligned, nodes first and in BREADTH-FIRST order so that
// any 128-byte-aligned prefix of the image is the top of the tree (the kernel stages as much of the
// image as fits in LDS: all of it for a small mesh, the top levels for a big one):
//
//   [nodes n4 x 128 B or 64 B][triangle records nt x 96 B, LEAF ORDER][material index per record][materials]
//
//   node (128 B): lo.x[4] hi.x[4] lo.y[4] hi.y[4] lo.z[4] hi.z[4]  (f32, padded conservatively like
//                 the BVH2 image: rtow_bvh.h make_scene_image)      offsets 0 16 32 48 64 80
//                 child[4] (u32)                                    offset 96
//                 16 B unused                                       offset 112
//     A lane reads the NEAR planes of all four children with one 16-byte load at  +axis*32 + (d<0 ? 16 : 0)
//     and the FAR planes at that address ^ 16: the slab test needs no min/max per plane.
//   half node (64 B; `half`, for a mesh whose image does not fit LDS whole — its nodes come from L2, where the
//   address path of the vector memory unit is what the walk waits for: 4 loads per node instead of 7):
//                 lo.x[4] hi.x[4] lo.y[4] hi.y[4] lo.z[4] hi.z[4]  (binary16)  offsets 0 8 16 24 32 40
//                 child[4] (u32)                                                offset 48
//     Planes are stored in the mesh's own frame, plane' = (plane - map_c) * map_s with the root box at +-1000, padded
//     like the f32 planes and then rounded OUTWARDS to binary16: a plane moves by at most 1/4000 of the mesh's extent.
//     The kernel feeds them to v_fma_mix_f32 as they are (binary16 operand, binary32 arithmetic: no conversion
//     instruction), with the ray's (o - map_c) / d and 1 / (d map_s) — t is the same number in both frames.
//   child word ("ref21", also what the traversal stack holds in its low 21 bits):
//     inner  : node index            (bit 20 = 0; byte offset = index << 7)
//     leaf   : 1<<20 | first << 2 | (count-1), records [first, first+count), count 1..4, first < 2^18 - 4
//     empty  : kRefNone (box inverted: +inf / -inf, never hit)
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "rtow_bvh.h"

namespace rtow {

constexpr uint32_t kBvh4NodeBytes = 128;      // binary32 planes (meshes whose whole image is staged in LDS)
constexpr uint32_t kBvh4HalfNodeBytes = 64;   // binary16 planes (bigger meshes: nodes are read from L2)
constexpr uint32_t kRefNone = 0x1fffffu;
constexpr uint32_t kRefLeaf = 1u << 20;
constexpr uint32_t kBvh4MaxNodes = 1u << 20;
constexpr uint32_t kBvh4MaxTris = (1u << 18) - 4u;

// binary16 with directed rounding (the planes of the half-precision node must not move inwards)
inline double half_value(uint16_t h) {
  const int e = (h >> 10) & 31, m = h & 1023;
  double v = e == 0 ? std::ldexp((double)m, -24) : (e == 31 ? (m ? NAN : INFINITY) : std::ldexp((double)(m | 1024), e - 25));
  return (h & 0x8000) ? -v : v;
}
// the largest binary16 <= x (dir < 0) or the smallest >= x (dir > 0).  Integer arithmetic on the binary64 pattern
// (the frexp / ldexp / ceil form this replaces was 19 ms of the 96.8k-triangle mesh's image: 646,000 planes).
inline uint16_t half_directed(double x, int dir) {
  if (std::isnan(x)) return 0x7e00;
  const bool neg = std::signbit(x);
  const double a = std::fabs(x);
  const bool mag_up = (dir > 0) != neg;  // away from zero?
  const uint16_t sign = neg ? 0x8000 : 0;
  if (a == 0.0) return sign;
  if (std::isinf(a)) return sign | 0x7c00;
  if (a > 65504.0) return sign | (mag_up ? 0x7c00 : 0x7bff);
  uint64_t b;
  std::memcpy(&b, &a, sizeof b);
  const int be = (int)(b >> 52);  // biased exponent; 0 = a binary64 denormal (< 2^-1022: far below binary16's 2^-24)
  if (be == 0) return sign | (uint16_t)(mag_up ? 1 : 0);
  const int e = be - 1023;                                       // a = m * 2^(e - 52), m in [2^52, 2^53)
  const uint64_t m = (b & ((1ull << 52) - 1ull)) | (1ull << 52);
  const int E = std::max(e, -14);  // denormals share the exponent of the smallest normal
  const int sh = 42 + (E - e);     // a in units of its binary16 ulp: q = a * 2^(10 - E) = m >> sh, exactly
  uint64_t q = sh >= 64 ? 0ull : (m >> sh);
  const bool inexact = sh >= 64 ? true : (m & ((1ull << sh) - 1ull)) != 0ull;
  const uint32_t qi = (uint32_t)q + ((mag_up && inexact) ? 1u : 0u);  // 0 .. 2048
  // (a carry, qi = 2048, lands in the next exponent by plain addition; so does 1024 from the denormals)
  const uint32_t bits = (E == -14 && qi < 1024u) ? qi : (((uint32_t)(E + 15) << 10) + (qi - 1024u));
  return sign | (uint16_t)std::min<uint32_t>(bits, 0x7c00u);
}

struct Bvh4Image {
  bool half = false;                                   // 64-byte nodes with binary16 planes
  double map_c[3] = {0, 0, 0}, map_s[3] = {1, 1, 1};  // half: plane' = (plane - map_c) * map_s
  uint32_t node_bytes() const { return half ? kBvh4HalfNodeBytes : kBvh4NodeBytes; }
  uint32_t child_off() const { return half ? 48u : 96u; }  // child[4] inside a node
  std::vector<unsigned char> blob;
  uint32_t off_tri = 0, off_pmat = 0, off_mats = 0;
  int32_t n_nodes = 0;
  int32_t depth = 0;       // levels of inner nodes (root = 1): bounds the traversal stack at 3*depth
  bool ok = false;
};

// `bvh`: SAH BVH2 over triangles only with leaves of <= 4 primitives; `tri` [n][12] (a e1 e2 n) and
// `pmat` in INSERTION order (permuted into leaf order here).
inline void make_bvh4_image(const HostBvh &bvh, const std::vector<double> &tri, const std::vector<int32_t> &pmat,
                            const std::vector<unsigned char> &mats_bytes, const double cam_origin[3], Bvh4Image &img,
                            bool half = false) {
  img.ok = false;
  img.half = half;
  const uint32_t node_bytes = img.node_bytes(), child_off = img.child_off();
  const size_t nt = tri.size() / 12;
  const int n2 = (int)(bvh.link.size() / 4);
  if (nt == 0 || n2 == 0 || nt > kBvh4MaxTris) return;
  auto is_leaf2 = [&](int nd) { return bvh.link[(size_t)nd * 4 + 1] > 0; };
  auto area2 = [&](int nd) {
    const double *b = &bvh.box[(size_t)nd * 6];
    const double dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return dx * dy + dy * dz + dz * dx;
  };
  for (int nd = 0; nd < n2; ++nd)
    if (is_leaf2(nd) && bvh.link[(size_t)nd * 4 + 1] > 4) return;  // leaf too big for the 2-bit count

  // pad rule of the BVH2 image (the f32 slab test sees origin and planes rounded to f32)
  double scale = 1.0;
  for (int k = 0; k < 3; ++k) {
    scale = std::max(scale, std::fabs(bvh.box[k]));
    scale = std::max(scale, std::fabs(bvh.box[3 + k]));
    scale = std::max(scale, std::fabs(cam_origin[k]));
  }

  // the mesh's own frame: the root box maps to [-1000, 1000] per axis (padded planes stay below 1024, where the
  // binary16 spacing doubles), so that a plane is at most 1/4000 of the mesh's extent away from where it should be
  // (1/12000 on average) and nothing is near a denormal
  for (int k = 0; half && k < 3; ++k) {
    const double lo = bvh.box[k], hi = bvh.box[3 + k];
    const double widest = std::max({bvh.box[3] - bvh.box[0], bvh.box[4] - bvh.box[1], bvh.box[5] - bvh.box[2]});
    const double half_k = std::max({0.5 * (hi - lo), 1e-4 * widest, 1e-30});  // (a flat mesh has no extent in one axis)
    img.map_c[k] = 0.5 * (lo + hi);
    img.map_s[k] = 1000.0 / half_k;
  }

  struct N4 { int child[4]; int n; };  // BVH2 node ids of the children
  std::vector<N4> nodes;               // breadth-first
  std::vector<int> src;                // BVH2 node each BVH4 node expands
  std::vector<int> level;
  src.push_back(0);
  level.push_back(1);
  int depth = 1;
  // a single-leaf tree: one node whose only child is that leaf
  for (size_t i = 0; i < src.size(); ++i) {
    N4 n4;
    n4.n = 0;
    const int root2 = src[i];
    if (is_leaf2(root2)) {
      n4.child[n4.n++] = root2;
    } else {
      const int l = bvh.link[(size_t)root2 * 4 + 0];
      n4.child[n4.n++] = l;
      n4.child[n4.n++] = l + 1;
      while (n4.n < 4) {
        int pick = -1;
        double best = -1.0;
        for (int c = 0; c < n4.n; ++c)
          if (!is_leaf2(n4.child[c]) && area2(n4.child[c]) > best) {
            best = area2(n4.child[c]);
            pick = c;
          }
        if (pick < 0) break;
        const int l2 = bvh.link[(size_t)n4.child[pick] * 4 + 0];
        n4.child[pick] = l2;
        n4.child[n4.n++] = l2 + 1;
      }
    }
    nodes.push_back(n4);
    for (int c = 0; c < n4.n; ++c)
      if (!is_leaf2(n4.child[c])) {
        src.push_back(n4.child[c]);
        level.push_back(level[i] + 1);
        depth = std::max(depth, level[i] + 1);
      }
    if (src.size() > kBvh4MaxNodes) return;
  }
  const size_t n4count = nodes.size();

  auto up16 = [](size_t v) { return (v + 15) / 16 * 16; };
  const size_t nodes_bytes = n4count * node_bytes;
  img.off_tri = (uint32_t)nodes_bytes;
  img.off_pmat = (uint32_t)up16(img.off_tri + nt * 96);
  img.off_mats = (uint32_t)up16(img.off_pmat + nt * 4);
  img.blob.assign(up16(img.off_mats + mats_bytes.size()), 0);
  img.n_nodes = (int32_t)n4count;
  img.depth = depth;
  unsigned char *B = img.blob.data();
  if (!mats_bytes.empty()) std::memcpy(B + img.off_mats, mats_bytes.data(), mats_bytes.size());

  // second pass in the same breadth-first order: inner children get consecutive node indices in
  // the order they were appended to `src`; leaves copy their records in leaf order
  uint32_t next_inner = 1, next_rec = 0;
  const float inf = INFINITY;
  for (size_t i = 0; i < n4count; ++i) {
    const N4 &n4 = nodes[i];
    float *f = reinterpret_cast<float *>(B + i * node_bytes);
    uint16_t *h = reinterpret_cast<uint16_t *>(B + i * node_bytes);
    uint32_t *cw = reinterpret_cast<uint32_t *>(B + i * node_bytes + child_off);
    for (int c = 0; c < 4; ++c) {
      if (c >= n4.n) {  // empty slot
        for (int k = 0; k < 3; ++k) {
          if (half) {
            h[k * 8 + c] = 0x7c00;      // +inf
            h[k * 8 + 4 + c] = 0xfc00;  // -inf
          } else {
            f[k * 8 + c] = inf;
            f[k * 8 + 4 + c] = -inf;
          }
        }
        cw[c] = kRefNone;
        continue;
      }
      const int nd = n4.child[c];
      for (int k = 0; k < 3; ++k) {
        const double lo = bvh.box[(size_t)nd * 6 + k], hi = bvh.box[(size_t)nd * 6 + 3 + k];
        const double pad = 2e-6 * scale + 2e-6 * std::max(std::fabs(lo), std::fabs(hi));
        if (half) {
          // the same pad (it covers the f32 rounding of the ray's side of the test: |c| <= scale), then outwards
          h[k * 8 + c] = half_directed((lo - pad - img.map_c[k]) * img.map_s[k], -1);
          h[k * 8 + 4 + c] = half_directed((hi + pad - img.map_c[k]) * img.map_s[k], +1);
        } else {
          f[k * 8 + c] = std::nextafterf((float)(lo - pad), -INFINITY);
          f[k * 8 + 4 + c] = std::nextafterf((float)(hi + pad), INFINITY);
        }
      }
      if (is_leaf2(nd)) {
        const int first = bvh.link[(size_t)nd * 4 + 0], cnt = bvh.link[(size_t)nd * 4 + 1];
        cw[c] = kRefLeaf | (next_rec << 2) | (uint32_t)(cnt - 1);
        for (int k = 0; k < cnt; ++k) {
          const int p = bvh.prim[(size_t)first + k];
          std::memcpy(B + img.off_tri + (size_t)(next_rec + k) * 96, &tri[(size_t)p * 12], 96);
          std::memcpy(B + img.off_pmat + (size_t)(next_rec + k) * 4, &pmat[p], 4);
        }
        next_rec += (uint32_t)cnt;
      } else {
        cw[c] = next_inner++;
      }
    }
  }
  img.ok = next_inner == n4count && next_rec == nt;
}

// What the kernel's termination and addressing rest on: child links point to LATER nodes (breadth-first
// order, so the tree is acyclic), leaves stay inside the record section and every record is referenced
// exactly once.
inline bool validate_bvh4_image(const Bvh4Image &img, size_t n_tri) {
  if (!img.ok || img.n_nodes <= 0) return false;
  std::vector<unsigned char> seen(n_tri, 0);
  std::vector<int> level((size_t)img.n_nodes, 0);
  level[0] = 1;
  int depth = 1;
  for (int i = 0; i < img.n_nodes; ++i) {
    const uint32_t *cw = reinterpret_cast<const uint32_t *>(img.blob.data() + (size_t)i * img.node_bytes() + img.child_off());
    if (level[i] == 0) return false;  // unreachable node
    for (int c = 0; c < 4; ++c) {
      const uint32_t r = cw[c];
      if (r == kRefNone) continue;
      if (r & ~0x1fffffu) return false;
      if (r & kRefLeaf) {
        const uint32_t first = (r & (kRefLeaf - 1u)) >> 2, cnt = (r & 3u) + 1u;
        if ((size_t)first + cnt > n_tri) return false;
        for (uint32_t k = 0; k < cnt; ++k) {
          if (seen[first + k]) return false;
          seen[first + k] = 1;
        }
      } else {
        if ((int)r <= i || (int)r >= img.n_nodes || level[r] != 0) return false;
        level[r] = level[i] + 1;
        depth = std::max(depth, level[r]);
      }
    }
  }
  for (size_t k = 0; k < n_tri; ++k)
    if (!seen[k]) return false;
  return depth == img.depth;
}

}  // namespace rtow
