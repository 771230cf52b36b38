// Host-side acceleration builders under AddressSanitizer / UBSan (CPU only; tests/test_host_builders.py).
// The builders are header-only (csrc/rtow_bvh.h, rtow_bvh4.h, rtow_grid.h), so this harness compiles them
// without HIP: SAH BVH2 -> threaded image, 4-wide image (triangle meshes), uniform grid with fat lists
// (static and moving spheres).  Exit code 0 = every image built and validated.
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <fstream>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "../../raytracing-one-weekend_amd/csrc/rtow_bvh.h"
#include "../../raytracing-one-weekend_amd/csrc/rtow_bvh4.h"
#include "../../raytracing-one-weekend_amd/csrc/rtow_grid.h"

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
      tri.insert(tri.end(), {a[0], a[1], a[2], e1[0], e1[1], e1[2], e2[0], e2[1], e2[2],
                             e1[1] * e2[2] - e2[1] * e1[2], e1[2] * e2[0] - e2[2] * e1[0], e1[0] * e2[1] - e2[0] * e1[1]});
    }
  }
  return tri;
}

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  const double cam[3] = {1, 0, -1};
  std::vector<unsigned char> mats(48, 0);
  // ---- triangle mesh: BVH2 image and 4-wide image, also with duplicated (coincident) triangles
  std::vector<double> tri = load_tris(argv[1]);
  if (tri.empty()) return 3;
  for (int rep = 0; rep < 2; ++rep) {
    const size_t nt = tri.size() / 12;
    std::vector<int32_t> pmat(nt, 0);
    std::vector<double> none;
    rtow::HostBvh bvh;
    rtow::build_bvh(none, none, none, tri, bvh, 4, 0.0, 0.0, 1.0);
    rtow::SceneImage img;
    rtow::make_scene_image(bvh, none, none, tri, cam, img, pmat, mats);
    if (!rtow::validate_scene_image(img, (int)nt)) return 10 + rep;
    rtow::Bvh4Image img4;
    rtow::make_bvh4_image(bvh, tri, pmat, mats, cam, img4);
    if (!img4.ok || !rtow::validate_bvh4_image(img4, nt)) return 20 + rep;
    std::printf("mesh %zu triangles: %d BVH2 nodes, %d BVH4 nodes, depth %d, %zu bytes\n", nt, img.n_nodes, img4.n_nodes,
                img4.depth, img4.blob.size());
    // the product's tree for meshes: pairs of triangles per leaf (leaf cap 2, level cost 1.5)
    rtow::HostBvh pairs;
    rtow::build_bvh(none, none, none, tri, pairs, 2, 1.5, 0.0, 1.0);
    rtow::Bvh4Image v;
    rtow::make_bvh4_image(pairs, tri, pmat, mats, cam, v);
    if (!v.ok || !rtow::validate_bvh4_image(v, nt)) return 60 + rep;
    std::printf("  pair leaves: %d BVH4 nodes, depth %d, %zu bytes\n", v.n_nodes, v.depth, v.blob.size());
    // the same tree with 64-byte nodes (binary16 planes in the mesh's frame): same links, and every plane at or
    // outside the binary32 one (give or take that one's own rounding), by no more than 1/3900 of the mesh's extent
    rtow::Bvh4Image vh;
    rtow::make_bvh4_image(pairs, tri, pmat, mats, cam, vh, /*half=*/true);
    if (!vh.ok || !vh.half || !rtow::validate_bvh4_image(vh, nt) || vh.n_nodes != v.n_nodes) return 80 + rep;
    if (vh.blob.size() != v.blob.size() - (size_t)v.n_nodes * 64) return 82 + rep;
    double worst = 0.0;
    for (int i = 0; i < v.n_nodes; ++i) {
      const float *f = reinterpret_cast<const float *>(v.blob.data() + (size_t)i * 128);
      const uint16_t *h = reinterpret_cast<const uint16_t *>(vh.blob.data() + (size_t)i * 64);
      if (std::memcmp(v.blob.data() + (size_t)i * 128 + 96, vh.blob.data() + (size_t)i * 64 + 48, 16) != 0) return 84 + rep;
      for (int k = 0; k < 3; ++k) {
        const double extent = 2000.0 / vh.map_s[k];
        for (int c = 0; c < 4; ++c) {
          const double lo32 = f[k * 8 + c], hi32 = f[k * 8 + 4 + c];
          const double lo16 = rtow::half_value(h[k * 8 + c]) / vh.map_s[k] + vh.map_c[k];
          const double hi16 = rtow::half_value(h[k * 8 + 4 + c]) / vh.map_s[k] + vh.map_c[k];
          if (std::isinf(lo32)) {  // empty slot: inverted in both
            if (!(std::isinf(lo16) && lo16 > 0 && std::isinf(hi16) && hi16 < 0)) return 86 + rep;
            continue;
          }
          const double slack32 = 4.0 * 1.2e-7 * std::max({std::fabs(lo32), std::fabs(hi32), 1.0});
          if (!(lo16 <= lo32 + slack32 && hi16 >= hi32 - slack32)) return 88 + rep;
          worst = std::max({worst, (lo32 - lo16) / extent, (hi16 - hi32) / extent});
        }
      }
    }
    if (worst > 1.0 / 3900.0) return 90 + rep;
    std::printf("  half nodes: %zu bytes, planes moved outwards by at most 1/%.0f of the extent\n", vh.blob.size(), 1.0 / worst);
    tri.insert(tri.end(), tri.begin(), tri.begin() + 12 * 50);  // second round: 50 coincident triangles
  }
  {  // concurrent subtrees (forced on this small mesh), a build whose every thread start fails, and the serial
     // build: one tree, byte-identical images (the advisor's finding on std::async across the C ABI)
    const size_t nt = tri.size() / 12;
    std::vector<int32_t> pmat(nt, 0);
    std::vector<double> none;
    std::vector<unsigned char> ref;
    for (int mode = 0; mode < 3; ++mode) {
      rtow::HostBvh bvh;
      rtow::build_bvh(none, none, none, tri, bvh, 2, 1.5, 0.0, 1.0, mode == 0 ? (1 << 30) : 64, mode == 2 ? 1 : 0);
      rtow::SceneImage img;
      rtow::make_scene_image(bvh, none, none, tri, cam, img, pmat, mats);
      if (!rtow::validate_scene_image(img, (int)nt)) return 70 + mode;
      if (mode == 0)
        ref = img.blob;
      else if (img.blob != ref)
        return 73 + mode;
    }
    std::printf("serial, concurrent and thread-start-failure builds give one image (%zu bytes)\n", ref.size());
  }
  {  // the concurrent-stretches helper of the image pass: every index once, whatever the split
    for (size_t n : {(size_t)0, (size_t)1, (size_t)7, (size_t)1000, (size_t)4097}) {
      std::vector<int> hits(n, 0);
      rtow::bvh_detail::for_stretches(n, 16, [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; ++i) hits[i] += 1;
      });
      for (size_t i = 0; i < n; ++i)
        if (hits[i] != 1) return 40;
    }
    std::printf("for_stretches: every index exactly once for 0, 1, 7, 1000 and 4097 items\n");
  }
  {  // one triangle: a single-leaf tree
    std::vector<double> one(tri.begin(), tri.begin() + 12), none;
    std::vector<int32_t> pmat(1, 0);
    rtow::HostBvh bvh;
    rtow::build_bvh(none, none, none, one, bvh, 4, 0.0, 0.0, 1.0);
    rtow::Bvh4Image img4;
    rtow::make_bvh4_image(bvh, one, pmat, mats, cam, img4);
    if (!img4.ok || !rtow::validate_bvh4_image(img4, 1)) return 30;
  }
  {  // binary16 with directed rounding (the half nodes' planes): every finite value maps to itself in both directions,
     // every point strictly between two neighbours goes down to the lower and up to the upper one
    auto ordered = [](int32_t k) { return k >= 0 ? (uint16_t)k : (uint16_t)(0x8000u | (uint32_t)(-k - 1)); };
    for (int32_t k = -0x7c00; k < 0x7bff; ++k) {  // -65504 .. the value below +65504
      const uint16_t a = ordered(k), b = ordered(k + 1);
      const double va = rtow::half_value(a), vb = rtow::half_value(b);
      if (rtow::half_value(rtow::half_directed(va, -1)) != va || rtow::half_value(rtow::half_directed(va, +1)) != va) return 95;
      if (va == vb) continue;  // -0 and +0
      const double mid = 0.5 * (va + vb), near_a = va + (vb - va) * 1e-9, near_b = vb - (vb - va) * 1e-9;
      for (double x : {mid, near_a, near_b})
        if (rtow::half_value(rtow::half_directed(x, -1)) != va || rtow::half_value(rtow::half_directed(x, +1)) != vb) return 96;
    }
    if (rtow::half_directed(1e9, +1) != 0x7c00 || rtow::half_directed(1e9, -1) != 0x7bff || rtow::half_directed(-1e9, -1) != 0xfc00 ||
        rtow::half_directed(-1e9, +1) != 0xfbff)
      return 97;
    std::printf("binary16 directed rounding: all %d finite values and the gaps between them\n", 2 * 0x7c00);
  }
  // ---- sphere scenes: grid with fat lists, static and moving
  for (int moving = 0; moving < 2; ++moving) {
    std::mt19937 rng(7);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    std::vector<double> sph, sph_r, mov, none;
    sph.insert(sph.end(), {0, -1000, 0, 1000.0 * 1000.0});
    sph_r.push_back(1000.0);
    for (int a = -11; a < 11; ++a)
      for (int b = -11; b < 11; ++b) {
        const double x = a + 0.9 * U(rng), z = b + 0.9 * U(rng);
        if (moving && U(rng) < 0.8) {
          mov.insert(mov.end(), {x, 0.2, z, 0.0, 0.5 * U(rng), 0.0, 0.04, 0.2});
        } else {
          sph.insert(sph.end(), {x, 0.2, z, 0.04});
          sph_r.push_back(0.2);
        }
      }
    const size_t n = sph_r.size() + mov.size() / 8;
    std::vector<int32_t> pmat(n, 0);
    rtow::GridImage g;
    const double cam2[3] = {13, 2, 3};
    rtow::build_grid_image(sph, sph_r, mov, none, cam2, g, 1.5, 4.0, 0.0, 1.0, pmat, mats);
    if (!g.ok) return 40 + moving;
    if ((g.fat_stride != (moving ? 80u : 48u)) || g.off_fat == 0) return 50 + moving;
    std::printf("spheres (%s): grid %dx%dx%d, image %zu bytes, fat entries of %u bytes\n", moving ? "moving" : "static", g.n[0],
                g.n[1], g.n[2], g.blob.size(), g.fat_stride);
  }
  return 0;
}
