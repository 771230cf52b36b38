// rtow_grid.h — host build of the uniform-grid scene image (kernel RTOW_KERNEL_GRID).
//
// Why a grid next to the BVH: in the reference's scenes (hundreds of small primitives
// spread over a ground plane, or one compact mesh) most rays start INSIDE the bounds of
// the small primitives, so a BVH walk first descends ~2·log2(N) boxes that merely contain
// the ray origin before it learns anything.  A 3D-DDA starts in the origin's cell and
// visits a handful of mostly empty cells.  Primitives much larger than the rest (the
// r = 1000 ground sphere) would occupy every cell, so they go to a short "large" list
// that every ray tests first.  Leaf tests are the same f64 code as the other kernels, and
// primitives are registered in every cell their PADDED bounds touch, so the set of
// candidates is conservative and the accepted (t, primitive) is identical.
//
// Blob layout (16-byte aligned sections):
//   [header 64 B][cells ncell x 4 B][ids][sphere 32 B][moving 64 B][triangle 96 B records]
//   header : f32 gmin[3], f32 cell[3], f32 inv_cell[3], i32 n[3], u32 n_large, u32 off_large,
//            pad (off_large = byte offset of the large-primitive id list inside [ids])
//   cell   : u32 (first << 8 | count) into the id section, 0 = empty
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace rtow {

struct GridImage {
  std::vector<unsigned char> blob;
  uint32_t off_cells = 0, off_ids = 0, off_sph = 0, off_mov = 0, off_tri = 0;
  uint32_t off_pmat = 0, off_mats = 0;  // material index per primitive, material records
  int32_t n[3] = {1, 1, 1};
  uint32_t n_large = 0;
  size_t total_bytes = 0;
  uint32_t off_fat = 0;
  uint32_t fat_stride = 0;  // bytes per fat list entry: 48 (static spheres) or 80 (scenes with moving spheres)
  bool ok = false;  // false: scene not suited (e.g. lists too long) — use the BVH
};

namespace grid_detail {
struct B3 {
  double mn[3], mx[3];
};
}  // namespace grid_detail

// Grid origin, resolution and cell size from the bounds of the small primitives — shared by the
// host builder below and the device builder (rtow_build_grid.hip), so both lay out the same grid.
// `scale_prims`: max |coordinate| over the small bounds (unpadded) and the large primitives' bounds.
// An axis whose extent is below this many cell sizes gets a single layer of cells instead of two
// thin ones (the slab of small spheres with moving spheres swept 0.5 upwards: 3.2 instead of 3.7 cell
// steps per segment, +1.6 %; no effect on the static scene).  RTOW_GRID_FLAT overrides; 1.0 = plain
// rounding up.
inline double grid_flat_ratio() {
  const char *e = std::getenv("RTOW_GRID_FLAT");
  return e ? std::atof(e) : 1.5;
}

struct GridHeader {
  double pad = 0.0;
  int32_t n[3] = {1, 1, 1};
  float gminf[3] = {0, 0, 0}, cellf[3] = {1, 1, 1}, invf[3] = {1, 1, 1};
};
inline void grid_header(const double gmn_in[3], const double gmx_in[3], double scale_prims, const double cam_origin[3],
                        size_t n_small, double cells_per_prim, GridHeader &h) {
  // the f32 DDA sees the ray rounded to f32: pad by more than that rounding can move the ray
  // or a cell boundary anywhere in the scene (same reasoning as the BVH boxes, rtow_bvh.h)
  double scale = std::max(1.0, scale_prims);
  for (int k = 0; k < 3; ++k) scale = std::max(scale, std::fabs(cam_origin[k]));
  h.pad = 4e-6 * scale;
  double gmn[3], gmx[3], ext[3];
  for (int k = 0; k < 3; ++k) {
    gmn[k] = gmn_in[k] - 2 * h.pad;
    gmx[k] = gmx_in[k] + 2 * h.pad;
    ext[k] = std::max(gmx[k] - gmn[k], 1e-9);
  }
  const double target = std::min(std::max(cells_per_prim * (double)n_small, 8.0), 32768.0);
  double s = std::cbrt(ext[0] * ext[1] * ext[2] / target);
  const double flat_ratio = grid_flat_ratio();
  bool flat[3];
  int thick = 0;
  double span = 1.0;
  for (int k = 0; k < 3; ++k) {
    flat[k] = ext[k] < flat_ratio * s;  // a thin slab: one layer (see grid_flat_ratio())
    if (!flat[k]) {
      ++thick;
      span *= ext[k];
    }
  }
  // the cell count is meant to be `target`: with an axis collapsed to one layer, the cell size follows from the
  // other axes alone (the cover scene, one layer of spheres: 35 x 1 x 35 cells for a target of 723 otherwise)
  if (thick > 0 && thick < 3) s = std::pow(span / target, 1.0 / thick);
  for (int k = 0; k < 3; ++k) {
    int nk = (int)std::ceil(ext[k] / s);
    if (flat[k]) nk = 1;
    nk = std::min(std::max(nk, 1), 128);
    h.n[k] = nk;
    h.gminf[k] = std::nextafterf((float)gmn[k], -INFINITY);
    // cell size as f32, rounded up so n*cell covers the extent
    h.cellf[k] = std::nextafterf((float)((gmx[k] - (double)h.gminf[k]) / nk), INFINITY);
    h.invf[k] = 1.0f / h.cellf[k];
  }
}

// Section offsets and everything but the header, the cell words and the id section.
inline void layout_grid_image(size_t ncell, size_t total_ids, size_t n_large, const std::vector<double> &sph,
                              const std::vector<double> &mov, const std::vector<double> &tri,
                              const std::vector<int32_t> &prim_mat, const std::vector<unsigned char> &mats_bytes,
                              GridImage &img, bool offsets_only = false, size_t fat_entries = 0,
                              uint32_t fat_stride = 48) {
  const size_t hdr = 64;
  img.off_cells = (uint32_t)hdr;
  const size_t cells_bytes = ((ncell * 4 + 15) / 16) * 16;
  img.off_ids = (uint32_t)(hdr + cells_bytes);
  const size_t ids_bytes = ((total_ids * 4 + 15) / 16) * 16;
  // "fat" cell lists: id + sphere record side by side (same order as the id list), so a cell test is ONE
  // round of LDS reads instead of id -> record.  48 B per entry [id . . .][cx cy][cz r2] when all small
  // spheres are static; 80 B [id . . .][c0x c0y][c0z dx][dy dz][r2 .] when some move (a static sphere
  // then has d = 0)
  img.off_fat = fat_entries ? (uint32_t)(img.off_ids + ids_bytes) : 0u;
  img.fat_stride = fat_entries ? fat_stride : 0u;
  img.off_sph = (uint32_t)(img.off_ids + ids_bytes + fat_entries * fat_stride);
  img.off_mov = img.off_sph + (uint32_t)(sph.size() * 8);
  img.off_tri = img.off_mov + (uint32_t)(mov.size() * 8);
  img.off_pmat = (uint32_t)((((size_t)img.off_tri + tri.size() * 8) + 15) / 16 * 16);
  img.off_mats = (uint32_t)((((size_t)img.off_pmat + prim_mat.size() * 4) + 15) / 16 * 16);
  const size_t total = (size_t)img.off_mats + mats_bytes.size();
  img.total_bytes = ((total + 15) / 16) * 16;
  img.n_large = (uint32_t)n_large;
  if (offsets_only) return;
  img.blob.assign(img.total_bytes, 0);
  if (!prim_mat.empty()) std::memcpy(img.blob.data() + img.off_pmat, prim_mat.data(), prim_mat.size() * 4);
  if (!mats_bytes.empty()) std::memcpy(img.blob.data() + img.off_mats, mats_bytes.data(), mats_bytes.size());
}

// the 64-byte image header (see the layout comment at the top)
inline void write_grid_header(unsigned char *h, const GridHeader &hd, uint32_t n_large, uint32_t off_large,
                              uint32_t off_fat = 0, uint32_t fat_stride = 0) {
  std::memset(h, 0, 64);
  std::memcpy(h + 56, &off_fat, 4);
  std::memcpy(h + 60, &fat_stride, 4);
  std::memcpy(h + 0, hd.gminf, 12);
  std::memcpy(h + 12, hd.cellf, 12);
  std::memcpy(h + 24, hd.invf, 12);
  std::memcpy(h + 36, hd.n, 12);
  std::memcpy(h + 48, &n_large, 4);
  std::memcpy(h + 52, &off_large, 4);
}

// Fat cell lists (id + sphere record per list entry): for scenes made of spheres only, and only if the
// image still fits the 160 KiB of LDS with them (shared by the host and the device builder).  Returns the
// entry size: 0 = no fat lists, 48 = static spheres only, 80 = some spheres move.
inline uint32_t grid_wants_fat_lists(int n_moving, int n_triangles, size_t ncell, size_t total_ids, size_t n_large,
                                     const std::vector<double> &sph, const std::vector<double> &mov,
                                     const std::vector<double> &tri, const std::vector<int32_t> &prim_mat,
                                     const std::vector<unsigned char> &mats_bytes, size_t n_cell_ids,
                                     double scale_small = 0.0) {
  if (n_triangles != 0 || n_cell_ids == 0 || std::getenv("RTOW_GRID_NO_FAT")) return 0u;
  // 48-byte entries carry k = |c|^2 - r^2 for the fast builds' 8-operation test, c = |o|^2 - 2 c.o + k, whose
  // cancellation is of the size of |o|^2 and |c|^2 in WORLD coordinates: 2^-52 * 2 scale^2 absolute, against r^2.
  // Measured on the cover scene moved off the origin (scripts/far_origin_check.py, profiles/r05_far_origin.log): at
  // coordinates of 1e4 the fast image differs from the strict one beyond 1e-9 in 45 % of the pixels (5e-7 per sample:
  // invisible, but not the build's usual agreement); at the origin in none.  A scene whose small primitives (and
  // camera) sit further out than 1,000 times its smallest sphere radius therefore takes the 80-byte entries — centre
  // and r^2 as they are, the 12-operation test on o - c — like a scene with moving spheres.  (Rendering in the frame
  // of the grid's centre instead keeps the 8 operations at any distance and costs 3 subtractions and a scalar-load
  // wait per segment: -1 % on the benchmark scene; scripts/experiments/r05_fat_entries_in_grid_frame.patch.)
  double min_r2 = INFINITY;
  for (size_t i = 0; i < sph.size() / 4; ++i) min_r2 = std::min(min_r2, std::fabs(sph[i * 4 + 3]));
  const bool far = scale_small * scale_small > 1e6 * min_r2;
  const uint32_t stride = (n_moving != 0 || far) ? 80u : 48u;
  if (stride == 80u && n_moving != 0 && std::getenv("RTOW_GRID_NO_FAT_MOVING")) return 0u;
  GridImage probe;
  layout_grid_image(ncell, total_ids, n_large, sph, mov, tri, prim_mat, mats_bytes, probe, true, n_cell_ids, stride);
  return probe.total_bytes <= 160u * 1024u ? stride : 0u;
}

// one fat entry (host builder; rtow_build_grid.hip has the device twin)
inline void write_fat_entry(unsigned char *dst, uint32_t stride, int32_t id, const std::vector<double> &sph,
                            const std::vector<double> &mov) {
  const size_t ns = sph.size() / 4;
  std::memset(dst, 0, stride);
  std::memcpy(dst, &id, 4);
  if (stride == 48u) {
    // [id . k][cx cy][cz r2]: k = |c|^2 - r^2, the ray-independent part of the fast builds' discriminant (they test
    // h = o.d - c.d, c' = |o|^2 - 2 c.o + k against a unit direction: 8 operations instead of 12; rtow_trace_bvh.h).
    // The strict build reads c and r2 only.  (-ffp-contract=off on both builders: the images stay byte-identical.)
    const double *q = &sph[(size_t)id * 4];
    const double k = (q[0] * q[0] + q[1] * q[1] + q[2] * q[2]) - std::fabs(q[3]);
    std::memcpy(dst + 8, &k, 8);
    std::memcpy(dst + 16, q, 32);
  } else if ((size_t)id < ns) {
    const double *q = &sph[(size_t)id * 4];
    const double rec[8] = {q[0], q[1], q[2], 0.0, 0.0, 0.0, q[3], 0.0};
    std::memcpy(dst + 16, rec, 64);
  } else {
    const double *q = &mov[((size_t)id - ns) * 8];  // c0xyz dxyz r2 r
    const double rec[8] = {q[0], q[1], q[2], q[3], q[4], q[5], q[6], 0.0};
    std::memcpy(dst + 16, rec, 64);
  }
}

// sph [n][4] cx cy cz r2, sph_r [n]; mov [n][8] c0 delta r2 r; tri [n][12] a e1 e2 n
inline void build_grid_image(const std::vector<double> &sph, const std::vector<double> &sph_r,
                             const std::vector<double> &mov, const std::vector<double> &tri,
                             const double cam_origin[3], GridImage &img, double cells_per_prim = 1.0,
                             double large_ratio = 4.0, double time0 = 0.0, double time1 = 1.0,
                             const std::vector<int32_t> &prim_mat = {},
                             const std::vector<unsigned char> &mats_bytes = {}) {
  using grid_detail::B3;
  const int ns = (int)sph_r.size(), nm = (int)(mov.size() / 8), nt = (int)(tri.size() / 12);
  const int np = ns + nm + nt;
  img.ok = false;
  if (np == 0) return;
  std::vector<B3> pb(np);
  for (int i = 0; i < ns; ++i) {
    const double r = std::fabs(sph_r[i]);
    for (int k = 0; k < 3; ++k) {
      pb[i].mn[k] = sph[(size_t)i * 4 + k] - r;
      pb[i].mx[k] = sph[(size_t)i * 4 + k] + r;
    }
  }
  for (int i = 0; i < nm; ++i) {  // centre(time) over the camera's shutter interval, widened a little
    const double *m = &mov[(size_t)i * 8];
    const double r = std::fabs(m[7]);
    const double w = 1e-6 * (1.0 + std::fabs(time0) + std::fabs(time1));
    const double ta = std::min(time0, time1) - w, tb = std::max(time0, time1) + w;
    for (int k = 0; k < 3; ++k) {
      const double a0 = m[k] + ta * m[3 + k], a1 = m[k] + tb * m[3 + k];
      pb[ns + i].mn[k] = std::min(a0, a1) - r;
      pb[ns + i].mx[k] = std::max(a0, a1) + r;
    }
  }
  for (int i = 0; i < nt; ++i) {
    const double *t = &tri[(size_t)i * 12];
    for (int k = 0; k < 3; ++k) {
      const double a = t[k], b = t[k] + t[3 + k], c = t[k] + t[6 + k];
      pb[ns + nm + i].mn[k] = std::min(a, std::min(b, c));
      pb[ns + nm + i].mx[k] = std::max(a, std::max(b, c));
    }
  }
  // large primitives: bounding-box diagonal > large_ratio x the median diagonal (cover scene:
  // the ground sphere and the three r = 1 spheres; the grid then only spans the thin slab of
  // r = 0.2 spheres, which most rays leave after one or two cells)
  std::vector<double> diag(np);
  for (int i = 0; i < np; ++i) {
    double s = 0;
    for (int k = 0; k < 3; ++k) s += (pb[i].mx[k] - pb[i].mn[k]) * (pb[i].mx[k] - pb[i].mn[k]);
    diag[i] = std::sqrt(s);
  }
  std::vector<double> sorted = diag;
  std::nth_element(sorted.begin(), sorted.begin() + np / 2, sorted.end());
  const double med = std::max(sorted[np / 2], 1e-12);
  std::vector<int32_t> large, small;
  for (int i = 0; i < np; ++i) (diag[i] > large_ratio * med ? large : small).push_back(i);
  if (small.empty() || large.size() > 64) return;

  // grid bounds = bounds of the small primitives
  double gmn[3] = {INFINITY, INFINITY, INFINITY}, gmx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i : small)
    for (int k = 0; k < 3; ++k) {
      gmn[k] = std::min(gmn[k], pb[i].mn[k]);
      gmx[k] = std::max(gmx[k], pb[i].mx[k]);
    }
  double scale = 1.0;
  for (int k = 0; k < 3; ++k) {
    scale = std::max(scale, std::max(std::fabs(gmn[k]), std::fabs(gmx[k])));
    for (int i : large) scale = std::max(scale, std::max(std::fabs(pb[i].mn[k]), std::fabs(pb[i].mx[k])));
  }
  GridHeader hd;
  grid_header(gmn, gmx, scale, cam_origin, small.size(), cells_per_prim, hd);
  const double pad = hd.pad;
  long long ncell = 1;
  float cellf[3], gminf[3];
  for (int k = 0; k < 3; ++k) {
    img.n[k] = hd.n[k];
    ncell *= hd.n[k];
    gminf[k] = hd.gminf[k];
    cellf[k] = hd.cellf[k];
  }
  // register every small primitive in all cells its padded bounds touch (computed with the
  // same f32 gmin/cell the kernel uses, widened by one ulp-ish margin through `pad`)
  std::vector<std::vector<int32_t>> lists((size_t)ncell);
  for (int i : small) {
    int lo[3], hi[3];
    for (int k = 0; k < 3; ++k) {
      const double a = (pb[i].mn[k] - pad - (double)gminf[k]) / (double)cellf[k];
      const double b = (pb[i].mx[k] + pad - (double)gminf[k]) / (double)cellf[k];
      lo[k] = std::min(std::max((int)std::floor(a), 0), img.n[k] - 1);
      hi[k] = std::min(std::max((int)std::floor(b), 0), img.n[k] - 1);
    }
    for (int z = lo[2]; z <= hi[2]; ++z)
      for (int y = lo[1]; y <= hi[1]; ++y)
        for (int x = lo[0]; x <= hi[0]; ++x)
          lists[((size_t)z * img.n[1] + y) * img.n[0] + x].push_back(i);
  }
  size_t total_ids = large.size();
  for (auto &l : lists) {
    if (l.size() > 255) return;  // too dense for the 8-bit count: let the BVH handle this scene
    total_ids += l.size();
  }
  if (total_ids >= (1u << 24)) return;

  // fat lists: scenes of spheres only, when the image still fits LDS with them
  const size_t n_cell_ids = total_ids - large.size();
  double scale_small = 0.0;  // how far from the origin the small primitives and the camera sit
  for (int k = 0; k < 3; ++k)
    scale_small = std::max({scale_small, std::fabs(gmn[k]), std::fabs(gmx[k]), std::fabs(cam_origin[k])});
  const uint32_t fat = grid_wants_fat_lists(nm, nt, (size_t)ncell, total_ids, large.size(), sph, mov, tri, prim_mat,
                                            mats_bytes, n_cell_ids, scale_small);
  layout_grid_image((size_t)ncell, total_ids, large.size(), sph, mov, tri, prim_mat, mats_bytes, img, false,
                    fat ? n_cell_ids : 0, fat ? fat : 48u);

  std::vector<int32_t> ids;
  ids.reserve(total_ids);
  std::vector<uint32_t> cells((size_t)ncell, 0u);
  for (size_t c = 0; c < (size_t)ncell; ++c) {
    if (lists[c].empty()) continue;
    cells[c] = ((uint32_t)ids.size() << 8) | (uint32_t)lists[c].size();
    ids.insert(ids.end(), lists[c].begin(), lists[c].end());
  }
  const uint32_t off_large = (uint32_t)(img.off_ids + ids.size() * 4);
  ids.insert(ids.end(), large.begin(), large.end());

  unsigned char *h = img.blob.data();
  write_grid_header(h, hd, img.n_large, off_large, img.off_fat, img.fat_stride);
  if (img.off_fat)
    for (size_t e = 0; e < n_cell_ids; ++e)
      write_fat_entry(h + img.off_fat + e * img.fat_stride, img.fat_stride, ids[e], sph, mov);
  std::memcpy(h + img.off_cells, cells.data(), cells.size() * 4);
  if (!ids.empty()) std::memcpy(h + img.off_ids, ids.data(), ids.size() * 4);
  if (!sph.empty()) std::memcpy(h + img.off_sph, sph.data(), sph.size() * 8);
  if (!mov.empty()) std::memcpy(h + img.off_mov, mov.data(), mov.size() * 8);
  if (!tri.empty()) std::memcpy(h + img.off_tri, tri.data(), tri.size() * 8);
  img.ok = true;
}

}  // namespace rtow
