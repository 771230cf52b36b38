// rtow_build_grid.hip — device-side build of the GRID kernel's scene image (the structure the
// sphere scenes of the reference are traced with; companion of rtow_build.hip).
//
// Same algorithm and the same arithmetic as the host builder (rtow_grid.h): exact binary64
// primitive bounds, large/small split at `large_ratio` x the median bounding-box diagonal, grid
// over the small primitives, every small primitive registered in all cells its padded bounds
// touch, per-cell id lists in ascending id order.  The host only does the few scalar steps
// between the phases (grid_header() of rtow_grid.h — shared with the host builder — and the
// section offsets), so the image is BYTE-IDENTICAL to the host-built one (tested).
//
//   phase 1  bounds, diagonals, median (radix sort of the diagonals), large/small flags, exact
//            min/max of the small bounds                     -> 72-byte read-back
//   phase 2  cells touched per primitive -> counts per cell, exclusive scan, checks (list <= 255)
//                                                                   -> 8-byte read-back
//   phase 3  fill (atomic cursors), sort each cell's list, cell words + id section + large list
//            into the image
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <math.h>
#include <stdint.h>

namespace rtow {

namespace {

struct GScratch {
  int n = 0, capacity = 0;
  double *pb = nullptr;      // [n][6] exact bounds
  double *diag = nullptr, *diag_sorted = nullptr;
  uint32_t *is_large = nullptr, *large_rank = nullptr;
  unsigned long long *glob = nullptr;  // [0..2] min, [3..5] max (ordered u64), [6] scale bits, [7] n_large,
                                       // [8] max list length, [9] total ids
  uint32_t *count = nullptr, *first = nullptr, *cursor = nullptr;  // per cell (capacity kMaxCells)
  uint32_t *ids_tmp = nullptr;
  size_t ids_capacity = 0;
  long long cell_capacity = 0;
  void *tmp = nullptr;
  size_t tmp_bytes = 0;
  int n_small = 0, n_large = 0;
};
constexpr int kMaxCells = 128 * 128 * 128;

__device__ __forceinline__ unsigned long long ordered64(double d) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(d);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__host__ __device__ inline double unordered64(unsigned long long o) {
  const unsigned long long b = (o >> 63) ? (o & 0x7fffffffffffffffull) : ~o;
  double d;
  memcpy(&d, &b, sizeof d);
  return d;
}

// exact primitive bounds, the same expressions as rtow_grid.h
__global__ void kg_bounds(const double *sph, const double *sph_r, const double *mov, const double *tri, int ns,
                          int nm, int nt, double time0, double time1, double *pb, double *diag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ns + nm + nt) return;
  double mn[3], mx[3];
  if (i < ns) {
    const double r = fabs(sph_r[i]);
    for (int k = 0; k < 3; ++k) {
      mn[k] = sph[(size_t)i * 4 + k] - r;
      mx[k] = sph[(size_t)i * 4 + k] + r;
    }
  } else if (i < ns + nm) {
    const double *m = mov + (size_t)(i - ns) * 8;
    const double r = fabs(m[7]);
    const double w = 1e-6 * (1.0 + fabs(time0) + fabs(time1));
    const double ta = fmin(time0, time1) - w, tb = fmax(time0, time1) + w;
    for (int k = 0; k < 3; ++k) {
      const double a0 = m[k] + ta * m[3 + k], a1 = m[k] + tb * m[3 + k];
      mn[k] = fmin(a0, a1) - r;
      mx[k] = fmax(a0, a1) + r;
    }
  } else {
    const double *t = tri + (size_t)(i - ns - nm) * 12;
    for (int k = 0; k < 3; ++k) {
      const double a = t[k], b = t[k] + t[3 + k], c = t[k] + t[6 + k];
      mn[k] = fmin(a, fmin(b, c));
      mx[k] = fmax(a, fmax(b, c));
    }
  }
  double s = 0;
  for (int k = 0; k < 3; ++k) {
    pb[(size_t)i * 6 + k] = mn[k];
    pb[(size_t)i * 6 + 3 + k] = mx[k];
    s += (mx[k] - mn[k]) * (mx[k] - mn[k]);
  }
  diag[i] = sqrt(s);
}

__global__ void kg_classify(const double *pb, const double *diag, const double *diag_sorted, int n, double large_ratio,
                            uint32_t *is_large, unsigned long long *glob) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double med = fmax(diag_sorted[n / 2], 1e-12);
  const bool large = diag[i] > large_ratio * med;
  is_large[i] = large ? 1u : 0u;
  double amax = 0.0;
  for (int k = 0; k < 3; ++k) {
    const double mn = pb[(size_t)i * 6 + k], mx = pb[(size_t)i * 6 + 3 + k];
    amax = fmax(amax, fmax(fabs(mn), fabs(mx)));
    if (!large) {
      atomicMin(&glob[k], ordered64(mn));
      atomicMax(&glob[3 + k], ordered64(mx));
    }
  }
  // scale: large primitives contribute their own extent, small ones through the grid bounds — the
  // max over all primitives is the same number
  atomicMax(&glob[6], (unsigned long long)__double_as_longlong(amax));  // non-negative doubles order like their bits
  if (large) atomicAdd(&glob[7], 1ull);
}

__device__ __forceinline__ void cell_range(const double *pb, int i, double pad, const float *gminf, const float *cellf,
                                           const int32_t *n, int lo[3], int hi[3]) {
  for (int k = 0; k < 3; ++k) {
    const double a = (pb[(size_t)i * 6 + k] - pad - (double)gminf[k]) / (double)cellf[k];
    const double b = (pb[(size_t)i * 6 + 3 + k] + pad - (double)gminf[k]) / (double)cellf[k];
    lo[k] = min(max((int)floor(a), 0), n[k] - 1);
    hi[k] = min(max((int)floor(b), 0), n[k] - 1);
  }
}

struct GridGeom {
  float gminf[3], cellf[3];
  int32_t n[3];
  double pad;
};

template <bool FILL>
__global__ void kg_register(const double *pb, const uint32_t *is_large, int n, GridGeom g, uint32_t *count,
                            const uint32_t *first, uint32_t *cursor, uint32_t *ids) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || is_large[i]) return;
  int lo[3], hi[3];
  cell_range(pb, i, g.pad, g.gminf, g.cellf, g.n, lo, hi);
  for (int z = lo[2]; z <= hi[2]; ++z)
    for (int y = lo[1]; y <= hi[1]; ++y)
      for (int x = lo[0]; x <= hi[0]; ++x) {
        const uint32_t c = (uint32_t)((z * g.n[1] + y) * g.n[0] + x);
        if constexpr (FILL)
          ids[first[c] + atomicAdd(&cursor[c], 1u)] = (uint32_t)i;
        else
          atomicAdd(&count[c], 1u);
      }
}

__global__ void kg_check(const uint32_t *count, const uint32_t *first, int ncell, unsigned long long *glob) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncell) return;
  atomicMax(&glob[8], (unsigned long long)count[c]);
  if (c == ncell - 1) glob[9] = (unsigned long long)first[c] + count[c];
}

// ascending ids per cell (the host appends primitives in ascending order), then the cell word
__global__ void kg_finish_cells(const uint32_t *count, const uint32_t *first, int ncell, uint32_t *ids_tmp,
                                unsigned char *blob, uint32_t off_cells, uint32_t off_ids) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncell) return;
  const uint32_t cnt = count[c], f = first[c];
  uint32_t *l = ids_tmp + f;
  for (uint32_t a = 1; a < cnt; ++a) {  // insertion sort: lists are short (<= 255, typically < 4)
    const uint32_t v = l[a];
    uint32_t b = a;
    while (b > 0 && l[b - 1] > v) {
      l[b] = l[b - 1];
      --b;
    }
    l[b] = v;
  }
  uint32_t *out = reinterpret_cast<uint32_t *>(blob + off_ids) + f;
  for (uint32_t a = 0; a < cnt; ++a) out[a] = l[a];
  reinterpret_cast<uint32_t *>(blob + off_cells)[c] = cnt ? (f << 8) | cnt : 0u;
}

// fat cell lists (rtow_grid.h): id + sphere record side by side, same order as the id list; 48-byte entries
// for static scenes, 80-byte entries [id . . .][c0x c0y][c0z dx][dy dz][r2 .] when some spheres move
__global__ void kg_fat_lists(const double *sph, const double *mov, int ns, unsigned long long total_ids,
                             unsigned char *blob, uint32_t off_ids, uint32_t off_fat, uint32_t stride) {
  const unsigned long long e = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total_ids) return;
  const uint32_t id = reinterpret_cast<const uint32_t *>(blob + off_ids)[e];
  unsigned char *dst = blob + off_fat + e * (unsigned long long)stride;
  reinterpret_cast<uint32_t *>(dst)[0] = id;
  reinterpret_cast<uint32_t *>(dst)[1] = 0u;
  reinterpret_cast<uint32_t *>(dst)[2] = 0u;
  reinterpret_cast<uint32_t *>(dst)[3] = 0u;
  double *rec = reinterpret_cast<double *>(dst + 16);
  if (stride == 48u) {
    const double *q = sph + (size_t)id * 4;
    for (int k = 0; k < 4; ++k) rec[k] = q[k];
    // k = |c|^2 - r^2 in the entry's padding (rtow_grid.h write_fat_entry: the same expression, no contraction)
    reinterpret_cast<double *>(dst + 8)[0] = (q[0] * q[0] + q[1] * q[1] + q[2] * q[2]) - fabs(q[3]);
  } else if ((int)id < ns) {
    const double *q = sph + (size_t)id * 4;
    rec[0] = q[0], rec[1] = q[1], rec[2] = q[2], rec[3] = 0.0, rec[4] = 0.0, rec[5] = 0.0, rec[6] = q[3], rec[7] = 0.0;
  } else {
    const double *q = mov + ((size_t)id - (size_t)ns) * 8;
    for (int k = 0; k < 7; ++k) rec[k] = q[k];
    rec[7] = 0.0;
  }
}

__global__ void kg_large_list(const uint32_t *is_large, const uint32_t *large_rank, int n, unsigned char *blob,
                              uint32_t off_large) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !is_large[i]) return;
  reinterpret_cast<uint32_t *>(blob + off_large)[large_rank[i]] = (uint32_t)i;
}

void grelease(GScratch *s) {
  if (!s) return;
  void *ptrs[] = {s->pb, s->diag, s->diag_sorted, s->is_large, s->large_rank, s->glob,
                  s->count, s->first, s->cursor, s->ids_tmp, s->tmp};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  delete s;
}

template <class T>
bool galloc(T *&p, size_t count) {
  return hipMalloc(reinterpret_cast<void **>(&p), (count ? count : 1) * sizeof(T)) == hipSuccess;
}

}  // namespace

// what phase 1 tells the host
struct GridBuildBounds {
  double gmn[3], gmx[3];  // exact bounds of the small primitives
  double scale_prims;     // max |coordinate| over all primitive bounds
  int32_t n_small, n_large;
};

int grid_build_phase1(const double *sph, const double *sph_r, const double *mov, const double *tri, int ns, int nm,
                      int nt, double time0, double time1, double large_ratio, void *stream, void **handle,
                      GridBuildBounds *out) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int n = ns + nm + nt;
  if (n <= 0) return 1;
  GScratch *s = static_cast<GScratch *>(*handle);
  if (!s || s->capacity < n) {
    grelease(s);
    *handle = nullptr;
    s = new GScratch;
    const size_t cap = (size_t)n + n / 8 + 64;
    bool ok = galloc(s->pb, cap * 6) && galloc(s->diag, cap) && galloc(s->diag_sorted, cap) &&
              galloc(s->is_large, cap) && galloc(s->large_rank, cap) && galloc(s->glob, 16);
    size_t a = 0, b = 0;
    if (ok) {
      ok = rocprim::radix_sort_keys(nullptr, a, s->diag, s->diag_sorted, cap, 0u, 64u, st) == hipSuccess &&
           rocprim::exclusive_scan(nullptr, b, s->is_large, s->large_rank, 0u, cap, rocprim::plus<uint32_t>(), st) == hipSuccess;
      s->tmp_bytes = a > b ? a : b;
      ok = ok && hipMalloc(&s->tmp, s->tmp_bytes ? s->tmp_bytes : 16) == hipSuccess;
    }
    if (!ok) {
      grelease(s);
      return 2;
    }
    s->capacity = (int)cap;
    *handle = s;
  }
  s->n = n;
  const unsigned long long g0[16] = {~0ull, ~0ull, ~0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0, 0, 0, 0, 0, 0};
  const int B = 256, G = (n + B - 1) / B;
  bool good = hipMemcpyAsync(s->glob, g0, sizeof g0, hipMemcpyHostToDevice, st) == hipSuccess;
  if (good) {
    hipLaunchKernelGGL(kg_bounds, dim3(G), dim3(B), 0, st, sph, sph_r, mov, tri, ns, nm, nt, time0, time1, s->pb,
                       s->diag);
    size_t tb = s->tmp_bytes;
    good = rocprim::radix_sort_keys(s->tmp, tb, s->diag, s->diag_sorted, (size_t)n, 0u, 64u, st) == hipSuccess;
  }
  if (good) {
    hipLaunchKernelGGL(kg_classify, dim3(G), dim3(B), 0, st, s->pb, s->diag, s->diag_sorted, n, large_ratio,
                       s->is_large, s->glob);
    size_t tb = s->tmp_bytes;
    good = rocprim::exclusive_scan(s->tmp, tb, s->is_large, s->large_rank, 0u, (size_t)n, rocprim::plus<uint32_t>(), st) == hipSuccess;
  }
  unsigned long long h[8];
  good = good && hipMemcpyAsync(h, s->glob, sizeof h, hipMemcpyDeviceToHost, st) == hipSuccess &&
         hipStreamSynchronize(st) == hipSuccess && hipGetLastError() == hipSuccess;
  if (!good) return 3;
  for (int k = 0; k < 3; ++k) {
    out->gmn[k] = unordered64(h[k]);
    out->gmx[k] = unordered64(h[3 + k]);
  }
  memcpy(&out->scale_prims, &h[6], sizeof(double));
  out->n_large = (int32_t)h[7];
  out->n_small = n - out->n_large;
  s->n_small = out->n_small;
  s->n_large = out->n_large;
  return 0;
}

// counts per cell + scan; returns the total number of (cell, primitive) entries and the longest list
int grid_build_phase2(void *handle, const float gminf[3], const float cellf[3], const int32_t n[3], double pad,
                      void *stream, unsigned long long *total_ids, unsigned long long *max_list) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  GScratch *s = static_cast<GScratch *>(handle);
  if (!s) return 1;
  const long long ncell = (long long)n[0] * n[1] * n[2];
  if (ncell <= 0 || ncell > kMaxCells) return 1;
  if (s->cell_capacity < ncell) {
    for (uint32_t **p : {&s->count, &s->first, &s->cursor}) {
      if (*p) (void)hipFree(*p);
      *p = nullptr;
    }
    s->cell_capacity = ncell + ncell / 4 + 64;
    if (!(galloc(s->count, (size_t)s->cell_capacity) && galloc(s->first, (size_t)s->cell_capacity) &&
          galloc(s->cursor, (size_t)s->cell_capacity)))
      return 2;
    size_t b = 0;
    if (rocprim::exclusive_scan(nullptr, b, s->count, s->first, 0u, (size_t)s->cell_capacity, rocprim::plus<uint32_t>(), st) != hipSuccess) return 2;
    if (b > s->tmp_bytes) {
      (void)hipFree(s->tmp);
      s->tmp = nullptr;
      if (hipMalloc(&s->tmp, b) != hipSuccess) return 2;
      s->tmp_bytes = b;
    }
  }
  GridGeom g;
  for (int k = 0; k < 3; ++k) {
    g.gminf[k] = gminf[k];
    g.cellf[k] = cellf[k];
    g.n[k] = n[k];
  }
  g.pad = pad;
  const int B = 256;
  bool good = hipMemsetAsync(s->count, 0, (size_t)ncell * 4, st) == hipSuccess &&
              hipMemsetAsync(s->cursor, 0, (size_t)ncell * 4, st) == hipSuccess;
  if (good) {
    hipLaunchKernelGGL(kg_register<false>, dim3((s->n + B - 1) / B), dim3(B), 0, st, s->pb, s->is_large, s->n, g,
                       s->count, s->first, s->cursor, (uint32_t *)nullptr);
    size_t tb = s->tmp_bytes;
    good = rocprim::exclusive_scan(s->tmp, tb, s->count, s->first, 0u, (size_t)ncell, rocprim::plus<uint32_t>(), st) == hipSuccess;
  }
  if (good)
    hipLaunchKernelGGL(kg_check, dim3(((int)ncell + B - 1) / B), dim3(B), 0, st, s->count, s->first, (int)ncell,
                       s->glob);
  unsigned long long h[2] = {0, 0};
  good = good && hipMemcpyAsync(h, s->glob + 8, sizeof h, hipMemcpyDeviceToHost, st) == hipSuccess &&
         hipStreamSynchronize(st) == hipSuccess && hipGetLastError() == hipSuccess;
  if (!good) return 3;
  *max_list = h[0];
  *total_ids = h[1];
  return 0;
}

// fill + sort the cell lists, write cell words, ids and the large list into the image
int grid_build_phase3(void *handle, const float gminf[3], const float cellf[3], const int32_t n[3], double pad,
                      unsigned long long total_ids, unsigned char *blob_dev, uint32_t off_cells, uint32_t off_ids,
                      void *stream, const double *sph, uint32_t off_fat, const double *mov, int ns, uint32_t fat_stride) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  GScratch *s = static_cast<GScratch *>(handle);
  if (!s) return 1;
  const int ncell = n[0] * n[1] * n[2];
  if (s->ids_capacity < total_ids + 1) {
    if (s->ids_tmp) (void)hipFree(s->ids_tmp);
    s->ids_tmp = nullptr;
    s->ids_capacity = (size_t)total_ids + total_ids / 4 + 64;
    if (!galloc(s->ids_tmp, s->ids_capacity)) return 2;
  }
  GridGeom g;
  for (int k = 0; k < 3; ++k) {
    g.gminf[k] = gminf[k];
    g.cellf[k] = cellf[k];
    g.n[k] = n[k];
  }
  g.pad = pad;
  const int B = 256;
  hipLaunchKernelGGL(kg_register<true>, dim3((s->n + B - 1) / B), dim3(B), 0, st, s->pb, s->is_large, s->n, g,
                     s->count, s->first, s->cursor, s->ids_tmp);
  hipLaunchKernelGGL(kg_finish_cells, dim3((ncell + B - 1) / B), dim3(B), 0, st, s->count, s->first, ncell,
                     s->ids_tmp, blob_dev, off_cells, off_ids);
  if (off_fat != 0u && total_ids > 0)
    hipLaunchKernelGGL(kg_fat_lists, dim3((unsigned)((total_ids + B - 1) / B)), dim3(B), 0, st, sph, mov, ns, total_ids,
                       blob_dev, off_ids, off_fat, fat_stride);
  hipLaunchKernelGGL(kg_large_list, dim3((s->n + B - 1) / B), dim3(B), 0, st, s->is_large, s->large_rank, s->n,
                     blob_dev, off_ids + 4u * (uint32_t)total_ids);
  const bool good = hipStreamSynchronize(st) == hipSuccess && hipGetLastError() == hipSuccess;
  return good ? 0 : 3;
}

void grid_build_release(void *handle) { grelease(static_cast<GScratch *>(handle)); }

}  // namespace rtow
