// Stream reduce: global_image = sum over streams of the partial images, added in
// stream order exactly like the reference joins its threads
// (src/render.cpp:176-180: global = done + global, futures taken in launch order).
#include <hip/hip_runtime.h>

#include "rtow_device.h"

namespace rtow {
namespace {
__global__ void __launch_bounds__(256) rtow_reduce_streams(const ReduceParams p) {
  const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
  if (idx >= p.npix3) return;
  // where this value sits in the (possibly tiled) item order of the trace kernel
  uint32_t src = idx;
  if (p.tile_h_log2 != 0u) {
    const uint32_t px = idx / 3u, c = idx - px * 3u;
    const uint32_t lr = px / p.W, j = px - lr * p.W;
    const uint32_t tr = lr >> p.tile_h_log2, tc = j >> p.tile_w_log2;
    const uint32_t w = ((lr & ((1u << p.tile_h_log2) - 1u)) << p.tile_w_log2) | (j & ((1u << p.tile_w_log2) - 1u));
    src = (((tr * p.tiles_per_row + tc) << 6) | w) * 3u + c;
  }
  double g = p.accumulate ? p.out[idx] : 0.0;
  for (int k = 0; k < p.nstreams; ++k) g = p.partials[(size_t)k * p.npix3 + src] + g;
  if (p.rgb8 != nullptr) {
    // write_color fused into the reduce (rtow_render_rgb8: the f64 sums never go to memory); the same
    // operations as rtow_tonemap_u8 below, so the bytes are the same
    double c = sqrt(g / p.spp);
    c = c < 0.0 ? 0.0 : (c > 0.999 ? 0.999 : c);
    const int v = (int)(256.0 * c);
    p.rgb8[idx] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
  } else {
    p.out[idx] = g;
  }
}
// write_color on the device (src/render.cpp:11-20): c = sqrt(sum / spp), clamp to [0, 0.999],
// int(256 * c) — one byte per channel.  f64 sqrt and division are the correctly rounded IEEE
// forms and this file is built with -ffp-contract=off, so the bytes equal the reference's.
__global__ void __launch_bounds__(256) rtow_tonemap_u8(const double *sums, unsigned char *rgb8, uint32_t n,
                                                       double spp) {
  const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
  if (idx >= n) return;
  double c = sqrt(sums[idx] / spp);
  c = c < 0.0 ? 0.0 : (c > 0.999 ? 0.999 : c);  // std::clamp(c, 0.0, 0.999); NaN stays NaN -> 0 below
  const int v = (int)(256.0 * c);
  rgb8[idx] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}
// Strips back to their rows, on the device (the multi-GPU output path, rtow_multi.cpp): `gathered` is
// [rank][max_rows][row_words] as one gather delivers it; global row i belongs to strip t = i / tile_rows, rank t % n,
// and is that rank's local row (t / n) * tile_rows + i % tile_rows.  One thread per word of the image.
template <class Word>
__global__ void __launch_bounds__(256) rtow_place_rows(const Word *gathered, Word *image, uint32_t n_ranks, uint32_t max_rows,
                                                       uint32_t row_words, uint32_t height, uint32_t tile_rows) {
  const uint32_t w = blockIdx.x * 256u + threadIdx.x, i = blockIdx.y;
  if (w >= row_words || i >= height) return;
  const uint32_t t = i / tile_rows, r = t % n_ranks;
  const uint32_t lr = (t / n_ranks) * tile_rows + (i - t * tile_rows);
  image[(size_t)i * row_words + w] = gathered[((size_t)r * max_rows + lr) * row_words + w];
}
}  // namespace

int launch_place_rows(const void *gathered, void *image, uint32_t n_ranks, uint32_t max_rows, uint32_t row_bytes,
                      uint32_t height, uint32_t tile_rows, void *stream) {
  if (row_bytes == 0 || height == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  // widest word that divides a row (rows of f64 sums: 8 bytes at least; W*3 bytes of an rgb8 row: often 16)
  if (row_bytes % 16u == 0u) {
    const uint32_t rw = row_bytes / 16u;
    hipLaunchKernelGGL(rtow_place_rows<uint4>, dim3((rw + 255u) / 256u, height), dim3(256), 0, st, (const uint4 *)gathered,
                       (uint4 *)image, n_ranks, max_rows, rw, height, tile_rows);
  } else if (row_bytes % 8u == 0u) {
    const uint32_t rw = row_bytes / 8u;
    hipLaunchKernelGGL(rtow_place_rows<uint2>, dim3((rw + 255u) / 256u, height), dim3(256), 0, st, (const uint2 *)gathered,
                       (uint2 *)image, n_ranks, max_rows, rw, height, tile_rows);
  } else if (row_bytes % 4u == 0u) {
    const uint32_t rw = row_bytes / 4u;
    hipLaunchKernelGGL(rtow_place_rows<uint32_t>, dim3((rw + 255u) / 256u, height), dim3(256), 0, st, (const uint32_t *)gathered,
                       (uint32_t *)image, n_ranks, max_rows, rw, height, tile_rows);
  } else {
    hipLaunchKernelGGL(rtow_place_rows<unsigned char>, dim3((row_bytes + 255u) / 256u, height), dim3(256), 0, st,
                       (const unsigned char *)gathered, (unsigned char *)image, n_ranks, max_rows, row_bytes, height, tile_rows);
  }
  return (int)hipGetLastError();
}

int launch_tonemap(const double *sums, unsigned char *rgb8, uint32_t n, double spp, void *stream) {
  const unsigned grid = (n + 255u) / 256u;
  if (grid == 0) return 0;
  hipLaunchKernelGGL(rtow_tonemap_u8, dim3(grid), dim3(256), 0, (hipStream_t)stream, sums, rgb8, n, spp);
  return (int)hipGetLastError();
}

int launch_reduce(const ReduceParams &p, void *stream) {
  const unsigned grid = (p.npix3 + 255u) / 256u;
  if (grid == 0) return 0;
  hipLaunchKernelGGL(rtow_reduce_streams, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  return (int)hipGetLastError();
}
}  // namespace rtow
