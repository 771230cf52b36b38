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
}  // namespace

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
