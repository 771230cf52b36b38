// rtow_trace_body.h — the trace kernels (included by rtow_trace_strict.hip and
// rtow_trace_fast.hip, which differ only in -ffp-contract and in RTOW_SUFFIX).
//
// What it computes: the per-pixel sample loop of the reference's render()
// (src/render.cpp:150-167) with ray_color()'s recursion (src/render.cpp:112-129)
// flattened into an iterative per-lane loop.
//
// Execution model (gfx950, wave64):
//   * persistent lanes.  A work item is (stream k, pixel): spt consecutive samples
//     of one pixel, summed in sample order exactly like one reference "thread"
//     (src/render.cpp:151-166).  A lane that finishes an item takes the next one
//     from a global counter; the fetch is wave-aggregated (one atomic per wave per
//     fetch round: __ballot + popcount ranks), so lanes stay dense until the
//     queue is empty, and the image does not depend on which lane traced what.
//   * ray regeneration.  Each trip of the main loop advances every live lane by
//     exactly one ray segment; a lane whose path ended starts its next sample in
//     the same trip, so the closest-hit loop always runs on a full wave.
//   * closest hit, streaming kernel: every lane tests every primitive.  The
//     primitive index is wave-uniform, so each record is fetched with ONE scalar
//     load into SGPRs and used directly as a VALU operand: the scene costs no
//     VGPRs and no LDS bandwidth.
//   * radiance.  The reference multiplies attenuations on the way back up the
//     recursion, a1*(a2*(...*(an*sky))).  To reproduce that order bit for bit the
//     lane records the material index of every bounce in a per-lane path stack in
//     HBM ([bounce][lane], coalesced) and folds it from the end when the path
//     escapes to the sky.  A path that ends black contributes an exact zero.
//   * RNG: Philox4x32-10, counter (draw>>1, sample, pixel, 0), key = seed; the two
//     64-bit halves of a block are consecutive draws, mapped to [0,1) exactly like
//     libstdc++'s generate_canonical<double,53> maps two mt19937 words
//     (src/random-utils.cpp:11-13).
//
// In the strict build (-ffp-contract=off) every expression below has the operand
// order of the reference expression it restates, f64 sqrt and division are the
// correctly rounded IEEE forms, and the result is bit-identical to oracle/.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rtow_device.h"

#ifndef RTOW_SUFFIX
#error "define RTOW_SUFFIX"
#endif
#define RTOW_CAT2(a, b) a##b
#define RTOW_CAT(a, b) RTOW_CAT2(a, b)

namespace rtow {
namespace {

// Scene arrays are immutable during a launch: reading them through the constant
// address space lets hipcc use scalar loads for wave-uniform indices.
#define RTOW_CONST __attribute__((address_space(4)))
typedef const RTOW_CONST double *cdptr;
typedef const RTOW_CONST int32_t *ciptr;

struct V3 {
  double x, y, z;
};
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
// glm: dot = x*x' + y*y' + z*z' (left to right); cross, normalize (v * 1/sqrt),
// reflect (I - N*dot(N,I)*2), refract — same definitions as oracle/rtow_oracle.cpp.
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 x, V3 y) {
  return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y};
}
__device__ __forceinline__ V3 normalize(V3 v) { return v * (1.0 / sqrt(dot(v, v))); }
__device__ __forceinline__ V3 reflect(V3 I, V3 N) { return I - N * dot(N, I) * 2.0; }
__device__ __forceinline__ V3 refract(V3 I, V3 N, double eta) {
  double d = dot(N, I);
  double k = 1.0 - eta * eta * (1.0 - d * d);
  if (k >= 0.0) return eta * I - (eta * d + sqrt(k)) * N;
  return {0.0, 0.0, 0.0};
}
__device__ __forceinline__ V3 ld3(const double *p) { return {p[0], p[1], p[2]}; }

// ------------------------------------------------------------------ Philox ---
struct Rng {
  uint32_t pixel, sample, d;
  uint32_t w2, w3;  // second half of the current block (valid when d is odd)
};

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t &o0, uint32_t &o1,
                                              uint32_t &o2, uint32_t &o3) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0;
    c1 = n1;
    c2 = n2;
    c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  o0 = c0;
  o1 = c1;
  o2 = c2;
  o3 = c3;
}

// (w0 + w1*2^32) / 2^64 with double rounding steps, < 1 enforced
__device__ __forceinline__ double canonical_from_words(uint32_t w0, uint32_t w1) {
  double sum = (double)w0 + (double)w1 * 4294967296.0;
  double r = sum * 0x1p-64;
  if (r >= 1.0) r = 0x1.fffffffffffffp-1;
  return r;
}

__device__ __forceinline__ double rng_canonical(Rng &g, uint32_t k0, uint32_t k1) {
  uint32_t w0, w1;
  if ((g.d & 1u) == 0u) {
    uint32_t o0, o1, o2, o3;
    philox4x32_10(g.d >> 1, g.sample, g.pixel, 0u, k0, k1, o0, o1, o2, o3);
    w0 = o0;
    w1 = o1;
    g.w2 = o2;
    g.w3 = o3;
  } else {
    w0 = g.w2;
    w1 = g.w3;
  }
  g.d += 1u;
  return canonical_from_words(w0, w1);
}

// src/random-utils.cpp:11-13: canonical*(b-a)+a
__device__ __forceinline__ double rng_range(Rng &g, uint32_t k0, uint32_t k1, double a, double b) {
  return rng_canonical(g, k0, k1) * (b - a) + a;
}

// src/random-utils.cpp:23-33: a point of [0,1)^3 inside the unit ball, not normalised
__device__ __forceinline__ V3 rng_unit_vector(Rng &g, uint32_t k0, uint32_t k1) {
  V3 v;
  for (;;) {
    v.x = rng_canonical(g, k0, k1);
    v.y = rng_canonical(g, k0, k1);
    v.z = rng_canonical(g, k0, k1);
    if (dot(v, v) >= 1.0) continue;
    break;
  }
  return v;
}

// ------------------------------------------------------- closest hit: stream ---
struct Closest {
  double t;  // closest accepted root so far (the shrinking tmax of src/render.cpp:57-65)
  int prim;  // class-major primitive id, -1 = miss
};

// sphere_hit_helper up to the accepted root (src/common-model.cpp:70-81);
// the hit point and normal are computed once, for the winner only.
__device__ __forceinline__ void sphere_test(V3 o, V3 d, double a, double cx, double cy, double cz,
                                            double r2, int id, double tmin, Closest &best) {
  V3 oc = {o.x - cx, o.y - cy, o.z - cz};
  double h = dot(oc, d);
  double c = dot(oc, oc) - r2;
  double disc = h * h - a * c;
  if (disc >= 0.0) {
    double sq = sqrt(disc);
    double root = (-h - sq) / a;
    bool ok = true;
    if (root < tmin || root > best.t) {
      root = (-h + sq) / a;
      if (root < tmin || root > best.t) ok = false;
    }
    if (ok) {
      best.t = root;
      best.prim = id;
    }
  }
}

// Triangle::hit (src/common-model.cpp:103-125) with e1, e2, n precomputed
__device__ __forceinline__ void triangle_test(V3 o, V3 d, V3 A, V3 e1, V3 e2, V3 n, int id,
                                              double tmin, Closest &best) {
  double det = -dot(d, n);
  double invdet = 1.0 / det;
  V3 ao = o - A;
  V3 dao = cross(ao, d);
  double u = dot(e2, dao) * invdet;
  double v = -dot(e1, dao) * invdet;
  double t = dot(ao, n) * invdet;
  if (det >= 1e-6 && t >= tmin && t <= best.t && u >= 0.0 && v >= 0.0 && (u + v) <= 1.0) {
    best.t = t;
    best.prim = id;
  }
}

__device__ __forceinline__ Closest closest_hit_stream(const DevScene &sc, V3 o, V3 d, double time) {
  Closest best;
  best.t = __builtin_huge_val();  // tmax = +inf, src/render.cpp:34
  best.prim = -1;
  const double tmin = 0.001;  // src/render.cpp:33
  const double a = dot(d, d);
  {
    cdptr g = (cdptr)sc.sph;
    const int n = sc.n_sph;
#pragma unroll 4
    for (int i = 0; i < n; ++i) {
      sphere_test(o, d, a, g[4 * i + 0], g[4 * i + 1], g[4 * i + 2], g[4 * i + 3], i, tmin, best);
    }
  }
  {
    cdptr g = (cdptr)sc.mov;
    const int n = sc.n_mov;
    const int base = sc.n_sph;
#pragma unroll 2
    for (int i = 0; i < n; ++i) {
      // center(time) = c0 + time*(c1-c0), src/oo-primitives.h:64-66 with t0=0, t1=1
      double cx = g[8 * i + 0] + time * g[8 * i + 3];
      double cy = g[8 * i + 1] + time * g[8 * i + 4];
      double cz = g[8 * i + 2] + time * g[8 * i + 5];
      sphere_test(o, d, a, cx, cy, cz, g[8 * i + 6], base + i, tmin, best);
    }
  }
  {
    cdptr g = (cdptr)sc.tri;
    const int n = sc.n_tri;
    const int base = sc.n_sph + sc.n_mov;
#pragma unroll 2
    for (int i = 0; i < n; ++i) {
      V3 A = {g[12 * i + 0], g[12 * i + 1], g[12 * i + 2]};
      V3 e1 = {g[12 * i + 3], g[12 * i + 4], g[12 * i + 5]};
      V3 e2 = {g[12 * i + 6], g[12 * i + 7], g[12 * i + 8]};
      V3 nn = {g[12 * i + 9], g[12 * i + 10], g[12 * i + 11]};
      triangle_test(o, d, A, e1, e2, nn, base + i, tmin, best);
    }
  }
  return best;
}

// --------------------------------------------------------------- the kernel ---
__device__ __forceinline__ unsigned lane_id() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

template <int KERNEL>
__global__ void __launch_bounds__(256) RTOW_CAT(rtow_trace_, RTOW_SUFFIX)(const TraceParams P) {
  const DevScene &sc = P.sc;
  const uint32_t k0 = P.seed_lo, k1 = P.seed_hi;
  const unsigned lane = lane_id();
  const uint32_t lane_g = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t npix_local = (uint32_t)P.local_rows * (uint32_t)P.W;

  // per-lane state
  bool done = false;
  bool need_sample = true;
  int s_left = 0;             // samples left in the current item
  uint32_t item = 0xffffffffu;
  uint32_t j = 0;             // column
  uint32_t gi = 0;            // global row (from the top)
  V3 acc = {0.0, 0.0, 0.0};   // pixel_color of this item (src/render.cpp:156)
  V3 ro = {0, 0, 0}, rd = {0, 0, 1};
  double rtime = 0.0;
  int depth = 0;              // remaining child rays
  int nb = 0;                 // bounces recorded on the path stack
  Rng g = {0, 0, 0, 0, 0};
  uint32_t nseg = 0;

  for (;;) {
    // ---- item bookkeeping ---------------------------------------------------
    bool need_item = false;
    if (!done && need_sample && s_left <= 0) {
      if (item != 0xffffffffu) {
        double *dst = P.partials + (size_t)item * 3;
        dst[0] = acc.x;
        dst[1] = acc.y;
        dst[2] = acc.z;
      }
      need_item = true;
    }
    const unsigned long long need_mask = __ballot(need_item);
    if (need_mask != 0ull) {
      const int leader = __ffsll((long long)need_mask) - 1;
      unsigned long long base = 0ull;
      if ((int)lane == leader) base = atomicAdd(&P.counters[0], (unsigned long long)__popcll(need_mask));
      base = __shfl(base, leader);
      if (need_item) {
        const unsigned long long mine = base + (unsigned long long)__popcll(need_mask & ((1ull << lane) - 1ull));
        if (mine >= (unsigned long long)P.n_items) {
          done = true;
        } else {
          item = (uint32_t)mine;
          const uint32_t k = item / npix_local;     // stream
          const uint32_t lp = item - k * npix_local; // local pixel
          const uint32_t lr = lp / (uint32_t)P.W;
          j = lp - lr * (uint32_t)P.W;
          // local row -> global row: this rank's q-th strip is global strip q*nranks+rank
          const uint32_t q = lr / (uint32_t)P.tile_rows;
          const uint32_t rr = lr - q * (uint32_t)P.tile_rows;
          gi = (q * (uint32_t)P.nranks + (uint32_t)P.rank) * (uint32_t)P.tile_rows + rr;
          g.pixel = gi * (uint32_t)P.W + j;
          g.sample = k * (uint32_t)P.spt;  // first sample index of this stream
          s_left = P.spt;
          acc = {0.0, 0.0, 0.0};
        }
      }
    }
    if (__ballot(!done) == 0ull) break;

    // (a lane whose fresh item has no samples — spt == 0 — goes straight back for the next one)
    if (!done && !(need_sample && s_left <= 0)) {
      // ---- new sample: pixel jitter + Camera::get_ray ------------------------
      if (need_sample) {
        g.d = 0u;
        // src/render.cpp:158-159
        const int from_top_i = P.H - (int)gi - 1;
        const double u = ((double)(int)j + rng_canonical(g, k0, k1)) / (double)(P.W - 1);
        const double v = ((double)from_top_i + rng_canonical(g, k0, k1)) / (double)(P.H - 1);
        // src/common-model.cpp:156-167; disk sample: y draws first (random-utils.cpp:36)
        double px, py;
        for (;;) {
          py = rng_range(g, k0, k1, -1.0, 1.0);
          px = rng_range(g, k0, k1, -1.0, 1.0);
          if (px * px + py * py + 0.0 * 0.0 >= 1.0) continue;
          break;
        }
        const double rdx = P.cam.lens_radius * px, rdy = P.cam.lens_radius * py;
        const V3 offset = ld3(P.cam.u) * rdx + ld3(P.cam.v) * rdy;
        const V3 from = ld3(P.cam.origin) + offset;
        rd = ld3(P.cam.llc) + u * ld3(P.cam.horizontal) + v * ld3(P.cam.vertical) - from;
        ro = from;
        rtime = rng_range(g, k0, k1, P.cam.t0, P.cam.t1);
        depth = P.max_child_rays;
        nb = 0;
        need_sample = false;
      }

      // ---- one ray segment: closest hit --------------------------------------
      ++nseg;
      const Closest best = closest_hit_stream(sc, ro, rd, rtime);

      if (best.prim >= 0) {
        if (depth <= 0) {
          need_sample = true;  // src/render.cpp:115: black
        } else {
          // rebuild the Hit of the winner (src/common-model.cpp:83-90, :121)
          V3 where = ro + rd * best.t;
          V3 normal;
          bool front = true;
          const int pid = best.prim;
          if (pid < sc.n_sph + sc.n_mov) {
            V3 center;
            double radius;
            if (pid < sc.n_sph) {
              const double *q = sc.sph + 4 * (size_t)pid;
              center = {q[0], q[1], q[2]};
              radius = sc.sph_r[pid];
            } else {
              const double *q = sc.mov + 8 * (size_t)(pid - sc.n_sph);
              center = {q[0] + rtime * q[3], q[1] + rtime * q[4], q[2] + rtime * q[5]};
              radius = q[7];
            }
            normal = normalize(where - center);
            front = (dot(rd, normal) < 0.0) ^ (radius < 0.0);
            normal = front ? normal : -normal;
          } else {
            const double *q = sc.tri + 12 * (size_t)(pid - sc.n_sph - sc.n_mov);
            normal = {q[9], q[10], q[11]};
          }
          const int mi = sc.prim_mat[pid];
          const DevMaterial *m = sc.mats + mi;
          const int kind = m->kind;

          // ---- Material::scatter (src/common-model.cpp:13-62) ------------------
          V3 dirbase = {0.0, 0.0, 0.0};
          if (kind == 2) {
            const double ir = m->ir;
            const V3 unit = normalize(rd);
            const double cos_theta = dot(-unit, normal);
            const double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
            const double ratio = front ? (1.0 / ir) : ir;
            bool refl = ratio * sin_theta > 1.0;
            if (!refl) {
              double r0 = (1.0 - ratio) / (1.0 + ratio);
              r0 = r0 * r0;
              const double x = 1.0 - cos_theta;
              const double x2 = x * x;
              const double R = r0 + (1.0 - r0) * (x2 * x2 * x);
              refl = R > rng_canonical(g, k0, k1);
            }
            dirbase = refl ? reflect(unit, normal) : refract(unit, normal, ratio);
          } else if (kind == 1) {
            dirbase = reflect(rd, normal);
          }
          const V3 rnd = rng_unit_vector(g, k0, k1);
          V3 dir;
          bool absorbed = false;
          if (kind == 0) {
            absorbed = fabs(normal.x - rnd.x) < 1e-8 && fabs(normal.y - rnd.y) < 1e-8 &&
                       fabs(normal.z - rnd.z) < 1e-8;
            dir = normal + rnd;
          } else {
            dir = dirbase + m->fuzz * rnd;
          }
          if (absorbed) {
            need_sample = true;  // src/render.cpp:120: black
          } else {
            P.stack[(size_t)nb * P.n_lanes + lane_g] = (uint32_t)mi;
            ++nb;
            --depth;
            ro = where;
            rd = dir;
          }
        }
      } else {
        // ---- background + unwind of the recursion (src/render.cpp:119,122-128) --
        const V3 unit = normalize(rd);
        const double t = 0.5 * (unit.y + +1.0);
        V3 c = (1.0 - t) * V3{1.0, 1.0, 1.0} + t * V3{0.5, 0.7, 1.0};
        for (int q = nb - 1; q >= 0; --q) {
          const uint32_t mi = P.stack[(size_t)q * P.n_lanes + lane_g];
          const DevMaterial *m = sc.mats + mi;
          c = V3{m->att[0], m->att[1], m->att[2]} * c;
        }
        acc = acc + c;  // pixel_color += ray_color(...)
        need_sample = true;
      }
      if (need_sample) {
        --s_left;
        ++g.sample;
      }
    }
  }

  // stats: one atomic per wave
  unsigned long long tot = nseg;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) tot += __shfl_down(tot, off);
  if (lane == 0) atomicAdd(&P.counters[1], tot);
}

}  // namespace

int RTOW_CAT(launch_trace_, RTOW_SUFFIX)(const TraceParams &p, int kernel, int grid, int block,
                                         void *stream) {
  (void)kernel;
  hipLaunchKernelGGL((RTOW_CAT(rtow_trace_, RTOW_SUFFIX) < 1 >), dim3(grid), dim3(block), 0,
                     (hipStream_t)stream, p);
  return (int)hipGetLastError();
}

int RTOW_CAT(trace_occupancy_, RTOW_SUFFIX)(int kernel, int block) {
  (void)kernel;
  int nb = 0;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(
      &nb, RTOW_CAT(rtow_trace_, RTOW_SUFFIX) < 1 >, block, 0);
  if (e != hipSuccess) return -1;
  return nb;
}

}  // namespace rtow
