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
//     the same trip, so the closest-hit step always runs on a full wave.
//   * closest hit, two strategies with identical results:
//       STREAM — every lane tests every primitive.  The primitive index is
//         wave-uniform, so each record is fetched with ONE scalar load into SGPRs and
//         used directly as a VALU operand: the scene costs no VGPRs, no LDS traffic.
//       BVH — every lane walks a threaded (stackless, skip-link) BVH whose nodes and
//         primitive records sit in LDS (one scene image per workgroup, staged with
//         coalesced 16-byte loads).  Node boxes are f32 and padded, and the slab test
//         is slackened, so culling is conservative; leaf primitives are tested with
//         the same f64 code as STREAM, so the accepted (t, primitive) is the same.
//         Leaves found during the walk are queued per lane and tested in a separate
//         phase, so the box loop and the primitive loop are each SIMT-dense.
//   * radiance.  The reference multiplies attenuations on the way back up the
//     recursion, a1*(a2*(...*(an*sky))).  To reproduce that order bit for bit the
//     lane records the material index of every bounce in a per-lane path stack in
//     HBM ([bounce][lane], coalesced) and folds it from the end when the path
//     escapes to the sky.  A path that ends black contributes an exact zero.
//   * RNG: Philox4x32-7 in REQUESTS, one block each, counter (request, sample, pixel,
//     0), key = seed: pixel jitter + shutter time share a block (42 bits each); every
//     disk candidate takes one (two doubles, each from two words like libstdc++'s
//     generate_canonical<double,53>, src/random-utils.cpp:11-13); every unit-ball
//     candidate takes one (32 bits per coordinate) and the dielectric coin rides in the
//     spare word of the bounce's first candidate.  Whole blocks per request keep the
//     rejection loops free of per-lane parity divergence; Philox is ~1/4 of the
//     kernel's VALU time, so blocks are not wasted.
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

// Arithmetic type of rays, hit tests on small primitives and shading.  binary64 (the
// reference's type) in the strict and fast builds; binary32 in the f32 build
// (rtow_trace_f32.hip), where the always-test large primitives and all spheres met by the
// STREAM/BVH kernels are still tested in binary64 (an r = 1000 sphere cancels catastrophically
// in binary32: SURVEY.md §7 "fp32 robustness") and pixel sums stay binary64.
#ifdef RTOW_REAL_F32
typedef float real;
#else
typedef double real;
#endif

template <class T>
struct Vec3 {
  T x, y, z;
};
typedef Vec3<real> V3;
typedef Vec3<double> V3d;
template <class T>
__device__ __forceinline__ Vec3<T> operator+(Vec3<T> a, Vec3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class T>
__device__ __forceinline__ Vec3<T> operator-(Vec3<T> a, Vec3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class T>
__device__ __forceinline__ Vec3<T> operator-(Vec3<T> a) { return {-a.x, -a.y, -a.z}; }
template <class T>
__device__ __forceinline__ Vec3<T> operator*(Vec3<T> a, Vec3<T> b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
template <class T>
__device__ __forceinline__ Vec3<T> operator*(Vec3<T> a, T s) { return {a.x * s, a.y * s, a.z * s}; }
template <class T>
__device__ __forceinline__ Vec3<T> operator*(T s, Vec3<T> a) { return {s * a.x, s * a.y, s * a.z}; }
// glm: dot = x*x' + y*y' + z*z' (left to right); cross, normalize (v * 1/sqrt),
// reflect (I - N*dot(N,I)*2), refract — same definitions as oracle/rtow_oracle.cpp.
template <class T>
__device__ __forceinline__ T dot(Vec3<T> a, Vec3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class T>
__device__ __forceinline__ Vec3<T> cross(Vec3<T> x, Vec3<T> y) {
  return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y};
}
__device__ __forceinline__ V3d to_f64(Vec3<float> v) { return {(double)v.x, (double)v.y, (double)v.z}; }
__device__ __forceinline__ V3d to_f64(V3d v) { return v; }
#ifdef RTOW_FAST_MATH
// fast build: hardware reciprocal-square-root seed (~2^-26) + two Newton steps instead of the
// correctly rounded sqrt and division (relative error ~1e-16; the strict build keeps IEEE forms)
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - 0.5 * x * y * y);
  y = y * (1.5 - 0.5 * x * y * y);
  return y;
}
__device__ __forceinline__ double fast_sqrt(double x) { return x > 0.0 ? x * fast_rsqrt(x) : 0.0; }
__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = y * (2.0 - x * y);
  y = y * (2.0 - x * y);
  return y;
}
__device__ __forceinline__ double fast_div(double n, double d) { return n * fast_rcp(d); }
#else
__device__ __forceinline__ double fast_div(double n, double d) { return n / d; }
__device__ __forceinline__ double fast_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ double fast_rcp(double x) { return 1.0 / x; }
__device__ __forceinline__ double fast_rsqrt(double x) { return 1.0 / sqrt(x); }
#endif
// binary32 forms (f32 build only): hardware rsq/rcp/sqrt (1 ulp) + one Newton step where it is cheap
__device__ __forceinline__ float fast_rsqrt(float x) {
  float y = __builtin_amdgcn_rsqf(x);
  return y * (1.5f - 0.5f * x * y * y);
}
__device__ __forceinline__ float fast_sqrt(float x) { return x > 0.0f ? __builtin_amdgcn_sqrtf(x) : 0.0f; }
__device__ __forceinline__ float fast_rcp(float x) {
  float y = __builtin_amdgcn_rcpf(x);
  return y * (2.0f - x * y);
}
__device__ __forceinline__ float fast_div(float n, float d) { return n * fast_rcp(d); }
template <class T>
__device__ __forceinline__ Vec3<T> normalize(Vec3<T> v) { return v * fast_rsqrt(dot(v, v)); }
template <class T>
__device__ __forceinline__ Vec3<T> reflect(Vec3<T> I, Vec3<T> N) { return I - N * dot(N, I) * T(2.0); }
template <class T>
__device__ __forceinline__ Vec3<T> refract(Vec3<T> I, Vec3<T> N, T eta) {
  T d = dot(N, I);
  T k = T(1.0) - eta * eta * (T(1.0) - d * d);
  if (k >= T(0.0)) return eta * I - (eta * d + fast_sqrt(k)) * N;
  return {T(0.0), T(0.0), T(0.0)};
}

// ------------------------------------------------------------------ Philox ---
struct Rng {
  uint32_t pixel, sample, r;  // r = next request index of this sample
};

// Philox4x32-7: the fastest member of the family reported Crush-resistant (Salmon et al.,
// SC'11); oracle/ uses the same round count (its tests pin the round function with the
// published 10-round known answers).
constexpr int kPhiloxRounds = 7;
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                           uint32_t k0, uint32_t k1, uint32_t &o0, uint32_t &o1,
                                           uint32_t &o2, uint32_t &o3) {
#pragma unroll
  for (int r = 0; r < kPhiloxRounds; ++r) {
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

// One request = one block (w0..w3); see oracle/rtow_oracle.cpp, struct PhiloxDraw.
// jitter + shutter time: 42 bits each (word k + 10 bits of word 3)
__device__ __forceinline__ void jitter_from_block(uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3, double &u,
                                                  double &v, double &t) {
  const double s42 = 0x1p-42;
  u = ((double)o0 + (double)(o3 & 1023u) * 4294967296.0) * s42;
  v = ((double)o1 + (double)((o3 >> 10) & 1023u) * 4294967296.0) * s42;
  t = ((double)o2 + (double)((o3 >> 20) & 1023u) * 4294967296.0) * s42;
}
// binary32 build: the top 24 bits of the same 42-bit values (truncated, so < 1 and within one
// binary32 ulp of the binary64 build's value: both builds sample the same lens/pixel positions)
__device__ __forceinline__ void jitter_from_block(uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3, float &u,
                                                  float &v, float &t) {
  u = (float)(((o3 & 1023u) << 14) | (o0 >> 18)) * 0x1p-24f;
  v = (float)((((o3 >> 10) & 1023u) << 14) | (o1 >> 18)) * 0x1p-24f;
  t = (float)((((o3 >> 20) & 1023u) << 14) | (o2 >> 18)) * 0x1p-24f;
}
__device__ __forceinline__ void rng_jitter(Rng &g, uint32_t k0, uint32_t k1, real &u, real &v, real &t) {
  uint32_t o0, o1, o2, o3;
  philox4x32(g.r, g.sample, g.pixel, 0u, k0, k1, o0, o1, o2, o3);
  g.r += 1u;
  jitter_from_block(o0, o1, o2, o3, u, v, t);
}
// disk candidate: two doubles, each from two words like the reference's doubles
__device__ __forceinline__ void rng_disk(Rng &g, uint32_t k0, uint32_t k1, real &a, real &b) {
  uint32_t o0, o1, o2, o3;
  philox4x32(g.r, g.sample, g.pixel, 0u, k0, k1, o0, o1, o2, o3);
  g.r += 1u;
#ifdef RTOW_REAL_F32
  a = (float)o1 * 0x1p-32f;  // the high words of the two doubles
  b = (float)o3 * 0x1p-32f;
#else
  a = canonical_from_words(o0, o1);
  b = canonical_from_words(o2, o3);
#endif
}
// unit-ball candidate (32 bits per coordinate); the spare word is the dielectric coin of
// the bounce when this is its first candidate
__device__ __forceinline__ V3 rng_scatter(Rng &g, uint32_t k0, uint32_t k1, real &coin) {
  uint32_t o0, o1, o2, o3;
  philox4x32(g.r, g.sample, g.pixel, 0u, k0, k1, o0, o1, o2, o3);
  g.r += 1u;
  const real s32 = real(0x1p-32);
  coin = (real)o3 * s32;
  return V3{(real)o0 * s32, (real)o1 * s32, (real)o2 * s32};
}

// -------------------------------------------------------- primitive hit tests ---
struct Closest {
  real t;    // closest accepted root so far (the shrinking tmax of src/render.cpp:57-65)
  int prim;  // class-major primitive id, -1 = miss
};

// sphere_hit_helper up to the accepted root (src/common-model.cpp:70-81);
// the hit point and normal are computed once, for the winner only.
// `inv_a` is 1/a, used only by the fast build (one reciprocal per ray instead of two
// divisions per candidate hit); the strict build divides like the reference.
// T is the arithmetic of the test: `real`, or double for large primitives in the f32 build.
__device__ __forceinline__ double rabs(double x) { return fabs(x); }
__device__ __forceinline__ float rabs(float x) { return fabsf(x); }

template <class T>
__device__ __forceinline__ T sphere_disc(Vec3<T> o, Vec3<T> d, T a, T cx, T cy, T cz, T r2, T &h) {
  Vec3<T> oc = {o.x - cx, o.y - cy, o.z - cz};
  h = dot(oc, d);
  T c = dot(oc, oc) - rabs(r2);  // r2 carries the radius' sign (see rtow_capi.cpp)
  return h * h - a * c;
}

template <class T>
__device__ __forceinline__ void sphere_resolve(T disc, T h, T a, T inv_a, int id, T tmin, Closest &best) {
  if (disc >= T(0.0)) {
    T sq = fast_sqrt(disc);
#if defined(RTOW_FAST_MATH)
    T root = (-h - sq) * inv_a;
    const T root2 = (-h + sq) * inv_a;
#else
    (void)inv_a;
    T root = (-h - sq) / a;
#endif
    bool ok = true;
    if (root < tmin || root > (T)best.t) {
#if defined(RTOW_FAST_MATH)
      root = root2;
#else
      root = (-h + sq) / a;
#endif
      if (root < tmin || root > (T)best.t) ok = false;
    }
    if (ok) {
      best.t = (real)root;
      best.prim = id;
    }
  }
}

template <class T>
__device__ __forceinline__ void sphere_test(Vec3<T> o, Vec3<T> d, T a, T inv_a, T cx, T cy, T cz, T r2, int id,
                                            T tmin, Closest &best) {
  T h;
  const T disc = sphere_disc(o, d, a, cx, cy, cz, r2, h);
  sphere_resolve(disc, h, a, inv_a, id, tmin, best);
}

// Triangle::hit (src/common-model.cpp:103-125) with e1, e2, n precomputed
template <class T>
__device__ __forceinline__ void triangle_test(Vec3<T> o, Vec3<T> d, Vec3<T> A, Vec3<T> e1, Vec3<T> e2, Vec3<T> n,
                                              int id, T tmin, Closest &best) {
  T det = -dot(d, n);
  T invdet = fast_rcp(det);  // strict build: 1.0 / det
  Vec3<T> ao = o - A;
  Vec3<T> dao = cross(ao, d);
  T u = dot(e2, dao) * invdet;
  T v = -dot(e1, dao) * invdet;
  T t = dot(ao, n) * invdet;
  if (det >= T(1e-6) && t >= tmin && t <= (T)best.t && u >= T(0.0) && v >= T(0.0) && (u + v) <= T(1.0)) {
    best.t = (real)t;
    best.prim = id;
  }
}

#define RTOW_TMIN 0.001  // src/render.cpp:33

// ------------------------------------------------------- closest hit: STREAM ---
// Always binary64 (in the f32 build the ray is widened once per segment): this kernel is for
// scenes of <= 16 primitives, which include the r = 1000 ground sphere.
__device__ __forceinline__ Closest closest_hit_stream(const DevScene &sc, V3d o, V3d d, double time) {
  Closest best;
  best.t = (real)__builtin_huge_val();  // tmax = +inf, src/render.cpp:34
  best.prim = -1;
  const double tmin = RTOW_TMIN;
  const double a = dot(d, d);
  const double inv_a = fast_rcp(a);  // used by the fast build only
  {
    cdptr g = (cdptr)sc.sph;
    const int n = sc.n_sph;
#pragma unroll 4
    for (int i = 0; i < n; ++i) {
      sphere_test<double>(o, d, a, inv_a, g[4 * i + 0], g[4 * i + 1], g[4 * i + 2], g[4 * i + 3], i, tmin, best);
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
      sphere_test<double>(o, d, a, inv_a, cx, cy, cz, g[8 * i + 6], base + i, tmin, best);
    }
  }
  {
    cdptr g = (cdptr)sc.tri;
    const int n = sc.n_tri;
    const int base = sc.n_sph + sc.n_mov;
#pragma unroll 2
    for (int i = 0; i < n; ++i) {
      V3d A = {g[12 * i + 0], g[12 * i + 1], g[12 * i + 2]};
      V3d e1 = {g[12 * i + 3], g[12 * i + 4], g[12 * i + 5]};
      V3d e2 = {g[12 * i + 6], g[12 * i + 7], g[12 * i + 8]};
      V3d nn = {g[12 * i + 9], g[12 * i + 10], g[12 * i + 11]};
      triangle_test<double>(o, d, A, e1, e2, nn, base + i, tmin, best);
    }
  }
  return best;
}

// Diagnostic region stamps (STAMPS build only, never in a timed run): wave cycles per
// region of the main loop, accumulated in scalar registers and added to
// counters[8 + region] once per wave.  Shares, not absolute times (each stamp drains
// the wave's outstanding memory operations).
enum { RG_FETCH = 0, RG_REGEN, RG_WALK, RG_SHADE, RG_LEAF, RG_COUNT };
template <bool ON>
struct Stamps {
  unsigned long long t[RG_COUNT] = {0, 0, 0, 0, 0};
  unsigned long long last = 0;
  unsigned long long iters = 0, trips = 0, phases = 0;  // wave-level loop counts
  __device__ __forceinline__ void start() {
    if constexpr (ON) last = now();
  }
  __device__ __forceinline__ void mark(int region) {
    if constexpr (ON) {
      const unsigned long long n = now();
      t[region] += n - last;
      last = n;
    }
  }
  static __device__ __forceinline__ unsigned long long now() {
    unsigned long long v;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return v;
  }
};

// ---------------------------------------------------------- closest hit: BVH ---
extern __shared__ __align__(16) unsigned char rtow_lds[];

// Scene image reader: LDS copy (ds_read_b128/b64) or the global blob (L1/L2).
template <bool LDS>
struct Image {
  const unsigned char *g;
  __device__ __forceinline__ float4 f4(uint32_t off) const {
    if constexpr (LDS)
      return *reinterpret_cast<const float4 *>(rtow_lds + off);
    else
      return *reinterpret_cast<const float4 *>(g + off);
  }
  __device__ __forceinline__ double2 d2(uint32_t off) const {
    if constexpr (LDS)
      return *reinterpret_cast<const double2 *>(rtow_lds + off);
    else
      return *reinterpret_cast<const double2 *>(g + off);
  }
  __device__ __forceinline__ uint32_t u32(uint32_t off) const {
    if constexpr (LDS)
      return *reinterpret_cast<const uint32_t *>(rtow_lds + off);
    else
      return *reinterpret_cast<const uint32_t *>(g + off);
  }
};

__device__ __forceinline__ float round_up_f32(double t) { return __double2float_ru(t); }
__device__ __forceinline__ float round_up_f32(float t) { return t; }

__device__ __forceinline__ float safe_inv(float d) {
  // axis-parallel rays: a huge finite reciprocal keeps the fma slab form free of NaNs
  const float big = 1e30f;
  return fabsf(d) < 1e-30f ? (__builtin_signbitf(d) ? -big : big) : 1.0f / d;
}

struct ImgOffsets {  // byte offsets of the id and record sections inside a scene image
  uint32_t ids, sph, mov, tri;
  uint32_t sph32, mov32;  // f32 build: binary32 copies of the sphere records (grid cells); `tri` then
                          // points at binary32 triangle records (48 B), the only triangle section
};

// The ray of one segment in the forms the tests need.  In the binary64 builds the two forms
// are the same values; in the f32 build the ray is widened once per segment for the tests
// that must run in binary64.
struct RayForms {
  V3 o, d;
  real a, inv_a, time;
  V3d o64, d64;
  double a64, inv_a64, time64;
};
__device__ __forceinline__ RayForms make_ray_forms(V3 o, V3 d, real time) {
  RayForms r;
  r.o = o;
  r.d = d;
  r.time = time;
  r.a = dot(d, d);
  r.inv_a = fast_rcp(r.a);  // used by the fast builds only
  r.o64 = to_f64(o);
  r.d64 = to_f64(d);
  r.time64 = (double)time;
#ifdef RTOW_REAL_F32
  r.a64 = dot(r.d64, r.d64);
  r.inv_a64 = fast_rcp(r.a64);
#else
  r.a64 = r.a;
  r.inv_a64 = r.inv_a;
#endif
  return r;
}

// Tests primitives ids[first .. first+count) of a scene image against the ray (the same code
// as the STREAM kernel, so the accepted (t, primitive) is the same).  SMALL: the spheres are
// grid-cell members, tested in binary32 by the f32 build (no effect in the binary64 builds).
template <bool LDS, bool SMALL>
__device__ __forceinline__ void leaf_test(const Image<LDS> &im, const DevScene &sc, ImgOffsets off,
                                          uint32_t first, uint32_t count, const RayForms &ray, Closest &best,
                                          uint32_t &nprim, int &last_id) {
  for (uint32_t k = 0; k < count; ++k) {
    const int id = (int)im.u32(off.ids + 4u * (first + k));
    // one-entry mailbox: a primitive spanning adjacent grid cells is listed in each of them
    if (id == last_id) continue;
    last_id = id;
    ++nprim;
    if (id < sc.n_sph) {
#ifdef RTOW_REAL_F32
      if constexpr (SMALL) {
        const float4 p = im.f4(off.sph32 + 16u * (uint32_t)id);
        sphere_test<float>(ray.o, ray.d, ray.a, ray.inv_a, p.x, p.y, p.z, p.w, id, (float)RTOW_TMIN, best);
        continue;
      }
#endif
      const uint32_t r = off.sph + 32u * (uint32_t)id;
      const double2 p0 = im.d2(r), p1 = im.d2(r + 16u);
      sphere_test<double>(ray.o64, ray.d64, ray.a64, ray.inv_a64, p0.x, p0.y, p1.x, p1.y, id, RTOW_TMIN, best);
    } else if (id < sc.n_sph + sc.n_mov) {
#ifdef RTOW_REAL_F32
      if constexpr (SMALL) {
        const uint32_t r = off.mov32 + 32u * (uint32_t)(id - sc.n_sph);
        const float4 p0 = im.f4(r), p1 = im.f4(r + 16u);  // c0xyz dx | dy dz r2 -
        sphere_test<float>(ray.o, ray.d, ray.a, ray.inv_a, p0.x + ray.time * p0.w, p0.y + ray.time * p1.x,
                           p0.z + ray.time * p1.y, p1.z, id, (float)RTOW_TMIN, best);
        continue;
      }
#endif
      const uint32_t r = off.mov + 64u * (uint32_t)(id - sc.n_sph);
      const double2 p0 = im.d2(r), p1 = im.d2(r + 16u), p2 = im.d2(r + 32u), p3 = im.d2(r + 48u);
      const double cx = p0.x + ray.time64 * p1.y;
      const double cy = p0.y + ray.time64 * p2.x;
      const double cz = p1.x + ray.time64 * p2.y;
      sphere_test<double>(ray.o64, ray.d64, ray.a64, ray.inv_a64, cx, cy, cz, p3.x, id, RTOW_TMIN, best);
    } else {
#ifdef RTOW_REAL_F32
      const uint32_t r = off.tri + 48u * (uint32_t)(id - sc.n_sph - sc.n_mov);
      const float4 q0 = im.f4(r), q1 = im.f4(r + 16u), q2 = im.f4(r + 32u);  // A e1 | e1 e2 | e2 n
      triangle_test<float>(ray.o, ray.d, V3{q0.x, q0.y, q0.z}, V3{q0.w, q1.x, q1.y}, V3{q1.z, q1.w, q2.x},
                           V3{q2.y, q2.z, q2.w}, id, (float)RTOW_TMIN, best);
#else
      const uint32_t r = off.tri + 96u * (uint32_t)(id - sc.n_sph - sc.n_mov);
      const double2 q0 = im.d2(r), q1 = im.d2(r + 16u), q2 = im.d2(r + 32u), q3 = im.d2(r + 48u),
                    q4 = im.d2(r + 64u), q5 = im.d2(r + 80u);
      triangle_test<double>(ray.o64, ray.d64, V3d{q0.x, q0.y, q1.x}, V3d{q1.y, q2.x, q2.y}, V3d{q3.x, q3.y, q4.x},
                            V3d{q4.y, q5.x, q5.y}, id, RTOW_TMIN, best);
#endif
    }
  }
}

template <bool LDS, bool ST>
__device__ __forceinline__ Closest closest_hit_bvh(const Image<LDS> &im, const DevScene &sc, V3 o,
                                                   V3 d, real time, bool active, uint32_t &nnode,
                                                   uint32_t &nprim, Stamps<ST> &stamps) {
  Closest best;
  best.t = (real)__builtin_huge_val();
  best.prim = -1;
  const RayForms ray = make_ray_forms(o, d, time);
  // f32 copy of the ray for the (conservative) box tests
  const float ix = safe_inv((float)d.x), iy = safe_inv((float)d.y), iz = safe_inv((float)d.z);
  const float oix = (float)o.x * ix, oiy = (float)o.y * iy, oiz = (float)o.z * iz;
  const float tmin32 = 0.0009f;   // < RTOW_TMIN
  const float slack = 1.00002f;   // relative slack on the far side of the interval
  float tmax32 = __builtin_huge_valf();
  const uint32_t END = (uint32_t)sc.n_nodes;  // skip links past the last node point here
  const ImgOffsets off = {sc.off_ids, sc.off_sph, sc.off_mov, sc.off_tri, sc.off_sph32, sc.off_mov32};
  int last_id = -1;
  uint32_t node = active ? 0u : END;  // the walk uses wave votes: idle lanes enter with nothing to do
  uint32_t q0 = 0u, q1 = 0u, q2 = 0u, q3 = 0u;  // queued leaves (0 = empty), oldest first
  // Termination: every link of the image points forward (node+1 or skip > node, checked
  // by the host at upload) and the walk stops at any index >= END, so a lane takes at
  // most n_nodes steps.  (A per-trip guard counter here cost 7 % of the kernel.)
  for (;;) {
    if constexpr (ST) stamps.iters += 1;
    if (node < END) {
      const float4 r0 = im.f4(node * 32u), r1 = im.f4(node * 32u + 16u);
      ++nnode;
      const float ax = fmaf(r0.x, ix, -oix), bx = fmaf(r0.w, ix, -oix);
      const float ay = fmaf(r0.y, iy, -oiy), by = fmaf(r1.x, iy, -oiy);
      const float az = fmaf(r0.z, iz, -oiz), bz = fmaf(r1.y, iz, -oiz);
      const float tnear = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), tmin32));
      const float tfar = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax32));
      const bool hit = tnear <= tfar * slack;
      const uint32_t skip = __float_as_uint(r1.z), leaf = __float_as_uint(r1.w);
      if (hit && leaf != 0u) {
        if (q0 == 0u)
          q0 = leaf;
        else if (q1 == 0u)
          q1 = leaf;
        else if (q2 == 0u)
          q2 = leaf;
        else
          q3 = leaf;
      }
      node = (hit && leaf == 0u) ? node + 1u : skip;
    }
    const bool any_walking = __any(node < END);
    if (__any(q3 != 0u) || !any_walking) {
      stamps.mark(RG_WALK);
      if constexpr (ST) stamps.phases += 1;
      // leaf phase: every lane tests the primitives of the OLDEST leaf it queued (most
      // lanes hold one; only the lanes whose queue filled hold two, and theirs moves up)
      if (q0 != 0u) leaf_test<LDS, false>(im, sc, off, q0 >> 3, q0 & 7u, ray, best, nprim, last_id);
      q0 = q1;
      q1 = q2;
      q2 = q3;
      q3 = 0u;
      // shrink the f32 interval (rounded up: never below the f64 value)
      tmax32 = round_up_f32(best.t);
      stamps.mark(RG_LEAF);
      if (!any_walking && !__any(q0 != 0u)) break;
    }
  }
  return best;
}

// --------------------------------------------------------- closest hit: GRID ---
// 3D-DDA over the uniform grid of rtow_grid.h.  Primitives far larger than the rest (the
// ground sphere) are not in the grid; every ray tests that short list first.  Cells are
// visited in order along the ray; a non-empty cell is queued like a BVH leaf and tested in
// the SIMT-dense leaf phase.  The walk ends when the ray leaves the grid (integer cell
// counters, so at most nx+ny+nz steps whatever the floats do) or when the exit distance of
// the current cell is beyond the closest hit so far.
template <bool LDS, bool ST>
__device__ __forceinline__ Closest closest_hit_grid(const Image<LDS> &im, const DevScene &sc, V3 o,
                                                    V3 d, real time, bool active, uint32_t &nnode,
                                                    uint32_t &nprim, Stamps<ST> &stamps) {
  Closest best;
  best.t = (real)__builtin_huge_val();
  best.prim = -1;
  const RayForms ray = make_ray_forms(o, d, time);
  const ImgOffsets off = {sc.g_off_ids, sc.g_off_sph, sc.g_off_mov, sc.g_off_tri, sc.g_off_sph32, sc.g_off_mov32};
  int last_id = -1;
  // header: wave-uniform scalar loads from the global copy of the image
  const RTOW_CONST float *hf = (const RTOW_CONST float *)sc.gblob;
  const RTOW_CONST int32_t *hi = (const RTOW_CONST int32_t *)sc.gblob;
  const float gx = hf[0], gy = hf[1], gz = hf[2];
  const float cx = hf[3], cy = hf[4], cz = hf[5];
  const float icx = hf[6], icy = hf[7], icz = hf[8];
  const int nx = hi[9], ny = hi[10], nz = hi[11];
  const uint32_t n_large = (uint32_t)hi[12], off_large = (uint32_t)hi[13];

  // the large primitives, for every ray.  Static spheres are taken four (then two) at a time:
  // all records are loaded and all discriminants computed before any hit branch, so LDS
  // latency and the f64 dependency chains of one test overlap the others.
  if (active && n_large != 0u) {
    const uint32_t lf = (off_large - off.ids) >> 2;
    uint32_t k = 0;
    for (; k + 3 < n_large; k += 4) {
      int id[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) id[j] = (int)im.u32(off.ids + 4u * (lf + k + j));
      if (id[0] < sc.n_sph && id[1] < sc.n_sph && id[2] < sc.n_sph && id[3] < sc.n_sph) {
        double dd[4], hh[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t r = off.sph + 32u * (uint32_t)id[j];
          const double2 p0 = im.d2(r), p1 = im.d2(r + 16u);
          dd[j] = sphere_disc<double>(ray.o64, ray.d64, ray.a64, p0.x, p0.y, p1.x, p1.y, hh[j]);
        }
        nprim += 4u;
#pragma unroll
        for (int j = 0; j < 4; ++j) sphere_resolve<double>(dd[j], hh[j], ray.a64, ray.inv_a64, id[j], RTOW_TMIN, best);
        last_id = id[3];
      } else {
        leaf_test<LDS, false>(im, sc, off, lf + k, 4u, ray, best, nprim, last_id);
      }
    }
    for (; k + 1 < n_large; k += 2) {
      const int ia = (int)im.u32(off.ids + 4u * (lf + k)), ib = (int)im.u32(off.ids + 4u * (lf + k + 1));
      if (ia < sc.n_sph && ib < sc.n_sph) {
        const uint32_t ra = off.sph + 32u * (uint32_t)ia, rb = off.sph + 32u * (uint32_t)ib;
        const double2 a0 = im.d2(ra), a1 = im.d2(ra + 16u), b0 = im.d2(rb), b1 = im.d2(rb + 16u);
        double ha, hb;
        const double da = sphere_disc<double>(ray.o64, ray.d64, ray.a64, a0.x, a0.y, a1.x, a1.y, ha);
        const double db = sphere_disc<double>(ray.o64, ray.d64, ray.a64, b0.x, b0.y, b1.x, b1.y, hb);
        nprim += 2u;
        sphere_resolve<double>(da, ha, ray.a64, ray.inv_a64, ia, RTOW_TMIN, best);
        sphere_resolve<double>(db, hb, ray.a64, ray.inv_a64, ib, RTOW_TMIN, best);
        last_id = ib;
      } else {
        leaf_test<LDS, false>(im, sc, off, lf + k, 2u, ray, best, nprim, last_id);
      }
    }
    if (k < n_large) leaf_test<LDS, false>(im, sc, off, lf + k, n_large - k, ray, best, nprim, last_id);
  }
  float tmax32 = round_up_f32(best.t);

  // clip the ray to the grid bounds (f32, conservative by the padding of rtow_grid.h)
  const float dx = (float)d.x, dy = (float)d.y, dz = (float)d.z;
  const float ox = (float)o.x, oy = (float)o.y, oz = (float)o.z;
  const float ix = safe_inv(dx), iy = safe_inv(dy), iz = safe_inv(dz);
  const float oix = ox * ix, oiy = oy * iy, oiz = oz * iz;
  const float hx = fmaf((float)nx, cx, gx), hy = fmaf((float)ny, cy, gy), hz = fmaf((float)nz, cz, gz);
  const float ax = fmaf(gx, ix, -oix), bx = fmaf(hx, ix, -oix);
  const float ay = fmaf(gy, iy, -oiy), by = fmaf(hy, iy, -oiy);
  const float az = fmaf(gz, iz, -oiz), bz = fmaf(hz, iz, -oiz);
  const float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0009f));
  const float t1 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax32));
  bool walking = active && t0 <= t1 * 1.00002f;

  // starting cell and DDA state
  const float px = fmaf(t0, dx, ox), py = fmaf(t0, dy, oy), pz = fmaf(t0, dz, oz);
  int c0 = (int)floorf((px - gx) * icx), c1 = (int)floorf((py - gy) * icy), c2 = (int)floorf((pz - gz) * icz);
  c0 = min(max(c0, 0), nx - 1);
  c1 = min(max(c1, 0), ny - 1);
  c2 = min(max(c2, 0), nz - 1);
  const bool fx = dx >= 0.0f, fy = dy >= 0.0f, fz = dz >= 0.0f;
  float tmx = fmaf(fmaf((float)(c0 + (fx ? 1 : 0)), cx, gx), ix, -oix);
  float tmy = fmaf(fmaf((float)(c1 + (fy ? 1 : 0)), cy, gy), iy, -oiy);
  float tmz = fmaf(fmaf((float)(c2 + (fz ? 1 : 0)), cz, gz), iz, -oiz);
  const float tdx = fabsf(cx * ix), tdy = fabsf(cy * iy), tdz = fabsf(cz * iz);
  int remx = fx ? nx - 1 - c0 : c0, remy = fy ? ny - 1 - c1 : c1, remz = fz ? nz - 1 - c2 : c2;
  const int incx = fx ? 1 : -1, incy = fy ? nx : -nx, incz = fz ? nx * ny : -(nx * ny);
  int idx = (c2 * ny + c1) * nx + c0;

  uint32_t q0 = 0u, q1 = 0u;
  for (;;) {
    if constexpr (ST) stamps.iters += 1;
    if (walking) {
      const uint32_t cw = im.u32(sc.g_off_cells + 4u * (uint32_t)idx);
      ++nnode;
      if (cw != 0u) {
        if (q0 == 0u)
          q0 = cw;
        else
          q1 = cw;
      }
      // leave through the nearest cell wall
      const bool sx = tmx <= tmy && tmx <= tmz;
      const bool sy = !sx && tmy <= tmz;
      const float tnext = sx ? tmx : (sy ? tmy : tmz);
      const int rem = sx ? remx : (sy ? remy : remz);
      walking = rem > 0 && !(tnext > tmax32);
      idx += sx ? incx : (sy ? incy : incz);
      tmx += sx ? tdx : 0.0f;
      tmy += sy ? tdy : 0.0f;
      tmz += (!sx && !sy) ? tdz : 0.0f;
      remx -= sx ? 1 : 0;
      remy -= sy ? 1 : 0;
      remz -= (!sx && !sy) ? 1 : 0;
    }
    const bool any_walking = __any(walking);
    if (__any(q1 != 0u) || !any_walking) {
      stamps.mark(RG_WALK);
      if constexpr (ST) stamps.phases += 1;
      if (q0 != 0u) leaf_test<LDS, true>(im, sc, off, q0 >> 8, q0 & 255u, ray, best, nprim, last_id);
      q0 = q1;
      q1 = 0u;
      tmax32 = round_up_f32(best.t);
      stamps.mark(RG_LEAF);
      if (!any_walking && !__any(q0 != 0u)) break;
    }
  }
  return best;
}

// --------------------------------------------------------------- the kernel ---
// n / d for a divisor fixed per launch: q = (((n - t) >> 1) + t) >> shift, t = mulhi(n, magic)
// (round-up method, exact for every 32-bit n; magic/shift come from the host).
__device__ __forceinline__ uint32_t fastdiv(uint32_t n, FastDiv f) {
  const uint32_t t = __umulhi(n, f.magic);
  return f.shift == 255u ? n : (((n - t) >> 1) + t) >> f.shift;  // shift 255: divisor 1
}

constexpr uint32_t kItemBatch = 64;  // work items fetched per global atomic (per wave)

__device__ __forceinline__ unsigned lane_id() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// cross-lane reads (ds_bpermute: lane i receives the value of lane src_i; all lanes active)
__device__ __forceinline__ uint32_t lane_read(uint32_t src, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)v);
}
__device__ __forceinline__ double lane_read(uint32_t src, double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const uint32_t lo = lane_read(src, (uint32_t)b), hi = lane_read(src, (uint32_t)(b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ uint32_t lanes_below(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// KERNEL: 1 = STREAM, 2 = BVH;  LDS: scene image staged in LDS (BVH only)
template <int KERNEL, bool LDS, bool STAMPS = false>
__global__ void __launch_bounds__(KERNEL >= 2 ? 1024 : 256)
    RTOW_CAT(rtow_trace_, RTOW_SUFFIX)(const TraceParams P) {
  const DevScene &sc = P.sc;
  const uint32_t k0 = P.seed_lo, k1 = P.seed_hi;
  const unsigned lane = lane_id();
  const uint32_t lane_g = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t npix_local = (uint32_t)P.local_rows * (uint32_t)P.W;

  Image<LDS> im;
  im.g = KERNEL == 3 ? sc.gblob : sc.blob;
  if constexpr (KERNEL >= 2 && LDS) {
    // stage the scene image: coalesced 16-byte loads, 16-byte LDS stores
    const uint4 *src = reinterpret_cast<const uint4 *>(im.g);
    uint4 *dst = reinterpret_cast<uint4 *>(rtow_lds);
    const uint32_t n16 = (KERNEL == 3 ? sc.gblob_bytes : sc.blob_bytes) / 16u;
    for (uint32_t i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
  }

  // per-lane state
  bool done = false;
  bool need_sample = true;
  int s_left = 0;             // samples left in the current item
  uint32_t item = 0xffffffffu;
  uint32_t j = 0;             // column
  uint32_t gi = 0;            // global row (from the top)
  V3d acc = {0.0, 0.0, 0.0};  // pixel_color of this item (src/render.cpp:156), binary64 in every build
  V3 ro = {0, 0, 0}, rd = {0, 0, 1};
  real rtime = 0;
  int depth = 0;              // remaining child rays
  int nb = 0;                 // bounces recorded on the path stack
  Rng g = {0, 0, 0};
  uint32_t nseg = 0, nnode = 0, nprim = 0;
  // end-of-launch sample donation (see "tail" below)
  bool helping = false;    // this lane traces a sample donated by another lane of the wave
  bool holding = false;    // ... has finished it and keeps its colour in `acc` until the owner adds it
  uint32_t partners = 0u;  // owner: stack of its helpers' lane ids (6 bits each, most recent lowest);
                           // helper: its owner's lane id
  int n_out = 0;           // owner: donated samples not yet added
  uint32_t tail_trips = 0u;
  uint32_t pool_next = 0, pool_end = 0;  // wave-uniform: this wave's batch of work items
  unsigned long long seen = 0ull;        // wave-uniform: queue head as of this wave's last fetch
  const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
  Stamps<STAMPS> stamps;
  stamps.start();
  unsigned long long t_empty = 0ull;  // diagnostic: when this wave first saw the queue empty
  unsigned long long trips_after_empty = 0ull;
  if constexpr (STAMPS) {
    if (lane == 0) atomicMin(P.t_origin, (unsigned long long)__builtin_amdgcn_s_memrealtime());
  }

  for (;;) {
    // ---- item bookkeeping ---------------------------------------------------
    bool need_item = false;
    if (!done && need_sample && s_left <= 0) {
      if (helping) {  // a donated sample is finished: keep its colour for the owner
        helping = false;
        holding = true;
        done = true;
      } else if (n_out == 0) {
        if (item != 0xffffffffu) {
          double *dst = P.partials + (size_t)item * 3;
          dst[0] = acc.x;
          dst[1] = acc.y;
          dst[2] = acc.z;
        }
        need_item = true;
      }  // else: an owner waiting for donated samples
    }
    const unsigned long long need_mask = __ballot(need_item);
    if (need_mask != 0ull) {
      // The item-decoding parameters are read here from the kernel-argument segment (scalar loads)
      // instead of living in SGPRs for the whole launch: the kernel is VALU-issue-bound and ran out
      // of SGPRs, so every one of them cost a v_readlane (VALU) per use.
      const RTOW_CONST TraceParams *kp = (const RTOW_CONST TraceParams *)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(kp));  // opaque per trip: keeps the loads from being hoisted out of the loop
      // Wave-local pool [pool_next, pool_end): one global atomic buys kItemBatch items,
      // which the lanes then take by ballot rank with no further traffic (a single hot
      // counter word saturates near 90 dequeues/us on this chip — one atomic per wave
      // trip was the bottleneck).  All of this is wave-uniform except `mine`.
      const uint32_t want = (uint32_t)__popcll(need_mask);
      const uint32_t avail = pool_end - pool_next;
      const uint32_t rank = (uint32_t)__popcll(need_mask & ((1ull << lane) - 1ull));
      unsigned long long mine = (unsigned long long)pool_next + rank;
      if (want > avail) {
        // guided self-scheduling: 64 items per atomic while the queue is long, shrinking to
        // exactly what this wave needs now as it drains (a wave that hoards items at the end
        // of the queue keeps the whole launch waiting: measured ~4 item durations per launch)
        const unsigned long long left = (unsigned long long)kp->n_items > seen ? (unsigned long long)kp->n_items - seen : 0ull;
        uint32_t batch = (uint32_t)(left / ((unsigned long long)n_waves * 4ull));
        batch = batch > kItemBatch ? kItemBatch : batch;
        batch = batch < want - avail ? want - avail : batch;
        const int leader = __ffsll((long long)need_mask) - 1;
        unsigned long long base = seen;
        // a wave that has seen the end of the queue stops polling it: at the end of a launch every
        // wave asks every trip, and the one counter word serves ~100 requests/us (measured: the
        // last trips of a launch took 34 us instead of 13)
        if (seen < (unsigned long long)kp->n_items) {
          if ((int)lane == leader) base = atomicAdd(&P.counters[0], (unsigned long long)batch);
          base = __shfl(base, leader);
        }
        seen = base + batch;  // how far the queue had advanced when this wave last looked
        if (rank >= avail) mine = base + (rank - avail);
        const unsigned long long nn = base + (want - avail), ne = base + batch;
        const unsigned long long cap = (unsigned long long)kp->n_items;
        pool_next = (uint32_t)(nn < cap ? nn : cap);
        pool_end = (uint32_t)(ne < cap ? ne : cap);
      } else {
        pool_next += want;
      }
      if (need_item) {
        if (mine >= (unsigned long long)kp->n_items) {
          done = true;
          if constexpr (STAMPS) {
            if (t_empty == 0ull) t_empty = __builtin_amdgcn_s_memrealtime();
          }
        } else {
          // Queue order.  Tiled mode: tile-major, stream-minor — all streams of a 64-pixel tile are
          // adjacent, tiles run top-to-bottom, and the queue is consumed from its far end, so a launch
          // ENDS on the top rows of the image for every stream.  In the reference's scenes that is sky
          // (the top 8 % of the cover image: one-segment paths), so most waves run out of work together:
          // waves finishing > 0.2 ms after they find the queue empty fell from 51 % to 5 % (+1.9 %).
          // (Stream-major order ended only the last stream on the sky; ending on the bottom rows —
          // near ground, short paths — measures the same.)  The partial-sum slot stays [stream][pixel].
          const uint32_t qi = kp->n_items - 1u - (uint32_t)mine;
          uint32_t k, lp, lr;
          if (kp->tile_h_log2 == 0u) {  // row-major, stream-major
            k = fastdiv(qi, FastDiv{kp->div_npix.magic, kp->div_npix.shift});
            lp = qi - k * npix_local;
            lr = fastdiv(lp, FastDiv{kp->div_w.magic, kp->div_w.shift});
            j = lp - lr * (uint32_t)kp->W;
          } else {  // 64-pixel tiles: a wave's batch of 64 items is one compact tile of one stream
            const uint32_t g64 = qi >> 6, w = qi & 63u;
            const uint32_t t = fastdiv(g64, FastDiv{kp->div_ns.magic, kp->div_ns.shift});
            k = g64 - t * (uint32_t)kp->nstreams;
            lp = (t << 6) | w;
            const uint32_t tr = fastdiv(t, FastDiv{kp->div_tpr.magic, kp->div_tpr.shift});
            const uint32_t tc = t - tr * (kp->div_tpr_n);
            lr = (tr << kp->tile_h_log2) + (w >> kp->tile_w_log2);
            j = (tc << kp->tile_w_log2) + (w & ((1u << kp->tile_w_log2) - 1u));
          }
          item = k * npix_local + lp;  // partial-sum slot
          // local row -> global row: this rank's q-th strip is global strip q*nranks+rank
          const uint32_t q = fastdiv(lr, FastDiv{kp->div_tile.magic, kp->div_tile.shift});
          const uint32_t rr = lr - q * (uint32_t)kp->tile_rows;
          gi = (q * (uint32_t)kp->nranks + (uint32_t)kp->rank) * (uint32_t)kp->tile_rows + rr;
          g.pixel = gi * (uint32_t)kp->W + j;
          g.sample = (k + (uint32_t)kp->stream_first) * (uint32_t)kp->spt;  // first sample index of this stream
          s_left = kp->spt;
          acc = {0.0, 0.0, 0.0};
        }
      }
    }
    // ---- tail: sample donation ------------------------------------------------------------
    // Once the queue is empty a wave is as slow as its slowest lane's item (up to spt samples of
    // up to max_child_rays segments each) while its other lanes idle.  An idle lane therefore
    // takes over the LAST unstarted
    // sample of a lane that still has two or more to go.  The owner adds the donated colours
    // after its own samples, in sample order (most recent donation first), so the pixel sum is
    // the sequential one bit for bit.  Only lanes of the same wave trade (registers + ds_bpermute).
    // Measured: +0.9 % on C2.  What remains of the tail is one PATH: a trip takes ~13.7 us with four
    // waves per SIMD, so a 50-bounce path started just before the queue empties runs ~0.2-0.7 ms.
    if (__ballot(done) != 0ull) {
      // (1) owners that have finished their own samples take the next donated colour, if ready
      const bool ready = !done && need_sample && s_left <= 0 && n_out > 0;
      if (__ballot(ready) != 0ull) {
        const uint32_t h = partners & 63u;
        const uint32_t src = ready ? h : lane;
        // (not `ready && lane_read(..)`: short-circuit evaluation would run the cross-lane read with
        // only the ready lanes active, and an inactive source lane reads as 0)
        const uint32_t partner_holds = lane_read(src, (uint32_t)holding);
        const bool take = ready && partner_holds != 0u;
        const V3d c = {lane_read(src, acc.x), lane_read(src, acc.y), lane_read(src, acc.z)};
        // a holding helper asks its owner whether it was the one taken
        const uint32_t ow = holding ? partners : lane;
        const bool o_take = lane_read(ow, (uint32_t)take) != 0u;
        const uint32_t o_h = lane_read(ow, h);
        if (take) {
          acc = acc + c;
          partners >>= 6;
          --n_out;
        }
        if (holding && o_take && o_h == lane) holding = false;
      }
      // (2) idle lanes take the last unstarted sample of lanes with >= 2 samples to go
      const bool can_give = !done && !helping && s_left >= 2 && n_out < 5;
      const bool can_help = done && !holding;
      const unsigned long long gm = __ballot(can_give), hm = __ballot(can_help);
      if (gm != 0ull && hm != 0ull) {
        const uint32_t ng = (uint32_t)__popcll(gm), nh = (uint32_t)__popcll(hm);
        const uint32_t cnt = ng < nh ? ng : nh;
        const uint32_t grank = lanes_below(gm), hrank = lanes_below(hm);
        // compaction (a permutation of the lanes): lane k learns the k-th giver / k-th helper
        const uint32_t giver_k = (uint32_t)__builtin_amdgcn_ds_permute(
            (int)((can_give ? grank : ng + (lane - grank)) << 2), (int)lane);
        const uint32_t helper_k = (uint32_t)__builtin_amdgcn_ds_permute(
            (int)((can_help ? hrank : nh + (lane - hrank)) << 2), (int)lane);
        const bool gives = can_give && grank < cnt;
        const bool helps = can_help && hrank < cnt;
        const uint32_t my_giver = lane_read(helps ? hrank : lane, giver_k);
        const uint32_t my_helper = lane_read(gives ? grank : lane, helper_k);
        const uint32_t gsrc = helps ? my_giver : lane;
        const uint32_t o_j = lane_read(gsrc, j), o_gi = lane_read(gsrc, gi);
        const uint32_t o_pixel = lane_read(gsrc, g.pixel), o_sample = lane_read(gsrc, g.sample);
        const uint32_t o_left = lane_read(gsrc, (uint32_t)s_left);
        if (gives) {
          s_left -= 1;
          partners = (partners << 6) | my_helper;
          ++n_out;
        }
        if (helps) {
          j = o_j;
          gi = o_gi;
          g.pixel = o_pixel;
          g.sample = o_sample + o_left - 1u;  // the giver's last sample
          s_left = 1;
          need_sample = true;
          acc = {0.0, 0.0, 0.0};
          helping = true;
          done = false;
          partners = my_giver;
        }
      }
      // structural bound on the tail (every wait above ends when a bounded path ends; this makes
      // the exit independent of that argument): give up donating, never hang
      if (++tail_trips > 4096u + 8u * (uint32_t)(P.max_child_rays + 2) * (uint32_t)(P.spt + 1)) {
        done = true;
        helping = false;
        holding = false;
        n_out = 0;
      }
    }
    if (__ballot(!done || holding) == 0ull) break;
    stamps.mark(RG_FETCH);

    // (a lane whose fresh item has no samples — spt == 0 — goes straight back for the next one)
    const bool live = !done && !(need_sample && s_left <= 0);

    // ---- new sample: pixel jitter + Camera::get_ray ----------------------------
    if (live && need_sample) {
      g.r = 0u;
      // src/render.cpp:158-159
      const int from_top_i = P.H - (int)gi - 1;
      real ju, jv, jt;
      rng_jitter(g, k0, k1, ju, jv, jt);
      const real u = fast_div((real)(int)j + ju, (real)(P.W - 1));
      const real v = fast_div((real)from_top_i + jv, (real)(P.H - 1));
      // src/common-model.cpp:156-167; disk sample: y draws first (random-utils.cpp:36)
      real px, py;
      for (;;) {
        real c0, c1;
        rng_disk(g, k0, k1, c0, c1);
        py = c0 * (real(1.0) - real(-1.0)) + real(-1.0);
        px = c1 * (real(1.0) - real(-1.0)) + real(-1.0);
        if (px * px + py * py + real(0.0) * real(0.0) >= real(1.0)) continue;
        break;
      }
      // camera block: wave-uniform scalar loads (origin u v horizontal vertical llc | lens t0 t1)
#ifdef RTOW_REAL_F32
      const RTOW_CONST float *cm = (const RTOW_CONST float *)P.cam32;
#else
      cdptr cm = (cdptr)(const double *)P.cam;
#endif
      const real lens = cm[18], ct0 = cm[19], ct1 = cm[20];
      const real rdx = lens * px, rdy = lens * py;
      const V3 offset = V3{cm[3], cm[4], cm[5]} * rdx + V3{cm[6], cm[7], cm[8]} * rdy;
      const V3 from = V3{cm[0], cm[1], cm[2]} + offset;
      rd = V3{cm[15], cm[16], cm[17]} + u * V3{cm[9], cm[10], cm[11]} + v * V3{cm[12], cm[13], cm[14]} - from;
      ro = from;
      rtime = jt * (ct1 - ct0) + ct0;
      depth = P.max_child_rays;
      nb = 0;
      need_sample = false;
    }

    stamps.mark(RG_REGEN);
    if constexpr (STAMPS) {
      stamps.trips += 1;
      if (t_empty != 0ull) trips_after_empty += 1;
    }
    // ---- one ray segment: closest hit --------------------------------------------
    Closest best;
    best.t = 0;
    best.prim = -1;
    if constexpr (KERNEL == 3) {
      best = closest_hit_grid<LDS, STAMPS>(im, sc, ro, rd, rtime, live, nnode, nprim, stamps);
    } else if constexpr (KERNEL == 2) {
      // the walk uses wave votes, so every lane of the wave enters it
      best = closest_hit_bvh<LDS, STAMPS>(im, sc, ro, rd, rtime, live, nnode, nprim, stamps);
    } else {
      if (live) best = closest_hit_stream(sc, to_f64(ro), to_f64(rd), (double)rtime);
    }

    stamps.mark(RG_WALK);
    if (live) {
      ++nseg;
      if (best.prim >= 0) {
        if (depth <= 0) {
          need_sample = true;  // src/render.cpp:115: black
        } else {
          // rebuild the Hit of the winner (src/common-model.cpp:83-90, :121).  The walking
          // kernels read the winner's record, its material index and the material from the
          // LDS scene image (a chain of three dependent loads: LDS latency, not L2's)
          V3 where = ro + rd * best.t;
          V3 normal;
          bool front = true;
          const int pid = best.prim;
          int mi, kind;
          real m_fuzz, m_ir;
          if constexpr (KERNEL >= 2) {
            const uint32_t o_sph = KERNEL == 3 ? sc.g_off_sph : sc.off_sph;
            const uint32_t o_mov = KERNEL == 3 ? sc.g_off_mov : sc.off_mov;
            const uint32_t o_tri = KERNEL == 3 ? sc.g_off_tri : sc.off_tri;
            const uint32_t o_pmat = KERNEL == 3 ? sc.g_off_pmat : sc.off_pmat;
            const uint32_t o_mats = KERNEL == 3 ? sc.g_off_mats : sc.off_mats;
            if (pid < sc.n_sph + sc.n_mov) {
              V3 center;
              bool inward;  // negative radius: only the sign of the signed r*r is used here
              if (pid < sc.n_sph) {
                const double2 p0 = im.d2(o_sph + 32u * (uint32_t)pid), p1 = im.d2(o_sph + 32u * (uint32_t)pid + 16u);
                center = {(real)p0.x, (real)p0.y, (real)p1.x};
                inward = p1.y < 0.0;
              } else {
                const uint32_t r = o_mov + 64u * (uint32_t)(pid - sc.n_sph);
                const double2 p0 = im.d2(r), p1 = im.d2(r + 16u), p2 = im.d2(r + 32u), p3 = im.d2(r + 48u);
#ifdef RTOW_REAL_F32
                center = {(real)p0.x + rtime * (real)p1.y, (real)p0.y + rtime * (real)p2.x, (real)p1.x + rtime * (real)p2.y};
#else
                center = {p0.x + rtime * p1.y, p0.y + rtime * p2.x, p1.x + rtime * p2.y};
#endif
                inward = p3.x < 0.0;
              }
              normal = normalize(where - center);
              front = (dot(rd, normal) < real(0.0)) ^ inward;
              normal = front ? normal : -normal;
            } else {
#ifdef RTOW_REAL_F32
              const uint32_t r = o_tri + 48u * (uint32_t)(pid - sc.n_sph - sc.n_mov);
              const float4 q2 = im.f4(r + 32u);
              normal = {q2.y, q2.z, q2.w};
#else
              const uint32_t r = o_tri + 96u * (uint32_t)(pid - sc.n_sph - sc.n_mov);
              const double2 q4 = im.d2(r + 64u), q5 = im.d2(r + 80u);
              normal = {q4.y, q5.x, q5.y};
#endif
            }
            mi = (int)im.u32(o_pmat + 4u * (uint32_t)pid);
            const uint32_t mr = o_mats + 48u * (uint32_t)mi;
            const double2 m1 = im.d2(mr + 16u), m2 = im.d2(mr + 32u);  // {att.z, fuzz}, {ir, kind|pad}
            m_fuzz = (real)m1.y;
            m_ir = (real)m2.x;
            kind = (int)(__double_as_longlong(m2.y) & 0xffffffffll);
          } else {
            if (pid < sc.n_sph + sc.n_mov) {
              V3 center;
              bool inward;
              if (pid < sc.n_sph) {
                const double *q = sc.sph + 4 * (size_t)pid;
                center = {(real)q[0], (real)q[1], (real)q[2]};
                inward = sc.sph_r[pid] < 0.0;
              } else {
                const double *q = sc.mov + 8 * (size_t)(pid - sc.n_sph);
#ifdef RTOW_REAL_F32
                center = {(real)q[0] + rtime * (real)q[3], (real)q[1] + rtime * (real)q[4], (real)q[2] + rtime * (real)q[5]};
#else
                center = {q[0] + rtime * q[3], q[1] + rtime * q[4], q[2] + rtime * q[5]};
#endif
                inward = q[7] < 0.0;
              }
              normal = normalize(where - center);
              front = (dot(rd, normal) < real(0.0)) ^ inward;
              normal = front ? normal : -normal;
            } else {
              const double *q = sc.tri + 12 * (size_t)(pid - sc.n_sph - sc.n_mov);
              normal = {(real)q[9], (real)q[10], (real)q[11]};
            }
            mi = sc.prim_mat[pid];
            const DevMaterial *m = sc.mats + mi;
            kind = m->kind;
            m_fuzz = (real)m->fuzz;
            m_ir = (real)m->ir;
          }

          // ---- Material::scatter (src/common-model.cpp:13-62) ------------------
          // first unit-ball candidate of this bounce; its block also carries the coin
          real coin;
          V3 rnd = rng_scatter(g, k0, k1, coin);
          V3 dirbase = {0, 0, 0};
          if (kind == 2) {
            const real ir = m_ir;
            const V3 unit = normalize(rd);
            const real cos_theta = dot(-unit, normal);
            const real sin_theta = fast_sqrt(real(1.0) - cos_theta * cos_theta);
            const real ratio = front ? fast_rcp(ir) : ir;
            bool refl = ratio * sin_theta > real(1.0);
            if (!refl) {
              real r0 = fast_div(real(1.0) - ratio, real(1.0) + ratio);
              r0 = r0 * r0;
              const real x = real(1.0) - cos_theta;
              const real x2 = x * x;
              const real R = r0 + (real(1.0) - r0) * (x2 * x2 * x);
              refl = R > coin;
            }
            dirbase = refl ? reflect(unit, normal) : refract(unit, normal, ratio);
          } else if (kind == 1) {
            dirbase = reflect(rd, normal);
          }
          // random_unit_vector(): reject candidates outside the unit ball (random-utils.cpp:23-33)
          while (dot(rnd, rnd) >= real(1.0)) {
            real unused;
            rnd = rng_scatter(g, k0, k1, unused);
          }
          V3 dir;
          bool absorbed = false;
          if (kind == 0) {
            absorbed = rabs(normal.x - rnd.x) < real(1e-8) && rabs(normal.y - rnd.y) < real(1e-8) &&
                       rabs(normal.z - rnd.z) < real(1e-8);
            dir = normal + rnd;
          } else {
            dir = dirbase + m_fuzz * rnd;
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
        const real t = real(0.5) * (unit.y + real(+1.0));
        V3 c = (real(1.0) - t) * V3{1, 1, 1} + t * V3{real(0.5), real(0.7), real(1.0)};
        for (int q = nb - 1; q >= 0; --q) {
          const uint32_t smi = P.stack[(size_t)q * P.n_lanes + lane_g];
          if constexpr (KERNEL >= 2) {
            const uint32_t mr = (KERNEL == 3 ? sc.g_off_mats : sc.off_mats) + 48u * smi;
            const double2 a0 = im.d2(mr), a1 = im.d2(mr + 16u);
            c = V3{(real)a0.x, (real)a0.y, (real)a1.x} * c;
          } else {
            const DevMaterial *m = sc.mats + smi;
            c = V3{(real)m->att[0], (real)m->att[1], (real)m->att[2]} * c;
          }
        }
        acc = acc + to_f64(c);  // pixel_color += ray_color(...)
        need_sample = true;
      }
      if (need_sample) {
        --s_left;
        ++g.sample;
      }
    }
    stamps.mark(RG_SHADE);
  }
  if constexpr (STAMPS) {
    if (lane == 0)
    {
      // wave lifetime on the 100 MHz constant clock: [4] sum of end times, [5] min, [6] max,
      // [7] sum of the times at which the wave first found the queue empty (relative to the
      // earliest start, kept in [8+RG_COUNT+3] as a min)
      const unsigned long long tend = __builtin_amdgcn_s_memrealtime();
      atomicAdd(&P.counters[4], tend - P.t_origin[0]);
      atomicMin(&P.counters[5], tend - P.t_origin[0]);
      atomicMax(&P.counters[6], tend - P.t_origin[0]);
      atomicAdd(&P.counters[7], t_empty - P.t_origin[0]);
      // [17..22]: histogram of (end - this wave's own queue-empty time) in 0.2 ms bins (last: >= 1 ms)
      {
        const unsigned long long after = tend - t_empty;  // 100 MHz ticks
        unsigned bin = (unsigned)(after / 20000ull);
        bin = bin > 5u ? 5u : bin;
        atomicAdd(&P.counters[17 + bin], 1ull);
        atomicMax(&P.counters[40], trips_after_empty);
        if (bin >= 2u) {  // stragglers: trips and time after the queue emptied
          atomicAdd(&P.counters[41], trips_after_empty);
          atomicAdd(&P.counters[42], after);
          atomicAdd(&P.counters[43], 1ull);
        }
      }
      for (int r = 0; r < RG_COUNT; ++r) atomicAdd(&P.counters[8 + r], stamps.t[r]);
      atomicAdd(&P.counters[13], stamps.iters);
      atomicAdd(&P.counters[14], stamps.trips);
      atomicAdd(&P.counters[15], stamps.phases);
    }
  }

  // stats: one atomic per wave and counter
  unsigned long long t0 = nseg, t1 = nprim, t2 = nnode;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    t0 += __shfl_down(t0, off);
    t1 += __shfl_down(t1, off);
    t2 += __shfl_down(t2, off);
  }
  if (lane == 0) {
    atomicAdd(&P.counters[1], t0);
    if (KERNEL >= 2) {
      atomicAdd(&P.counters[2], t1);
      atomicAdd(&P.counters[3], t2);
    }
  }
}

}  // namespace

// kernel: 1 STREAM, 2 BVH, 3 GRID; +16 = diagnostic region stamps (LDS variants only)
template <int K, bool L, bool S>
static int launch_one(const TraceParams &p, int grid, int block, unsigned lds_bytes, hipStream_t st) {
  auto k = RTOW_CAT(rtow_trace_, RTOW_SUFFIX)<K, L, S>;
  if (lds_bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds_bytes, st, p);
  return (int)hipGetLastError();
}

int RTOW_CAT(launch_trace_, RTOW_SUFFIX)(const TraceParams &p, int kernel, int grid, int block,
                                         unsigned lds_bytes, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  const bool lds = lds_bytes > 0;
  switch (kernel) {
    case 1: return launch_one<1, false, false>(p, grid, block, 0, st);
    case 2: return lds ? launch_one<2, true, false>(p, grid, block, lds_bytes, st)
                       : launch_one<2, false, false>(p, grid, block, 0, st);
    case 3: return lds ? launch_one<3, true, false>(p, grid, block, lds_bytes, st)
                       : launch_one<3, false, false>(p, grid, block, 0, st);
    case 2 + 16: return launch_one<2, true, true>(p, grid, block, lds_bytes, st);
    case 3 + 16: return launch_one<3, true, true>(p, grid, block, lds_bytes, st);
    default: return (int)hipErrorInvalidValue;
  }
}

// Workgroups per CU that stay resident: min over the register file (512 VGPRs per
// SIMD lane, allocated in granules of 8), the 32-wave CU limit and the 160 KiB of LDS.
// (The runtime's occupancy query ignores LDS above 64 KiB per CU on this stack; a grid
// that turns out larger than resident only queues the surplus workgroups, which then
// find the work queue empty — there is no inter-workgroup dependency.)
int RTOW_CAT(trace_occupancy_, RTOW_SUFFIX)(int kernel, int block, unsigned lds_bytes) {
  const void *fn;
  const bool lds = lds_bytes > 0;
  if (kernel == 3)
    fn = lds ? reinterpret_cast<const void *>(RTOW_CAT(rtow_trace_, RTOW_SUFFIX)<3, true, false>)
             : reinterpret_cast<const void *>(RTOW_CAT(rtow_trace_, RTOW_SUFFIX)<3, false, false>);
  else if (kernel == 2)
    fn = lds ? reinterpret_cast<const void *>(RTOW_CAT(rtow_trace_, RTOW_SUFFIX)<2, true, false>)
             : reinterpret_cast<const void *>(RTOW_CAT(rtow_trace_, RTOW_SUFFIX)<2, false, false>);
  else
    fn = reinterpret_cast<const void *>(RTOW_CAT(rtow_trace_, RTOW_SUFFIX)<1, false, false>);
  if (lds_bytes > 48 * 1024)
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  hipFuncAttributes fa;
  if (hipFuncGetAttributes(&fa, fn) != hipSuccess) return -1;
  const int regs = fa.numRegs > 0 ? fa.numRegs : 128;
  const int alloc = ((regs + 7) / 8) * 8;
  int waves_per_simd = 512 / alloc;
  if (waves_per_simd > 8) waves_per_simd = 8;
  if (waves_per_simd < 1) waves_per_simd = 1;
  const int waves_per_block = block / 64;
  int nb = (waves_per_simd * 4) / waves_per_block;
  if (lds_bytes > 0) {
    const int by_lds = (int)((160u * 1024u) / lds_bytes);
    if (by_lds < nb) nb = by_lds;
  }
  if (nb < 1) nb = 1;
  return nb;
}

}  // namespace rtow
