// rtow_trace_math.h — part of the trace kernels (included by rtow_trace_body.h inside namespace rtow::{anonymous};
// see that file for the execution model).  Arithmetic type `real`, vector math, fast/strict sqrt, reciprocal and division.
#pragma once

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
// A valid but unspecified value at no cost: the initialiser of a variable that only SOME lanes assign and only those
// lanes read.  A zero there is a v_mov per register in every trip of the main loop (the merge of "assigned" and "not
// assigned" needs a value on both sides); an empty asm statement that claims to define the register gives the merge
// a value without an instruction.  (__builtin_nondeterministic_value — freeze poison — is folded to zero.)
template <class T>
__device__ __forceinline__ T anyv(T) {
  T x;
  asm volatile("" : "=v"(x));  // (volatile: one definition per variable — a shared one is COPIED into each)
  return x;
}
__device__ __forceinline__ V3 anyv3() { return V3{anyv(real(0)), anyv(real(0)), anyv(real(0))}; }
// Issue priority of the wave by STAGE (s_setprio, 0..3; one scalar instruction).  The kernel is bound by VALU issue and a
// SIMD's waves are each in a different stage of their trip.  Left alone, the arbiter treats them alike; told which stage
// is latency-bound (LDS chains: highest), which is mixed (the default) and which is pure arithmetic (the random-number
// window: lowest), it lets the memory requests leave first and fills the wait with the arithmetic of the others.
// Round 4, C2: +3.9 % (every other placement that was tried — a constant priority, the reverse order, priority by path
// depth — measured the same as none or worse).  Scheduling only: the image cannot change.
constexpr int kPrioRng = 0, kPrioSetup = 1, kPrioStage = 2, kPrioLeaf = 3;
template <int P>
__device__ __forceinline__ void stage_prio() {
#ifndef RTOW_NO_STAGE_PRIO
  __builtin_amdgcn_s_setprio(P);
#endif
}
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
// fast build: hardware reciprocal / reciprocal-square-root seed (relative error 5e-8 measured,
// scripts/experiments/seed_accuracy.hip) + ONE refinement step instead of the correctly rounded division and sqrt:
// rcp   y (1 + e + e^2), e = 1 - x y                    3 fma           max relative error 1.1e-16
// rsq   y + (y/2) e,     e = 1 - (x y) y                mul fma mul fma (Goldschmidt, second order)  ~4e-15
// sqrt  s + (s/2)... as h = y/2, s = x y, e = 1/2 - s h, s + s e        mul mul fma fma              ~4e-15
// Round 4: the square roots dropped from the third-order polynomial y (15/8 - 5/4 z + 3/8 z^2) (3e-16, seven
// instructions of which two materialise the constant -5/4 in the fmac's destination) to the second-order step:
// four instructions, inline constants only, no register pair held for 15/8.  The seed's 5e-8 squared is 2.5e-15:
// twenty ulps of a ray parameter whose consumers (hit / miss decisions at tmin = 1e-3, the shading point) already
// differ from the strict build by FMA contraction at the same order; the tolerance tests are the judge.  The strict
// build keeps the IEEE forms.
__device__ __forceinline__ double fast_rsqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
#ifdef RTOW_SQRT_ORDER3
  const double z = (x * y) * y;
  return y * __builtin_fma(z, __builtin_fma(z, 0.375, -1.25), 1.875);
#else
  const double e = __builtin_fma(-(x * y), y, 1.0);
  return __builtin_fma(y * 0.5, e, y);
#endif
}
// (x > 0 only: x = 0 gives NaN — the callers below either know x > 0 or treat NaN as "no")
__device__ __forceinline__ double fast_sqrt_pos(double x) {
  const double y = __builtin_amdgcn_rsq(x);
#ifdef RTOW_SQRT_ORDER3
  const double s = x * y, z = s * y;
  return s * __builtin_fma(z, __builtin_fma(z, 0.375, -1.25), 1.875);
#else
  const double s = x * y, h = y * 0.5;
  const double e = __builtin_fma(-s, h, 0.5);
  return __builtin_fma(s, e, s);
#endif
}
__device__ __forceinline__ double fast_sqrt(double x) {
  if (!(x > 0.0)) return 0.0;
  return fast_sqrt_pos(x);
}
__device__ __forceinline__ double fast_rcp(double x) {
  const double y = __builtin_amdgcn_rcp(x);
  const double e = __builtin_fma(-x, y, 1.0);
  return __builtin_fma(y, __builtin_fma(e, e, e), y);
}
__device__ __forceinline__ double fast_div(double n, double d) { return n * fast_rcp(d); }
#else
__device__ __forceinline__ double fast_div(double n, double d) { return n / d; }
__device__ __forceinline__ double fast_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ double fast_sqrt_pos(double x) { return sqrt(x); }
__device__ __forceinline__ double fast_rcp(double x) { return 1.0 / x; }
__device__ __forceinline__ double fast_rsqrt(double x) { return 1.0 / sqrt(x); }
#endif
// binary32 forms (f32 build only): hardware rsq/rcp/sqrt (1 ulp) + one Newton step where it is cheap
__device__ __forceinline__ float fast_rsqrt(float x) {
  float y = __builtin_amdgcn_rsqf(x);
  return y * (1.5f - 0.5f * x * y * y);
}
__device__ __forceinline__ float fast_sqrt(float x) { return x > 0.0f ? __builtin_amdgcn_sqrtf(x) : 0.0f; }
__device__ __forceinline__ float fast_sqrt_pos(float x) { return __builtin_amdgcn_sqrtf(x); }
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

// ---- typed access to LDS and global memory -------------------------------------------------------------------------
// the dynamic LDS block of the trace kernels: a scene image (BVH / GRID), the top of the 4-wide tree and the traversal
// stacks (BVH4), or the STREAM kernel's per-wave triangle tiles
extern __shared__ __align__(16) unsigned char rtow_lds[];
// LDS is addressed with 32-bit offsets through address-space-3 pointers, global memory through
// address-space-1 pointers: with generic pointers the compiler merges the two sides of an
// "in LDS or in global memory" choice into flat loads (and 64-bit address arithmetic).
typedef float vf4 __attribute__((ext_vector_type(4)));      // native vectors: HIP's float4 class cannot be
typedef uint32_t vu4 __attribute__((ext_vector_type(4)));   // read through an address-space pointer
typedef double vd2 __attribute__((ext_vector_type(2)));
typedef _Float16 vh4 __attribute__((ext_vector_type(4)));   // four binary16 planes: the operands of v_fma_mix_f32
#define RTOW_AS_LDS __attribute__((address_space(3)))
#define RTOW_AS_GLB __attribute__((address_space(1)))
// `off` IS the LDS address: the trace kernels have no static LDS, so the dynamic block (rtow_lds) starts at 0 —
// checked on the host (trace_occupancy_*: hipFuncAttributes::sharedSizeBytes == 0).  Written as `rtow_lds + off` the
// compiler kept a `v_add_u32 v, 0, v` per access (the symbol's address, resolved too late to fold): nine per node
// step of the 4-wide walk.
template <class T>
__device__ __forceinline__ T lds_read(uint32_t off) {
#ifdef RTOW_LDS_SYMBOL
  return *(const RTOW_AS_LDS T *)((const RTOW_AS_LDS unsigned char *)rtow_lds + off);
#else
  return *(const RTOW_AS_LDS T *)(uintptr_t)off;
#endif
}
template <class T>
__device__ __forceinline__ void lds_write(uint32_t off, T v) {
#ifdef RTOW_LDS_SYMBOL
  *(RTOW_AS_LDS T *)((RTOW_AS_LDS unsigned char *)rtow_lds + off) = v;
#else
  *(RTOW_AS_LDS T *)(uintptr_t)off = v;
#endif
}
template <class T>
__device__ __forceinline__ T glb_read(const unsigned char *base, uint32_t off) {
  return *(const RTOW_AS_GLB T *)((const RTOW_AS_GLB unsigned char *)base + off);
}
