// rtow_trace_bvh.h — part of the trace kernels (included by rtow_trace_body.h inside namespace rtow::{anonymous};
// see that file for the execution model).  Scene-image reader, leaf tests and the threaded BVH walk.
#pragma once
// ---------------------------------------------------------- closest hit: BVH ---
// (rtow_lds, the dynamic LDS block of the trace kernels, is declared in rtow_trace_math.h)

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
  __device__ __forceinline__ uint4 u4(uint32_t off) const {
    if constexpr (LDS)
      return *reinterpret_cast<const uint4 *>(rtow_lds + off);
    else
      return *reinterpret_cast<const uint4 *>(g + off);
  }
  __device__ __forceinline__ double d1(uint32_t off) const {
    if constexpr (LDS)
      return *reinterpret_cast<const double *>(rtow_lds + off);
    else
      return *reinterpret_cast<const double *>(g + off);
  }
};

// Fast binary64 build, GRID walk: the walk runs on the UNIT direction (round 4).  With |d| = 1 the quadratic of a
// sphere is t^2 + 2 h t + c = 0: no a in the discriminant, no 1/a in the roots; and with the ray-independent
// k = |C|^2 - r^2 stored beside a cell-list sphere (rtow_grid.h) h = o.d - C.d and c = |o|^2 - 2 C.o + k take three fma
// each from per-ray constants — 8 operations to the discriminant instead of 12, 2 less per resolved root.  The ray
// parameter inside the walk is then a distance; the walk converts the closest hit back (t / |d|) when it returns, so
// everything outside it keeps the reference's un-normalised rays (src/common-model.cpp:24-31 depends on |d|).
// Rounding differs from the strict build's expressions at the 1e-12 level (cancellation in c), i.e. like any other
// re-association of the fast build: parity by tolerance.  The strict build never takes this path.
#if defined(RTOW_FAST_MATH) && !defined(RTOW_REAL_F32)
#define RTOW_UNIT_RAYS 1
#endif

// An f32 upper bound of the closest hit so far, for the conservative f32 culling tests (boxes, grid cells).  Not the
// correctly rounded-up conversion (no such instruction: __double2float_ru is a dozen instructions of integer
// fix-up) but round-to-nearest times (1 + 2^-21): the conversion is at most half an ulp (2^-24 relative) below t and
// the product, after its own rounding, at least (1 + 2^-22) times the converted value.  t > 0 here; +inf stays +inf.
__device__ __forceinline__ float round_up_f32(double t) { return (float)t * 1.00000048f; }
__device__ __forceinline__ float round_up_f32(float t) { return t; }

// 1 / d for the slab and DDA forms: the hardware reciprocal (1 ulp; the tests that use it are conservative by padding
// and slack) clamped to +-1e30 — rcp(+-0) and rcp(denormal) are +-inf and become +-1e30, which keeps fma(plane, inv,
// -o * inv) free of NaNs for axis-parallel rays.  Two instructions; the IEEE division `1.0f / d` this replaces was
// eleven plus a branch, three times per segment, in both builds (-freciprocal-math does not reach it).
__device__ __forceinline__ float safe_inv(float d) {
  const float big = 1e30f;
  return __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d), -big, big);
}

struct ImgOffsets {  // byte offsets of the id and record sections inside a scene image
  uint32_t ids, sph, mov, tri;
  uint32_t fat;           // grid image: id + sphere record side by side per list entry (0 = none)
  uint32_t fat_stride;    // ... bytes per entry: 48 (static spheres) or 80 (c0, delta, r2: scenes with moving spheres)
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
  double tmin;  // RTOW_TMIN in the walk's ray parameter
#ifdef RTOW_UNIT_RAYS
  bool unit;    // d is a unit vector (a = 1): the fields below are set
  double od, oo;  // o.d, |o|^2
  V3d o2;         // -2 o
#endif
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
  r.tmin = RTOW_TMIN;
#ifdef RTOW_UNIT_RAYS
  r.unit = false;
  r.od = r.oo = 0.0;
  r.o2 = V3d{0, 0, 0};
#endif
  return r;
}
#ifdef RTOW_UNIT_RAYS
// the forms of a ray whose direction `du` has been normalised; `len` = |d| of the original direction
__device__ __forceinline__ RayForms make_unit_ray_forms(V3d o, V3d du, double time, double len) {
  RayForms r;
  r.o = r.o64 = o;
  r.d = r.d64 = du;
  r.time = r.time64 = time;
  r.a = r.a64 = 1.0;
  r.inv_a = r.inv_a64 = 1.0;
  r.tmin = RTOW_TMIN * len;
  r.unit = true;
  r.od = dot(o, du);
  r.oo = dot(o, o);
  r.o2 = V3d{-2.0 * o.x, -2.0 * o.y, -2.0 * o.z};
  return r;
}
#endif

// Tests primitives ids[first .. first+count) of a scene image against the ray (the same code
// as the STREAM kernel, so the accepted (t, primitive) is the same).  SMALL: the spheres are
// grid-cell members, tested in binary32 by the f32 build (no effect in the binary64 builds).
// SPEC (rtow_device.h, kSpec*): what the host knows about the scene — the classes that cannot occur and the list
// format that is there are told to the compiler, which drops their code (and the registers it would hold).
template <bool LDS, bool SMALL, int SPEC = 0>
__device__ __forceinline__ void leaf_test(const Image<LDS> &im, const DevScene &sc, ImgOffsets off,
                                          uint32_t first, uint32_t count, const RayForms &ray, Closest &best,
                                          uint32_t &nprim, int &last_id) {
#ifndef RTOW_REAL_F32
  if constexpr (SMALL) {
    if constexpr (SPEC == 1) __builtin_assume(off.fat != 0u && off.fat_stride == 48u);
    if constexpr (SPEC == 2) __builtin_assume(off.fat != 0u && off.fat_stride == 80u);
    if (off.fat != 0u) {  // wave-uniform: one round of LDS reads per entry instead of id -> record
      if (off.fat_stride == 48u) {
#ifdef RTOW_UNIT_RAYS
        if (ray.unit) {
          for (uint32_t k = 0; k < count; ++k) {
            const uint32_t e = off.fat + 48u * (first + k);
            uint4 hd = im.u4(e);  // id . k
            double2 p0 = im.d2(e + 16u);
            double cz = im.d1(e + 32u);
            asm volatile("" : "+v"(hd.x), "+v"(hd.z), "+v"(hd.w), "+v"(p0.x), "+v"(p0.y), "+v"(cz));
            const int id = (int)hd.x;
            if (id == last_id) continue;
            last_id = id;
            ++nprim;
            const double kk = __longlong_as_double((long long)(((unsigned long long)hd.w << 32) | hd.z));
            const double h = __builtin_fma(-p0.x, ray.d64.x, __builtin_fma(-p0.y, ray.d64.y, __builtin_fma(-cz, ray.d64.z, ray.od)));
            const double c = __builtin_fma(p0.x, ray.o2.x, __builtin_fma(p0.y, ray.o2.y, __builtin_fma(cz, ray.o2.z, ray.oo + kk)));
            const double disc = __builtin_fma(h, h, -c);
            if (disc >= 0.0) {
              const double sq = fast_sqrt_pos(disc);  // (disc == 0: NaN roots, no hit — see sphere_resolve)
              const double root1 = -h - sq, root2 = sq - h;
              const double root = root1 >= ray.tmin ? root1 : root2;
              if (root >= ray.tmin && root <= (double)best.t) {
                best.t = (real)root;
                best.prim = id;
              }
            }
          }
          return;
        }
#endif
        for (uint32_t k = 0; k < count; ++k) {
          const uint32_t e = off.fat + 48u * (first + k);
          const int id = (int)im.u32(e);
          double2 p0 = im.d2(e + 16u), p1 = im.d2(e + 32u);
          // (the record is read together with the id, not after the mailbox test: one LDS round trip per entry)
          asm volatile("" : "+v"(p0.x), "+v"(p0.y), "+v"(p1.x), "+v"(p1.y));
          if (id == last_id) continue;
          last_id = id;
          ++nprim;
          sphere_test<double>(ray.o64, ray.d64, ray.a64, ray.inv_a64, p0.x, p0.y, p1.x, p1.y, id, ray.tmin, best);
        }
      } else {  // entries with motion: centre(time) = c0 + time * delta (src/oo-primitives.h:64-66; delta = 0 if static)
        for (uint32_t k = 0; k < count; ++k) {
          const uint32_t e = off.fat + 80u * (first + k);
          const int id = (int)im.u32(e);
          double2 p0 = im.d2(e + 16u), p1 = im.d2(e + 32u), p2 = im.d2(e + 48u), p3 = im.d2(e + 64u);
          asm volatile("" : "+v"(p0.x), "+v"(p0.y), "+v"(p1.x), "+v"(p1.y), "+v"(p2.x), "+v"(p2.y), "+v"(p3.x));
          if (id == last_id) continue;
          last_id = id;
          ++nprim;
          // (a static sphere's delta is +0: c0 + time * 0 is c0 exactly, time being finite)
          const double cx = p0.x + ray.time64 * p1.y;
          const double cy = p0.y + ray.time64 * p2.x;
          const double cz = p1.x + ray.time64 * p2.y;
          sphere_test<double>(ray.o64, ray.d64, ray.a64, ray.inv_a64, cx, cy, cz, p3.x, id, ray.tmin, best);
        }
      }
      return;
    }
  }
#endif
  for (uint32_t k = 0; k < count; ++k) {
    const int id = (!SMALL && sc.leaf_direct) ? (int)(first + k) : (int)im.u32(off.ids + 4u * (first + k));
    if constexpr (SPEC == 1) __builtin_assume(id < sc.n_sph);
    if constexpr (SPEC == 2) __builtin_assume(id < sc.n_sph + sc.n_mov);
    // one-entry mailbox: a primitive spanning adjacent grid cells is listed in each of them
    if (id == last_id) continue;
    last_id = id;
    ++nprim;
    if (id < sc.n_sph) {
#ifdef RTOW_REAL_F32
      if constexpr (SMALL) {
        const float4 p = im.f4(off.sph32 + 16u * (uint32_t)id);
        sphere_test<float>(ray.o, ray.d, ray.a, ray.inv_a, p.x, p.y, p.z, p.w, id, (float)ray.tmin, best);
        continue;
      }
#endif
      const uint32_t r = off.sph + 32u * (uint32_t)id;
      const double2 p0 = im.d2(r), p1 = im.d2(r + 16u);
      sphere_test<double>(ray.o64, ray.d64, ray.a64, ray.inv_a64, p0.x, p0.y, p1.x, p1.y, id, ray.tmin, best);
    } else if (id < sc.n_sph + sc.n_mov) {
#ifdef RTOW_REAL_F32
      if constexpr (SMALL) {
        const uint32_t r = off.mov32 + 32u * (uint32_t)(id - sc.n_sph);
        const float4 p0 = im.f4(r), p1 = im.f4(r + 16u);  // c0xyz dx | dy dz r2 -
        sphere_test<float>(ray.o, ray.d, ray.a, ray.inv_a, p0.x + ray.time * p0.w, p0.y + ray.time * p1.x,
                           p0.z + ray.time * p1.y, p1.z, id, (float)ray.tmin, best);
        continue;
      }
#endif
      const uint32_t r = off.mov + 64u * (uint32_t)(id - sc.n_sph);
      const double2 p0 = im.d2(r), p1 = im.d2(r + 16u), p2 = im.d2(r + 32u), p3 = im.d2(r + 48u);
      const double cx = p0.x + ray.time64 * p1.y;
      const double cy = p0.y + ray.time64 * p2.x;
      const double cz = p1.x + ray.time64 * p2.y;
      sphere_test<double>(ray.o64, ray.d64, ray.a64, ray.inv_a64, cx, cy, cz, p3.x, id, ray.tmin, best);
    } else {
#ifdef RTOW_REAL_F32
      const uint32_t r = off.tri + 48u * (uint32_t)(id - sc.n_sph - sc.n_mov);
      const float4 q0 = im.f4(r), q1 = im.f4(r + 16u), q2 = im.f4(r + 32u);  // A e1 | e1 e2 | e2 n
      triangle_test<float>(ray.o, ray.d, V3{q0.x, q0.y, q0.z}, V3{q0.w, q1.x, q1.y}, V3{q1.z, q1.w, q2.x},
                           V3{q2.y, q2.z, q2.w}, id, (float)ray.tmin, best);
#else
      const uint32_t r = off.tri + 96u * (uint32_t)(id - sc.n_sph - sc.n_mov);
      const double2 q0 = im.d2(r), q1 = im.d2(r + 16u), q2 = im.d2(r + 32u), q3 = im.d2(r + 48u),
                    q4 = im.d2(r + 64u), q5 = im.d2(r + 80u);
      triangle_test<double>(ray.o64, ray.d64, V3d{q0.x, q0.y, q1.x}, V3d{q1.y, q2.x, q2.y}, V3d{q3.x, q3.y, q4.x},
                            V3d{q4.y, q5.x, q5.y}, id, ray.tmin, best);
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
  const ImgOffsets off = {sc.off_ids, sc.off_sph, sc.off_mov, sc.off_tri, 0u, 0u, sc.off_sph32, sc.off_mov32};
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

