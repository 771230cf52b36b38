// rtow_trace_hit.h — part of the trace kernels (included by rtow_trace_body.h inside namespace rtow::{anonymous};
// see that file for the execution model).  Sphere and triangle hit tests (templated on the arithmetic of the test) and the STREAM closest hit.
#pragma once
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
#if defined(RTOW_FAST_MATH)
    // (disc == 0 exactly — a tangent ray — gives NaN roots here, which fail both comparisons below: no hit, where the
    // strict build accepts the double root; the two instructions of a zero guard are not worth that measure-zero case)
    const T sq = fast_sqrt_pos(disc);
    // both roots at once (one reciprocal per ray): root1 <= root2, so a root1 beyond best.t rules out root2 as
    // well and "the nearer root if it is at or beyond tmin, else the farther one" is the reference's choice
    // (src/common-model.cpp:76-81) with one comparison less
    const T root1 = (-h - sq) * inv_a, root2 = (-h + sq) * inv_a;
    const T root = root1 >= tmin ? root1 : root2;
    if (root >= tmin && root <= (T)best.t) {
      best.t = (real)root;
      best.prim = id;
    }
#else
    (void)inv_a;
    const T sq = fast_sqrt(disc);
    T root = (-h - sq) / a;
    bool ok = true;
    if (root < tmin || root > (T)best.t) {
      root = (-h + sq) / a;
      if (root < tmin || root > (T)best.t) ok = false;
    }
    if (ok) {
      best.t = (real)root;
      best.prim = id;
    }
#endif
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
#if defined(RTOW_FAST_MATH) && !defined(RTOW_TRI_DIVIDE)
  // Fast builds: the same inequalities multiplied through by det (> 0 where they matter) — no reciprocal and no
  // scaled u, v, t for the lanes that miss, which is nearly all of them; the one division is paid by a hit.
  {
    const Vec3<T> ao = o - A;
    const Vec3<T> dao = cross(ao, d);
    const T ud = dot(e2, dao), vd = -dot(e1, dao), td = dot(ao, n);
    if (det >= T(1e-6) && td >= tmin * det && td <= (T)best.t * det && ud >= T(0.0) && vd >= T(0.0) && (ud + vd) <= det) {
      best.t = (real)(td * fast_rcp(det));
      best.prim = id;
    }
    return;
  }
#endif
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
// the tiled triangle loop: records per tile (32 x 96 B = 3 KB = three 16-byte pieces per lane), and from how many
// triangles on the host provisions LDS for it (rtow_capi.cpp)
constexpr uint32_t kStreamTile = 32u, kStreamTileBytes = kStreamTile * 96u;
// Always binary64 (in the f32 build the ray is widened once per segment): this kernel is for
// scenes of <= 16 primitives, which include the r = 1000 ground sphere.
// Every lane of the wave calls this together (`active`: the lane has a ray to advance): the tiled triangle loop
// below stages its tiles with all 64 lanes.
__device__ __forceinline__ Closest closest_hit_stream(const DevScene &sc, V3d o, V3d d, double time, bool active) {
  Closest best;
  best.t = (real)__builtin_huge_val();  // tmax = +inf, src/render.cpp:34
  best.prim = -1;
  const double tmin = RTOW_TMIN;
  const double a = dot(d, d);
  const double inv_a = fast_rcp(a);  // used by the fast build only
  if (active) {
    cdptr g = (cdptr)sc.sph;
    const int n = sc.n_sph;
#pragma unroll 4
    for (int i = 0; i < n; ++i) {
      sphere_test<double>(o, d, a, inv_a, g[4 * i + 0], g[4 * i + 1], g[4 * i + 2], g[4 * i + 3], i, tmin, best);
    }
  }
  if (active) {
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
  if (sc.stream_tile_lds != 0u) {
    // Triangles in LDS-STAGED TILES (a scene of kStreamTileMin triangles or more; the host gives every wave
    // 2 x kStreamTileBytes of LDS).  The scalar-load loop below waits for memory once per record: scalar loads return
    // out of order, so their one counter can only be waited down to zero and a record's loads get exactly one test of
    // cover (57 % of the wave cycles inside s_waitcnt on the 96,800-triangle stress, profiles/r05_pmc_stream.json).
    // Here the 64 lanes of the wave fetch a tile of 32 records with three COALESCED 16-byte loads each (1 KB per
    // instruction, the records as they lie in HBM: `tri`, 96 B apiece) while the previous tile is being tested, park
    // it in the wave's own slice of LDS (double-buffered: no barrier, a wave's LDS operations execute in order) and
    // read every record back with six broadcast ds_read_b128 — a wave-uniform address, conflict-free — whose latency
    // the next record's reads cover.  Same test, same operands, same order of primitives: the image cannot change.
    const int n = sc.n_tri;
    const int base = sc.n_sph + sc.n_mov;
    if (__ballot(active) != 0ull) {  // (wave-uniform)
      const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      const uint32_t slice = (threadIdx.x >> 6) * (2u * kStreamTileBytes);
      const unsigned char *src = (const unsigned char *)sc.tri;
      const uint32_t total = (uint32_t)n * 96u;
      const uint32_t ntiles = ((uint32_t)n + kStreamTile - 1u) / kStreamTile;
      vu4 r0, r1, r2;  // this lane's three 16-byte pieces of the tile in flight
      auto fetch = [&](uint32_t tile) {
        const uint32_t p = tile * kStreamTileBytes + 16u * lane;
        const vu4 z = {0u, 0u, 0u, 0u};
        r0 = p < total ? glb_read<vu4>(src, p) : z;
        r1 = p + 1024u < total ? glb_read<vu4>(src, p + 1024u) : z;
        r2 = p + 2048u < total ? glb_read<vu4>(src, p + 2048u) : z;
      };
      auto park = [&](uint32_t buf) {
        const uint32_t q = slice + buf * kStreamTileBytes + 16u * lane;
        lds_write<vu4>(q, r0);
        lds_write<vu4>(q + 1024u, r1);
        lds_write<vu4>(q + 2048u, r2);
      };
      fetch(0u);
      park(0u);
      for (uint32_t tile = 0; tile < ntiles; ++tile) {
        if (tile + 1u < ntiles) fetch(tile + 1u);  // in flight while this tile is tested
        const uint32_t tb = slice + (tile & 1u) * kStreamTileBytes;
        const int first = (int)(tile * kStreamTile);
        const int cnt = n - first < (int)kStreamTile ? n - first : (int)kStreamTile;
        // Two records in flight (register sets a / b, alternating): the reads of record k + 1 are issued before
        // record k is tested, so the LDS latency of one record hides behind the arithmetic of the other.  Of a record's
        // six 16-byte pieces the test needs four at once (A and n: the determinant and the ray parameter, where nearly
        // every triangle is rejected); the other two (e1, e2) are read by the lanes that get that far.
        struct Rec {
          vd2 p0, p1, p4, p5;
          uint32_t e;
        };
        auto load = [&](int k) {
          Rec r;
          r.e = tb + 96u * (uint32_t)k;
          r.p0 = lds_read<vd2>(r.e), r.p1 = lds_read<vd2>(r.e + 16u), r.p4 = lds_read<vd2>(r.e + 64u),
          r.p5 = lds_read<vd2>(r.e + 80u);
          return r;
        };
        auto test = [&](const Rec &r, int k) {
          if (active) {
            const vd2 p2 = lds_read<vd2>(r.e + 32u), p3 = lds_read<vd2>(r.e + 48u);  // (sunk into the branch that uses them)
            triangle_test<double>(o, d, V3d{r.p0.x, r.p0.y, r.p1.x}, V3d{r.p1.y, p2.x, p2.y}, V3d{p3.x, p3.y, r.p4.x},
                                  V3d{r.p4.y, r.p5.x, r.p5.y}, base + first + k, tmin, best);
          }
        };
        Rec a = load(0);
        int k = 0;
        for (; k + 1 < cnt; k += 2) {
          const Rec b = load(k + 1);
          test(a, k);
          a = load(k + 2 < cnt ? k + 2 : cnt - 1);  // (the last round reloads a record it does not use: in the tile)
          test(b, k + 1);
        }
        if (k < cnt) test(a, k);
        if (tile + 1u < ntiles) park((tile + 1u) & 1u);  // (the buffer tile - 1 was tested from)
      }
    }
  } else if (active) {
    // Triangles, software-pipelined over records at a 128-byte stride (round 5).  The PMC pass of the 96,800-triangle
    // streaming stress (profiles/r05_pmc_stream_before.json) showed VALU issue utilisation 0.28 with 57 % of the wave
    // cycles inside s_waitcnt: a 96-byte record at a 96-byte stride is only 32-byte aligned, so it arrived as five
    // or six scalar loads, and the compiler waited (s_waitcnt lgkmcnt(0)) right behind every record's loads because
    // the test's early-out is a branch.  Now a record is ONE x16 and ONE x8 load, and the loads of the next record are
    // issued before the current one is tested (two register sets, alternating: no copies).  Same operations on the
    // same operands in the same order: the image cannot change.
    cdptr g = (cdptr)sc.tri16;
    const int n = sc.n_tri;
    const int base = sc.n_sph + sc.n_mov;
    if (n > 0) {
      double a[12], b[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) a[k] = g[k];
      int i = 0;
      for (; i + 1 < n; i += 2) {
#pragma unroll
        for (int k = 0; k < 12; ++k) b[k] = g[16 * (i + 1) + k];
        triangle_test<double>(o, d, V3d{a[0], a[1], a[2]}, V3d{a[3], a[4], a[5]}, V3d{a[6], a[7], a[8]},
                              V3d{a[9], a[10], a[11]}, base + i, tmin, best);
        const int j = i + 2 < n ? i + 2 : n - 1;  // (the last round reloads a record it does not use: in bounds)
#pragma unroll
        for (int k = 0; k < 12; ++k) a[k] = g[16 * j + k];
        triangle_test<double>(o, d, V3d{b[0], b[1], b[2]}, V3d{b[3], b[4], b[5]}, V3d{b[6], b[7], b[8]},
                              V3d{b[9], b[10], b[11]}, base + i + 1, tmin, best);
      }
      if (i < n)
        triangle_test<double>(o, d, V3d{a[0], a[1], a[2]}, V3d{a[3], a[4], a[5]}, V3d{a[6], a[7], a[8]},
                              V3d{a[9], a[10], a[11]}, base + i, tmin, best);
    }
  }
  return best;
}

