// rtow_oracle.cpp — CPU oracle: a restatement of the reference's sample loop.
//
// TEST INFRASTRUCTURE ONLY (see rtow_oracle.h).  Nothing on the product path may
// include, link or call this file.
//
// Every function cites the reference lines it restates (paths are relative to
// the reference checkout, joaotavora/raytracing-one-weekend @ v1).  The
// arithmetic follows the reference expression by expression — same operand
// order, same comparisons, no FMA contraction (build with -ffp-contract=off) —
// because the first gate is a byte-identical PPM against the md5 sums recorded
// in SURVEY.md §8c.  glm (pinned glm/cci.20220420 in conanfile.txt:2) is not in
// this image; the handful of glm functions the path uses are restated from
// their published definitions in the `glm-like` block below.
//
// Two RNG policies share the same integrator:
//   MtGlobal  — the reference's process-global default-seeded std::mt19937 and
//               libstdc++'s uniform_real_distribution mapping
//               (src/random-utils.cpp:6-13): two 32-bit draws per double.
//   PhiloxDraw— the device path's counter-based stream.  Random numbers are drawn
//               in REQUESTS, one Philox4x32-7 block each, counter (r, s, p, 0), key =
//               seed, a pure function of (seed, pixel p, sample s, request r):
//               request 0 of a sample = pixel jitter, shutter time and the lens point,
//               request 1 + b = bounce b's point of the unit ball and its dielectric coin.
//               Under this policy the lens point and the "unit vector" are drawn DIRECTLY from
//               their block (same distributions as the reference's rejection loops, which
//               MtGlobal keeps: see struct PhiloxDraw and the overloads of
//               random_in_unit_disk / random_in_unit_sphere).

#include "rtow_oracle.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------- glm-like ---
// vec3 = glm::dvec3 (src/vec3.h:6-8).  Definitions restated from glm 0.9.9:
//   dot(a,b)      = a.x*b.x + a.y*b.y + a.z*b.z   (left-to-right adds)
//   cross(x,y)    = (x.y*y.z - y.y*x.z, x.z*y.x - y.z*x.x, x.x*y.y - y.x*x.y)
//   normalize(v)  = v * (1 / sqrt(dot(v,v)))      (multiply by reciprocal)
//   reflect(I,N)  = I - N * dot(N,I) * 2
//   refract(I,N,e): d = dot(N,I); k = 1 - e*e*(1 - d*d);
//                   k >= 0 ? e*I - (e*d + sqrt(k))*N : 0
struct V3 {
  double x, y, z;
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 x, V3 y) {
  return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y};
}
inline double length(V3 v) { return std::sqrt(dot(v, v)); }
inline V3 normalize(V3 v) { return v * (1.0 / std::sqrt(dot(v, v))); }
inline V3 reflect(V3 I, V3 N) { return I - N * dot(N, I) * 2.0; }
inline V3 refract(V3 I, V3 N, double eta) {
  double d = dot(N, I);
  double k = 1.0 - eta * eta * (1.0 - d * d);
  if (k >= 0.0) return eta * I - (eta * d + std::sqrt(k)) * N;
  return {0, 0, 0};
}
inline double comp(V3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }
inline V3 load3(const double *p) { return {p[0], p[1], p[2]}; }
inline void store3(double *p, V3 v) {
  p[0] = v.x;
  p[1] = v.y;
  p[2] = v.z;
}

// ------------------------------------------------------------------- RNGs ---
// std::mt19937 (32-bit Mersenne twister, default seed 5489) written out.
struct Mt19937 {
  uint32_t s[624];
  int idx;
  Mt19937() { seed(5489u); }
  void seed(uint32_t v) {
    s[0] = v;
    for (int i = 1; i < 624; ++i) s[i] = 1812433253u * (s[i - 1] ^ (s[i - 1] >> 30)) + (uint32_t)i;
    idx = 624;
  }
  void twist() {
    for (int i = 0; i < 624; ++i) {
      uint32_t y = (s[i] & 0x80000000u) | (s[(i + 1) % 624] & 0x7fffffffu);
      uint32_t v = s[(i + 397) % 624] ^ (y >> 1);
      if (y & 1u) v ^= 0x9908b0dfu;
      s[i] = v;
    }
    idx = 0;
  }
  uint32_t next() {
    if (idx >= 624) twist();
    uint32_t y = s[idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  }
};

// libstdc++ generate_canonical<double,53>(mt19937): two draws, the first is the
// low word: (u0 + u1*2^32) / 2^64, each step rounded in double; a result that
// rounds up to 1.0 is replaced by nextafter(1,0).
inline double canonical_from_words(uint32_t w0, uint32_t w1) {
  double sum = (double)w0 + (double)w1 * 4294967296.0;
  double r = sum / 18446744073709551616.0;
  if (r >= 1.0) r = 0x1.fffffffffffffp-1;
  return r;
}

Mt19937 &the_generator() {  // src/random-utils.cpp:6-9 (one global stream)
  static Mt19937 g;
  return g;
}

struct MtGlobal {
  uint64_t ndraws = 0;
  double canonical() {
    Mt19937 &g = the_generator();
    uint32_t w0 = g.next();
    uint32_t w1 = g.next();
    ++ndraws;
    return canonical_from_words(w0, w1);
  }
  void begin_sample(uint32_t, uint32_t) {}
  // the global stream has no request structure
  void request_jitter() {}
  void request_disk() {}
  void request_time() {}
  void request_coin() {}
  void request_scatter() {}
  void begin_scatter() {}
};

// Philox4x32-R (Salmon et al., SC'11).  The device path uses R = 7, the fastest member
// of the family that the paper reports as Crush-resistant; R = 10 is kept for the
// published known-answer vectors.
constexpr int kPhiloxRounds = 7;
inline void philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4], int rounds) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < rounds; ++r) {
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
  out[0] = c0;
  out[1] = c1;
  out[2] = c2;
  out[3] = c3;
}

// One request = one Philox block = four 32-bit words w0..w3 = everything the request needs (round 5: no request draws
// a second block; the layouts are pinned word by word in tests/test_oracle_units.py and restated by the device in
// raytracing-one-weekend_amd/csrc/rtow_trace_rng.h):
//   request 0 of a sample     : u, v, time = top 21 bits of w0, w1, w2 (* 2^-21); the two 32-bit uniforms of the lens
//                               point = w3 and the 11+11+10 low bits of w0..w2
//   request 1 + b (bounce b)  : z = top 24 bits of w0, azimuth = top 24 bits of w1, three radius uniforms = the two
//                               halves of w2 and the low half of w3 (16 bits each), dielectric coin (32 bits) = the high
//                               half of w3 and the low bytes of w0, w1
// The reference draws its lens point and its "unit vector" by REJECTION (random_in_unit_disk, random_in_unit_sphere:
// src/random-utils.cpp:23-41).  This policy — the device's — draws the same two DISTRIBUTIONS directly, one block per
// request (random_in_unit_disk / random_in_unit_sphere overloads below): uniform on the unit disk as (sqrt(U1), 2 pi U2)
// in polar coordinates; uniform in the positive octant of the unit ball (what the loop over [0,1)^3 returns) as radius
// max(U1, U2, U3) times a direction uniform on the octant of the sphere (z = U4, azimuth (pi/2) U5).  The reference's own
// loops are the MtGlobal policy's, which reproduces the reference's images byte for byte (tests/test_oracle_golden.py);
// what ties the two policies together is the statistical test T3 (tests/test_oracle_units.py), as it always was.
struct PhiloxDraw {
  uint64_t seed = 0;
  uint32_t pixel = 0, sample = 0, r = 0;
  uint64_t ndraws = 0;
  double buf[3];
  int have = 0, pos = 0;
  double stash_time = 0.0;     // third value of the jitter request
  uint32_t lens_a = 0, lens_b = 0;  // the lens point's two 32-bit uniforms (same block)
  uint32_t w[4] = {0, 0, 0, 0};     // the words of the bounce's block (request_scatter)
  void begin_sample(uint32_t p, uint32_t s) {
    pixel = p;
    sample = s;
    r = 0;
    have = pos = 0;
  }
  void block(uint32_t req, uint32_t o[4]) const {
    uint32_t ctr[4] = {req, sample, pixel, 0u};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    philox4x32(ctr, key, o, kPhiloxRounds);
  }
  // The first block of a sample: pixel jitter and shutter time, 21 bits each (the top bits of words 0..2), and the two
  // uniforms of the lens point, 32 bits each (word 3 and the 11+11+10 low bits left in words 0..2).
  void request_jitter() {
    uint32_t o[4];
    block(r++, o);
    const double s21 = 0x1p-21;
    buf[0] = (double)(o[0] >> 11) * s21;
    buf[1] = (double)(o[1] >> 11) * s21;
    stash_time = (double)(o[2] >> 11) * s21;
    lens_a = o[3];
    lens_b = (o[0] & 0x7ffu) | ((o[1] & 0x7ffu) << 11) | ((o[2] & 0x3ffu) << 22);
    have = 2;
    pos = 0;
  }
  void request_disk() {}  // (the lens point's uniforms came with the jitter block)
  void request_time() {
    buf[0] = stash_time;
    have = 1;
    pos = 0;
  }
  void begin_scatter() {}
  static double coin_of(const uint32_t o[4]) {
    return (double)(((o[3] >> 16) << 16) | ((o[0] & 0xffu) << 8) | (o[1] & 0xffu)) * 0x1p-32;
  }
  void request_coin() {  // peek at the bounce's block (request_scatter consumes it)
    uint32_t o[4];
    block(r, o);
    buf[0] = coin_of(o);
    have = 1;
    pos = 0;
  }
  void request_scatter() { block(r++, w); }
  double canonical() {
    if (pos >= have) std::abort();  // a draw outside a request: oracle bug
    ++ndraws;
    return buf[pos++];
  }
};

// sin on [0, pi/2]: x (c0 + x^2 (c1 + x^2 (c2 + x^2 (c3 + x^2 c4)))), a degree-9 Chebyshev fit, |error| < 7e-9 — the
// device's polynomial, same coefficients, same order of operations (rtow_trace_rng.h, sin_quarter); cos x = sin(pi/2 - x)
inline double sin_quarter(double x) {
  const double x2 = x * x;
  return x * (0x1.ffffffdb33084p-1 +
              x2 * (-0x1.555549a9260fdp-3 + x2 * (0x1.110eb1f04c8ffp-7 + x2 * (-0x1.9f6d0201a288bp-13 + x2 * 0x1.5da8d4e70fe23p-19))));
}
constexpr double kHalfPi = 1.5707963267948966;

// src/random-utils.cpp:11-13 — uniform_real_distribution{a,b}: canonical*(b-a)+a
template <class R>
inline double random_double(R &rng, double a = 0.0, double b = 1.0) {
  return rng.canonical() * (b - a) + a;
}
// src/random-utils.cpp:19-22 — brace-init: x, y, z in draw order
template <class R>
inline V3 random_vec3(R &rng, double lo = 0.0, double hi = 1.0) {
  double x = random_double(rng, lo, hi);
  double y = random_double(rng, lo, hi);
  double z = random_double(rng, lo, hi);
  return {x, y, z};
}
// src/random-utils.cpp:23-33 — samples [0,1)^3 (the defaults), rejects |v|^2>=1,
// and "random_unit_vector" returns it WITHOUT normalising.
template <class R>
inline V3 random_in_unit_sphere(R &rng) {
  rng.begin_scatter();
  for (;;) {
    rng.request_scatter();
    V3 v = random_vec3(rng);
    if (dot(v, v) >= 1) continue;
    return v;
  }
}
// The Philox policy's direct form of the same distribution (uniform in the positive octant of the unit ball): radius =
// the largest of three uniforms, direction uniform on the octant of the sphere.  One block, no loop.
inline V3 random_in_unit_sphere(PhiloxDraw &rng) {
  rng.request_scatter();
  rng.ndraws += 5;
  const uint32_t *o = rng.w;
  const double z = (double)(o[0] >> 8) * 0x1p-24;
  const double phi = (double)(o[1] >> 8) * (0x1p-24 * kHalfPi);
  const uint32_t rm = std::max(std::max(o[2] & 0xffffu, o[2] >> 16), o[3] & 0xffffu);
  const double r = (double)rm * 0x1p-16;
  const double sn = sin_quarter(phi), cs = sin_quarter(kHalfPi - phi);
  const double rs = r * std::sqrt(1.0 - z * z);
  return {rs * cs, rs * sn, r * z};
}
template <class R>
inline V3 random_unit_vector(R &rng) {
  return random_in_unit_sphere(rng);
}
// src/random-utils.cpp:34-41 — vec3(rd(-1,1), rd(-1,1), 0) is a paren-init
// (function-call arguments): g++ evaluates them right to left, so y takes the
// first draw and x the second (measured).
template <class R>
inline V3 random_in_unit_disk(R &rng) {
  for (;;) {
    rng.request_disk();
    double py = random_double(rng, -1, 1);
    double px = random_double(rng, -1, 1);
    V3 p{px, py, 0};
    if (dot(p, p) >= 1) continue;
    return p;
  }
}

// The Philox policy's direct form: uniform on the unit disk in polar coordinates (radius sqrt(U1), angle 2 pi U2 — the
// top two bits of U2 choose the quadrant, the other thirty the angle inside it).
inline V3 random_in_unit_disk(PhiloxDraw &rng) {
  rng.ndraws += 2;
  const uint32_t b = rng.lens_b;
  const double rho = std::sqrt((double)rng.lens_a * 0x1p-32);
  const double th = (double)(b & 0x3fffffffu) * (0x1p-30 * kHalfPi);
  const double sn = sin_quarter(th), cs = sin_quarter(kHalfPi - th);
  const uint32_t q = b >> 30;
  const double cx = (q & 1u) ? sn : cs, sy = (q & 1u) ? cs : sn;
  const double px = rho * ((q == 1u || q == 2u) ? -cx : cx);
  const double py = rho * (q >= 2u ? -sy : sy);
  return {px, py, 0};
}

// ------------------------------------------------------------------ model ---
struct Ray {  // src/common-model.h:17-33
  V3 o, d;
  double time;
  V3 at(double t) const { return o + d * t; }
};

struct Material {
  int kind;
  V3 albedo;
  double fuzz, ir;
};

struct Prim {  // Sphere / MovingSphere / Triangle, src/oo-primitives.h:26-88
  int kind;
  V3 a, b, c;  // sphere: a=center; moving: a=center0, b=center1; triangle: a,b,c
  double radius;
  int mat;
  int cls_index;
};

struct Hit {  // src/common-model.h:42-61
  int prim;
  V3 where;
  double at;
  V3 normal;
  bool front;
};

struct Aabb {  // src/common-model.h:63-89 — default-constructed = two zero points
  V3 mn{0, 0, 0}, mx{0, 0, 0};
  double volume() const { return (mx.x - mn.x) * (mx.y - mn.y) * (mx.z - mn.z); }
};

// src/common-model.h:71-84
inline bool aabb_hit(const Aabb &b, const Ray &r, double t_min, double t_max) {
  for (int a = 0; a < 3; a++) {
    double inv = 1.0 / comp(r.d, a);  // 1.0F promoted to double
    double t0 = (comp(b.mn, a) - comp(r.o, a)) * inv;
    double t1 = (comp(b.mx, a) - comp(r.o, a)) * inv;
    if (inv < 0.0) std::swap(t0, t1);
    t_min = t0 > t_min ? t0 : t_min;
    t_max = t1 < t_max ? t1 : t_max;
    if (t_max <= t_min) return false;
  }
  return true;
}

// src/common-model.cpp:185-195
inline Aabb surrounding_box(const Aabb &b0, const Aabb &b1) {
  Aabb r;
  r.mn = {std::fmin(b0.mn.x, b1.mn.x), std::fmin(b0.mn.y, b1.mn.y), std::fmin(b0.mn.z, b1.mn.z)};
  r.mx = {std::fmax(b0.mx.x, b1.mx.x), std::fmax(b0.mx.y, b1.mx.y), std::fmax(b0.mx.z, b1.mx.z)};
  return r;
}

// MovingSphere::center(time), src/oo-primitives.h:64-66 with t0=0, t1=1
inline V3 moving_center(const Prim &p, double time) {
  return p.a + ((time - 0.0) / (1.0 - 0.0)) * (p.b - p.a);
}

inline Aabb bounding_box(const Prim &p) {
  Aabb r;
  if (p.kind == RTOW_PRIM_SPHERE) {  // src/common-model.cpp:168-171
    V3 rv{p.radius, p.radius, p.radius};
    r.mn = p.a - rv;
    r.mx = p.a + rv;
  } else if (p.kind == RTOW_PRIM_MOVING_SPHERE) {  // src/common-model.cpp:197-207
    V3 rv{p.radius, p.radius, p.radius};
    Aabb b0, b1;
    V3 c0 = moving_center(p, 0.0), c1 = moving_center(p, 1.0);
    b0.mn = c0 - rv;
    b0.mx = c0 + rv;
    b1.mn = c1 - rv;
    b1.mx = c1 + rv;
    r = surrounding_box(b0, b1);
  } else {  // src/common-model.cpp:127-134: corners pass through glm::vec3 = float
    auto f = [](double v) { return (double)(float)v; };
    r.mn = {f(std::min({p.a.x, p.b.x, p.c.x})), f(std::min({p.a.y, p.b.y, p.c.y})),
            f(std::min({p.a.z, p.b.z, p.c.z}))};
    r.mx = {f(std::max({p.a.x, p.b.x, p.c.x})), f(std::max({p.a.y, p.b.y, p.c.y})),
            f(std::max({p.a.z, p.b.z, p.c.z}))};
  }
  return r;
}

// src/common-model.cpp:64-91
inline bool sphere_hit_helper(const Ray &r, double tmin, double tmax, V3 center, double radius,
                              int prim, Hit &out) {
  V3 oc = r.o - center;
  double a = dot(r.d, r.d);
  double h = dot(oc, r.d);
  double c = dot(oc, oc) - radius * radius;
  double discriminant = h * h - a * c;
  if (discriminant < 0.0) return false;
  double root = (-h - std::sqrt(discriminant)) / a;
  if (root < tmin || root > tmax) {
    root = (-h + std::sqrt(discriminant)) / a;
    if (root < tmin || root > tmax) return false;
  }
  V3 hitpoint = r.at(root);
  V3 normal = normalize(hitpoint - center);
  bool front = (dot(r.d, normal) < 0) ^ (radius < 0);
  normal = front ? normal : -normal;
  out = Hit{prim, hitpoint, root, normal, front};
  return true;
}

// src/common-model.cpp:103-125
inline bool triangle_hit(const Ray &r, double tmin, double tmax, V3 A, V3 B, V3 C, int prim,
                         Hit &out) {
  V3 e1 = B - A;
  V3 e2 = C - A;
  V3 n = cross(e1, e2);
  double det = -dot(r.d, n);
  double invdet = 1.0 / det;
  V3 ao = r.o - A;
  V3 dao = cross(ao, r.d);
  double u = dot(e2, dao) * invdet;
  double v = -dot(e1, dao) * invdet;
  double t = dot(ao, n) * invdet;
  if (det >= 1e-6 && t >= tmin && t <= tmax && u >= 0.0 && v >= 0.0 && (u + v) <= 1.0) {
    out = Hit{prim, r.at(t), t, n, true};
    return true;
  }
  return false;
}

struct Counters {
  uint64_t segments = 0, prim_tests = 0, node_tests = 0;
};

// optional ray log (single-threaded renders only): one record of 12 doubles per segment
// {pixel, sample, segment index, ox,oy,oz, dx,dy,dz, time, t_hit (inf = miss), prim}
double *g_raylog = nullptr;
uint64_t g_raylog_cap = 0, g_raylog_n = 0;
thread_local uint32_t g_cur_pixel = 0, g_cur_sample = 0, g_cur_seg = 0;

struct World {
  std::vector<Prim> prims;  // insertion order, then permuted in place by the BVH build
  std::vector<Material> mats;
  rtow_camera_t cam;
};

inline bool prim_hit(const World &w, int pi, const Ray &r, double tmin, double tmax, Hit &out) {
  const Prim &p = w.prims[pi];
  switch (p.kind) {
    case RTOW_PRIM_SPHERE:  // src/common-model.cpp:93-96
      return sphere_hit_helper(r, tmin, tmax, p.a, p.radius, pi, out);
    case RTOW_PRIM_MOVING_SPHERE:  // src/common-model.cpp:98-101
      return sphere_hit_helper(r, tmin, tmax, moving_center(p, r.time), p.radius, pi, out);
    default:
      return triangle_hit(r, tmin, tmax, p.a, p.b, p.c, pi, out);
  }
}

// ------------------------------------------------------------------- BVH ----
struct BvhNode {  // src/render.cpp:22-34
  int lo = 0, hi = 0;  // leaf: [lo,hi) into world.prims; inner: empty range
  Aabb box;
  int left = -1, right = -1;
};

struct FastTree;
struct Bvh {
  std::vector<BvhNode> nodes;
  int root = -1;
  const FastTree *fast = nullptr;  // checker's own tree (orc_render_ex accel = 1), see below
};

// src/render.cpp:73-110 — the primitive array itself is sorted in place.
int bvh_build(World &w, Bvh &bvh, int lo, int hi) {
  int me = (int)bvh.nodes.size();
  bvh.nodes.push_back(BvhNode{});
  int n = hi - lo;
  if (n >= 1 && n <= 6) {
    Aabb box;  // starts as the default box (two zero points), src/render.cpp:76-78
    for (int i = lo; i < hi; ++i) box = surrounding_box(box, bounding_box(w.prims[i]));
    bvh.nodes[me].lo = lo;
    bvh.nodes[me].hi = hi;
    bvh.nodes[me].box = box;
    return me;
  }
  Aabb pbeg = bounding_box(w.prims[lo]);
  Aabb pend = bounding_box(w.prims[hi - 1]);
  V3 delta = pend.mn - pbeg.mn;
  int axis;
  if (std::fabs(delta.x) > std::fabs(delta.y)) {
    axis = (std::fabs(delta.x) > std::fabs(delta.z)) ? 0 : 2;
  } else {
    axis = (std::fabs(delta.y) > std::fabs(delta.z)) ? 1 : 2;
  }
  // std::sort (libstdc++ introsort) — unstable; ties resolve as in the reference
  // build because the comparator, the algorithm and the input order are the same.
  std::sort(w.prims.begin() + lo, w.prims.begin() + hi, [axis](const Prim &p1, const Prim &p2) {
    return comp(bounding_box(p1).mn, axis) < comp(bounding_box(p2).mn, axis);
  });
  int leftn = n / 2;
  int l = bvh_build(w, bvh, lo, lo + leftn);
  int r = bvh_build(w, bvh, lo + leftn, hi);
  bvh.nodes[me].left = l;
  bvh.nodes[me].right = r;
  bvh.nodes[me].box = surrounding_box(bvh.nodes[l].box, bvh.nodes[r].box);
  return me;
}

// src/render.cpp:36-50
double stupid_volume(const Bvh &b, int ni) {
  const BvhNode &n = b.nodes[ni];
  double myown = n.box.volume();
  double childrens = 0;
  if (n.left >= 0 && n.right >= 0) {
    myown -= b.nodes[n.left].box.volume();
    myown -= b.nodes[n.right].box.volume();
    childrens += stupid_volume(b, n.left);
    childrens += stupid_volume(b, n.right);
  } else {
    myown = 0;
  }
  if (myown < 0)
    myown = -myown;
  else
    myown = 0;
  return myown + childrens;
}

// src/render.cpp:52-71
bool bvh_hit(const World &w, const Bvh &b, int ni, const Ray &r, double tmin, double tmax, Hit &out,
             Counters &cnt) {
  const BvhNode &n = b.nodes[ni];
  ++cnt.node_tests;
  if (!aabb_hit(n.box, r, tmin, tmax)) return false;
  if (n.hi > n.lo) {
    bool any = false;
    double upper = tmax;
    for (int i = n.lo; i < n.hi; ++i) {
      Hit probe;
      ++cnt.prim_tests;
      if (prim_hit(w, i, r, tmin, upper, probe)) {
        out = probe;
        any = true;
        upper = probe.at;
      }
    }
    return any;
  }
  Hit lh, rh;
  bool l = bvh_hit(w, b, n.left, r, tmin, tmax, lh, cnt);
  bool rr = bvh_hit(w, b, n.right, r, tmin, l ? lh.at : tmax, rh, cnt);
  if (rr) {
    out = rh;
    return true;
  }
  if (l) {
    out = lh;
    return true;
  }
  return false;
}

// ------------------------------------------------ checker's own tree (not the reference's) ---
// The reference's median-split tree culls poorly on big meshes (the 96,800-triangle stand-in for
// BASELINE config C5 costs ~4,200 box + ~11,000 triangle tests per segment), which puts oracle
// renders at BASELINE sizes out of reach.  `orc_render_ex(..., accel = 1)` therefore finds the
// closest hit with a tree of the checker's own — exact f64 bounds (padded), SAH-binned build,
// near-child-first descent — while every primitive test is still prim_hit() above, in ascending
// position order inside a leaf, with the same shrinking [tmin, tmax].  The closest hit does not
// depend on the tree (exact ties in t aside), so the image is the reference-tree image bit for
// bit; tests/test_oracle_units.py checks that on every scene class before any GPU test leans on it.
struct FastNode {
  double mn[3], mx[3];
  int left = -1;   // inner: children left, left + 1
  int lo = 0, hi = 0;  // leaf: positions [lo, hi) of FastTree::order
};
struct FastTree {
  std::vector<FastNode> nodes;
  std::vector<int> order;  // positions into World::prims, leaf ranges ascending inside a leaf
};

inline void fast_bounds(const World &w, int pi, double mn[3], double mx[3]) {
  const Prim &p = w.prims[pi];
  V3 lo, hi;
  if (p.kind == RTOW_PRIM_TRIANGLE) {
    lo = {std::min({p.a.x, p.b.x, p.c.x}), std::min({p.a.y, p.b.y, p.c.y}), std::min({p.a.z, p.b.z, p.c.z})};
    hi = {std::max({p.a.x, p.b.x, p.c.x}), std::max({p.a.y, p.b.y, p.c.y}), std::max({p.a.z, p.b.z, p.c.z})};
  } else {
    const double r = std::fabs(p.radius);
    V3 c0 = p.a, c1 = p.a;
    if (p.kind == RTOW_PRIM_MOVING_SPHERE) {  // ray times are drawn from [cam.t0, cam.t1]
      const double ta = std::min(w.cam.t0, w.cam.t1) - 1e-6, tb = std::max(w.cam.t0, w.cam.t1) + 1e-6;
      c0 = moving_center(p, ta);
      c1 = moving_center(p, tb);
    }
    lo = {std::min(c0.x, c1.x) - r, std::min(c0.y, c1.y) - r, std::min(c0.z, c1.z) - r};
    hi = {std::max(c0.x, c1.x) + r, std::max(c0.y, c1.y) + r, std::max(c0.z, c1.z) + r};
  }
  const double l[3] = {lo.x, lo.y, lo.z}, h[3] = {hi.x, hi.y, hi.z};
  for (int k = 0; k < 3; ++k) {
    const double pad = 1e-9 * (1.0 + std::max(std::fabs(l[k]), std::fabs(h[k])));
    mn[k] = l[k] - pad;
    mx[k] = h[k] + pad;
  }
}

inline void fast_build(const World &w, FastTree &t) {
  const int n = (int)w.prims.size();
  std::vector<double> bmn((size_t)n * 3), bmx((size_t)n * 3), cen((size_t)n * 3);
  for (int i = 0; i < n; ++i) {
    fast_bounds(w, i, &bmn[(size_t)i * 3], &bmx[(size_t)i * 3]);
    for (int k = 0; k < 3; ++k) cen[(size_t)i * 3 + k] = 0.5 * (bmn[(size_t)i * 3 + k] + bmx[(size_t)i * 3 + k]);
  }
  t.order.resize(n);
  for (int i = 0; i < n; ++i) t.order[i] = i;
  t.nodes.clear();
  t.nodes.push_back(FastNode{});
  struct Job { int node, lo, hi; };
  std::vector<Job> jobs{{0, 0, n}};
  auto area = [](const double *a, const double *b) {
    const double dx = b[0] - a[0], dy = b[1] - a[1], dz = b[2] - a[2];
    return (dx >= 0 && dy >= 0 && dz >= 0) ? dx * dy + dy * dz + dz * dx : 0.0;
  };
  while (!jobs.empty()) {
    const Job jb = jobs.back();
    jobs.pop_back();
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    double cmn[3] = {INFINITY, INFINITY, INFINITY}, cmx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = jb.lo; i < jb.hi; ++i) {
      const int p = t.order[i];
      for (int k = 0; k < 3; ++k) {
        mn[k] = std::min(mn[k], bmn[(size_t)p * 3 + k]);
        mx[k] = std::max(mx[k], bmx[(size_t)p * 3 + k]);
        cmn[k] = std::min(cmn[k], cen[(size_t)p * 3 + k]);
        cmx[k] = std::max(cmx[k], cen[(size_t)p * 3 + k]);
      }
    }
    FastNode nd;
    std::memcpy(nd.mn, mn, sizeof mn);
    std::memcpy(nd.mx, mx, sizeof mx);
    const int cnt = jb.hi - jb.lo;
    int mid = -1;
    if (cnt > 2) {
      constexpr int kBins = 12;
      double best = INFINITY;
      int best_ax = -1, best_k = -1;
      for (int ax = 0; ax < 3; ++ax) {
        if (!(cmx[ax] > cmn[ax])) continue;
        double bl[kBins][3], bh[kBins][3];
        int bc[kBins] = {0};
        for (int b = 0; b < kBins; ++b)
          for (int k = 0; k < 3; ++k) bl[b][k] = INFINITY, bh[b][k] = -INFINITY;
        const double sc = kBins / (cmx[ax] - cmn[ax]);
        for (int i = jb.lo; i < jb.hi; ++i) {
          const int p = t.order[i];
          const int b = std::min(std::max((int)((cen[(size_t)p * 3 + ax] - cmn[ax]) * sc), 0), kBins - 1);
          ++bc[b];
          for (int k = 0; k < 3; ++k) {
            bl[b][k] = std::min(bl[b][k], bmn[(size_t)p * 3 + k]);
            bh[b][k] = std::max(bh[b][k], bmx[(size_t)p * 3 + k]);
          }
        }
        double ra[kBins];
        int rc[kBins];
        double al[3] = {INFINITY, INFINITY, INFINITY}, ah[3] = {-INFINITY, -INFINITY, -INFINITY};
        int c = 0;
        for (int b = kBins - 1; b >= 1; --b) {
          for (int k = 0; k < 3; ++k) al[k] = std::min(al[k], bl[b][k]), ah[k] = std::max(ah[k], bh[b][k]);
          c += bc[b];
          ra[b] = area(al, ah);
          rc[b] = c;
        }
        for (int k = 0; k < 3; ++k) al[k] = INFINITY, ah[k] = -INFINITY;
        c = 0;
        for (int b = 0; b + 1 < kBins; ++b) {
          for (int k = 0; k < 3; ++k) al[k] = std::min(al[k], bl[b][k]), ah[k] = std::max(ah[k], bh[b][k]);
          c += bc[b];
          if (c == 0 || rc[b + 1] == 0) continue;
          const double cost = area(al, ah) * c + ra[b + 1] * rc[b + 1];
          if (cost < best) best = cost, best_ax = ax, best_k = b;
        }
      }
      if (best_ax >= 0) {
        const double sc = kBins / (cmx[best_ax] - cmn[best_ax]);
        auto it = std::partition(t.order.begin() + jb.lo, t.order.begin() + jb.hi, [&](int p) {
          return std::min(std::max((int)((cen[(size_t)p * 3 + best_ax] - cmn[best_ax]) * sc), 0), kBins - 1) <= best_k;
        });
        mid = (int)(it - t.order.begin());
      }
      if (mid <= jb.lo || mid >= jb.hi) {  // coincident centroids: split the range in half
        mid = cnt > 4 ? jb.lo + cnt / 2 : -1;
      }
    }
    if (mid < 0) {
      std::sort(t.order.begin() + jb.lo, t.order.begin() + jb.hi);  // ascending position inside a leaf
      nd.lo = jb.lo;
      nd.hi = jb.hi;
      t.nodes[jb.node] = nd;
      continue;
    }
    nd.left = (int)t.nodes.size();
    t.nodes[jb.node] = nd;
    t.nodes.push_back(FastNode{});
    t.nodes.push_back(FastNode{});
    jobs.push_back({nd.left, jb.lo, mid});
    jobs.push_back({nd.left + 1, mid, jb.hi});
  }
}

// entry distance of the ray into a box, or +inf if it misses [tmin, tmax] (conservative: touching counts)
inline double fast_slab(const FastNode &n, const double o[3], const double d[3], double tmin, double tmax) {
  double t0 = tmin, t1 = tmax;
  for (int k = 0; k < 3; ++k) {
    if (d[k] == 0.0) {
      if (o[k] < n.mn[k] || o[k] > n.mx[k]) return INFINITY;
      continue;
    }
    const double inv = 1.0 / d[k];
    double a = (n.mn[k] - o[k]) * inv, b = (n.mx[k] - o[k]) * inv;
    if (a > b) std::swap(a, b);
    // one ulp-scale slack each way: the primitive tests decide, never the box arithmetic
    a -= std::fabs(a) * 4e-16;
    b += std::fabs(b) * 4e-16;
    t0 = a > t0 ? a : t0;
    t1 = b < t1 ? b : t1;
    if (t0 > t1) return INFINITY;
  }
  return t0;
}

inline bool fast_hit(const World &w, const FastTree &t, const Ray &r, double tmin, double tmax, Hit &out,
                     Counters &cnt) {
  const double o[3] = {r.o.x, r.o.y, r.o.z}, d[3] = {r.d.x, r.d.y, r.d.z};
  int stack[128];
  int sp = 0;
  bool any = false;
  double upper = tmax;
  ++cnt.node_tests;
  if (fast_slab(t.nodes[0], o, d, tmin, upper) == INFINITY) return false;
  stack[sp++] = 0;
  while (sp > 0) {
    const FastNode &n = t.nodes[stack[--sp]];
    if (n.left < 0) {
      for (int i = n.lo; i < n.hi; ++i) {
        Hit probe;
        ++cnt.prim_tests;
        if (prim_hit(w, t.order[i], r, tmin, upper, probe)) {
          out = probe;
          any = true;
          upper = probe.at;
        }
      }
      continue;
    }
    cnt.node_tests += 2;
    const double ta = fast_slab(t.nodes[n.left], o, d, tmin, upper);
    const double tb = fast_slab(t.nodes[n.left + 1], o, d, tmin, upper);
    // far child first onto the stack; boxes are re-checked against the shrunken `upper` only
    // through their entry distance here (a stale entry costs tests, never a wrong hit)
    const int near = ta <= tb ? n.left : n.left + 1, far = ta <= tb ? n.left + 1 : n.left;
    const double tn = ta <= tb ? ta : tb, tf = ta <= tb ? tb : ta;
    // (a full stack would drop a subtree and give a wrong-but-green reference image: hard error instead;
    // the SAH tree's depth is ~2 log2 n, far below 128, so this never fires on a sane tree)
    if ((tf != INFINITY) + (tn != INFINITY) + sp > 128) std::abort();
    if (tf != INFINITY) stack[sp++] = far;
    if (tn != INFINITY) stack[sp++] = near;
  }
  return any;
}

// -------------------------------------------------------------- materials ---
struct Scatter {
  Ray r;
  V3 attenuation;
};

// src/common-model.cpp:13-22
template <class R>
bool scatter_lambertian(const Material &m, const Ray &rin, const Hit &hit, R &rng, Scatter &out) {
  V3 rnd = random_unit_vector(rng);
  V3 n = hit.normal;
  if (std::fabs(n.x - rnd.x) < 1e-8 && std::fabs(n.y - rnd.y) < 1e-8 &&
      std::fabs(n.z - rnd.z) < 1e-8)
    return false;
  V3 dir = n + rnd;
  out = Scatter{Ray{hit.where, dir, rin.time}, m.albedo};
  return true;
}

// src/common-model.cpp:24-31
template <class R>
bool scatter_metal(const Material &m, const Ray &rin, const Hit &hit, R &rng, Scatter &out) {
  V3 reflected = reflect(rin.d, hit.normal);
  V3 dir = reflected + m.fuzz * random_unit_vector(rng);
  out = Scatter{Ray{hit.where, dir, rin.time}, m.albedo};
  return true;
}

// src/common-model.cpp:33-38
inline double reflectance(double cosine, double ref_idx) {
  double r0 = (1 - ref_idx) / (1 + ref_idx);
  r0 = r0 * r0;
  return r0 + (1 - r0) * std::pow((1 - cosine), 5);
}

// src/common-model.cpp:40-62
template <class R>
bool scatter_dielectric(const Material &m, const Ray &rin, const Hit &rec, R &rng, Scatter &out) {
  V3 unit = normalize(rin.d);
  double cos_theta = dot(-unit, rec.normal);
  double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
  double ratio = rec.front ? (1.0 / m.ir) : m.ir;
  bool cannot_refract = ratio * sin_theta > 1.0;
  V3 direction;
  if (!cannot_refract) rng.request_coin();  // the coin is drawn only if refraction is possible
  if (cannot_refract || reflectance(cos_theta, ratio) > random_double(rng))
    direction = reflect(unit, rec.normal);
  else
    direction = refract(unit, rec.normal, ratio);
  V3 dir = direction + m.fuzz * random_unit_vector(rng);
  out = Scatter{Ray{rec.where, dir, rin.time}, V3{1.0, 1.0, 1.0}};
  return true;
}

template <class R>
bool scatter(const World &w, const Ray &rin, const Hit &hit, R &rng, Scatter &out) {
  const Material &m = w.mats[w.prims[hit.prim].mat];
  switch (m.kind) {
    case RTOW_MAT_LAMBERTIAN:
      return scatter_lambertian(m, rin, hit, rng, out);
    case RTOW_MAT_METAL:
      return scatter_metal(m, rin, hit, rng, out);
    default:
      return scatter_dielectric(m, rin, hit, rng, out);
  }
}

// -------------------------------------------------------------- integrator ---
// src/render.cpp:112-129 (recursive, attenuation * recurse)
template <class R>
V3 ray_color(const World &w, const Bvh &b, const Ray &ray, long max_depth, R &rng, Counters &cnt) {
  ++cnt.segments;
  Hit hit;
  const bool did_hit =
      b.fast ? fast_hit(w, *b.fast, ray, 0.001, std::numeric_limits<double>::infinity(), hit, cnt)
             : bvh_hit(w, b, b.root, ray, 0.001, std::numeric_limits<double>::infinity(), hit, cnt);
  if (g_raylog && g_raylog_n < g_raylog_cap) {
    double *r = g_raylog + 12 * g_raylog_n++;
    r[0] = g_cur_pixel; r[1] = g_cur_sample; r[2] = g_cur_seg++;
    r[3] = ray.o.x; r[4] = ray.o.y; r[5] = ray.o.z;
    r[6] = ray.d.x; r[7] = ray.d.y; r[8] = ray.d.z;
    r[9] = ray.time;
    r[10] = did_hit ? hit.at : std::numeric_limits<double>::infinity();
    r[11] = did_hit ? (double)w.prims[hit.prim].cls_index : -1.0;
  }
  if (did_hit) {
    if (max_depth <= 0) return {0, 0, 0};
    Scatter sc;
    if (scatter(w, ray, hit, rng, sc))
      return sc.attenuation * ray_color(w, b, sc.r, max_depth - 1, rng, cnt);
    return {0, 0, 0};
  }
  V3 unit = normalize(ray.d);
  double t = 0.5 * (unit.y + +1.0);
  return (1.0 - t) * V3{1.0, 1.0, 1.0} + t * V3{0.5, 0.7, 1.0};
}

// src/common-model.cpp:156-167
template <class R>
Ray camera_get_ray(const rtow_camera_t &c, double s, double t, R &rng) {
  V3 rd = c.lens_radius * random_in_unit_disk(rng);
  V3 offset = load3(c.u) * rd.x + load3(c.v) * rd.y;
  V3 from = load3(c.origin) + offset;
  V3 direction = load3(c.lower_left_corner) + s * load3(c.horizontal) + t * load3(c.vertical) - from;
  rng.request_time();
  double when = random_double(rng, c.t0, c.t1);
  return Ray{from, direction, when};
}

// src/common-model.cpp:136-154
rtow_camera_t make_camera(V3 lookfrom, V3 lookat, V3 vup, double fov, double aspect,
                          double aperture, bool has_focus, double focus_dist, double t0,
                          double t1) {
  rtow_camera_t c;
  V3 w = normalize(lookfrom - lookat);
  V3 u = normalize(cross(vup, w));
  V3 v = normalize(cross(w, u));
  double viewport_height = 2.0 * std::tan(fov * 3.141592653589793238462643383279502884 / 180 / 2);
  double viewport_width = aspect * viewport_height;
  double fd = has_focus ? focus_dist : length(lookfrom - lookat);
  V3 horizontal = fd * viewport_width * u;
  V3 vertical = fd * viewport_height * v;
  V3 llc = lookfrom - horizontal / 2.0 - vertical / 2.0 - fd * w;
  store3(c.origin, lookfrom);
  store3(c.u, u);
  store3(c.v, v);
  store3(c.w, w);
  store3(c.horizontal, horizontal);
  store3(c.vertical, vertical);
  store3(c.lower_left_corner, llc);
  c.lens_radius = aperture / 2;
  c.t0 = t0;
  c.t1 = t1;
  return c;
}

// ----------------------------------------------------------- scene plumbing ---
struct SceneBuilder {  // accumulates the flat arrays of an rtow_scene_t
  rtow_camera_t cam;
  std::vector<double> sg, mg, tg;
  std::vector<int32_t> sm, mm, tm, pk, pi;
  std::vector<rtow_material_t> mats;
  int add_material(int kind, V3 albedo, double fuzz, double ir) {
    rtow_material_t m;
    std::memset(&m, 0, sizeof m);
    store3(m.albedo, albedo);
    m.fuzz = fuzz;
    m.ir = ir;
    m.kind = kind;
    mats.push_back(m);
    return (int)mats.size() - 1;
  }
  void add_sphere(V3 c, double r, int mat) {
    pk.push_back(RTOW_PRIM_SPHERE);
    pi.push_back((int)sm.size());
    sg.insert(sg.end(), {c.x, c.y, c.z, r});
    sm.push_back(mat);
  }
  void add_moving(V3 c0, V3 c1, double r, int mat) {
    pk.push_back(RTOW_PRIM_MOVING_SPHERE);
    pi.push_back((int)mm.size());
    mg.insert(mg.end(), {c0.x, c0.y, c0.z, c1.x, c1.y, c1.z, r, 0.0});
    mm.push_back(mat);
  }
  void add_triangle(V3 a, V3 b, V3 c, int mat) {
    pk.push_back(RTOW_PRIM_TRIANGLE);
    pi.push_back((int)tm.size());
    tg.insert(tg.end(), {a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z});
    tm.push_back(mat);
  }
  template <class T>
  static T *dup(const std::vector<T> &v) {
    T *p = (T *)std::malloc(sizeof(T) * (v.size() ? v.size() : 1));
    if (!v.empty()) std::memcpy(p, v.data(), sizeof(T) * v.size());
    return p;
  }
  rtow_scene_t *finish() const {
    rtow_scene_t *s = (rtow_scene_t *)std::calloc(1, sizeof(rtow_scene_t));
    s->camera = cam;
    s->n_spheres = (int)sm.size();
    s->sphere_geom = dup(sg);
    s->sphere_mat = dup(sm);
    s->n_moving = (int)mm.size();
    s->moving_geom = dup(mg);
    s->moving_mat = dup(mm);
    s->n_triangles = (int)tm.size();
    s->triangle_geom = dup(tg);
    s->triangle_mat = dup(tm);
    s->n_materials = (int)mats.size();
    s->materials = dup(mats);
    s->n_prims = (int)pk.size();
    s->prim_kind = dup(pk);
    s->prim_index = dup(pi);
    return s;
  }
};

double clamp_fuzz(double f) { return std::clamp(f, 0.0, 1.0); }  // src/common-model.h:134,145

bool world_from_scene(const rtow_scene_t *s, World &w) {
  if (!s || s->n_prims <= 0) return false;
  if (s->n_prims != s->n_spheres + s->n_moving + s->n_triangles) return false;
  w.cam = s->camera;
  w.mats.resize(s->n_materials);
  for (int i = 0; i < s->n_materials; ++i) {
    const rtow_material_t &m = s->materials[i];
    w.mats[i] = Material{m.kind, load3(m.albedo), m.fuzz, m.ir};
  }
  w.prims.resize(s->n_prims);
  for (int i = 0; i < s->n_prims; ++i) {
    Prim p{};
    p.kind = s->prim_kind[i];
    int k = s->prim_index[i];
    p.cls_index = k;
    if (p.kind == RTOW_PRIM_SPHERE) {
      const double *g = s->sphere_geom + 4 * k;
      p.a = load3(g);
      p.radius = g[3];
      p.mat = s->sphere_mat[k];
    } else if (p.kind == RTOW_PRIM_MOVING_SPHERE) {
      const double *g = s->moving_geom + 8 * k;
      p.a = load3(g);
      p.b = load3(g + 3);
      p.radius = g[6];
      p.mat = s->moving_mat[k];
    } else if (p.kind == RTOW_PRIM_TRIANGLE) {
      const double *g = s->triangle_geom + 9 * k;
      p.a = load3(g);
      p.b = load3(g + 3);
      p.c = load3(g + 6);
      p.mat = s->triangle_mat[k];
    } else {
      return false;
    }
    if (p.mat < 0 || p.mat >= s->n_materials) return false;
    w.prims[i] = p;
  }
  return true;
}

// rows of the image owned by (rank, nranks, tile_rows): strip t = row / tile_rows
// belongs to rank t % nranks.
std::vector<int> local_rows(const rtow_config_t &c) {
  std::vector<int> rows;
  int nranks = c.nranks > 0 ? c.nranks : 1;
  int tile = c.tile_rows > 0 ? c.tile_rows : 1;
  for (int i = 0; i < c.image_height; ++i)
    if ((i / tile) % nranks == c.rank) rows.push_back(i);
  return rows;
}

// One sample of the hot loop, src/render.cpp:157-162.
template <class R>
inline V3 trace_sample(const World &w, const Bvh &bvh, const rtow_config_t &cfg, int i, int j,
                       R &rng, Counters &cnt) {
  int from_top_i = cfg.image_height - i - 1;
  rng.request_jitter();
  double u = (j + random_double(rng)) / (cfg.image_width - 1);
  double v = (from_top_i + random_double(rng)) / (cfg.image_height - 1);
  Ray r = camera_get_ray(w.cam, u, v, rng);
  return ray_color(w, bvh, r, cfg.max_child_rays, rng, cnt);
}

}  // namespace

// =============================================================== C interface ===
extern "C" {

void orc_mt_reset(void) { the_generator().seed(5489u); }

// Ray log for offline experiments (tests/tools/sim_traversal.py); pass NULL to switch it off.
void orc_set_raylog(double *buf, uint64_t capacity_records) {
  g_raylog = buf;
  g_raylog_cap = capacity_records;
  g_raylog_n = 0;
}
uint64_t orc_raylog_count(void) { return g_raylog_n; }

void orc_mt_burn(uint64_t n) {
  MtGlobal g;
  for (uint64_t i = 0; i < n; ++i) (void)random_double(g);
}

double orc_mt_random_double(double a, double b) {
  MtGlobal g;
  return random_double(g, a, b);
}

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  philox4x32(ctr, key, out, 10);
}
void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4], int rounds) {
  philox4x32(ctr, key, out, rounds);
}
int orc_philox_rounds(void) { return kPhiloxRounds; }

// value k of request `request`: kind 0 = the block of a new sample (k = 0, 1 jitter u, v; 2 shutter time; 3, 4 the
// lens point x, y), kind 2 = the block of a bounce (k = 0..2 the point of the unit ball x, y, z; 3 the dielectric coin);
// the raw words of the block: orc_philox4x32 with counter (request, sample, pixel, 0)
double orc_philox_request(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t request, int kind,
                          int k) {
  PhiloxDraw g;
  g.seed = seed;
  g.begin_sample(pixel, sample);
  g.r = request;
  if (kind == 0) {
    g.request_jitter();
    if (k >= 3) {
      const V3 p = random_in_unit_disk(g);
      return k == 3 ? p.x : p.y;
    }
    return k == 2 ? g.stash_time : g.buf[k];
  }
  if (kind == 2) {
    if (k == 3) {
      g.request_coin();
      return g.buf[0];
    }
    const V3 v = random_in_unit_sphere(g);
    return k == 0 ? v.x : (k == 1 ? v.y : v.z);
  }
  return std::nan("");
}

// lots_of_balls(), src/main.cpp:23-83.
int orc_scene_cover(int nsqrt, double aspect_ratio, int moving_spheres, rtow_scene_t **out) {
  if (!out) return RTOW_EINVAL;
  MtGlobal rng;
  SceneBuilder sb;
  sb.cam = make_camera({13, 2, 3}, {0, 0, 0}, {0, 1, 0}, 20.0, aspect_ratio, 0.1, true, 10.0, 0, 1);
  int ground = sb.add_material(RTOW_MAT_LAMBERTIAN, {0.5, 0.5, 0.5}, 0, 0);
  sb.add_sphere({0, -1000, 0}, 1000.0, ground);
  for (int a = -nsqrt; a < nsqrt; a++) {
    for (int b = -nsqrt; b < nsqrt; b++) {
      double choose_mat = random_double(rng);
      // point center(a + 0.9*rd(), 0.2, b + 0.9*rd()) — paren-init, evaluated
      // right to left by g++: the z term draws first (src/main.cpp:46).
      double cz = b + 0.9 * random_double(rng);
      double cx = a + 0.9 * random_double(rng);
      V3 center{cx, 0.2, cz};
      if (length(center - V3{4, 0.2, 0}) > 0.9) {
        if (choose_mat < 0.8) {
          V3 a1 = random_vec3(rng);
          V3 a2 = random_vec3(rng);
          V3 albedo = a1 * a2;  // component-wise product commutes: draw order is immaterial
          int m = sb.add_material(RTOW_MAT_LAMBERTIAN, albedo, 0, 0);
          if (moving_spheres) {
            V3 center2 = center + V3{0, random_double(rng, 0, .5), 0};
            sb.add_moving(center, center2, 0.2, m);
          } else {
            sb.add_sphere(center, 0.2, m);
          }
        } else if (choose_mat < 0.95) {
          V3 albedo = random_vec3(rng, 0.5, 1);
          double fuzz = random_double(rng, 0, 0.5);
          int m = sb.add_material(RTOW_MAT_METAL, albedo, clamp_fuzz(fuzz), 0);
          sb.add_sphere(center, 0.2, m);
        } else {
          int m = sb.add_material(RTOW_MAT_DIELECTRIC, {0, 0, 0}, clamp_fuzz(0), 1.5);
          sb.add_sphere(center, 0.2, m);
        }
      }
    }
  }
  int glass = sb.add_material(RTOW_MAT_DIELECTRIC, {0, 0, 0}, 0, 1.5);
  int reddish = sb.add_material(RTOW_MAT_LAMBERTIAN, {0.4, 0.2, 0.1}, 0, 0);
  int reddish_metal = sb.add_material(RTOW_MAT_METAL, {0.7, 0.6, 0.5}, clamp_fuzz(0), 0);
  sb.add_sphere({0, 1, 0}, 1.0, glass);
  sb.add_sphere({-4, 1, 0}, 1.0, reddish);
  sb.add_sphere({4, 1, 0}, 1.0, reddish_metal);
  *out = sb.finish();
  return RTOW_OK;
}

// foo(), src/main.cpp:85-136.  tinyobjloader 1.0.6 is not in this image; the
// restatement reads `v x y z` and `f i[/..] j[/..] k[/..]` of the first shape.
int orc_scene_obj(const char *path, double aspect_ratio, rtow_scene_t **out) {
  if (!out || !path) return RTOW_EINVAL;
  MtGlobal rng;
  // rt::random_int() — uniform_int_distribution{0,1}: one 32-bit draw with
  // libstdc++ 11 (src/main.cpp:86).
  (void)the_generator().next();
  SceneBuilder sb;
  sb.cam = make_camera({1, 0, -1}, {0, 0, 0}, {0, 1, 0}, 35.0, aspect_ratio, 0.01, false, 0.0, 0, 1);
  int boring = sb.add_material(RTOW_MAT_LAMBERTIAN, {0.5, 0.5, 0.5}, 0, 0);
  FILE *f = std::fopen(path, "r");
  if (!f) return RTOW_EINVAL;
  std::vector<V3> verts;
  char line[4096];
  bool seen_face = false, done = false;
  int rc = RTOW_OK;
  while (!done && std::fgets(line, sizeof line, f)) {
    char *p = line;
    while (*p == ' ' || *p == '\t') ++p;
    if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
      char *q = p + 1;
      double x = std::strtod(q, &q), y = std::strtod(q, &q), z = std::strtod(q, &q);
      verts.push_back({x, y, z});
    } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
      seen_face = true;
      std::vector<long> idx;
      char *q = p + 1;
      for (;;) {
        while (*q == ' ' || *q == '\t') ++q;
        if (*q == '\0' || *q == '\n' || *q == '\r') break;
        char *e;
        long vi = std::strtol(q, &e, 10);
        if (e == q) break;
        idx.push_back(vi);
        q = e;
        while (*q && *q != ' ' && *q != '\t' && *q != '\n' && *q != '\r') ++q;
      }
      // tinyobj 1.0.6 LoadObj is called with its default triangulate = true (src/main.cpp:109):
      // polygons arrive as the fan (v0, vi, vi+1); fewer than three vertices is the
      // "Oops found a face that isn't a triangle" error of src/main.cpp:130
      if (idx.size() < 3) {
        rc = RTOW_EINVAL;
        break;
      }
      std::vector<V3> fv(idx.size());
      for (size_t k = 0; k < idx.size(); ++k) {
        long vi = idx[k] > 0 ? idx[k] - 1 : (long)verts.size() + idx[k];
        if (vi < 0 || vi >= (long)verts.size()) {
          rc = RTOW_EINVAL;
          break;
        }
        fv[k] = verts[vi];
      }
      if (rc != RTOW_OK) break;
      for (size_t k = 1; k + 1 < fv.size(); ++k) sb.add_triangle(fv[0], fv[k], fv[k + 1], boring);
    } else if ((p[0] == 'o' || p[0] == 'g') && (p[1] == ' ' || p[1] == '\t')) {
      if (seen_face) done = true;  // shapes[0] only (src/main.cpp:115)
    }
  }
  std::fclose(f);
  (void)rng;
  if (rc != RTOW_OK) return rc;
  *out = sb.finish();
  return RTOW_OK;
}

void orc_scene_free(rtow_scene_t *s) {
  if (!s) return;
  std::free((void *)s->sphere_geom);
  std::free((void *)s->sphere_mat);
  std::free((void *)s->moving_geom);
  std::free((void *)s->moving_mat);
  std::free((void *)s->triangle_geom);
  std::free((void *)s->triangle_mat);
  std::free((void *)s->materials);
  std::free((void *)s->prim_kind);
  std::free((void *)s->prim_index);
  std::free(s);
}

void orc_free(void *p) { std::free(p); }

// render(), src/render.cpp:135-191 (without the I/O): BVH build, the sample
// loop per stream ("thread"), and the in-order sum of the partial images.
int orc_render(const rtow_scene_t *scene, const rtow_config_t *cfg, int rng_mode, int nthreads,
               double *rgb_sums, orc_stats_t *stats) {
  return orc_render_ex(scene, cfg, rng_mode, nthreads, 0, rgb_sums, stats);
}

int orc_render_ex(const rtow_scene_t *scene, const rtow_config_t *cfg, int rng_mode, int nthreads, int accel,
                  double *rgb_sums, orc_stats_t *stats) {
  if (!scene || !cfg || !rgb_sums) return RTOW_EINVAL;
  if (cfg->image_width <= 0 || cfg->image_height <= 0 || cfg->nstreams <= 0) return RTOW_EINVAL;
  World w;
  if (!world_from_scene(scene, w)) return scene && scene->n_prims == 0 ? RTOW_EEMPTY : RTOW_EINVAL;
  Bvh bvh;
  bvh.root = bvh_build(w, bvh, 0, (int)w.prims.size());  // also sorts the primitives like the reference
  FastTree fast;
  if (accel) {
    fast_build(w, fast);
    bvh.fast = &fast;
  }

  const int W = cfg->image_width;
  const std::vector<int> rows = local_rows(*cfg);
  const int spt = cfg->samples_per_pixel / cfg->nstreams;  // src/render.cpp:174
  const size_t npx = rows.size() * (size_t)W;
  // stream range / accumulation (include/rtow.h): streams [k_first, k_end), continuing from the
  // caller's sums when cfg->accumulate is set
  const int k_first = cfg->stream_count > 0 ? cfg->stream_first : 0;
  const int k_end = cfg->stream_count > 0 ? cfg->stream_first + cfg->stream_count : cfg->nstreams;
  if (!cfg->accumulate) std::fill(rgb_sums, rgb_sums + npx * 3, 0.0);
  Counters total;
  uint64_t rng_doubles = 0;

  if (rng_mode == ORC_RNG_MT19937) {
    // streams run one after another on the single global generator; with
    // nstreams == 1 this is exactly the reference at `-t 1`.
    MtGlobal rng;
    std::vector<double> local(npx * 3);
    for (int k = k_first; k < k_end; ++k) {
      for (size_t li = 0; li < rows.size(); ++li) {
        int i = rows[li];
        for (int j = 0; j < W; ++j) {
          V3 pixel{0, 0, 0};
          for (int s = 0; s < spt; ++s) pixel = pixel + trace_sample(w, bvh, *cfg, i, j, rng, total);
          store3(&local[(li * W + j) * 3], pixel);
        }
      }
      for (size_t q = 0; q < npx * 3; ++q) rgb_sums[q] = local[q] + rgb_sums[q];  // :178-179
    }
    rng_doubles = rng.ndraws;
  } else {
    int nt = nthreads > 0 ? nthreads : 1;
    std::vector<Counters> cnts(nt);
    std::vector<uint64_t> draws(nt, 0);
    std::atomic<size_t> next_row{0};
    auto work = [&](int tid) {
      PhiloxDraw rng;
      rng.seed = cfg->seed;
      Counters &cnt = cnts[tid];
      for (;;) {
        size_t li = next_row.fetch_add(1);
        if (li >= rows.size()) break;
        int i = rows[li];
        for (int j = 0; j < W; ++j) {
          uint32_t pixel_id = (uint32_t)(i * W + j);
          V3 global = load3(&rgb_sums[(li * W + j) * 3]);
          for (int k = k_first; k < k_end; ++k) {
            V3 partial{0, 0, 0};
            for (int s = 0; s < spt; ++s) {
              rng.begin_sample(pixel_id, (uint32_t)(k * spt + s));
              g_cur_pixel = pixel_id;
              g_cur_sample = (uint32_t)(k * spt + s);
              g_cur_seg = 0;
              partial = partial + trace_sample(w, bvh, *cfg, i, j, rng, cnt);
            }
            global = partial + global;
          }
          store3(&rgb_sums[(li * W + j) * 3], global);
        }
      }
      draws[tid] = rng.ndraws;
    };
    if (nt == 1) {
      work(0);
    } else {
      std::vector<std::thread> th;
      for (int t = 0; t < nt; ++t) th.emplace_back(work, t);
      for (auto &t : th) t.join();
    }
    for (int t = 0; t < nt; ++t) {
      total.segments += cnts[t].segments;
      total.prim_tests += cnts[t].prim_tests;
      total.node_tests += cnts[t].node_tests;
      rng_doubles += draws[t];
    }
  }
  if (stats) {
    stats->samples = (uint64_t)npx * (uint64_t)spt * (uint64_t)(k_end - k_first);
    stats->segments = total.segments;
    stats->prim_tests = total.prim_tests;
    stats->node_tests = total.node_tests;
    stats->rng_doubles = rng_doubles;
    stats->bvh_stupid_volume = stupid_volume(bvh, bvh.root);
    stats->bvh_nodes = (int)bvh.nodes.size();
    int leaves = 0;
    for (auto &n : bvh.nodes) leaves += n.hi > n.lo;
    stats->bvh_leaves = leaves;
  }
  return RTOW_OK;
}

// write_color (src/render.cpp:11-20) and the P3 framing (:182-186).
int orc_ppm(const double *rgb, int width, int height, int spp, char **out_text, uint64_t *out_len) {
  if (!rgb || !out_text || !out_len) return RTOW_EINVAL;
  std::string s;
  s.reserve((size_t)width * height * 12 + 32);
  s += "P3\n" + std::to_string(width) + ' ' + std::to_string(height) + "\n255\n";
  char buf[64];
  for (size_t p = 0; p < (size_t)width * height; ++p) {
    int v[3];
    for (int c = 0; c < 3; ++c) {
      double x = std::sqrt(rgb[p * 3 + c] / static_cast<double>(spp));
      v[c] = static_cast<int>(256 * std::clamp(x, 0.0, 0.999));
    }
    int n = std::snprintf(buf, sizeof buf, "%d %d %d\n", v[0], v[1], v[2]);
    s.append(buf, n);
  }
  char *p = (char *)std::malloc(s.size() + 1);
  std::memcpy(p, s.data(), s.size());
  p[s.size()] = 0;
  *out_text = p;
  *out_len = s.size();
  return RTOW_OK;
}

int orc_sphere_hit(const double center[3], double radius, const double ro[3], const double rd[3],
                   double tmin, double tmax, double *t, double *p, double *n, int *front) {
  Hit h;
  Ray r{load3(ro), load3(rd), 0.0};
  if (!sphere_hit_helper(r, tmin, tmax, load3(center), radius, 0, h)) return 0;
  *t = h.at;
  store3(p, h.where);
  store3(n, h.normal);
  *front = h.front;
  return 1;
}

int orc_triangle_hit(const double a[3], const double b[3], const double c[3], const double ro[3],
                     const double rd[3], double tmin, double tmax, double *t, double *p, double *n) {
  Hit h;
  Ray r{load3(ro), load3(rd), 0.0};
  if (!triangle_hit(r, tmin, tmax, load3(a), load3(b), load3(c), 0, h)) return 0;
  *t = h.at;
  store3(p, h.where);
  store3(n, h.normal);
  return 1;
}

int orc_aabb_hit(const double bmin[3], const double bmax[3], const double ro[3],
                 const double rd[3], double tmin, double tmax) {
  Aabb b;
  b.mn = load3(bmin);
  b.mx = load3(bmax);
  Ray r{load3(ro), load3(rd), 0.0};
  return aabb_hit(b, r, tmin, tmax) ? 1 : 0;
}

}  // extern "C"
