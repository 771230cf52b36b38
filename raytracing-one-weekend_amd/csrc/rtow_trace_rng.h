// rtow_trace_rng.h — part of the trace kernels (included by rtow_trace_body.h inside namespace rtow::{anonymous};
// see that file for the execution model).  Philox4x32-7 and the mapping of one block per request to jitter, lens and scatter samples.
#pragma once
// ------------------------------------------------------------------ Philox ---
struct Rng {
  uint32_t pixel, sample, r;  // r = next request index of this sample
};

// Philox4x32-7: the fastest member of the family reported Crush-resistant (Salmon et al.,
// SC'11); oracle/ uses the same round count (its tests pin the round function with the
// published 10-round known answers).
#ifndef RTOW_PHILOX_ROUNDS
#define RTOW_PHILOX_ROUNDS 7
#endif
constexpr int kPhiloxRounds = RTOW_PHILOX_ROUNDS;  // (the macro exists for timing experiments only)
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

// One request = one block (w0..w3) = everything the request needs: no request ever draws a second block (round 5).
// The layouts are restated in oracle/rtow_oracle.cpp, struct PhiloxDraw, and pinned word by word in
// tests/test_oracle_units.py.
//
// The reference draws its random points by REJECTION: random_in_unit_disk() and random_in_unit_sphere()
// (src/random-utils.cpp:23-41) loop until a point of the square / cube lies inside, accepting pi/4 and pi/6 of their
// candidates.  On a 64-lane wave that loop runs as long as the wave's unluckiest lane: with two candidates per block
// 2.35 further block evaluations per trip for 7 lanes each (profiles/r05_stamps.log), 9-11 % of the cover scene's
// run time (profiles/r05_ab_no_rejection_ceiling.log).  The device — and the oracle's Philox policy with it — draws
// the SAME DISTRIBUTIONS directly instead:
//   unit disk (lens)     radius = sqrt(U1), angle = 2 pi U2: uniform on the disk;
//   unit ball, positive  radius = max(U1, U2, U3) (distribution function r^3), direction uniform on the octant of the
//   octant (scatter)     sphere: z = U4, azimuth = (pi/2) U5 (Archimedes: the area of a zone depends on its height
//                        alone) — what the reference's loop over [0,1)^3 returns, un-normalised, as its "unit vector"
// with sine and cosine of the azimuth from ONE fixed polynomial (below), evaluated in the same order by the oracle, so
// the strict build stays bit-identical to it.  The reference's own loops live on in the oracle's mt19937 policy, which
// reproduces its images byte for byte; the statistical test between the two policies (tests/test_oracle_units.py, T3)
// is what ties this stream to the reference's, as before.
//
// sin on [0, pi/2]:  x (c0 + x^2 (c1 + x^2 (c2 + x^2 (c3 + x^2 c4)))), a degree-9 Chebyshev fit, |error| < 7e-9
// (an un-normalised direction offset of 2^-21 granularity took this place before); cos x = sin(pi/2 - x).
__device__ __forceinline__ real sin_quarter(real x) {
  const real x2 = x * x;
  return x * (real(0x1.ffffffdb33084p-1) +
              x2 * (real(-0x1.555549a9260fdp-3) +
                    x2 * (real(0x1.110eb1f04c8ffp-7) + x2 * (real(-0x1.9f6d0201a288bp-13) + x2 * real(0x1.5da8d4e70fe23p-19)))));
}
constexpr double kHalfPi = 1.5707963267948966;  // the binary64 nearest pi / 2
// The samplers' angle is a product (bits x pi/2 x 2^-k) and its complement pi/2 - angle feeds the cosine: a fast build
// would fuse that subtraction with the product where the compiler sees both (these helpers) and not where the angle
// arrives through a merge of two kinds of lanes (the trip kernels' shared evaluation, rtow_trace_body.h) — and the trip
// kernels and the state machine must draw the SAME numbers.  An empty asm statement makes the product a value the
// subtraction cannot reach into; no instruction.
__device__ __forceinline__ void opaque(real &x) { asm volatile("" : "+v"(x)); }

// First block of a sample (request 0): pixel jitter and shutter time, 21 bits each (the top bits of words 0..2), and the
// two 32-bit uniforms of the lens point: word 3, and the 11 + 11 + 10 low bits left over in words 0..2.
__device__ __forceinline__ void jitter_from_block(uint32_t o0, uint32_t o1, uint32_t o2, real &u, real &v, real &t) {
  const real s21 = real(0x1p-21);
  u = (real)(o0 >> 11) * s21;
  v = (real)(o1 >> 11) * s21;
  t = (real)(o2 >> 11) * s21;
}
// random_in_unit_disk (src/random-utils.cpp:34-41), directly: radius sqrt(word 3 * 2^-32); angle from the 32 low bits —
// their top two choose the quadrant, the other thirty the angle inside it
__device__ __forceinline__ void lens_from_block(uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3, real &px, real &py) {
  const uint32_t b = (o0 & 0x7ffu) | ((o1 & 0x7ffu) << 11) | ((o2 & 0x3ffu) << 22);
  const real rho = fast_sqrt((real)o3 * real(0x1p-32));
  real th = (real)(b & 0x3fffffffu) * real(0x1p-30 * kHalfPi);
  opaque(th);  // (the angle as a ROUNDED product: see opaque())
  const real sn = sin_quarter(th), cs = sin_quarter(real(kHalfPi) - th);
  const uint32_t q = b >> 30;
  const real cx = (q & 1u) ? sn : cs, sy = (q & 1u) ? cs : sn;
  px = rho * ((q == 1u || q == 2u) ? -cx : cx);
  py = rho * (q >= 2u ? -sy : sy);
}
// The block of a bounce (request 1 + bounce): z = the top 24 bits of word 0, azimuth = the top 24 bits of word 1, the
// three radius uniforms = the halves of word 2 and the low half of word 3 (16 bits each), and the dielectric coin (32
// bits) = the high half of word 3 and the low bytes of words 0 and 1.
// random_in_unit_sphere (src/random-utils.cpp:23-29), directly; returned un-normalised like the reference's
// "random_unit_vector" (:31-33)
__device__ __forceinline__ V3 ball_from_block(uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3) {
  const real z = (real)(o0 >> 8) * real(0x1p-24);
  real phi = (real)(o1 >> 8) * real(0x1p-24 * kHalfPi);
  opaque(phi);
  const uint32_t rm = max(max(o2 & 0xffffu, o2 >> 16), o3 & 0xffffu);
  const real r = (real)rm * real(0x1p-16);
  const real sn = sin_quarter(phi), cs = sin_quarter(real(kHalfPi) - phi);
  const real rs = r * fast_sqrt_pos(real(1.0) - z * z);  // (z <= 1 - 2^-24: the argument is positive)
  return V3{rs * cs, rs * sn, r * z};
}
__device__ __forceinline__ real coin_from_block(uint32_t o0, uint32_t o1, uint32_t o3) {
  return (real)(((o3 >> 16) << 16) | ((o0 & 0xffu) << 8) | (o1 & 0xffu)) * real(0x1p-32);
}
