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

// One request = one block (w0..w3); see oracle/rtow_oracle.cpp, struct PhiloxDraw.
// First block of a sample: pixel jitter + shutter time, 21 bits each (top bits of words 0..2), and
// the first lens-disk candidate, 32 bits per coordinate (word 3; the 11+11+10 low bits of words 0..2).
__device__ __forceinline__ void jitter_from_block(uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3, real &u, real &v,
                                                  real &t, real &da, real &db) {
  const real s21 = real(0x1p-21), s32 = real(0x1p-32);
  u = (real)(o0 >> 11) * s21;
  v = (real)(o1 >> 11) * s21;
  t = (real)(o2 >> 11) * s21;
  da = (real)o3 * s32;
  db = (real)((o0 & 0x7ffu) | ((o1 & 0x7ffu) << 11) | ((o2 & 0x3ffu) << 22)) * s32;
}
__device__ __forceinline__ void rng_jitter(Rng &g, uint32_t k0, uint32_t k1, real &u, real &v, real &t, real &da,
                                           real &db) {
  uint32_t o0, o1, o2, o3;
  philox4x32(g.r, g.sample, g.pixel, 0u, k0, k1, o0, o1, o2, o3);
  g.r += 1u;
  jitter_from_block(o0, o1, o2, o3, u, v, t, da, db);
}
// a further lens-disk block: two candidates, (w0, w1) and (w2, w3), 32 bits per coordinate
__device__ __forceinline__ void rng_disk2(Rng &g, uint32_t k0, uint32_t k1, real &a0, real &b0, real &a1, real &b1) {
  uint32_t o0, o1, o2, o3;
  philox4x32(g.r, g.sample, g.pixel, 0u, k0, k1, o0, o1, o2, o3);
  g.r += 1u;
  const real s32 = real(0x1p-32);
  a0 = (real)o0 * s32;
  b0 = (real)o1 * s32;
  a1 = (real)o2 * s32;
  b1 = (real)o3 * s32;
}
// Unit-ball candidates: 21 bits per coordinate, one candidate per pair of words (x, y: the top 21
// bits of the two words; z: the 11 + 10 low bits left over).  The first block of a bounce carries
// a candidate (words 0, 1) and EITHER the dielectric coin (word 2, 32 bits) OR, for a bounce that draws no coin, a
// second candidate (words 2, 3); a further block carries two candidates.
// The candidate as its three 21-bit integers (X, Y, Z) = 2^21 (x, y, z).  The rejection test of random_in_unit_sphere
// (src/random-utils.cpp:23-29: length2 >= 1) runs on them: x*x + y*y + z*z >= 1 <=> X*X + Y*Y + Z*Z >= 2^42, and the
// binary64 form is EXACT for these operands (each square has 42 significant bits, the sums stay below 2^44 multiples
// of 2^-42), so the integer test takes the same decision as the reference expression bit for bit — in three
// v_mad_u64_u32 and a compare instead of three conversions, three scalings, three multiply-adds and a compare.
// Only the ACCEPTED candidate is converted to floating point (round 4: -6 binary64-rate instructions per candidate
// inside the rejection loop).
struct BallCand {
  uint32_t x, y, z;
};
__device__ __forceinline__ BallCand ball_ints(uint32_t lo, uint32_t hi) {
  return BallCand{lo >> 11, hi >> 11, (lo & 0x7ffu) | ((hi & 0x3ffu) << 11)};
}
__device__ __forceinline__ bool ball_outside(BallCand c) {
  const unsigned long long n2 = (unsigned long long)c.x * c.x + (unsigned long long)c.y * c.y + (unsigned long long)c.z * c.z;
  return (uint32_t)(n2 >> 32) >= (1u << 10);  // n2 >= 2^42
}
__device__ __forceinline__ V3 ball_point(BallCand c) {
  const real s21 = real(0x1p-21);
  return V3{(real)c.x * s21, (real)c.y * s21, (real)c.z * s21};
}
__device__ __forceinline__ V3 ball_from_pair(uint32_t lo, uint32_t hi) { return ball_point(ball_ints(lo, hi)); }
// first block of a bounce: candidate (w0, w1), the coin (w2) and — for a bounce that does not use the coin — a second
// candidate (w2, w3)
__device__ __forceinline__ void rng_scatter_first(Rng &g, uint32_t k0, uint32_t k1, real &coin, BallCand &a, BallCand &b) {
  uint32_t o0, o1, o2, o3;
  philox4x32(g.r, g.sample, g.pixel, 0u, k0, k1, o0, o1, o2, o3);
  g.r += 1u;
  coin = (real)o2 * real(0x1p-32);
  a = ball_ints(o0, o1);
  b = ball_ints(o2, o3);
}
__device__ __forceinline__ void rng_scatter2i(Rng &g, uint32_t k0, uint32_t k1, BallCand &a, BallCand &b) {
  uint32_t o0, o1, o2, o3;
  philox4x32(g.r, g.sample, g.pixel, 0u, k0, k1, o0, o1, o2, o3);
  g.r += 1u;
  a = ball_ints(o0, o1);
  b = ball_ints(o2, o3);
}
