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

