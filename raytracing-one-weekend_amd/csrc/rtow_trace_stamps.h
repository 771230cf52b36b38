// rtow_trace_stamps.h — part of the trace kernels (included by rtow_trace_body.h inside namespace rtow::{anonymous};
// see that file for the execution model).  Region stamps of the diagnostic build.
#pragma once
// Diagnostic region stamps (STAMPS build only, never in a timed run): wave cycles per
// region of the main loop, accumulated in scalar registers and added to
// counters[8 + region] once per wave.  Shares, not absolute times (each stamp drains
// the wave's outstanding memory operations).
enum { RG_FETCH = 0, RG_REGEN, RG_WALK, RG_SHADE, RG_LEAF, RG_COUNT };
template <bool ON>
struct Stamps {
  unsigned long long t[RG_COUNT] = {0, 0, 0, 0, 0};
  unsigned long long last = 0;
  unsigned long long tt[RG_COUNT] = {0, 0, 0, 0, 0};    // the same cycles, only for the trips after this wave found the queue empty
  bool tail = false;
  unsigned long long lt[RG_COUNT] = {0, 0, 0, 0, 0};    // the same, weighted by the lanes the region worked for
  unsigned long long iters = 0, trips = 0, phases = 0;  // wave-level loop counts
  unsigned long long blocks = 0, block_lanes = 0;       // Philox block evaluations of the new-ray stage (one per trip) / unused since the rejection loops went
  unsigned long long step_lanes = 0, leaf_lanes = 0;    // lanes stepping per step-loop iteration / testing per leaf phase
  unsigned long long iters_cam = 0, phases_cam = 0;     // GRID: step iterations / leaf phases that only camera rays needed
  unsigned long long primary = 0;                       // lanes of this wave whose ray is a camera ray (set by the caller)
  __device__ __forceinline__ void start() {
    if constexpr (ON) last = now();
  }
  __device__ __forceinline__ void mark(int region) {
    if constexpr (ON) {
      const unsigned long long n = now();
      t[region] += n - last;
      if (tail) tt[region] += n - last;
      last = n;
    }
  }
  __device__ __forceinline__ void mark(int region, unsigned long long lanes) {
    if constexpr (ON) {
      const unsigned long long n = now();
      t[region] += n - last;
      if (tail) tt[region] += n - last;
      lt[region] += (n - last) * (unsigned long long)__popcll(lanes);
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

