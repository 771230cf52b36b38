// Micro-benchmark: cycles per wave-instruction of a few VALU ops on gfx950 at 1..8 waves/SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/valu_rate.hip -o gpurun_out/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define N_ITER 2000
#define UNROLL 16

template <int OP>
__global__ void k(float *outf, double *outd, unsigned long long *cyc, int dummy) {
  float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
  uint32_t u0 = threadIdx.x + dummy, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
  const float fb = 1.000001f, fc = 1e-7f;
  const double db = 1.000001, dc = 1e-7;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < N_ITER; ++i) {
#pragma unroll
    for (int j = 0; j < UNROLL / 8; ++j) {
      if (OP == 0) { a0 = fmaf(a0, fb, fc); a1 = fmaf(a1, fb, fc); a2 = fmaf(a2, fb, fc); a3 = fmaf(a3, fb, fc); a4 = fmaf(a4, fb, fc); a5 = fmaf(a5, fb, fc); a6 = fmaf(a6, fb, fc); a7 = fmaf(a7, fb, fc); }
      if (OP == 1) { d0 = fma(d0, db, dc); d1 = fma(d1, db, dc); d2 = fma(d2, db, dc); d3 = fma(d3, db, dc); d4 = fma(d4, db, dc); d5 = fma(d5, db, dc); d6 = fma(d6, db, dc); d7 = fma(d7, db, dc); }
      if (OP == 2) { uint64_t p;
        p = (uint64_t)u0 * 0xD2511F53u; u0 = (uint32_t)(p >> 32) ^ (uint32_t)p; p = (uint64_t)u1 * 0xD2511F53u; u1 = (uint32_t)(p >> 32) ^ (uint32_t)p;
        p = (uint64_t)u2 * 0xD2511F53u; u2 = (uint32_t)(p >> 32) ^ (uint32_t)p; p = (uint64_t)u3 * 0xD2511F53u; u3 = (uint32_t)(p >> 32) ^ (uint32_t)p;
        p = (uint64_t)u4 * 0xD2511F53u; u4 = (uint32_t)(p >> 32) ^ (uint32_t)p; p = (uint64_t)u5 * 0xD2511F53u; u5 = (uint32_t)(p >> 32) ^ (uint32_t)p;
        p = (uint64_t)u6 * 0xD2511F53u; u6 = (uint32_t)(p >> 32) ^ (uint32_t)p; p = (uint64_t)u7 * 0xD2511F53u; u7 = (uint32_t)(p >> 32) ^ (uint32_t)p; }
      if (OP == 3) { a0 = fminf(a0, a1 + fc); a1 = fmaxf(a1, a2); a2 = fminf(a2, a3); a3 = fmaxf(a3, a4); a4 = fminf(a4, a5); a5 = fmaxf(a5, a6); a6 = fminf(a6, a7); a7 = fmaxf(a7, a0); }
      if (OP == 4) { d0 = d0 + dc; d1 = d1 * db; d2 = d2 + dc; d3 = d3 * db; d4 = d4 + dc; d5 = d5 * db; d6 = d6 + dc; d7 = d7 * db; }
      if (OP == 5) { u0 ^= u1 + 1; u1 ^= u2 + 3; u2 ^= u3 + 5; u3 ^= u4 + 7; u4 ^= u5 + 9; u5 ^= u6 + 11; u6 ^= u7 + 13; u7 ^= u0 + 15; }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  outf[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  outd[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;
  outf[blockIdx.x * blockDim.x + threadIdx.x] += (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char *name, int insts_per_unroll8) {
  float *of; double *od; unsigned long long *cy;
  hipMalloc(&of, 1 << 22); hipMalloc(&od, 1 << 23); hipMalloc(&cy, 8 * 4096);
  for (int wps : {1, 2, 4, 8}) {
    int block = 64 * 4 * wps;  // 4 SIMDs per CU -> wps waves per SIMD with one block per CU
    if (block > 1024) { block = 1024; }
    int blocks_per_cu = (64 * 4 * wps) / block;
    int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(block), 0, 0, of, od, cy, 1);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(block), 0, 0, of, od, cy, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), cy, grid * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= grid;
    double ninst = (double)N_ITER * (UNROLL / 8) * insts_per_unroll8;
    // s_memtime ticks at 100 MHz-based constant clock? report both per-wave ticks and wall-derived cycles at 2.4 GHz
    double wall_cycles = ms * 1e-3 * 2.4e9;
    printf("%-14s waves/SIMD %d: memtime ticks/inst (one wave) %.2f | wall: %.2f cycles(2.4GHz) per wave-inst per SIMD => %.2f cyc/inst throughput\n",
           name, wps, avg / ninst, wall_cycles / ninst, wall_cycles / (ninst * wps));
  }
  hipFree(of); hipFree(od); hipFree(cy);
}

int main() {
  run<0>("v_fma_f32", 8);
  run<1>("v_fma_f64", 8);
  run<2>("mad_u64_u32+xor", 8);   // 8 x (v_mad_u64_u32 + v_xor)
  run<3>("v_min/max_f32", 8);
  run<4>("v_add/mul_f64", 8);
  run<5>("int add+xor", 8);        // 8 x (v_add + v_xor) -> 16 insts; reported per pair
  return 0;
}
