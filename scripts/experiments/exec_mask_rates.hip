// exec_mask_rates.hip — does a vector instruction get cheaper when most of its lanes are masked off?
// The trace kernels run at ~0.5 lane activity and full issue utilisation; if the SIMD skipped the 16-lane passes of
// an instruction whose EXEC bits are all zero, packing the active lanes into whole quarters of a wave would pay.
// One 1024-lane workgroup per CU (4 waves per SIMD, as the kernels run), eight independent chains per lane, the loop
// entered only by the lanes of `mask`.  Prints cycles per wave-instruction per SIMD for several masks.
//   hipcc --offload-arch=gfx950 -O2 scripts/experiments/exec_mask_rates.hip -o /tmp/exec_mask_rates && /tmp/exec_mask_rates
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                  \
  do {                                                            \
    hipError_t e = (x);                                           \
    if (e != hipSuccess) {                                        \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); \
      std::exit(1);                                               \
    }                                                             \
  } while (0)

constexpr int kIters = 4096;

#define REP8(op)                                                                                                  \
  asm volatile(op " %0, %0, %8, %0\n\t" op " %1, %1, %8, %1\n\t" op " %2, %2, %8, %2\n\t" op " %3, %3, %8, %3\n\t" \
               op " %4, %4, %8, %4\n\t" op " %5, %5, %8, %5\n\t" op " %6, %6, %8, %6\n\t" op " %7, %7, %8, %7"     \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])    \
               : "v"(k))

#define KERNEL(name, T, op)                                                              \
  __global__ __launch_bounds__(1024) void name(T *out, int iters, unsigned long long mask, unsigned long long mask_odd, \
                                               int by_block) {                            \
    T a[8];                                                                              \
    for (int i = 0; i < 8; ++i) a[i] = (T)1 + (T)(threadIdx.x + i);                      \
    T k = (T)0.999;                                                                      \
    const int w_ = threadIdx.x >> 6;                                                     \
    const bool odd = by_block == 1 ? (blockIdx.x & 1) != 0 : (by_block == 2 ? ((w_ >> 2) & 1) != 0 : (by_block == 3 ? w_ != 0 : (w_ & 1) != 0)); \
    if (((odd ? mask_odd : mask) >> (threadIdx.x & 63)) & 1ull) {                        \
      for (int it = 0; it < iters; ++it) {                                               \
        REP8(op); REP8(op); REP8(op); REP8(op);                                          \
      }                                                                                  \
    }                                                                                    \
    T s = a[0];                                                                          \
    for (int i = 1; i < 8; ++i) s += a[i];                                               \
    if (s == (T)12345) out[threadIdx.x] = s;                                             \
  }

KERNEL(k_fma_f64, double, "v_fma_f64")
// the same wave alternates between a sparse and a full EXEC inside the loop: 16 instructions under `mask`, 16 under all lanes
__global__ __launch_bounds__(1024) void k_alternate(double *out, int iters, unsigned long long mask, unsigned long long, int) {
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = 1.0 + (double)(threadIdx.x + i);
  double k = 0.999;
  const bool in = ((mask >> (threadIdx.x & 63)) & 1ull) != 0ull;
  for (int it = 0; it < iters; ++it) {
    if (in) {
      REP8("v_fma_f64"); REP8("v_fma_f64");
    }
    REP8("v_fma_f64"); REP8("v_fma_f64");
  }
  double s = a[0];
  for (int i = 1; i < 8; ++i) s += a[i];
  if (s == 12345.0) out[threadIdx.x] = s;
}
KERNEL(k_fma_f32, float, "v_fma_f32")

template <class K, class T>
double run(K kern, T *buf, int blocks, unsigned long long mask, unsigned long long mask_odd = 0ull, int by_block = -1) {
  if (by_block < 0) mask_odd = mask, by_block = 0;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), 0, 0, buf, 64, mask, mask_odd, by_block);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), 0, 0, buf, kIters, mask, mask_odd, by_block);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  return best;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  void *buf;
  CHECK(hipMalloc(&buf, 1 << 16));
  std::printf("# %s, %d CUs, clock %d kHz; one 1024-lane workgroup per CU (4 waves per SIMD); cycles per wave-instruction per SIMD\n",
              prop.gcnArchName, cus, prop.clockRate);
  auto cyc = [&](double ms) { return ms * 1e-3 * prop.clockRate * 1e3 / (32.0 * kIters * 4.0); };
  struct { const char *name; unsigned long long m; } masks[] = {
      {"all 64 lanes", ~0ull},
      {"lanes 0-31", 0xffffffffull},
      {"lanes 32-63", 0xffffffff00000000ull},
      {"lanes 0-15", 0xffffull},
      {"lanes 0-15 and 32-47", 0x0000ffff0000ffffull},
      {"every other lane (32)", 0x5555555555555555ull},
      {"34 lanes, scattered", 0x9b6d3a5e4c72f189ull},
  };
  std::printf("%-26s %10s %10s\n", "EXEC", "v_fma_f64", "v_fma_f32");
  for (auto &mk : masks)
    std::printf("%-26s %10.2f %10.2f\n", mk.name, cyc(run(k_fma_f64, (double *)buf, cus, mk.m)), cyc(run(k_fma_f32, (float *)buf, cus, mk.m)));
  // how the cost moves with the NUMBER of active lanes: the lowest n lanes, and n lanes spread over the wave
  std::printf("\n%-8s %12s %12s %12s %12s\n", "lanes", "f64 low n", "f64 spread", "f32 low n", "f32 spread");
  for (int n : {1, 2, 3, 4, 6, 8, 10, 12, 14, 16, 20, 24, 32, 48, 64}) {
    const unsigned long long low = n == 64 ? ~0ull : ((1ull << n) - 1ull);
    unsigned long long spread = 0ull;
    for (int i = 0; i < n; ++i) spread |= 1ull << ((i * 64) / n);
    std::printf("%-8d %12.2f %12.2f %12.2f %12.2f\n", n, cyc(run(k_fma_f64, (double *)buf, cus, low)),
                cyc(run(k_fma_f64, (double *)buf, cus, spread)), cyc(run(k_fma_f32, (float *)buf, cus, low)),
                cyc(run(k_fma_f32, (float *)buf, cus, spread)));
  }
  // Is the cliff below ~10 active lanes a property of the instruction or of the whole chip's activity (clocks)?
  // Waves with 4 active lanes next to full waves on the SAME SIMD (even / odd waves of a workgroup), and on
  // DIFFERENT CUs (even / odd workgroups).  The figure is still kernel time / (32 x iterations x 4 waves).
  const unsigned long long full = ~0ull, four = 0x0001000100010001ull;
  std::printf("\n%-44s %10s\n", "mix (v_fma_f64)", "cycles");
  std::printf("%-44s %10.2f\n", "all waves full", cyc(run(k_fma_f64, (double *)buf, cus, full, full, 0)));
  std::printf("%-44s %10.2f\n", "all waves 4 lanes", cyc(run(k_fma_f64, (double *)buf, cus, four, four, 0)));
  std::printf("%-44s %10.2f\n", "even waves full, odd waves 4 lanes", cyc(run(k_fma_f64, (double *)buf, cus, full, four, 0)));
  std::printf("%-44s %10.2f\n", "even waves full, odd waves idle", cyc(run(k_fma_f64, (double *)buf, cus, full, 0ull, 0)));
  std::printf("%-44s %10.2f\n", "even waves 4 lanes, odd waves idle", cyc(run(k_fma_f64, (double *)buf, cus, four, 0ull, 0)));
  std::printf("%-44s %10.2f\n", "even CUs full, odd CUs 4 lanes", cyc(run(k_fma_f64, (double *)buf, cus, full, four, 1)));
  std::printf("%-44s %10.2f\n", "even CUs 4 lanes, odd CUs idle", cyc(run(k_fma_f64, (double *)buf, cus, four, 0ull, 1)));
  std::printf("%-44s %10.2f\n", "per SIMD: 2 waves full + 2 waves 4 lanes", cyc(run(k_fma_f64, (double *)buf, cus, full, four, 2)));
  std::printf("%-44s %10.2f\n", "one wave of 16 full, 15 waves 4 lanes", cyc(run(k_fma_f64, (double *)buf, cus, full, four, 3)));
  std::printf("%-44s %10.2f\n", "one wave of 16 with 10 lanes, 15 with 4", cyc(run(k_fma_f64, (double *)buf, cus, 0x3ffull, four, 3)));
  std::printf("%-44s %10.2f\n", "one wave of 16 with 4 lanes, 15 idle", cyc(run(k_fma_f64, (double *)buf, cus, four, 0ull, 3)));
  std::printf("%-44s %10.2f\n", "one wave of 16 full, 15 idle", cyc(run(k_fma_f64, (double *)buf, cus, full, 0ull, 3)));
  std::printf("%-44s %10.2f\n", "every wave: 16 instr. 4 lanes, 16 instr. full", cyc(run(k_alternate, (double *)buf, cus, four, four, 0)));
  std::printf("%-44s %10.2f\n", "every wave: 32 instr. full (same kernel)", cyc(run(k_alternate, (double *)buf, cus, full, full, 0)));
  return 0;
}
