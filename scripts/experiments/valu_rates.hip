// valu_rates.hip — issue cost of the vector instructions the trace kernels are made of, measured the way
// the kernels run them: 1024-lane workgroups, one per CU (4 waves per SIMD), eight independent chains per
// lane.  Prints cycles per wave-instruction per SIMD relative to v_fma_f32 (= 2 at 32 lanes per cycle:
// /opt/skills/guides/MI355X_MICROARCH.md, constants table).
//   hipcc --offload-arch=gfx950 -O2 scripts/experiments/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x)                                                              \
  do {                                                                        \
    hipError_t e = (x);                                                       \
    if (e != hipSuccess) {                                                    \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));             \
      std::exit(1);                                                           \
    }                                                                         \
  } while (0)

constexpr int kIters = 4096;

// eight independent instructions per repetition, four repetitions per loop trip
#define REP8_F32(op) \
  asm volatile(op " %0, %0, %8, %0\n\t" op " %1, %1, %8, %1\n\t" op " %2, %2, %8, %2\n\t" op " %3, %3, %8, %3\n\t" \
               op " %4, %4, %8, %4\n\t" op " %5, %5, %8, %5\n\t" op " %6, %6, %8, %6\n\t" op " %7, %7, %8, %7"   \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(k))
#define REP8_2(op) \
  asm volatile(op " %0, %0, %8\n\t" op " %1, %1, %8\n\t" op " %2, %2, %8\n\t" op " %3, %3, %8\n\t" \
               op " %4, %4, %8\n\t" op " %5, %5, %8\n\t" op " %6, %6, %8\n\t" op " %7, %7, %8"   \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(k))
#define REP8_1(op) \
  asm volatile(op " %0, %0\n\t" op " %1, %1\n\t" op " %2, %2\n\t" op " %3, %3\n\t" \
               op " %4, %4\n\t" op " %5, %5\n\t" op " %6, %6\n\t" op " %7, %7"   \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]))

#define KERNEL(name, T, init, kinit, body)                                        \
  __global__ __launch_bounds__(1024) void name(T *out, int iters) {               \
    T a[8];                                                                       \
    for (int i = 0; i < 8; ++i) a[i] = (T)(init) + (T)(threadIdx.x + i);          \
    T k = (T)(kinit);                                                             \
    for (int it = 0; it < iters; ++it) {                                          \
      body; body; body; body;                                                     \
    }                                                                             \
    T s = a[0];                                                                   \
    for (int i = 1; i < 8; ++i) s += a[i];                                        \
    if (s == (T)12345) out[threadIdx.x] = s;                                      \
  }

KERNEL(k_fma_f32, float, 1.0f, 0.999f, REP8_F32("v_fma_f32"))
KERNEL(k_mul_f32, float, 1.0f, 0.999f, REP8_2("v_mul_f32"))
KERNEL(k_fma_f64, double, 1.0, 0.999, REP8_F32("v_fma_f64"))
KERNEL(k_mul_f64, double, 1.0, 0.999, REP8_2("v_mul_f64"))
KERNEL(k_add_f64, double, 1.0, 0.999, REP8_2("v_add_f64"))
KERNEL(k_min_f64, double, 1.0, 0.999, REP8_2("v_min_f64"))
KERNEL(k_rcp_f64, double, 1.5, 0.999, REP8_1("v_rcp_f64"))
KERNEL(k_rsq_f64, double, 1.5, 0.999, REP8_1("v_rsq_f64"))
KERNEL(k_sqrt_f64, double, 1.5, 0.999, REP8_1("v_sqrt_f64"))
KERNEL(k_rcp_f32, float, 1.5f, 0.999f, REP8_1("v_rcp_f32"))
KERNEL(k_mov_b64, double, 1.5, 0.999, REP8_1("v_mov_b64"))
KERNEL(k_mov_b32, float, 1.5f, 0.999f, REP8_1("v_mov_b32"))
KERNEL(k_mul_lo_u32, unsigned, 3u, 0x9E3779B9u, REP8_2("v_mul_lo_u32"))
KERNEL(k_mul_hi_u32, unsigned, 3u, 0x9E3779B9u, REP8_2("v_mul_hi_u32"))
KERNEL(k_xor_b32, unsigned, 3u, 0x9E3779B9u, REP8_2("v_xor_b32"))
KERNEL(k_add_u32, unsigned, 3u, 0x9E3779B9u, REP8_2("v_add_u32"))
KERNEL(k_alignbit, unsigned, 3u, 13u, REP8_F32("v_alignbit_b32"))
KERNEL(k_pk_fma_f32, double, 1.0, 0.999, REP8_F32("v_pk_fma_f32"))
KERNEL(k_pk_mul_f32, double, 1.0, 0.999, REP8_2("v_pk_mul_f32"))
KERNEL(k_pk_add_f32, double, 1.0, 0.999, REP8_2("v_pk_add_f32"))
KERNEL(k_bpermute, unsigned, 3u, 4u, REP8_2("ds_bpermute_b32"); asm volatile("s_waitcnt lgkmcnt(0)"))

// v_mad_u64_u32 vdst(64), sdst(carry), src0, src1, src2(64)
__global__ __launch_bounds__(1024) void k_mad_u64_u32(unsigned long long *out, int iters) {
  unsigned long long a[8];
  for (int i = 0; i < 8; ++i) a[i] = 3ull + threadIdx.x + i;
  unsigned k = 0x9E3779B9u + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      asm volatile("v_mad_u64_u32 %0, vcc, %8, %8, %0\n\tv_mad_u64_u32 %1, vcc, %8, %8, %1\n\tv_mad_u64_u32 %2, vcc, %8, %8, %2\n\t"
                   "v_mad_u64_u32 %3, vcc, %8, %8, %3\n\tv_mad_u64_u32 %4, vcc, %8, %8, %4\n\tv_mad_u64_u32 %5, vcc, %8, %8, %5\n\t"
                   "v_mad_u64_u32 %6, vcc, %8, %8, %6\n\tv_mad_u64_u32 %7, vcc, %8, %8, %7"
                   : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                   : "v"(k)
                   : "vcc");
  }
  unsigned long long s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 12345ull) out[threadIdx.x] = s;
}
// compare + select (the pattern of a closest-hit update)
__global__ __launch_bounds__(1024) void k_cmp_f64_cndmask(double *out, int iters) {
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = 1.0 + threadIdx.x + i;
  double k = 0.999;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      asm volatile("v_cmp_gt_f64 vcc, %0, %8\n\tv_cmp_gt_f64 vcc, %1, %8\n\tv_cmp_gt_f64 vcc, %2, %8\n\tv_cmp_gt_f64 vcc, %3, %8\n\t"
                   "v_cmp_gt_f64 vcc, %4, %8\n\tv_cmp_gt_f64 vcc, %5, %8\n\tv_cmp_gt_f64 vcc, %6, %8\n\tv_cmp_gt_f64 vcc, %7, %8"
                   : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                   : "v"(k)
                   : "vcc");
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 12345.0) out[threadIdx.x] = s;
}

template <class K, class T>
double run(K kern, T *buf, int blocks, int threads) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, buf, 64);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, buf, kIters);
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
  std::printf("# %s, %d CUs, clock %d kHz; one workgroup per CU; 32 instructions x %d trips per lane\n", prop.gcnArchName, cus,
              prop.clockRate, kIters);
  for (int threads : {256, 512, 1024}) {
    const double base = run(k_fma_f32, (float *)buf, cus, threads);
    const double wps = threads / 256.0;
    auto cyc = [&](double ms) { return ms * 1e-3 * prop.clockRate * 1e3 / (32.0 * kIters * wps); };
    std::printf("\n## %d lanes per workgroup = %.0f wave(s) per SIMD; cycles per wave-instruction per SIMD at the reported clock (v_fma_f32 %.3f ms)\n",
                threads, wps, base);
#define ROW(name, kern, T) std::printf("%-18s %6.2f\n", name, cyc(run(kern, (T *)buf, cus, threads)))
    ROW("v_fma_f32", k_fma_f32, float);
    ROW("v_mul_f32", k_mul_f32, float);
    ROW("v_pk_fma_f32", k_pk_fma_f32, double);
    ROW("v_pk_mul_f32", k_pk_mul_f32, double);
    ROW("v_pk_add_f32", k_pk_add_f32, double);
    ROW("v_fma_f64", k_fma_f64, double);
    ROW("v_mul_f64", k_mul_f64, double);
    ROW("v_add_f64", k_add_f64, double);
    ROW("v_min_f64", k_min_f64, double);
    ROW("v_cmp_gt_f64", k_cmp_f64_cndmask, double);
    ROW("v_rcp_f64", k_rcp_f64, double);
    ROW("v_rsq_f64", k_rsq_f64, double);
    ROW("v_sqrt_f64", k_sqrt_f64, double);
    ROW("v_rcp_f32", k_rcp_f32, float);
    ROW("v_mov_b32", k_mov_b32, float);
    ROW("v_mov_b64", k_mov_b64, double);
    ROW("v_xor_b32", k_xor_b32, unsigned);
    ROW("v_add_u32", k_add_u32, unsigned);
    ROW("v_alignbit_b32", k_alignbit, unsigned);
    ROW("v_mul_lo_u32", k_mul_lo_u32, unsigned);
    ROW("v_mul_hi_u32", k_mul_hi_u32, unsigned);
    ROW("v_mad_u64_u32", k_mad_u64_u32, unsigned long long);
    ROW("ds_bpermute_b32", k_bpermute, unsigned);
  }
  return 0;
}
