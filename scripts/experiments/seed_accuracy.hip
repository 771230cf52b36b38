// seed_accuracy.hip — relative error of v_rcp_f64 / v_rsq_f64 and of the Newton-refined values the fast build
// uses (rtow_trace_math.h), over 4 M random arguments.
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=fast scripts/experiments/seed_accuracy.hip -o seed_accuracy.bin
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

__global__ void k(const double *x, double *out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double r0 = __builtin_amdgcn_rcp(v);
  double r1 = r0 * (2.0 - v * r0);
  double r2 = r1 * (2.0 - v * r1);
  double q0 = __builtin_amdgcn_rsq(v);
  double h = 0.5 * v;
  double q1 = q0 * (1.5 - h * q0 * q0);
  double q2 = q1 * (1.5 - h * q1 * q1);
  // one third-order step instead (what rtow_trace_math.h uses since the end of round 2)
  const double e = __builtin_fma(-v, r0, 1.0);
  const double r3 = __builtin_fma(r0, __builtin_fma(e, e, e), r0);
  const double s0 = v * q0, z = s0 * q0;
  const double poly = __builtin_fma(z, __builtin_fma(z, 0.375, -1.25), 1.875);
  const double q3 = q0 * poly;
  out[8 * i + 0] = r0, out[8 * i + 1] = r1, out[8 * i + 2] = r2, out[8 * i + 3] = r3;
  out[8 * i + 4] = q0, out[8 * i + 5] = q1, out[8 * i + 6] = q2, out[8 * i + 7] = q3;
}

int main() {
  const int n = 1 << 22;
  std::mt19937_64 rng(1);
  std::uniform_real_distribution<double> U(-12.0, 12.0);
  std::vector<double> x(n), out(8 * (size_t)n);
  for (auto &v : x) v = std::exp2(U(rng)) * (1.0 + 0.37 * U(rng) / 12.0);
  for (auto &v : x) v = std::fabs(v) + 1e-300;
  double *dx, *dout;
  hipMalloc(&dx, n * 8);
  hipMalloc(&dout, 8 * (size_t)n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, dx, dout, n);
  hipMemcpy(out.data(), dout, 8 * (size_t)n * 8, hipMemcpyDeviceToHost);
  double mx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const long double v = x[i];
    const long double rr = 1.0L / v, qq = 1.0L / sqrtl(v);
    for (int j = 0; j < 4; ++j) mx[j] = std::fmax(mx[j], (double)fabsl((out[8 * (size_t)i + j] - rr) / rr));
    for (int j = 4; j < 8; ++j) mx[j] = std::fmax(mx[j], (double)fabsl((out[8 * (size_t)i + j] - qq) / qq));
  }
  std::printf("max relative error over %d arguments (2^-53 = 1.1e-16)\n", n);
  std::printf("v_rcp_f64 %.3e   + 1 Newton step %.3e   + 2 steps %.3e   one third-order step %.3e\n", mx[0], mx[1], mx[2], mx[3]);
  std::printf("v_rsq_f64 %.3e   + 1 Newton step %.3e   + 2 steps %.3e   one third-order step %.3e\n", mx[4], mx[5], mx[6], mx[7]);
  return 0;
}
