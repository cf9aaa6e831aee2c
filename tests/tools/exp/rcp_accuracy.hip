// Raw accuracy of v_rcp_f64 / v_rsq_f64 on gfx950 and after one / two Newton steps (the forms fast_rcp / fast_rsqrt use).
// hipcc -O3 --offload-arch=gfx950 -ffp-contract=off rcp_accuracy.hip -o rcp_accuracy && ./rcp_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
__global__ void k(const double* x, double* o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double d = x[i];
  double r0 = __builtin_amdgcn_rcp(d);
  double r1 = fma(fma(-d, r0, 1.0), r0, r0);
  double r2 = fma(fma(-d, r1, 1.0), r1, r1);
  double s0 = __builtin_amdgcn_rsq(d);
  const double hx = 0.5 * d;
  double s1 = fma(fma(-hx * s0, s0, 0.5), s0, s0);
  double s2 = fma(fma(-hx * s1, s1, 0.5), s1, s1);
  o[6 * i] = r0; o[6 * i + 1] = r1; o[6 * i + 2] = r2; o[6 * i + 3] = s0; o[6 * i + 4] = s1; o[6 * i + 5] = s2;
}
int main() {
  const int n = 1 << 22;
  std::vector<double> x(n), o(6 * (size_t)n);
  std::mt19937_64 g(1);
  std::uniform_real_distribution<double> m(1.0, 2.0), e(-40, 40);
  for (int i = 0; i < n; ++i) x[i] = m(g) * std::exp2(std::floor(e(g)));
  double *dx, *dout;
  hipMalloc(&dx, n * 8); hipMalloc(&dout, 6 * (size_t)n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
  hipMemcpy(o.data(), dout, 6 * (size_t)n * 8, hipMemcpyDeviceToHost);
  double mx[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const long double tr = 1.0L / (long double)x[i], ts = 1.0L / sqrtl((long double)x[i]);
    for (int j = 0; j < 6; ++j) {
      const long double t = j < 3 ? tr : ts;
      const double rel = (double)fabsl(((long double)o[6 * (size_t)i + j] - t) / t);
      if (rel > mx[j]) mx[j] = rel;
    }
  }
  const double ulp = 1.1102230246251565e-16;
  printf("max relative error over %d doubles (units of 2^-53):\n  v_rcp_f64 %.3g  +1 Newton %.3g  +2 Newton %.3g\n  v_rsq_f64 %.3g  +1 Newton %.3g  +2 Newton %.3g\n",
         n, mx[0] / ulp, mx[1] / ulp, mx[2] / ulp, mx[3] / ulp, mx[4] / ulp, mx[5] / ulp);
  return 0;
}
