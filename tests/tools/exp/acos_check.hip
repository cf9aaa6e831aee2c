// accuracy of acos_branchfree against the host libm on [-1, 1] (dense near +-1 and +-0.5)
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -o acos_check acos_check.hip && ./acos_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include "../armenv_kin.h"
using namespace armenv;
__global__ void k(const double *x, double *y, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = acos_branchfree(x[i]);
}
int main() {
  const int n = 1 << 21;
  double *hx = new double[n], *hy = new double[n];
  for (int i = 0; i < n; ++i) {
    double u = (i + 0.5) / n;
    int m = i & 3;
    hx[i] = m == 0 ? 2 * u - 1 : m == 1 ? 1 - u * u * 1e-3 : m == 2 ? -1 + u * u * 1e-3 : (i & 4 ? 0.5 : -0.5) + (u - 0.5) * 1e-6;
  }
  hx[0] = 1.0; hx[1] = -1.0; hx[2] = 0.0; hx[3] = 0.5; hx[4] = -0.5;
  double *dx, *dy; hipMalloc(&dx, 8 * n); hipMalloc(&dy, 8 * n);
  hipMemcpy(dx, hx, 8 * n, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, dy, n);
  hipMemcpy(hy, dy, 8 * n, hipMemcpyDeviceToHost);
  double e = 0; long flips = 0;
  for (int i = 0; i < n; ++i) { double r = acos(hx[i]); e = fmax(e, fabs(hy[i] - r)); flips += (float)(2 * hy[i]) != (float)(2 * r); }
  printf("max |acos err| %.3g over %d points; float(2 acos) differs from libm's in %ld points; acos(1)=%g acos(-1)-pi=%g\n", e, n, flips, hy[0], hy[1] - M_PI);
  return 0;
}
