// accuracy of the f32 engine's lockstep sincos against double-precision sin / cos on [-8, 8]
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I.. -o sincosf_check sincosf_check.hip && ./sincosf_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include "../armenv_kin.h"
using namespace armenv;
__global__ void k(const float *x, float *s, float *c, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float q[NJ], cq[NJ], sq[NJ];
  for (int j = 0; j < NJ; ++j) q[j] = x[i] + 0.37f * j;
  sincos_all<float>(q, cq, sq);
  s[i] = sq[3]; c[i] = cq[3];
}
int main() {
  const int n = 1 << 20;
  float *hx = new float[n], *hs = new float[n], *hc = new float[n];
  for (int i = 0; i < n; ++i) hx[i] = -8.0f + 16.0f * i / n;
  float *dx, *ds, *dc; hipMalloc(&dx, 4 * n); hipMalloc(&ds, 4 * n); hipMalloc(&dc, 4 * n);
  hipMemcpy(dx, hx, 4 * n, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, ds, dc, n);
  hipMemcpy(hs, ds, 4 * n, hipMemcpyDeviceToHost); hipMemcpy(hc, dc, 4 * n, hipMemcpyDeviceToHost);
  double es = 0, ec = 0;
  for (int i = 0; i < n; ++i) { double a = (double)(hx[i] + 0.37f * 3); es = fmax(es, fabs(hs[i] - sin(a))); ec = fmax(ec, fabs(hc[i] - cos(a))); }
  printf("max |sin err| %.3g  max |cos err| %.3g  (f32 ulp at 1 = 1.19e-7)\n", es, ec);
  return 0;
}
