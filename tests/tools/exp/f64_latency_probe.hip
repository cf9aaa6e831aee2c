// Dependent-chain vs independent-chain cost of v_fma_f64 / v_fma_f32 / v_rcp_f64 on one wave per SIMD (gfx950).
// hipcc --offload-arch=gfx950 -O3 -o f64_latency_probe f64_latency_probe.hip && ./f64_latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T, int CHAINS>
__global__ void probe(T *out, int iters, T a, T b) {
  T x[CHAINS];
  for (int c = 0; c < CHAINS; ++c) x[c] = (T)threadIdx.x * (T)1e-3 + (T)c;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) x[c] = __builtin_fma(x[c], a, b);
  }
  long long t1 = clock64();
  T s = 0;
  for (int c = 0; c < CHAINS; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long *>(out)[4096] = t1 - t0;
}
__global__ void probe_rcp(double *out, int iters, int chains) {
  double x0 = threadIdx.x * 1e-3 + 1.5, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  long long t0 = clock64();
  if (chains == 1) for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) x0 = __builtin_amdgcn_rcp(x0) + 1.0;
  } else for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { x0 = __builtin_amdgcn_rcp(x0) + 1.0; x1 = __builtin_amdgcn_rcp(x1) + 1.0; x2 = __builtin_amdgcn_rcp(x2) + 1.0; x3 = __builtin_amdgcn_rcp(x3) + 1.0; }
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long *>(out)[4096] = t1 - t0;
}
template <typename T, int CHAINS> void run(const char *name) {
  T *d; hipMalloc(&d, 1 << 20);
  const int iters = 2000;
  probe<T, CHAINS><<<1024, 64>>>(d, iters, (T)0.999, (T)0.001);   // one wave per SIMD
  hipDeviceSynchronize();
  long long cyc; hipMemcpy(&cyc, reinterpret_cast<long long *>(d) + 4096, 8, hipMemcpyDeviceToHost);
  printf("%-28s %6.2f cycles per instruction (wave 0)\n", name, (double)cyc / (iters * 16.0 * CHAINS));
  hipFree(d);
}
int main() {
  run<double, 1>("fma f64, 1 dependent chain"); run<double, 2>("fma f64, 2 chains"); run<double, 4>("fma f64, 4 chains");
  run<float, 1>("fma f32, 1 dependent chain"); run<float, 2>("fma f32, 2 chains"); run<float, 4>("fma f32, 4 chains");
  double *d; hipMalloc(&d, 1 << 20);
  for (int ch : {1, 4}) {
    probe_rcp<<<1024, 64>>>(d, 2000, ch); hipDeviceSynchronize();
    long long cyc; hipMemcpy(&cyc, reinterpret_cast<long long *>(d) + 4096, 8, hipMemcpyDeviceToHost);
    printf("rcp f64 + add, %d chain(s)       %6.2f cycles per (rcp, add) pair\n", ch, (double)cyc / (2000 * 16.0));
  }
  return 0;
}
