// Does a wave64 f64 VALU instruction get cheaper when most of EXEC is off?  (It would make the extra IK trips of a wave's
// last unconverged lane cheap.)  One wave per SIMD, 4 independent v_fma_f64 chains, lanes selected by a mask.
// hipcc -O3 --offload-arch=gfx950 exec_mask_probe.hip -o exec_mask_probe && ./exec_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(64) void k(double* out, unsigned long long mask, int iters, long long* cycles) {
  const int lane = threadIdx.x;
  double a = 1.0 + lane * 1e-9, b = 0.999999, c0 = 0.1, c1 = 0.2, c2 = 0.3, c3 = 0.4;
  long long t0 = 0, t1 = 0;
  if ((mask >> lane) & 1ull) {
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5"
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(b), "v"(a));
      }
    }
    t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + lane] = c0 + c1 + c2 + c3;
    if (lane == __ffsll((long long)mask) - 1 && blockIdx.x == 0) cycles[0] = t1 - t0;
  }
}

int main() {
  double* out; long long* cyc;
  CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&cyc, 8));
  const int iters = 2000;
  struct { const char* name; unsigned long long m; } cases[] = {
      {"all 64 lanes", ~0ull}, {"lanes 0-31", 0xffffffffull}, {"lanes 0-15", 0xffffull}, {"lane 0", 1ull},
      {"lanes 0,16,32,48", 0x0001000100010001ull}, {"lanes 48-63", 0xffff000000000000ull}, {"lanes 16-31", 0xffff0000ull}};
  for (int grid : {1, 1024}) {
    printf("grid %d workgroups of one wave:\n", grid);
    for (auto& c : cases) {
      hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, out, c.m, iters, cyc);
      CK(hipDeviceSynchronize());
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, out, c.m, iters, cyc);
      CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
      const double n = (double)iters * 64;
      printf("  %-18s %8.1f us   %6.2f ns per v_fma_f64  (s_memtime ticks %lld)\n", c.name, ms * 1e3, ms * 1e6 / n, h);
    }
  }
  return 0;
}
