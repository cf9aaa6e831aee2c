// Launch-cost probe: what does a dependent launch of G workgroups x 256 threads cost on gfx950 as a function of the
// wave's register footprint, its kernarg size, its run time and the bytes it leaves dirty?  50 launches per hipGraph.
// hipcc -O3 --offload-arch=gfx950 launch_probe.hip -o launch_probe && ./launch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Big { double t[160]; };   // 1280 B of kernarg

__device__ __forceinline__ void spin(unsigned long long ticks) {   // s_memtime ticks at 100 MHz
  if (ticks == 0) return;
  unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < ticks * 20) { __builtin_amdgcn_s_sleep(2); }
}

__global__ __launch_bounds__(256) void k_small(float* out, int spin_t, int store) {
  spin(spin_t);
  if (store) out[blockIdx.x * 256 + threadIdx.x] = threadIdx.x;
}
__global__ __launch_bounds__(256) void k_regs(float* out, int spin_t, int store) {
  asm volatile("v_mov_b32 v255, 0\n v_accvgpr_write_b32 a255, v255" ::: "v255", "a255");
  spin(spin_t);
  if (store) out[blockIdx.x * 256 + threadIdx.x] = threadIdx.x;
}
__global__ __launch_bounds__(256) void k_regs_karg(float* out, int spin_t, int store, Big b) {
  asm volatile("v_mov_b32 v255, 0\n v_accvgpr_write_b32 a255, v255" ::: "v255", "a255");
  spin(spin_t);
  if (store) out[blockIdx.x * 256 + threadIdx.x] = threadIdx.x + (float)b.t[threadIdx.x % 160];
}
// many stores per thread: `rows` floats, column layout
__global__ __launch_bounds__(256) void k_dirty(float* out, int rows, int n) {
  asm volatile("v_mov_b32 v255, 0\n v_accvgpr_write_b32 a255, v255" ::: "v255", "a255");
  int i = blockIdx.x * 256 + threadIdx.x;
  float v = out[i];
  for (int r = 0; r < rows; ++r) out[(size_t)r * n + i] = v + r;
}

// one 64-bit atomic add per wave: mode 0 = all waves on one address, 1 = one address per wave (64 B apart),
// 2 = plain read-modify-write of the wave's own slot
__global__ __launch_bounds__(256) void k_atomic(unsigned long long* ctr, int mode, int per_wave) {
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if ((threadIdx.x & 63) != 0) return;
  for (int r = 0; r < per_wave; ++r) {
    if (mode == 0) atomicAdd(&ctr[r], 1ull);
    else if (mode == 1) atomicAdd(&ctr[8 * wave + r], 1ull);
    else ctr[8 * wave + r] += 1ull;
  }
}

template <class F> static double timed(hipStream_t s, F launch) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < 50; ++i) launch();
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, s));
  CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return ms * 1e3 / 1000;
}

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  float* out; CK(hipMalloc(&out, (size_t)64 << 20)); CK(hipMemset(out, 0, (size_t)64 << 20));
  Big b{}; for (int i = 0; i < 160; ++i) b.t[i] = i;
  printf("%6s %10s %10s %10s %10s %10s %10s %10s\n", "WGs", "small", "small+st", "regs", "regs+st", "regs+karg", "regs5us", "small5us");
  for (int g : {1, 16, 64, 128, 256, 512, 1024}) {
    double a = timed(s, [&] { hipLaunchKernelGGL(k_small, dim3(g), dim3(256), 0, s, out, 0, 0); });
    double a2 = timed(s, [&] { hipLaunchKernelGGL(k_small, dim3(g), dim3(256), 0, s, out, 0, 1); });
    double r = timed(s, [&] { hipLaunchKernelGGL(k_regs, dim3(g), dim3(256), 0, s, out, 0, 0); });
    double r2 = timed(s, [&] { hipLaunchKernelGGL(k_regs, dim3(g), dim3(256), 0, s, out, 0, 1); });
    double k = timed(s, [&] { hipLaunchKernelGGL(k_regs_karg, dim3(g), dim3(256), 0, s, out, 0, 1, b); });
    double r5 = timed(s, [&] { hipLaunchKernelGGL(k_regs, dim3(g), dim3(256), 0, s, out, 500, 1); });
    double s5 = timed(s, [&] { hipLaunchKernelGGL(k_small, dim3(g), dim3(256), 0, s, out, 500, 1); });
    printf("%6d %10.2f %10.2f %10.2f %10.2f %10.2f %10.2f %10.2f\n", g, a, a2, r, r2, k, r5, s5);
  }
  printf("\ndirty bytes per launch (256 WGs, 512-register waves): rows x 65536 x 4 B\n");
  for (int rows : {1, 8, 16, 32, 64}) {
    double d = timed(s, [&] { hipLaunchKernelGGL(k_dirty, dim3(256), dim3(256), 0, s, out, rows, 65536); });
    printf("  rows %3d  (%5.1f MB)  %8.2f us/launch\n", rows, rows * 65536 * 4 / 1e6, d);
  }
  printf("\none atomicAdd(u64) per wave at kernel exit, 50 dependent launches per graph, us/launch\n%6s %12s %12s %12s %14s\n", "WGs", "one address", "per-wave addr", "plain rmw", "4 x one addr");
  unsigned long long* ctr = reinterpret_cast<unsigned long long*>(out);
  for (int g : {16, 64, 256, 512, 1024}) {
    double a0 = timed(s, [&] { hipLaunchKernelGGL(k_atomic, dim3(g), dim3(256), 0, s, ctr, 0, 1); });
    double a1 = timed(s, [&] { hipLaunchKernelGGL(k_atomic, dim3(g), dim3(256), 0, s, ctr, 1, 1); });
    double a2 = timed(s, [&] { hipLaunchKernelGGL(k_atomic, dim3(g), dim3(256), 0, s, ctr, 2, 1); });
    double a3 = timed(s, [&] { hipLaunchKernelGGL(k_atomic, dim3(g), dim3(256), 0, s, ctr, 0, 4); });
    printf("%6d %12.2f %12.2f %12.2f %14.2f\n", g, a0, a1, a2, a3);
  }
  return 0;
}
