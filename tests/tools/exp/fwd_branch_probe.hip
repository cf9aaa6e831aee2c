// What a TAKEN FORWARD branch costs a wave that has its SIMD to itself: a loop body of 512 independent v_fma_f64 with 16
// always-taken `s_cbranch_scc1` skips of M instructions spread through it (M = 1, 4, 16, 64, 256), against the same body with
// the branches never taken (the skipped block is then executed: M v_nop) and with no branches at all.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fwd_branch_probe tests/tools/exp/fwd_branch_probe.hip && /tmp/fwd_branch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
#include <utility>

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

#define NOP1 "v_nop\n\t"
#define NOP4 NOP1 NOP1 NOP1 NOP1
#define NOP16 NOP4 NOP4 NOP4 NOP4
#define NOP64 NOP16 NOP16 NOP16 NOP16
#define NOP256 NOP64 NOP64 NOP64 NOP64

template <int M>
__device__ __forceinline__ void skip(int flag) {   // taken when flag == 0
  if constexpr (M == 1) asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\t" NOP1 "1:" ::"s"(flag) : "scc");
  else if constexpr (M == 4) asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\t" NOP4 "1:" ::"s"(flag) : "scc");
  else if constexpr (M == 16) asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\t" NOP16 "1:" ::"s"(flag) : "scc");
  else if constexpr (M == 64) asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\t" NOP64 "1:" ::"s"(flag) : "scc");
  else if constexpr (M == 256) asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\t" NOP256 "1:" ::"s"(flag) : "scc");
}

template <int M>
__global__ __launch_bounds__(256) void probe(double *out, int iters, double a, double b, int flag) {
  double x[16];
  static_for<0, 16>([&](auto I) { constexpr int t = I; x[t] = a * (double)(t + (int)threadIdx.x); });
  double av = a, bv = b;
  asm volatile("" : "+v"(av), "+v"(bv));
  flag = __builtin_amdgcn_readfirstlane(flag);
#pragma unroll 1
  for (int i = 0; i < iters; ++i)
    static_for<0, 512>([&](auto I) {
      constexpr int t = I % 16;
      x[t] = __builtin_fma(x[t], av, bv);
      asm volatile("" : "+v"(x[t]));
      if constexpr (M > 0 && I % 32 == 31) skip<M>(flag);
    });
  double s = 0;
  static_for<0, 16>([&](auto I) { constexpr int t = I; s += x[t]; });
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int M>
static double run(int flag) {   // ns per loop trip (512 FMAs + 16 branch sites)
  const int blocks = 256;
  double *out;
  (void)hipMalloc(&out, sizeof(double) * 256 * blocks);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 128;
  double best = 1e30;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0, 0);
    for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(probe<M>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.999, 1e-3, flag);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / 5 / iters;
    if (ns < best) best = ns;
  }
  (void)hipFree(out);
  return best;
}

template <int M>
static void row(double base) {
  const double taken = run<M>(0), fall = run<M>(1);
  printf("skip of %3d instructions: taken %.0f ns per trip -> %.1f ns = %.0f cycles at 2.4 GHz per taken branch; not taken (block executed) %.0f ns per trip -> %.1f ns per site\n",
         M, taken, (taken - base) / 16, (taken - base) / 16 * 2.4, fall, (fall - base) / 16);
}

int main() {
  {   // clocks
    double *w; (void)hipMalloc(&w, sizeof(double) * 256 * 256);
    for (int k = 0; k < 300; ++k) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), 0, 0, w, 128, 0.999, 1e-3, 0);
    (void)hipDeviceSynchronize(); (void)hipFree(w);
  }
  const double base = run<0>(0);
  printf("512 v_fma_f64 per trip, no branch sites: %.0f ns per trip = %.3f ns per instruction\n", base, base / 512);
  row<1>(base); row<4>(base); row<16>(base); row<64>(base); row<256>(base);
  return 0;
}
