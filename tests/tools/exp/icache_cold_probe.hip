// Does a kernel's instruction fetch start cold at every launch, and what does that cost a one-wave-per-SIMD kernel?  The same N
// independent v_fma_f64 per wave as (a) straight-line code (N instructions of code, every one fetched once) and (b) a loop over a
// 512-instruction body (4 KB of code, fetched once, N / 512 - 1 back edges of ~47 ns each), for N = 2 048, 4 096, 8 192: per-launch
// time of back-to-back launches on one stream (256 workgroups of 256: one wave per SIMD), and an empty kernel beside them.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/icache_cold_probe tests/tools/exp/icache_cold_probe.hip && /tmp/icache_cold_probe
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
// blocks of 256 to keep the template recursion shallow
template <int BLOCKS>
__device__ __forceinline__ void fmas(double (&x)[16], double av, double bv) {
  static_for<0, BLOCKS>([&](auto) {
    static_for<0, 256>([&](auto I) { constexpr int t = I % 16; x[t] = __builtin_fma(x[t], av, bv); asm volatile("" : "+v"(x[t])); });
  });
}
template <int N, bool LOOP>
__global__ __launch_bounds__(256) void probe(double *out, double a, double b) {
  double x[16];
  static_for<0, 16>([&](auto I) { constexpr int t = I; x[t] = a * (double)(t + (int)threadIdx.x); });
  double av = a, bv = b;
  asm volatile("" : "+v"(av), "+v"(bv));
  if constexpr (LOOP) {
#pragma unroll 1
    for (int i = 0; i < N / 512; ++i) fmas<2>(x, av, bv);
  } else {
    fmas<N / 256>(x, av, bv);
  }
  double s = 0;
  static_for<0, 16>([&](auto I) { constexpr int t = I; s += x[t]; });
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void empty_kernel(double *out) { if (out == nullptr) __builtin_trap(); }

template <class K>
static double time_launches(K launch) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  double best = 1e30;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0, 0);
    for (int k = 0; k < 200; ++k) launch();
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms * 1e3 / 200 < best) best = ms * 1e3 / 200;
  }
  return best;   // us per launch
}

template <int N>
static void row(double *out, double empty) {
  const double s = time_launches([&] { hipLaunchKernelGGL((probe<N, false>), dim3(256), dim3(256), 0, 0, out, 0.999, 1e-3); });
  const double l = time_launches([&] { hipLaunchKernelGGL((probe<N, true>), dim3(256), dim3(256), 0, 0, out, 0.999, 1e-3); });
  const int edges = N / 512 - 1;
  printf("%5d v_fma_f64 per wave (%3d KB straight-line): straight %.2f us per launch, loop of 512 %.2f us (%d back edges ~ %.2f us); empty kernel %.2f us; "
         "issue alone %.2f us -> straight-line minus (loop minus back edges) = %.2f us\n",
         N, N * 8 / 1024, s, l, edges, edges * 0.047, empty, N * 2.06e-3, s - (l - edges * 0.047));
}

int main() {
  double *out;
  (void)hipMalloc(&out, sizeof(double) * 256 * 256);
  for (int k = 0; k < 2000; ++k) hipLaunchKernelGGL((probe<8192, true>), dim3(256), dim3(256), 0, 0, out, 0.999, 1e-3);   // clocks
  (void)hipDeviceSynchronize();
  const double empty = time_launches([&] { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, 0, out); });
  row<2048>(out, empty);
  row<4096>(out, empty);
  row<8192>(out, empty);
  return 0;
}
