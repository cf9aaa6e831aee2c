// What a taken backward branch costs a wave that has its SIMD to itself (the env kernels: one wave per SIMD): the same 65 536
// independent v_fma_f64 (sixteen chains) as loops of 16 / 32 / 64 / 128 / 256 / 512 instructions per trip, one and two waves per
// SIMD.  ns per instruction = a + b / BODY: a is the issue interval, b the cost of the loop's back edge (s_add, s_cmp, taken
// s_cbranch + the refetch).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/branch_cost_probe tests/tools/exp/branch_cost_probe.hip && /tmp/branch_cost_probe
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

template <int BODY, typename T>
__global__ __launch_bounds__(256) void probe(T *out, int iters, T a, T b) {
  T x[16];
  static_for<0, 16>([&](auto I) { constexpr int t = I; x[t] = a * (T)(t + (int)threadIdx.x); });
  T av = a, bv = b;
  asm volatile("" : "+v"(av), "+v"(bv));
#pragma unroll 1
  for (int i = 0; i < iters; ++i)
    static_for<0, BODY>([&](auto I) { constexpr int t = I % 16; x[t] = __builtin_fma(x[t], av, bv); asm volatile("" : "+v"(x[t])); });
  T s = T(0);
  static_for<0, 16>([&](auto I) { constexpr int t = I; s += x[t]; });
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int BODY, typename T>
static double run(int blocks) {
  T *out;
  (void)hipMalloc(&out, sizeof(T) * 256 * blocks);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 65536 / BODY;
  double best = 1e30;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0, 0);
    for (int k = 0; k < 5; ++k) hipLaunchKernelGGL((probe<BODY, T>), dim3(blocks), dim3(256), 0, 0, out, iters, T(0.999), T(1e-3));
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / 5 / 65536.0 / (blocks / 256);
    if (ns < best) best = ns;
  }
  (void)hipFree(out);
  return best;
}

template <typename T>
static void sweep(const char *name) {
  for (int w = 1; w <= 2; ++w) {
    const double r[6] = {run<16, T>(256 * w), run<32, T>(256 * w), run<64, T>(256 * w), run<128, T>(256 * w), run<256, T>(256 * w), run<512, T>(256 * w)};
    printf("%s, %d wave(s) per SIMD: ns per instruction per SIMD at 16 / 32 / 64 / 128 / 256 / 512 instructions per trip: %.3f %.3f %.3f %.3f %.3f %.3f"
           "  -> back edge = %.0f ns = %.0f cycles at 2.4 GHz per wave (from 16 vs 512)\n", name, w, r[0], r[1], r[2], r[3], r[4], r[5],
           (r[0] - r[5]) * 16 * w / (1.0 - 16.0 / 512.0), (r[0] - r[5]) * 16 * w / (1.0 - 16.0 / 512.0) * 2.4);
  }
}

int main() {
  {   // the clocks ramp for ~30 ms after an idle period
    double *w; (void)hipMalloc(&w, sizeof(double) * 256 * 256);
    for (int k = 0; k < 300; ++k) hipLaunchKernelGGL((probe<128, double>), dim3(256), dim3(256), 0, 0, w, 512, 0.999, 1e-3);
    (void)hipDeviceSynchronize(); (void)hipFree(w);
  }
  sweep<double>("v_fma_f64");
  sweep<float>("v_fma_f32");
  return 0;
}
