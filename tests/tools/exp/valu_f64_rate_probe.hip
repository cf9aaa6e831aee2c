// Issue rate of the vector pipe as the env kernels use it: ONE wave per SIMD (or two), back-to-back independent instructions on
// sixteen chains, nothing else in the loop.  Prints ns per wave instruction against the nominal 4 cycles at 2.4 GHz
// (f64 FMA: 78.6 TFLOP/s on 256 CUs).  Variants: v_fma_f64 with a scalar multiplier, with vector operands only, v_mul_f64,
// v_add_f64, v_fma_f32, and f64 / f32 alternating (does a 32-bit instruction fit into the gap behind an f64 one?).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_f64_rate_probe tests/tools/exp/valu_f64_rate_probe.hip && /tmp/valu_f64_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND>
__global__ __launch_bounds__(256) void probe(double *out, int iters, double a, double b) {
  double x[16];
  float xf[16];
  for (int t = 0; t < 16; ++t) { x[t] = a * (double)(t + threadIdx.x); xf[t] = (float)x[t]; }
  double av = a, bv = b;
  float af = (float)a, bf = (float)b;
  if constexpr (KIND != 0) asm volatile("" : "+v"(av), "+v"(bv));     // vector-register operands
  asm volatile("" : "+v"(af), "+v"(bf));
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      if constexpr (KIND == 0) x[t] = __builtin_fma(x[t], a, b);
      else if constexpr (KIND == 1) x[t] = __builtin_fma(x[t], av, bv);
      else if constexpr (KIND == 2) x[t] = x[t] * av;
      else if constexpr (KIND == 3) x[t] = x[t] + bv;
      else if constexpr (KIND == 4) xf[t] = __builtin_fmaf(xf[t], af, bf);
      else { x[t] = __builtin_fma(x[t], av, bv); xf[t] = __builtin_fmaf(xf[t], af, bf); }   // alternating f64 / f32
    }
  }
  double s = 0;
  for (int t = 0; t < 16; ++t) s += x[t] + (double)xf[t];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
static void run(const char *name, int blocks, double a, double b) {
  double *out;
  (void)hipMalloc(&out, sizeof(double) * 256 * blocks);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 8192;
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, a, b);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, a, b);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / 5 / ((double)iters * 16) / (blocks > 256 ? blocks / 256 : 1);   // per instruction per SIMD
    if (ns < best) best = ns;
  }
  printf("%-28s %4d workgroups of 4 waves: %.3f ns per wave instruction per SIMD = %.1f cycles at 2.4 GHz\n", name, blocks, best, best * 2.4);
  (void)hipFree(out);
}

int main() {
  {   // the clocks ramp for ~30 ms after an idle period: warm up first
    double *w; (void)hipMalloc(&w, sizeof(double) * 256 * 256);
    for (int k = 0; k < 300; ++k) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256), 0, 0, w, 8192, 0.999, 1e-3);
    (void)hipDeviceSynchronize(); (void)hipFree(w);
  }
  run<0>("v_fma_f64 (scalar a, b)", 256, 0.999, 1e-3);
  run<1>("v_fma_f64 (vector operands)", 256, 0.999, 1e-3);
  run<1>("v_fma_f64 (vector operands)", 32, 0.999, 1e-3);
  run<1>("v_fma_f64 (vector operands)", 512, 0.999, 1e-3);
  run<1>("v_fma_f64 (vector operands)", 1024, 0.999, 1e-3);
  run<0>("v_fma_f64 (scalar a, b)", 512, 0.999, 1e-3);
  run<0>("v_fma_f64 (scalar a, b)", 1024, 0.999, 1e-3);
  run<2>("v_mul_f64", 256, 1.0, 0.0);
  run<3>("v_add_f64", 256, 1.0, 1e-3);
  run<4>("v_fma_f32", 256, 0.999, 1e-3);
  run<4>("v_fma_f32", 512, 0.999, 1e-3);
  run<5>("v_fma_f64 + v_fma_f32 (per pair / 2)", 256, 0.999, 1e-3);
  return 0;
}
