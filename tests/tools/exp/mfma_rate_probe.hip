// Issue rate of the matrix pipe as the fused actors use it: ONE wave per SIMD, back-to-back MFMAs on eight independent
// accumulator tiles, constant operands, nothing else in the loop.  Prints ns per MFMA and the share of the nominal rate
// (2.4 GHz x 8 passes of 4 cycles for v_mfma_f32_32x32x16_f16, 16 passes for v_mfma_f32_32x32x2_f32).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate_probe tests/tools/exp/mfma_rate_probe.hip && /tmp/mfma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(256) void probe(float *out, int iters, float seed) {
  f32x16 acc[8];
  for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = seed * (float)(t + r);
  half8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(seed + (float)j); b[j] = (_Float16)(seed - (float)j); }
  const float af = seed, bf = seed * 0.5f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if constexpr (KIND == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
      else acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[t], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
static void run(const char *name, int blocks, double nominal_ns) {
  float *out;
  hipMalloc(&out, sizeof(float) * 256 * blocks);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4096;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.0f);   // zero operands: no overflow
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.0f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / 5 / ((double)iters * 8);
    printf("%s, %d workgroups of 4 waves: %.2f ns per MFMA per wave = %.0f %% of the nominal %.2f ns\n", name, blocks, ns, 100.0 * nominal_ns / ns, nominal_ns);
  }
  hipFree(out);
}

int main() {
  {   // the clocks ramp for ~30 ms after an idle period (profiles/r01_launch_costs.txt): warm up for a few hundred ms first
    float *w; (void)hipMalloc(&w, sizeof(float) * 256 * 256);
    for (int k = 0; k < 400; ++k) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), 0, 0, w, 4096, 0.0f);
    (void)hipDeviceSynchronize(); (void)hipFree(w);
  }
  run<0>("v_mfma_f32_32x32x16_f16", 256, 32 / 2.4);
  run<0>("v_mfma_f32_32x32x16_f16", 32, 32 / 2.4);
  run<1>("v_mfma_f32_32x32x2_f32", 256, 64 / 2.4);
  run<1>("v_mfma_f32_32x32x2_f32", 32, 64 / 2.4);
  run<0>("v_mfma_f32_32x32x16_f16", 256, 32 / 2.4);     // again, last: order does not matter once warm
  return 0;
}
