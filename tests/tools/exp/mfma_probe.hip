// MFMA issue-rate probe for v_mfma_f32_32x32x2_f32 (exp only).  hipcc --offload-arch=gfx950 -O3 mfma_probe.hip -o mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, bool VARY>
__global__ __launch_bounds__(256) void probe(float *out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(VARY ? a + i : a, b, acc[i], 0, 0, 0);
    if (VARY) { a = a * 1.0001f; b = b + 0.5f; }
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, bool VARY> void run(const char *name, int blocks) {
  float *out; hipMalloc(&out, sizeof(float) * blocks * 256);
  const int iters = 4096;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<NACC, VARY><<<blocks, 256>>>(out, iters, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<NACC, VARY><<<blocks, 256>>>(out, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double mf = (double)blocks * 4 * iters * NACC;           // MFMAs per wave x waves
  double tf = mf * 4096 / (ms * 1e-3) / 1e12;
  double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * NACC) / ((blocks * 4) / 1024.0 > 1 ? (blocks * 4) / 1024.0 : 1);
  printf("%-28s blocks %4d: %.3f ms  %.1f TF  ~%.0f cycles/MFMA/SIMD (at 2.4 GHz)\n", name, blocks, ms, tf, cyc);
  hipFree(out);
}
// operands streamed from memory like the actor: A from a [iters][64] table, B computed by a short VALU chain
template <int NACC>
__global__ __launch_bounds__(256) void probe_mem(float *out, const float4 *tab, int iters, unsigned long long *clk) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int lane = threadIdx.x & 63;
  unsigned long long c0 = clock64(), w0 = wall_clock64();
  float4 cur = tab[lane];
  float h = 0.25f;
  for (int it = 0; it < iters; ++it) {
    const float4 nxt = tab[((it + 1) & 127) * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
    const float av[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i & 3], (i & 1) ? h : -h, acc[i], 0, 0, 0);
    h = fmaxf(fmaf(cur.x, 0.37f, fmaf(cur.y, -0.21f, cur.z * 0.11f)), 0.f) + 0.01f;
    __builtin_amdgcn_sched_barrier(0);
    cur = nxt;
  }
  unsigned long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
template <int NACC> void run_mem(const char *name, int blocks, bool zeros) {
  float *out; hipMalloc(&out, sizeof(float) * blocks * 256);
  float4 *tab; hipMalloc(&tab, sizeof(float4) * 128 * 64);
  unsigned long long *clk; hipMalloc(&clk, 16);
  float4 *h = new float4[128 * 64];
  unsigned x = 12345;
  auto rnd = [&]() { x = x * 1664525u + 1013904223u; return zeros ? 0.f : ((x >> 8) * (1.f / 8388608.f) - 1.f); };
  for (int i = 0; i < 128 * 64; ++i) h[i] = make_float4(rnd(), rnd(), rnd(), rnd());
  hipMemcpy(tab, h, sizeof(float4) * 128 * 64, hipMemcpyHostToDevice);
  const int iters = 4096;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe_mem<NACC><<<blocks, 256>>>(out, tab, iters, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe_mem<NACC><<<blocks, 256>>>(out, tab, iters, clk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long hc[2]; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
  double ghz = (double)hc[0] / ((double)hc[1] / 100e6) / 1e9;
  double mf = (double)blocks * 4 * iters * NACC;
  printf("%-34s blocks %4d: %.3f ms  %.1f TF  shader clock %.2f GHz  %.0f shader cycles/MFMA\n", name, blocks, ms,
         mf * 4096 / (ms * 1e-3) / 1e12, ghz, (double)hc[0] / ((double)iters * NACC));
}
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
template <int NACC, int MODE>
__global__ __launch_bounds__(256) void probe_h(float *out, const half8 *tab, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int lane = threadIdx.x & 63;
  half8 a[4], b[2];
  for (int i = 0; i < 4; ++i) a[i] = tab[i * 64 + lane];
  for (int i = 0; i < 2; ++i) b[i] = tab[(4 + i) * 64 + lane];
  for (int it = 0; it < iters; ++it) {
    if (MODE == 1) {   // stream A operands from memory
      for (int i = 0; i < 4; ++i) a[i] = tab[((it & 15) * 8 + i) * 64 + lane];
    }
    if (MODE == 2) {   // perturb B with VALU
      for (int i = 0; i < 2; ++i) b[i][it & 7] = (_Float16)((float)b[i][it & 7] * 0.999f);
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[i & 1], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int MODE> void run_h(const char *name, int blocks, bool zeros) {
  float *out; hipMalloc(&out, sizeof(float) * blocks * 256);
  half8 *tab; hipMalloc(&tab, 16 * 16 * 8 * 64);
  _Float16 *h = new _Float16[16 * 8 * 64 * 8];
  unsigned x = 777;
  for (int i = 0; i < 16 * 8 * 64 * 8; ++i) { x = x * 1664525u + 1013904223u; h[i] = zeros ? (_Float16)0.f : (_Float16)(((x >> 8) * (1.f / 8388608.f) - 1.f) * 0.1f); }
  hipMemcpy(tab, h, 16 * 8 * 64 * 8 * 2, hipMemcpyHostToDevice);
  const int iters = 4096;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe_h<NACC, MODE><<<blocks, 256>>>(out, tab, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe_h<NACC, MODE><<<blocks, 256>>>(out, tab, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double mf = (double)blocks * 4 * iters * NACC;
  printf("f16 32x32x16 %-30s blocks %4d: %.3f ms  %.0f TF  %.0f ns-cycles@2.4/MFMA/SIMD\n", name, blocks, ms, mf * 32768 / (ms * 1e-3) / 1e12,
         ms * 1e-3 * 2.4e9 / ((double)iters * NACC) / ((blocks * 4) / 1024.0 > 1 ? (blocks * 4) / 1024.0 : 1));
}
int main() {
  run_h<8, 0>("fixed operands, random data", 256, false);
  run_h<8, 0>("fixed operands, zero data", 256, true);
  run_h<8, 1>("A streamed, random", 256, false);
  run_h<8, 2>("B perturbed by VALU", 256, false);
  run_h<8, 0>("fixed, 2 waves/SIMD", 512, false);

  run_mem<8>("mem operands, random data", 256, false);
  run_mem<8>("mem operands, zero data", 256, true);
  run_mem<8>("mem operands, random, 2 waves/SIMD", 512, false);

  run<8, false>("8 acc, fixed operands", 256);
  run<8, true>("8 acc, varying operands", 256);
  run<4, false>("4 acc, fixed operands", 256);
  run<2, false>("2 acc, fixed operands", 256);
  run<1, false>("1 acc, fixed operands", 256);
  run<8, false>("8 acc, fixed, 2 waves/SIMD", 512);
  return 0;
}
