// Operand / result layout of v_mfma_f32_4x4x1_16b_f32 on gfx950 (the layer-3 epilogue of the fused actor, armenv_actor.h):
// 16 independent 4x4x1 outer products per instruction.  Expected (and asserted here):
//   lane l supplies A[block = l / 4][row i = l % 4] and B[block = l / 4][column j = l % 4];
//   D register r of lane l = C + A[block][row r] * B[block][column l % 4].
// hipcc -O3 --offload-arch=gfx950 tests/tools/exp/mfma4x4_probe.hip -o /tmp/mfma4x4 && /tmp/mfma4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float *out, unsigned long long *cyc) {
  const int l = threadIdx.x;
  const float a = 100.f + l;          // A[block][i = l % 4]
  const float b = 1.f + 0.001f * l;   // B[block][j = l % 4]
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[4 * l + r] = c[r];
  // issue rate: 256 back-to-back instructions on four independent accumulators
  f32x4 d0 = c, d1 = c, d2 = c, d3 = c;
  const unsigned long long t0 = clock64();
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d0, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d1, 0, 0, 0);
    d2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d2, 0, 0, 0);
    d3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d3, 0, 0, 0);
  }
  const unsigned long long t1 = clock64();
  if (l == 0) cyc[0] = t1 - t0;
  out[256 + l] = d0[0] + d1[1] + d2[2] + d3[3];
}
int main() {
  float *o; unsigned long long *c;
  hipMalloc(&o, 4 * 512); hipMalloc(&c, 8);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, c);
  float h[320]; unsigned long long hc;
  hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const float want = (100.f + (4 * (l / 4) + r)) * (1.f + 0.001f * l);
      if (fabsf(h[4 * l + r] - want) > 1e-4f * want) { if (bad < 8) printf("lane %d reg %d: got %g want %g\n", l, r, h[4 * l + r], want); ++bad; }
    }
  printf("layout %s; 256 v_mfma_f32_4x4x1_16b_f32 on four accumulators: %llu clock64 ticks (%.1f per instruction)\n", bad ? "DIFFERENT" : "as expected", hc, hc / 256.0);
  return bad != 0;
}
