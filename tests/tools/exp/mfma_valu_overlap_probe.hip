// Does f64 VALU work of one wave issue under the f16 MFMAs of ANOTHER wave on the same SIMD?  (exp only)
//   hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap_probe.hip -o mfma_valu_overlap_probe && ./mfma_valu_overlap_probe
// 512-thread workgroups, two waves per SIMD (amdgpu_waves_per_eu(2, 2)), one workgroup per CU.  Waves 0..3 run back-to-back
// v_mfma_f32_32x32x16_f16 on eight independent accumulator tiles (the f16x3 actor's k-step: 24 MFMAs, no accumulator reused within 8);
// waves 4..7 run v_fma_f64 in CHAINS independent chains (16: issue-bound; 2: latency-bound like the IK's LDL^T).
// role mask: 1 = MFMA waves work, 2 = VALU waves work, 3 = both.  Reported: kernel time and the SIMD each wave ran on.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int CHAINS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void probe(float *out, unsigned *where, int role, int mfma_iters,
                                                                                   int valu_iters, double x0) {
  const int wave = threadIdx.x >> 6;
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) where[wave] = hw;
  float res = 0.f;
  if (wave < 4) {
    if (!(role & 1)) return;
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.01f * (threadIdx.x + j)); b[j] = (_Float16)(0.02f * j); }
    for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
      for (int m = 0; m < 24; ++m) acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 7], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) res += acc[i][r];
  } else {
    if (!(role & 2)) return;
    double c[CHAINS];
    for (int i = 0; i < CHAINS; ++i) c[i] = x0 + i + threadIdx.x;
    const double m = 1.0000001, d = 1e-9;
    for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
      for (int u = 0; u < 64 / CHAINS; ++u)
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) c[i] = __builtin_fma(c[i], m, d);
    }
    double s = 0;
    for (int i = 0; i < CHAINS; ++i) s += c[i];
    res = (float)s;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = res;
}

template <int CHAINS> float run(int role, int mi, int vi, float *out, unsigned *where) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<CHAINS><<<256, 512>>>(out, where, role, mi, vi, 1.0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<CHAINS><<<256, 512>>>(out, where, role, mi, vi, 1.0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float *out; unsigned *where;
  hipMalloc(&out, sizeof(float) * 256 * 512); hipMalloc(&where, 32);
  const int mi = 4096;                       // 4096 x 24 MFMAs x 32 cycles = 3.1 M cycles = 1.3 ms at 2.4 GHz
  for (int warm = 0; warm < 3; ++warm) run<16>(3, mi, 20000, out, where);
  unsigned hw[8]; hipMemcpy(hw, where, 32, hipMemcpyDeviceToHost);
  printf("wave -> (simd, cu) of workgroup 0:");
  for (int w = 0; w < 8; ++w) printf(" %d:(%u,%u)", w, (hw[w] >> 4) & 3, (hw[w] >> 8) & 15);
  printf("\n");
  const float tm = run<16>(1, mi, 0, out, where);
  printf("MFMA waves alone:                         %.3f ms  (%.1f cycles per MFMA at 2.4 GHz)\n", tm, tm * 1e-3 * 2.4e9 / (mi * 24.0));
  for (int chains : {16, 4, 2}) {
    // calibrate the VALU leg to about the MFMA leg's time
    int vi = 20000;
    auto rv = [&](int role, int v) { return chains == 16 ? run<16>(role, mi, v, out, where) : (chains == 4 ? run<4>(role, mi, v, out, where) : run<2>(role, mi, v, out, where)); };
    float tv = rv(2, vi);
    vi = (int)(vi * tm / tv);
    tv = rv(2, vi);
    const float tb = rv(3, vi);
    printf("f64 chains %2d: VALU waves alone %.3f ms (%.2f cycles per v_fma_f64), both %.3f ms  -> overlap %.0f %% (0 = serial sum, 100 = max of the two)\n",
           chains, tv, tv * 1e-3 * 2.4e9 / (vi * 64.0), tb, 100.0 * (tm + tv - tb) / (tm + tv - (tm > tv ? tm : tv)));
  }
  return 0;
}
