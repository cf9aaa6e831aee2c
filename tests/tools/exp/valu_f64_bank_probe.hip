// Does the f64 vector rate of a single wave depend on which VGPR banks its operands sit in?  (bank = register index mod 4; a
// 64-bit operand is an even-aligned pair, i.e. banks {0,1} or {2,3}.)  Hand-allocated registers, eight independent chains.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_f64_bank_probe tests/tools/exp/valu_f64_bank_probe.hip && /tmp/valu_f64_bank_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHAINS_MUL(M)                                                                                     \
  "v_mul_f64 v[8:9], v[8:9], " M "\n v_mul_f64 v[12:13], v[12:13], " M "\n v_mul_f64 v[16:17], v[16:17], " M "\n" \
  "v_mul_f64 v[20:21], v[20:21], " M "\n v_mul_f64 v[24:25], v[24:25], " M "\n v_mul_f64 v[28:29], v[28:29], " M "\n" \
  "v_mul_f64 v[32:33], v[32:33], " M "\n v_mul_f64 v[36:37], v[36:37], " M "\n"
#define CHAINS_FMA(M, C)                                                                                  \
  "v_fma_f64 v[8:9], v[8:9], " M ", " C "\n v_fma_f64 v[12:13], v[12:13], " M ", " C "\n v_fma_f64 v[16:17], v[16:17], " M ", " C "\n" \
  "v_fma_f64 v[20:21], v[20:21], " M ", " C "\n v_fma_f64 v[24:25], v[24:25], " M ", " C "\n v_fma_f64 v[28:29], v[28:29], " M ", " C "\n" \
  "v_fma_f64 v[32:33], v[32:33], " M ", " C "\n v_fma_f64 v[36:37], v[36:37], " M ", " C "\n"
#define CLOB "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v12", "v13", "v16", "v17", "v20", "v21", "v24", "v25", "v28", "v29", "v32", "v33", "v36", "v37"

// chains live in v[8:9], v[12:13], ...: banks {0,1}.  v[4:5]: banks {0,1} (same as the chains), v[2:3] / v[6:7]: banks {2,3}.
template <int KIND>
__global__ __launch_bounds__(256) void probe(double *out, int iters, double a, double b) {
  asm volatile("v_mov_b32 v2, %0\n v_mov_b32 v3, %1\n v_mov_b32 v4, %0\n v_mov_b32 v5, %1\n v_mov_b32 v6, %2\n v_mov_b32 v7, %3\n"
               "v_mov_b32 v8, %0\n v_mov_b32 v9, %1\n v_mov_b32 v12, %0\n v_mov_b32 v13, %1\n v_mov_b32 v16, %0\n v_mov_b32 v17, %1\n"
               "v_mov_b32 v20, %0\n v_mov_b32 v21, %1\n v_mov_b32 v24, %0\n v_mov_b32 v25, %1\n v_mov_b32 v28, %0\n v_mov_b32 v29, %1\n"
               "v_mov_b32 v32, %0\n v_mov_b32 v33, %1\n v_mov_b32 v36, %0\n v_mov_b32 v37, %1\n"
               :: "v"((int)__double2loint(a)), "v"((int)__double2hiint(a)), "v"((int)__double2loint(b)), "v"((int)__double2hiint(b)) : CLOB);
  for (int i = 0; i < iters; ++i) {
    if constexpr (KIND == 0) asm volatile(CHAINS_MUL("v[2:3]") CHAINS_MUL("v[2:3]") ::: CLOB);                   // other bank pair
    else if constexpr (KIND == 1) asm volatile(CHAINS_MUL("v[4:5]") CHAINS_MUL("v[4:5]") ::: CLOB);              // same bank pair
    else if constexpr (KIND == 2) asm volatile(CHAINS_FMA("v[2:3]", "v[6:7]") CHAINS_FMA("v[2:3]", "v[6:7]") ::: CLOB);   // x{0,1} m{2,3} c{2,3}
    else if constexpr (KIND == 3) asm volatile(CHAINS_FMA("v[4:5]", "v[6:7]") CHAINS_FMA("v[4:5]", "v[6:7]") ::: CLOB);   // x{0,1} m{0,1} c{2,3}
    else if constexpr (KIND == 4) asm volatile(CHAINS_FMA("v[4:5]", "v[4:5]") CHAINS_FMA("v[4:5]", "v[4:5]") ::: CLOB);   // all {0,1}
    else asm volatile(CHAINS_FMA("v[2:3]", "1.0") CHAINS_FMA("v[2:3]", "1.0") ::: CLOB);                          // x{0,1} m{2,3} inline constant
  }
  double r;
  asm volatile("v_mov_b32 %0, v8\n v_mov_b32 %1, v9" : "=v"(((int *)&r)[0]), "=v"(((int *)&r)[1]) :: CLOB);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND>
static void run(const char *name) {
  const int blocks = 256, iters = 8192;
  double *out;
  (void)hipMalloc(&out, sizeof(double) * 256 * blocks);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 0.0);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 0.0);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / 5 / ((double)iters * 16);
    if (ns < best) best = ns;
  }
  printf("%-58s %.3f ns per wave instruction = %.1f cycles at 2.4 GHz\n", name, best, best * 2.4);
  (void)hipFree(out);
}

int main() {
  for (int k = 0; k < 3; ++k) { double *w; (void)hipMalloc(&w, 8 * 65536); for (int j = 0; j < 100; ++j) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), 0, 0, w, 8192, 1.0, 0.0); (void)hipDeviceSynchronize(); (void)hipFree(w); }
  run<0>("v_mul_f64  x{0,1} * m{2,3}");
  run<1>("v_mul_f64  x{0,1} * m{0,1}");
  run<2>("v_fma_f64  x{0,1} * m{2,3} + c{2,3}");
  run<3>("v_fma_f64  x{0,1} * m{0,1} + c{2,3}");
  run<4>("v_fma_f64  x{0,1} * m{0,1} + c{0,1} (m = c)");
  run<5>("v_fma_f64  x{0,1} * m{2,3} + 1.0 (inline constant)");
  return 0;
}
