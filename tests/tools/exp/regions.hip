// Static instruction counts of the regions of one IK trip (f64, KUKA fast path), each region in a kernel of its own, for
// tests/tools/region_counts.py (the two-lanes-per-env model of VERDICT r03 #3).  hipcc --offload-arch=gfx950 -c, never run.
#include <hip/hip_runtime.h>
#include "../../../drl-on-robot-arm_amd/csrc/armenv_kin.h"
using namespace armenv;
using T = double;
#define LD(i) in[(i) * 64 + threadIdx.x]
__global__ void region_fk(const T *in, T *out) {
  T c[NJ], s[NJ];
  for (int j = 0; j < NJ; ++j) { c[j] = LD(j); s[j] = LD(7 + j); }
  FKState<T> S; ChainDev<T> ch{};
  fk<KukaChain, T>(ch, c, s, S);
  int o = 0;
  for (int j = 0; j < NJ; ++j) for (int k = 0; k < 3; ++k) { out[(o++) * 64 + threadIdx.x] = S.z[j][k]; out[(o++) * 64 + threadIdx.x] = S.pj[j][k]; }
  for (int k = 0; k < 9; ++k) out[(o++) * 64 + threadIdx.x] = S.W[k];
}
__global__ void region_orientation(const T *in, T *out, IKParams<T> P) {
  T W[9], qc[4], e[3];
  for (int k = 0; k < 9; ++k) W[k] = LD(k);
  quat_from_frame<T>(W, qc);
  orientation_error<T>(P.tq, qc, P.angle_f32, e);
  for (int k = 0; k < 3; ++k) out[k * 64 + threadIdx.x] = e[k];
}
__device__ __forceinline__ void load_frame(const T *in, FKState<T> &S, T (&e)[6]) {
  int o = 0;
  for (int j = 0; j < NJ; ++j) for (int k = 0; k < 3; ++k) { S.z[j][k] = LD(o); ++o; S.pj[j][k] = LD(o); ++o; }
  for (int k = 0; k < 3; ++k) { S.p[k] = LD(o); ++o; }
  for (int k = 0; k < 6; ++k) { e[k] = LD(o); ++o; }
}
__global__ void region_dls(const T *in, T *out, IKParams<T> P) {
  FKState<T> S; T e[6], d[NJ], mp = 1e30;
  load_frame(in, S, e);
  dls_update<KukaChain, T, 0>(S, S.p, e, P, d, mp);
  for (int j = 0; j < NJ; ++j) out[j * 64 + threadIdx.x] = d[j];
}
__global__ void region_dls_7col(const T *in, T *out, IKParams<T> P) {   // all seven Jacobian columns (the MODE 2 build)
  FKState<T> S; T e[6], d[NJ], mp = 1e30, pe[3];
  load_frame(in, S, e);
  for (int k = 0; k < 3; ++k) pe[k] = LD(200 + k);
  dls_update<KukaChain, T, 2>(S, pe, e, P, d, mp);
  for (int j = 0; j < NJ; ++j) out[j * 64 + threadIdx.x] = d[j];
}
template <int N> __global__ void region_rotate(const T *in, T *out) {
  T c[N], s[N], d[N];
  for (int j = 0; j < N; ++j) { c[j] = LD(j); s[j] = LD(7 + j); d[j] = LD(14 + j); }
  for (int j = 0; j < N; ++j) rotate_small<T>(c[j], s[j], d[j]);
  for (int j = 0; j < N; ++j) { out[j * 64 + threadIdx.x] = c[j]; out[(7 + j) * 64 + threadIdx.x] = s[j]; }
}
template __global__ void region_rotate<7>(const T *, T *);
template __global__ void region_rotate<4>(const T *, T *);
template __global__ void region_rotate<1>(const T *, T *);
__global__ void region_io_only(const T *in, T *out) {   // the load / store scaffolding of the kernels above, to subtract
  T acc = 0;
  for (int j = 0; j < 21; ++j) acc += LD(j);
  out[threadIdx.x] = acc;
}
// a lane pair's exchange: v_permlane32_swap moves one dword between lane i and lane i + 32; an f64 costs two
__global__ void region_swap8(const T *in, T *out) {
  T v[8];
  for (int j = 0; j < 8; ++j) v[j] = LD(j);
  for (int j = 0; j < 8; ++j) {
    unsigned lo = (unsigned)__double2loint(v[j]), hi = (unsigned)__double2hiint(v[j]), lo2 = lo, hi2 = hi;
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(lo2));
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(hi), "+v"(hi2));
    v[j] = __hiloint2double((int)hi, (int)lo) + __hiloint2double((int)hi2, (int)lo2);
  }
  for (int j = 0; j < 8; ++j) out[j * 64 + threadIdx.x] = v[j];
}
