// Static instruction counts of the IK building blocks (exp only): each block in its own kernel, f64 and f32.
#include <hip/hip_runtime.h>
#include "../armenv_kin.h"
using namespace armenv;
template <typename T> __global__ void k_sincos(const T *in, T *out) {
  T q[NJ], c[NJ], s[NJ];
  for (int j = 0; j < NJ; ++j) q[j] = in[j * 64 + threadIdx.x];
  sincos_all<T>(q, c, s);
  for (int j = 0; j < NJ; ++j) { out[j * 64 + threadIdx.x] = c[j]; out[(7 + j) * 64 + threadIdx.x] = s[j]; }
}
template <typename T> __global__ void k_fk(const T *in, T *out) {
  T c[NJ], s[NJ];
  for (int j = 0; j < NJ; ++j) { c[j] = in[j * 64 + threadIdx.x]; s[j] = in[(7 + j) * 64 + threadIdx.x]; }
  FKState<T> S; ChainDev<T> ch{};
  fk<KukaChain, T>(ch, c, s, S);
  T acc = 0;
  for (int j = 0; j < NJ; ++j) for (int k = 0; k < 3; ++k) acc += S.z[j][k] + S.pj[j][k];
  for (int k = 0; k < 9; ++k) acc += S.W[k];
  out[threadIdx.x] = acc + S.p[0] + S.p[1] + S.p[2];
}
template <typename T> __global__ void k_orient(const T *in, T *out, IKParams<T> P) {
  T W[9], qc[4], e[3];
  for (int k = 0; k < 9; ++k) W[k] = in[k * 64 + threadIdx.x];
  quat_from_frame<T>(W, qc);
  orientation_error<T>(P.tq, qc, P.angle_f32, e);
  out[threadIdx.x] = e[0] + e[1] + e[2];
}
template <typename T> __global__ void k_dls(const T *in, T *out, IKParams<T> P) {
  FKState<T> S;
  int o = 0;
  for (int j = 0; j < NJ; ++j) for (int k = 0; k < 3; ++k) { S.z[j][k] = in[(o++) * 64 + threadIdx.x]; S.pj[j][k] = in[(o++) * 64 + threadIdx.x]; }
  for (int k = 0; k < 3; ++k) S.p[k] = in[(o++) * 64 + threadIdx.x];
  T e[6], d[NJ];
  for (int k = 0; k < 6; ++k) e[k] = in[(o++) * 64 + threadIdx.x];
  dls_update<KukaChain, T>(S, e, P, d);
  for (int j = 0; j < NJ; ++j) out[j * 64 + threadIdx.x] = d[j];
}
template <typename T> __global__ void k_rotate(const T *in, T *out) {
  T c[NJ], s[NJ], d[NJ];
  for (int j = 0; j < NJ; ++j) { c[j] = in[j * 64 + threadIdx.x]; s[j] = in[(7 + j) * 64 + threadIdx.x]; d[j] = in[(14 + j) * 64 + threadIdx.x]; }
  for (int j = 0; j < NJ; ++j) rotate_small<T>(c[j], s[j], d[j]);
  for (int j = 0; j < NJ; ++j) { out[j * 64 + threadIdx.x] = c[j]; out[(7 + j) * 64 + threadIdx.x] = s[j]; }
}
template __global__ void k_sincos<double>(const double *, double *);
template __global__ void k_fk<double>(const double *, double *);
template __global__ void k_orient<double>(const double *, double *, IKParams<double>);
template __global__ void k_dls<double>(const double *, double *, IKParams<double>);
template __global__ void k_rotate<double>(const double *, double *);
template __global__ void k_sincos<float>(const float *, float *);
template __global__ void k_fk<float>(const float *, float *);
template __global__ void k_orient<float>(const float *, float *, IKParams<float>);
template __global__ void k_dls<float>(const float *, float *, IKParams<float>);
template __global__ void k_rotate<float>(const float *, float *);
