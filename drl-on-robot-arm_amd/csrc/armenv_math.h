// armenv_math.h -- scalar helpers for the gfx950 env kernels: compile-time loops, f32/f64 math
// traits, Philox4x32-10.  Device code only (hipcc --offload-arch=gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

#define AE_DEV __device__ __forceinline__
#define AE_HD __host__ __device__ __forceinline__

namespace armenv {

constexpr int NJ = 7;

// Fully unrolled loop with a compile-time index: every array subscript below is a constant, so
// per-lane arrays stay in VGPRs (a runtime subscript would send them to scratch).
template <int B, int E, class F>
AE_DEV void static_for(F &&f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

template <typename T> struct Mth;

template <> struct Mth<double> {
  static AE_DEV void sincos(double x, double &s, double &c) { ::sincos(x, &s, &c); }
  static AE_DEV double sqrt(double x) { return ::sqrt(x); }
  static AE_DEV double acos(double x) { return ::acos(x); }
  static AE_DEV double fabs(double x) { return ::fabs(x); }
  static AE_DEV double fma(double a, double b, double c) { return ::fma(a, b, c); }
  static AE_DEV double fmax(double a, double b) { return ::fmax(a, b); }
  static AE_DEV double fmin(double a, double b) { return ::fmin(a, b); }
  static AE_DEV bool finite(double x) { return ::isfinite(x); }
  static constexpr double eps = 2.220446049250313e-16;
  static constexpr double pi = 3.14159265358979323846;
};

template <> struct Mth<float> {
  static AE_DEV void sincos(float x, float &s, float &c) { ::sincosf(x, &s, &c); }
  static AE_DEV float sqrt(float x) { return ::sqrtf(x); }
  static AE_DEV float acos(float x) { return ::acosf(x); }
  static AE_DEV float fabs(float x) { return ::fabsf(x); }
  static AE_DEV float fma(float a, float b, float c) { return ::fmaf(a, b, c); }
  static AE_DEV float fmax(float a, float b) { return ::fmaxf(a, b); }
  static AE_DEV float fmin(float a, float b) { return ::fminf(a, b); }
  static AE_DEV bool finite(float x) { return ::isfinite(x); }
  static constexpr float eps = 1.1920929e-07f;
  static constexpr float pi = 3.14159265358979323846f;
};

// Philox4x32-10 (Salmon et al., SC'11).  Counter-based: the goal of (env, episode) is a pure
// function of (seed, env_id, episode), independent of launch geometry or sharding.
AE_HD void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

// 53-bit uniform in [0,1), the resolution of CPython's random.random().
AE_HD double u53(uint32_t hi, uint32_t lo) {
  uint64_t v = (((uint64_t)hi << 32) | lo) >> 11;
  return (double)v * (1.0 / 9007199254740992.0);
}

// Two uniforms from block (3*draw + b) of stream (seed, env_id, episode).
AE_HD void philox_pair(uint64_t seed, uint64_t env_id, uint32_t episode, uint32_t block, double &u0, double &u1) {
  uint32_t c[4] = {(uint32_t)env_id, (uint32_t)(env_id >> 32), episode, block};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  u0 = u53(c[0], c[1]);
  u1 = u53(c[2], c[3]);
}

}  // namespace armenv
