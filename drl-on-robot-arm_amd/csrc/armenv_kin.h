// armenv_kin.h -- per-lane kinematics for a 7-revolute +z chain: forward kinematics, geometric
// Jacobian, Bullet-style orientation error and the damped-least-squares update.
//
// One env per lane; everything lives in VGPRs (all subscripts are compile-time constants).
// Replaces the third-party engine calls of /root/reference/envs/rl_reach_env.py:
//   p.getLinkState(kuka, 6)[4]  (:202,237,271)            -> fk()
//   p.calculateInverseKinematics(...)  (:244-250)           -> ik_move()
#pragma once
#include "armenv_math.h"
#include "chains_builtin.h"

namespace armenv {

// Build modes of the per-lane env code (the MODE template parameter of the lanes, ik_trip, dls_update, ik_limits):
//   0  the default kernels;
//   1  the bookkeeping build: parity-fence counters, LDL^T pivot minimum, per-step ik_updates / diag outputs
//      (ArmEnvConfig.fence_counters);
//   2  the bookkeeping build with the IK evaluated at ArmEnvConfig.ik_tip_offset instead of the URDF link-7 frame, and with the
//      f64 step diagnostics (StepIO::diag; ArmEnvConfig.fence_counters = 2).  Kept apart from mode 1: one more nullable output
//      pointer in the bookkeeping build cost it 5-10 % (two scalar registers live across the IK loop, armenv_env.h step_tail).
// Compile-time, not run-time: a branch in the trip loop cost the default path 3 %.
constexpr bool kFenceOf(int mode) { return mode >= 1; }
constexpr bool kTipOf(int mode) { return mode == 2; }

// Runtime chain (generic path): joint-origin translation and row-major 3x3 rotation per joint.
template <typename T> struct ChainDev {
  T xyz[NJ][3];
  T R[NJ][9];
  T base_p[3];
  T base_R[9];  // row-major
};

struct GenericChain {
  static constexpr bool kGeneric = true;
  static constexpr const char *kName = "generic";
};

// IK / task scalars shared by every kernel (uniform -> SGPRs).
template <typename T> struct IKParams {
  T tq[4];          // target orientation xyzw
  T lambda;
  T residual;
  T max_dtheta;
  int32_t max_iters;
  int32_t exit_mode;
  int32_t angle_f32;
  int32_t clamp_limits;
  int32_t pad0;
  T fence_pivot;    // an IK call whose damped system J J^T + lambda I had an LDL^T pivot below this is ill-conditioned
  // ArmEnvConfig.ik_tip_offset: the point of link 7 (in the link-7 frame) at which the IK takes its position error and its
  // linear Jacobian.  Read by the MODE 2 builds of the kernels only (kTipOf): the default builds are compiled for the URDF
  // link frame itself (offset 0), where the last joint's lever arm is an exact zero and is left out.
  T tip[3];
  // URDF joint limits: lim[0..6] lower, lim[7..13] upper, in DEVICE MEMORY (EnvCold) -- 28 scalar registers the IK loop
  // needs for its own constants otherwise (measured: +70 instructions per trip from s_mov rematerialisation and
  // v_readlane spills with the limits held as kernel arguments).  lim_min = min over joints of min(-lower, upper) (or -1
  // when a joint's range does not straddle 0): no joint can be outside its limits while max |q_j| <= lim_min, so the
  // per-step test is six v_max and one compare, and the table is read only by waves that hold a lane beyond it.
  // lim[14] = limit_erp: the share of a violation one step of the push-back model (clamp_limits == 2) removes.
  const T *lim;
  T lim_min;
};

// World frame of a link: W holds the rotation as three column vectors W[3*col + row].
template <typename T> struct FKState {
  T W[9];
  T p[3];
  T z[NJ][3];   // world joint axes
  T pj[NJ][3];  // world joint pivots
};

// Compile-time structure of a built-in chain's FK.  fk() starts from W = I, and the first joints only rotate about
// axes of that frame, so some entries of W are EXACT zeros or +-1 whatever the joint angles (KUKA: W after joint 1 is
// [c -s 0; s c 0; 0 0 1]).  FkKinds propagates that through the chain at compile time -- kind 0 = run-time value,
// 1 = exact zero, 2 = exactly +1, 3 = exactly -1 -- for W before each joint (Wk) and for the permuted frame A = W R_origin
// the joint rotation acts on (Ak).  fk() then leaves out every multiplication by an exact zero and every multiplication by
// one: x * 1 is x and acc + 0 is acc, so the values are the same (an exact-zero entry may come out as +0 where the full
// arithmetic gives -0), and one FK loses ~25 of its 144 instructions -- four FKs per env step.
template <class C> struct FkKinds {
  int Wk[NJ + 1][9];
  int Ak[NJ][9];
  constexpr FkKinds() : Wk{}, Ak{} {
    for (int i = 0; i < 9; ++i) Wk[0][i] = (i % 4 == 0) ? 2 : 1;
    for (int j = 0; j < NJ; ++j) {
      for (int c = 0; c < 3; ++c) {
        const int k = C::perm[j][c], sg = C::sgn[j][c];
        for (int r = 0; r < 3; ++r) {
          const int w = Wk[j][3 * k + r];
          Ak[j][3 * c + r] = (sg > 0 || w < 2) ? w : (w == 2 ? 3 : 2);
        }
      }
      for (int r = 0; r < 3; ++r) {
        const bool z = Ak[j][r] == 1 && Ak[j][3 + r] == 1;
        Wk[j + 1][r] = z ? 1 : 0;
        Wk[j + 1][3 + r] = z ? 1 : 0;
        Wk[j + 1][6 + r] = Ak[j][6 + r];
      }
    }
  }
};
// c * a + s * b with the rounding of fma(c, a, s * b), given the compile-time kinds of a and b
template <int KA, int KB, typename T>
AE_DEV T rot_term(T c, T a, T s, T b) {
  using M = Mth<T>;
  if constexpr (KA == 1 && KB == 1) return T(0);
  else if constexpr (KA == 1) return KB == 2 ? s : (KB == 3 ? -s : s * b);
  else if constexpr (KB == 1) return KA == 2 ? c : (KA == 3 ? -c : c * a);
  else if constexpr (KA >= 2) return (KA == 2 ? c : -c) + (KB == 2 ? s : (KB == 3 ? -s : s * b));
  else return M::fma(c, a, KB == 2 ? s : (KB == 3 ? -s : s * b));
}

// Forward kinematics.  Built-in chains: column j of (W * R_origin) is sgn[j] * W[:, perm[j]]
// (register renaming + sign), the translation touches only the non-zero origin components, and
// the joint rotation is a 2x2 rotation of two columns: 15 VALU ops + one sincos per joint.
// Generic chains take the full 3x3 products.
template <class C, typename T>
AE_DEV void fk(const ChainDev<T> &ch, const T (&cq)[NJ], const T (&sq)[NJ], FKState<T> &S) {
  using M = Mth<T>;
  if constexpr (C::kGeneric) {
    static_for<0, 9>([&](auto I) { constexpr int i = I; S.W[3 * (i % 3) + (i / 3)] = ch.base_R[i]; });
    static_for<0, 3>([&](auto I) { constexpr int i = I; S.p[i] = ch.base_p[i]; });
  } else {
    static_for<0, 9>([&](auto I) { constexpr int i = I; S.W[i] = (i % 4 == 0) ? T(1) : T(0); });
    static_for<0, 3>([&](auto I) { constexpr int i = I; S.p[i] = T(0); });
  }
  static_for<0, NJ>([&](auto JI) {
    constexpr int j = JI;
    T A[9];
    if constexpr (C::kGeneric) {
      static_for<0, 3>([&](auto RI) {
        constexpr int r = RI;
        S.p[r] = M::fma(S.W[0 + r], ch.xyz[j][0], M::fma(S.W[3 + r], ch.xyz[j][1], M::fma(S.W[6 + r], ch.xyz[j][2], S.p[r])));
      });
      // A[:,c] = sum_k W[:,k] * Ro[k][c]
      static_for<0, 3>([&](auto CI) {
        constexpr int c = CI;
        static_for<0, 3>([&](auto RI) {
          constexpr int r = RI;
          A[3 * c + r] = M::fma(S.W[0 + r], ch.R[j][0 + c], M::fma(S.W[3 + r], ch.R[j][3 + c], S.W[6 + r] * ch.R[j][6 + c]));
        });
      });
    } else {
      static_for<0, 3>([&](auto KI) {
        constexpr int k = KI;
        constexpr double t = C::xyz[j][k];
        if constexpr (t != 0.0) {
          static_for<0, 3>([&](auto RI) {
            constexpr int r = RI;
            constexpr int kw = FkKinds<C>().Wk[j][3 * k + r];
            if constexpr (kw == 0) S.p[r] = M::fma(S.W[3 * k + r], T(t), S.p[r]);
            else if constexpr (kw == 2) S.p[r] = S.p[r] + T(t);
            else if constexpr (kw == 3) S.p[r] = S.p[r] - T(t);
          });
        }
      });
      static_for<0, 3>([&](auto CI) {
        constexpr int c = CI;
        constexpr int k = C::perm[j][c];
        constexpr int s = C::sgn[j][c];
        static_for<0, 3>([&](auto RI) { constexpr int r = RI; A[3 * c + r] = (s > 0) ? S.W[3 * k + r] : -S.W[3 * k + r]; });
      });
    }
    const T sn = sq[j], cs = cq[j];
    static_for<0, 3>([&](auto RI) {
      constexpr int r = RI;
      if constexpr (C::kGeneric) {
        S.W[0 + r] = M::fma(cs, A[0 + r], sn * A[3 + r]);
        S.W[3 + r] = M::fma(cs, A[3 + r], -(sn * A[0 + r]));
      } else {
        constexpr int k0 = FkKinds<C>().Ak[j][0 + r], k3 = FkKinds<C>().Ak[j][3 + r];
        constexpr int k0n = k0 < 2 ? k0 : (k0 == 2 ? 3 : 2);   // kind of -A[0 + r]
        S.W[0 + r] = rot_term<k0, k3, T>(cs, A[0 + r], sn, A[3 + r]);
        S.W[3 + r] = rot_term<k3, k0n, T>(cs, A[3 + r], sn, -A[0 + r]);
      }
      S.W[6 + r] = A[6 + r];
      S.z[j][r] = A[6 + r];
      S.pj[j][r] = S.p[r];
    });
  });
}

// fdlibm __kernel_sin / __kernel_cos on [-pi/4, pi/4] (< 1 ulp)
template <typename T>
AE_DEV void sincos_kernel(T r, T &s, T &c) {
  using M = Mth<T>;
  const T z = r * r;
  T ps = T(1.58969099521155010221e-10);
  ps = M::fma(ps, z, T(-2.50507602534068634195e-08));
  ps = M::fma(ps, z, T(2.75573137070700676789e-06));
  ps = M::fma(ps, z, T(-1.98412698298579493134e-04));
  ps = M::fma(ps, z, T(8.33333333332248946124e-03));
  ps = M::fma(ps, z, T(-1.66666666666666324348e-01));
  s = M::fma(r * z, ps, r);
  T pc = T(-1.13596475577881948265e-11);
  pc = M::fma(pc, z, T(2.08757232129817482790e-09));
  pc = M::fma(pc, z, T(-2.75573143513906633035e-07));
  pc = M::fma(pc, z, T(2.48015872894767294178e-05));
  pc = M::fma(pc, z, T(-1.38888888888741095749e-03));
  pc = M::fma(pc, z, T(4.16666666666666019037e-02));
  c = M::fma(z * z, pc, M::fma(T(-0.5), z, T(1)));
}

// sin and cos of a joint angle: three-term Cody-Waite reduction by pi/2 (exact products for |x| up to ~1e5, far beyond
// any joint angle) + the kernels above + quadrant fix-up: half the instructions of the library sincos, whose
// Payne-Hanek path for huge arguments is dead weight here.  Non-finite or absurdly large inputs fall back to the library.
AE_DEV void sincos_joint(double x, double &s, double &c) {
  if (!(::fabs(x) < 1.0e5)) { ::sincos(x, &s, &c); return; }
  const double k = ::rint(x * 6.36619772367581382433e-01);
  double r = ::fma(-k, 1.57079632673412561417e+00, x);
  r = ::fma(-k, 6.07710050630396597660e-11, r);
  r = ::fma(-k, 2.02226624871116645580e-21, r);
  r = ::fma(-k, 8.47842766036889956997e-32, r);
  double sr, cr;
  sincos_kernel<double>(r, sr, cr);
  const int n = (int)k & 3;
  const double s1 = (n & 1) ? cr : sr, c1 = (n & 1) ? sr : cr;
  s = (n & 2) ? -s1 : s1;
  c = ((n + 1) & 2) ? -c1 : c1;
}
AE_DEV void sincos_joint(float x, float &s, float &c) { ::sincosf(x, &s, &c); }

template <typename T>
AE_DEV void sincos_all(const T (&q)[NJ], T (&cq)[NJ], T (&sq)[NJ]) {
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; sincos_joint(q[j], sq[j], cq[j]); });
}
// f64: the seven joints in lockstep.  Same operations per joint as sincos_joint (bitwise the same results), but stage by
// stage across the joints: every polynomial coefficient is materialised once instead of once per joint (a 64-bit
// literal costs two s_mov_b32, and one wave per SIMD pays for every scalar instruction) and the seven independent chains
// hide the 6-cycle latency of a dependent v_fma_f64.  No library fallback for huge arguments: the fused reduction stays
// exactly rounded while k = rint(2 q / pi) is an exact integer (|q| < 2^50), far beyond any joint angle a live env can
// hold, and the Payne-Hanek path it replaces was 1 400 instructions of never-executed code per call site -- three sites
// per kernel, sitting between the hot regions of a kernel whose instruction fetch starts cold at every launch.
template <>
AE_DEV void sincos_all<double>(const double (&q)[NJ], double (&cq)[NJ], double (&sq)[NJ]) {
  double k[NJ], r[NJ], z[NJ], ps[NJ], pc[NJ];
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; k[j] = ::rint(q[j] * 6.36619772367581382433e-01); });
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; r[j] = ::fma(-k[j], 1.57079632673412561417e+00, q[j]); });
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; r[j] = ::fma(-k[j], 6.07710050630396597660e-11, r[j]); });
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; r[j] = ::fma(-k[j], 2.02226624871116645580e-21, r[j]); });
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; r[j] = ::fma(-k[j], 8.47842766036889956997e-32, r[j]); z[j] = r[j] * r[j]; });
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; ps[j] = ::fma(1.58969099521155010221e-10, z[j], -2.50507602534068634195e-08); });
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; ps[j] = ::fma(ps[j], z[j], 2.75573137070700676789e-06); });
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; ps[j] = ::fma(ps[j], z[j], -1.98412698298579493134e-04); });
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; ps[j] = ::fma(ps[j], z[j], 8.33333333332248946124e-03); });
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; ps[j] = ::fma(ps[j], z[j], -1.66666666666666324348e-01); });
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; pc[j] = ::fma(-1.13596475577881948265e-11, z[j], 2.08757232129817482790e-09); });
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; pc[j] = ::fma(pc[j], z[j], -2.75573143513906633035e-07); });
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; pc[j] = ::fma(pc[j], z[j], 2.48015872894767294178e-05); });
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; pc[j] = ::fma(pc[j], z[j], -1.38888888888741095749e-03); });
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; pc[j] = ::fma(pc[j], z[j], 4.16666666666666019037e-02); });
  static_for<0, NJ>([&](auto JI) {
    constexpr int j = JI;
    const double sr = ::fma(r[j] * z[j], ps[j], r[j]);
    const double cr = ::fma(z[j] * z[j], pc[j], ::fma(-0.5, z[j], 1.0));
    const int n = (int)k[j] & 3;
    const double s1 = (n & 1) ? cr : sr, c1 = (n & 1) ? sr : cr;
    sq[j] = (n & 2) ? -s1 : s1;
    cq[j] = ((n + 1) & 2) ? -c1 : c1;
  });
}
// f32 engine: the same structure in single precision (three-part pi/2, fdlibm's k_sinf / k_cosf minimax coefficients):
// within 1 ulp of sincosf at 1 on |q| < 100 (7e-8 measured), far inside the f32 engine's 1e-4 step tolerance; accuracy
// degrades gradually beyond (float k stays exact to 2^24).
template <>
AE_DEV void sincos_all<float>(const float (&q)[NJ], float (&cq)[NJ], float (&sq)[NJ]) {
  float k[NJ], r[NJ], z[NJ], ps[NJ], pc[NJ];
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; k[j] = ::rintf(q[j] * 0.636619772f); });
  // pi/2 = 1.5703125 + 4.837512969970703125e-4 + 7.54978995489188216e-8 (Cody-Waite: the first two products are exact)
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; r[j] = ::fmaf(-k[j], 1.5703125f, q[j]); });
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; r[j] = ::fmaf(-k[j], 4.837512969970703125e-4f, r[j]); });
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; r[j] = ::fmaf(-k[j], 7.54978995489188216e-8f, r[j]); z[j] = r[j] * r[j]; });
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; ps[j] = ::fmaf(2.7183114939898219064e-6f, z[j], -1.98393348360966317347e-4f); });
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; ps[j] = ::fmaf(ps[j], z[j], 8.3333293858894631756e-3f); });
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; ps[j] = ::fmaf(ps[j], z[j], -1.66666666416265235595e-1f); });
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; pc[j] = ::fmaf(2.43904487962774090654e-5f, z[j], -1.38867637746099294692e-3f); });
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; pc[j] = ::fmaf(pc[j], z[j], 4.16666233237390631894e-2f); });
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; pc[j] = ::fmaf(pc[j], z[j], -4.99999997251031003120e-1f); });
  static_for<0, NJ>([&](auto JI) {
    constexpr int j = JI;
    const float sr = ::fmaf(r[j] * z[j], ps[j], r[j]);
    const float cr = ::fmaf(z[j], pc[j], 1.0f);
    const int n = (int)k[j] & 3;
    const float s1 = (n & 1) ? cr : sr, c1 = (n & 1) ? sr : cr;
    sq[j] = (n & 2) ? -s1 : s1;
    cq[j] = ((n + 1) & 2) ? -c1 : c1;
  });
}

// (c,s) <- (cos(q+d), sin(q+d)) from (cos q, sin q) for |d| <= pi/4 (the DLS scale-back bounds every update by
// max_dtheta), with the fdlibm __kernel_sin/__kernel_cos minimax polynomials (< 1 ulp on that interval): no
// range reduction, ~22 VALU ops instead of a full sincos per joint per IK trip.
template <typename T>
AE_DEV void rotate_small(T &c, T &s, T d) {
  using M = Mth<T>;
  T sd, cd;
  sincos_kernel<T>(d, sd, cd);
  const T cn = M::fma(c, cd, -(s * sd));
  const T sn = M::fma(s, cd, c * sd);
  c = cn;
  s = sn;
}

// reciprocal: v_rcp + two Newton steps (f64) / one (f32); used where a few-ulp quotient is enough (LDL^T pivots)
AE_DEV double hw_rcp(double x) { return __builtin_amdgcn_rcp(x); }
AE_DEV float hw_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
AE_DEV double hw_rsq(double x) { return __builtin_amdgcn_rsq(x); }
AE_DEV float hw_rsq(float x) { return __builtin_amdgcn_rsqf(x); }

// ARMENV_EXACT_RCP / _RSQRT / _ACOS / _ROTATE: one-change builds for tests/tools/accuracy_attribution.py (make variants): each
// replaces ONE of the kernel's fast paths by the library / IEEE operation it approximates.  Never defined in the product build.
template <typename T>
AE_DEV T fast_rcp(T d) {
#ifdef ARMENV_EXACT_RCP
  return T(1) / d;
#endif
  T r = hw_rcp(d);
  r = Mth<T>::fma(Mth<T>::fma(-d, r, T(1)), r, r);
  if constexpr (sizeof(T) == 8) r = Mth<T>::fma(Mth<T>::fma(-d, r, T(1)), r, r);
  return r;
}

// 1/sqrt(x): v_rsq + Newton steps (two in f64, one in f32); a few ulp, one short dependent chain instead of the
// sqrt-then-divide pair (each a long quarter-rate sequence in f64)
template <typename T>
AE_DEV T fast_rsqrt(T x) {
#ifdef ARMENV_EXACT_RSQRT
  return T(1) / Mth<T>::sqrt(x);
#endif
  T r = hw_rsq(x);
  const T hx = T(0.5) * x;
  r = Mth<T>::fma(Mth<T>::fma(-hx * r, r, T(0.5)), r, r);
  if constexpr (sizeof(T) == 8) r = Mth<T>::fma(Mth<T>::fma(-hx * r, r, T(0.5)), r, r);
  return r;
}

// btMatrix3x3::getRotation (Bullet src/LinearMath/btMatrix3x3.h): rotation matrix -> quaternion xyzw.
// m(r,c) = W[3*c + r].
template <typename T>
AE_DEV void quat_from_frame(const T (&W)[9], T (&q)[4]) {
  const T m00 = W[0], m10 = W[1], m20 = W[2], m01 = W[3], m11 = W[4], m21 = W[5], m02 = W[6], m12 = W[7], m22 = W[8];
  // Same case selection as getRotation (trace > 0, else the largest diagonal element), evaluated with selects
  // so that a wave whose lanes disagree on the case pays one sqrt and one divide, not four branches.
  // Every candidate is computed first and the ternaries below choose between VALUES: written as nested expressions
  // hipcc evaluates them lazily behind divergent branches (a dozen exec-mask regions per IK trip).
  const T trace = m00 + m11 + m22;
  const bool cw = trace > T(0);
  const bool cz = !cw & (m00 < m11 ? (m11 < m22) : (m00 < m22));
  const bool cy = !cw & !cz & (m00 < m11);
  // cx otherwise
  const T tw = trace + T(1), tz = m22 - m00 - m11 + T(1), ty = m11 - m22 - m00 + T(1), tx = m00 - m11 - m22 + T(1);
  const T t = cw ? tw : (cz ? tz : (cy ? ty : tx));
  const T h = T(0.5) * fast_rsqrt<T>(t);   // 0.5 / sqrt(t)
  const T big = t * h;                       // sqrt(t) * 0.5
  const T d21 = (m21 - m12) * h, d02 = (m02 - m20) * h, d10 = (m10 - m01) * h;
  T s10 = (m10 + m01) * h, s20 = (m20 + m02) * h, s21 = (m21 + m12) * h;
  asm("" : "+v"(s10), "+v"(s20), "+v"(s21));   // keep the three sums out of a `!cw` branch the compiler would build for them
  q[0] = cw ? d21 : (cz ? s20 : (cy ? s10 : big));
  q[1] = cw ? d02 : (cz ? s21 : (cy ? big : s10));
  q[2] = cw ? d10 : (cz ? big : (cy ? s21 : s20));
  q[3] = cw ? big : (cz ? d10 : (cy ? d02 : d21));
}

// acos on [-1, 1] without branches: fdlibm's e_acos.c rational approximation R(z) = p(z) / q(z), evaluated ONCE on
// z = x^2 (|x| < 0.5) or z = (1 - |x|) / 2 (otherwise), and the three result forms chosen with selects.  The library
// routine is the same arithmetic behind three divergent regions; inside the IK trip that costs a wave the exec-mask
// bookkeeping of every region on every trip and keeps the error chain out of the scheduler's reach.  ~1 ulp, and the
// engine rounds the angle through float anyway (IKParams::angle_f32).
AE_DEV double acos_branchfree(double x) {
#ifdef ARMENV_EXACT_ACOS
  return ::acos(x);
#endif
  const double ax = ::fabs(x);
  const bool small = ax < 0.5;
  const double z = small ? x * x : ::fma(-0.5, ax, 0.5);
  double p = ::fma(3.47933107596021167570e-05, z, 7.91534994289814532176e-04);
  p = ::fma(p, z, -4.00555345006794114027e-02);
  p = ::fma(p, z, 2.01212532134862925881e-01);
  p = ::fma(p, z, -3.25565818622400915405e-01);
  p = ::fma(p, z, 1.66666666666666657415e-01);
  p = p * z;
  double q = ::fma(7.70381505559019352791e-02, z, -6.88283971605453293030e-01);
  q = ::fma(q, z, 2.02094576023350569471e+00);
  q = ::fma(q, z, -2.40339491173441421878e+00);
  q = ::fma(q, z, 1.0);
  const double R = p * fast_rcp<double>(q);
  const double sr = z * fast_rsqrt<double>(small ? 1.0 : (z > 0.0 ? z : 1.0));   // sqrt(z) for the two outer forms
  const double s = z > 0.0 ? sr : 0.0;
  const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
  const double inner = pio2_hi - (x - ::fma(-x, R, pio2_lo));              // |x| < 0.5
  const double w = ::fma(R, s, s);                                          // s + s R
  const double outer = x < 0.0 ? ::fma(-2.0, w - pio2_lo, 3.14159265358979311600e+00) : 2.0 * w;
  return small ? inner : outer;
}

// IKTrajectoryHelper::computeIK orientation part: deltaQ = endQ * startQ^-1, angle = 2 acos(w) wrapped to
// (-pi, pi], e = angle * normalize(axis)  (axis = (1,0,0) when 1 - w^2 < 10 eps).
template <typename T>
AE_DEV void orientation_error(const T (&tq)[4], const T (&qc)[4], int angle_f32, T (&e)[3]) {
  using M = Mth<T>;
  const T bx = -qc[0], by = -qc[1], bz = -qc[2], bw = qc[3];
  const T ax = tq[0], ay = tq[1], az = tq[2], aw = tq[3];
  // explicit fma chains: the build uses -ffp-contract=off so that every kernel that inlines this code (step,
  // rollout, ik) rounds identically
  const T dx = M::fma(aw, bx, M::fma(ax, bw, M::fma(ay, bz, -(az * by))));
  const T dy = M::fma(aw, by, M::fma(ay, bw, M::fma(az, bx, -(ax * bz))));
  const T dz = M::fma(aw, bz, M::fma(az, bw, M::fma(ax, by, -(ay * bx))));
  const T dw = M::fma(aw, bw, -M::fma(ax, bx, M::fma(ay, by, az * bz)));
  if constexpr (sizeof(T) == 4) {
    // f32: 2*acos(w) is useless near convergence (w = 1 - k*6e-8 quantises the angle to ~7e-4*sqrt(k) rad),
    // so the f32 engine uses the well-conditioned equivalent angle = 2*atan2(|v|, |w|) on the short arc.
    const T sg = dw < T(0) ? T(-1) : T(1);
    const T n = M::sqrt(M::fma(dx, dx, M::fma(dy, dy, dz * dz)));
    const T ang = T(2) * ::atan2f(n, sg * dw);
    const T k = n > T(1e-30) ? sg * ang / n : T(0);
    e[0] = k * dx; e[1] = k * dy; e[2] = k * dz;
    return;
  }
  const T wc = dw < T(-1) ? T(-1) : (dw > T(1) ? T(1) : dw);
  T angle = T(2) * (T)acos_branchfree((double)wc);
  // axis = v / sqrt(1 - w^2), renormalised (btQuaternion::getAxis + btVector3::normalize): together v / |v|; the
  // degenerate branch (1 - w^2 < 10 eps -> axis (1,0,0)) is a select so that the whole function is one basic block
  // and its long dependent chain (acos, rsqrt) can be scheduled under the Jacobian / J J^T arithmetic
  const T s2 = M::fma(-dw, dw, T(1));
  const T v2 = M::fma(dx, dx, M::fma(dy, dy, dz * dz));
  const bool degenerate = (s2 < T(10) * M::eps) || !(v2 > T(0));
  const T rv = fast_rsqrt<T>(degenerate ? T(1) : v2);
  if (angle_f32) angle = (T)(float)angle;
  angle = angle > M::pi ? angle - T(2) * M::pi : angle;
  if (angle_f32) angle = (T)(float)angle;
  e[0] = degenerate ? angle : angle * (dx * rv);
  e[1] = degenerate ? T(0) : angle * (dy * rv);
  e[2] = degenerate ? T(0) : angle * (dz * rv);
}

// One damped-least-squares update in the dual 6x6 form  dtheta = J^T (J J^T + lambda I)^-1 e,
// algebraically identical to BussIK's (J^T J + lambda I) dtheta = J^T e (Jacobian::CalcDeltaThetasDLS2)
// but SPD, pivot-free (LDL^T) and well conditioned in f32.  J columns: [z_i x (p - p_i) ; z_i].
// First joint of a built-in chain: fk() starts from W = I, so its axis is the compile-time unit vector
// z_0 = sgn * e_m (m = C::perm[0][2]) and its Jacobian column [z_0 x r ; z_0] has three entries that are exact zeros
// and one that is exactly +-1.  k0_kind(row) classifies row `row` of that column: 0 run-time value, 1 exact zero,
// 2 exactly +1, 3 exactly -1.  Terms with a zero factor are left out and a unit factor becomes an add: the same bits
// (x * 1 is x, acc + 0 is acc), 25 instructions per trip fewer (jj_term below).  Generic chains (run-time base pose) keep every
// term.
template <class C, bool G = C::kGeneric> struct Axis0 { static constexpr int m = 2, sg = 1; };
template <class C> struct Axis0<C, false> { static constexpr int m = C::perm[0][2], sg = C::sgn[0][2]; };
template <class C> constexpr int k0_kind(int row) {
  if (C::kGeneric) return 0;
  const int m = Axis0<C>::m, sg = Axis0<C>::sg;
  if (row < 3) return row == m ? 1 : 0;
  return (row - 3 == m) ? (sg > 0 ? 2 : 3) : 1;
}
// the same for the later joints: their axes z_k = A[:,2] at joint k keep the exact zeros FkKinds knows about (KUKA:
// z_1 = (-s0, c0, 0)); the linear part z_k x r_k is treated as run-time values
template <class C, bool G = C::kGeneric> struct ZKind { static constexpr int of(int, int) { return 0; } };
template <class C> struct ZKind<C, false> { static constexpr int of(int k, int r) { return FkKinds<C>().Ak[k][6 + r]; } };
template <class C> constexpr int col_kind(int k, int row) {
  if (k == 0) return k0_kind<C>(row);
  return row < 3 ? 0 : ZKind<C>::of(k, row - 3);
}
template <class C, typename T, int K, int R, int CC>
AE_DEV T jj_term(T a, T b, T acc) {
  constexpr int ka = col_kind<C>(K, R), kb = col_kind<C>(K, CC);
  if constexpr (ka == 1 || kb == 1) return acc;
  else if constexpr (ka >= 2 && kb >= 2) return acc + ((ka == kb) ? T(1) : T(-1));
  else if constexpr (ka == 2) return acc + b;
  else if constexpr (ka == 3) return acc - b;
  else if constexpr (kb == 2) return acc + a;
  else if constexpr (kb == 3) return acc - a;
  else return Mth<T>::fma(a, b, acc);
}
// FENCE (the bookkeeping builds of the kernels, ArmEnvConfig.fence_counters): minpiv is lowered to the smallest of the six
// LDL^T pivots.  A compile-time switch: as a run-time branch in this loop body it cost the one-launch-per-step kernel 3 %
// with the bookkeeping OFF (A/B inside one GPU session, tests/tools/ab_time.sh).
// pe: the point the IK works at -- S.p (MODE 0 / 1) or S.p + W tip (MODE 2, where the last joint's lever arm is a run-time value
// and its column stays in).
template <class C, typename T, int MODE = 0>
AE_DEV void dls_update(const FKState<T> &S, const T (&pe)[3], const T (&e)[6], const IKParams<T> &P, T (&dth)[NJ], T &minpiv) {
  using M = Mth<T>;
  constexpr bool FENCE = kFenceOf(MODE);
  // fk() defines the end-effector point as the LAST joint's pivot (S.p == S.pj[NJ-1], the same values), so the lever arm of
  // the last joint is x - x = +0 and its linear Jacobian column an exact zero: every term it enters adds +-0.  That column
  // is left out of the build, of J J^T and of J^T y -- 27 instructions per trip, the same bits for every finite state.
  constexpr int NL = kTipOf(MODE) ? NJ : NJ - 1;
  T Jl[NL][3];
  static_for<0, NL>([&](auto II) {
    constexpr int i = II;
    const T r0 = pe[0] - S.pj[i][0], r1 = pe[1] - S.pj[i][1], r2 = pe[2] - S.pj[i][2];
    if constexpr (i == 0 && !C::kGeneric) {
      // z_0 = sg e_m:  (z_0 x r)_m = 0,  (z_0 x r)_(m+1) = -sg r_(m+2),  (z_0 x r)_(m+2) = +sg r_(m+1)
      constexpr int m = Axis0<C>::m, sg = Axis0<C>::sg;
      const T r[3] = {r0, r1, r2};
      Jl[0][m] = T(0);
      Jl[0][(m + 1) % 3] = sg > 0 ? -r[(m + 2) % 3] : r[(m + 2) % 3];
      Jl[0][(m + 2) % 3] = sg > 0 ? r[(m + 1) % 3] : -r[(m + 1) % 3];
    } else {
      // fma(z_a, r_b, -(z_c r_d)) = rot_term(r_b, z_a, r_d, -z_c): the kinds sit on the z factors
      constexpr int kz0 = ZKind<C>::of(i, 0), kz1 = ZKind<C>::of(i, 1), kz2 = ZKind<C>::of(i, 2);
      constexpr auto neg = [](int k) constexpr { return k < 2 ? k : (k == 2 ? 3 : 2); };
      Jl[i][0] = rot_term<kz1, neg(kz2), T>(r2, S.z[i][1], r1, -S.z[i][2]);
      Jl[i][1] = rot_term<kz2, neg(kz0), T>(r0, S.z[i][2], r2, -S.z[i][0]);
      Jl[i][2] = rot_term<kz0, neg(kz1), T>(r1, S.z[i][0], r0, -S.z[i][1]);
    }
  });
  // A = J J^T + lambda I, lower triangle, A[r][c] with row r of J = (r<3 ? Jl[.][r] : z[.][r-3])
  T A[6][6];
  static_for<0, 6>([&](auto RI) {
    constexpr int r = RI;
    static_for<0, r + 1>([&](auto CI) {
      constexpr int c = CI;
      T acc = (r == c) ? P.lambda : T(0);
      static_for<0, (r < 3 || c < 3) ? NL : NJ>([&](auto KI) {
        constexpr int k = KI;
        const T a = (r < 3) ? Jl[k < NL ? k : 0][r % 3] : S.z[k][r % 3];
        const T b = (c < 3) ? Jl[k < NL ? k : 0][c % 3] : S.z[k][c % 3];
        acc = jj_term<C, T, k, r, c>(a, b, acc);
      });
      A[r][c] = acc;
    });
  });
  // LDL^T (no square roots, no pivoting: A is SPD)
  T L[6][6], invD[6], D[6];
  static_for<0, 6>([&](auto JI) {
    constexpr int j = JI;
    T v[6];
    T d = A[j][j];
    static_for<0, j>([&](auto KI) {
      constexpr int k = KI;
      v[k] = L[j][k] * D[k];
      d = M::fma(-L[j][k], v[k], d);
    });
    D[j] = d;
    invD[j] = fast_rcp<T>(d);
    static_for<j + 1, 6>([&](auto II) {
      constexpr int i = II;
      T s = A[i][j];
      static_for<0, j>([&](auto KI) {
        constexpr int k = KI;
        s = M::fma(-L[i][k], v[k], s);
      });
      L[i][j] = s * invD[j];
    });
  });
  if constexpr (FENCE) {
    const T m = M::fmin(M::fmin(M::fmin(D[0], D[1]), M::fmin(D[2], D[3])), M::fmin(D[4], D[5]));
    minpiv = M::fmin(minpiv, m);
  }
  // L zf = e ; w = zf / D ; L^T y = w
  T y[6];
  static_for<0, 6>([&](auto II) {
    constexpr int i = II;
    T s = e[i];
    static_for<0, i>([&](auto KI) { constexpr int k = KI; s = M::fma(-L[i][k], y[k], s); });
    y[i] = s;
  });
  static_for<0, 6>([&](auto II) { constexpr int i = II; y[i] *= invD[i]; });
  static_for<0, 6>([&](auto II) {
    constexpr int i = 5 - II;
    T s = y[i];
    static_for<i + 1, 6>([&](auto KI) { constexpr int k = KI; s = M::fma(-L[k][i], y[k], s); });
    y[i] = s;
  });
  // dtheta = J^T y, then Jacobian::MaxAngleDLS scale-back
  T mx = T(0);
  static_for<0, NJ>([&](auto II) {
    constexpr int i = II;
    T s;
    if constexpr (i == 0 && !C::kGeneric) {
      // the same chain of six terms with the exact zeros left out and the unit factor as an add
      s = T(0);
      bool first = true;
      static_for<0, 6>([&](auto RI) {
        constexpr int r = RI;
        constexpr int kind = k0_kind<C>(r);
        const T a = (r < 3) ? Jl[0][r % 3] : S.z[0][r % 3];
        if constexpr (kind == 0) { s = first ? a * y[r] : M::fma(a, y[r], s); first = false; }
        else if constexpr (kind == 2) { s = first ? y[r] : s + y[r]; first = false; }
        else if constexpr (kind == 3) { s = first ? -y[r] : s - y[r]; first = false; }
      });
    } else if constexpr (i < NL) {
      s = Jl[i][0] * y[0];
      s = M::fma(Jl[i][1], y[1], s);
      s = M::fma(Jl[i][2], y[2], s);
      static_for<0, 3>([&](auto RI) {
        constexpr int r = RI;
        constexpr int kz = ZKind<C>::of(i, r);
        if constexpr (kz == 0) s = M::fma(S.z[i][r], y[3 + r], s);
        else if constexpr (kz == 2) s = s + y[3 + r];
        else if constexpr (kz == 3) s = s - y[3 + r];
      });
    } else {
      s = S.z[i][0] * y[3];
      s = M::fma(S.z[i][1], y[4], s);
      s = M::fma(S.z[i][2], y[5], s);
    }
    dth[i] = s;
    mx = M::fmax(mx, M::fabs(s));
  });
  // Jacobian::MaxAngleDLS scale-back as a select (no branch: keeps the caller's update in one basic block)
  const bool over = mx > P.max_dtheta;
  const T sc = over ? P.max_dtheta * fast_rcp<T>(over ? mx : T(1)) : T(1);
  // one select on the factor instead of seven on the products: x * 1 is x, bit for bit
  static_for<0, NJ>([&](auto II) { constexpr int i = II; dth[i] = dth[i] * sc; });
}

// The arm move of one env step.  With FROM_ACTION the Cartesian target is built from the first FK:
//   tgt = clip(p(q) + dv * a, box)           /root/reference/envs/rl_reach_env.py:231-242
// then pybullet's IK loop runs (Bullet PhysicsServerCommandProcessor::processCalculateInverseKinematicsCommand):
//   for (i = 0; i < maxIter && currentDiff > residual; ++i) { currentDiff = |p(q) - tgt|; q += dls(q); }
// Each trip of the loop below does exactly one FK; the trip that decides to stop leaves S = FK(q_final),
// which is the post-step FK of _reward() (:271).  Returns the number of updates applied.
// START_F32: the start position is rounded through float before the action is added (rl_pick_env.py:328 casts
// getLinkState's tuple to np.float32; reach / push keep the f64 tuple).
// The three pieces of the arm move, shared by the lockstep loop of ik_move() below and by the lane-asynchronous rollout
// kernel (armenv_env.h env_rollout_async_kernel), where every lane walks through its own (step, trip) sequence.
//
// ik_target: tgt = clip(p(q) + dv * a, box) from the frame S of the start pose (rl_reach_env.py:231-242).
template <typename T, bool START_F32>
AE_DEV void ik_target(const FKState<T> &S, const T (&a)[3], T dv, const T (&box_lo)[3], const T (&box_hi)[3], T (&tgt)[3]) {
  using M = Mth<T>;
  static_for<0, 3>([&](auto KI) {
    constexpr int k = KI;
    T v = M::fma(a[k], dv, START_F32 ? (T)(float)S.p[k] : S.p[k]);
    v = v < box_lo[k] ? box_lo[k] : v;   // clip_val, rl_reach_env.py:225-230
    v = v > box_hi[k] ? box_hi[k] : v;
    tgt[k] = v;
  });
}
// ik_trip: one trip of Bullet's loop on the pose whose frame is S.  Returns true when the loop stops (nothing changes);
// otherwise applies one DLS update to q and (cq, sq), leaves S = FK(q) of the updated pose and counts the update in `it`.
// minpiv (fence bookkeeping only): running minimum of the LDL^T pivots of the call's damped systems -- the damped solve
// amplifies rounding differences by ~1 / pivot, so a call that passes through a near-singular pose (stretched elbow at the
// edge of the arm's reach, aligned wrist) is where two implementations' trajectories start to part.
// The trip in two pieces, for callers that steer the wave themselves (env_step's lockstep loop, armenv_env.h):
//   ik_stop    position error e[0..2] at the IK point pe of the frame S, its square diff2, and Bullet's loop test -- true when the
//              loop stops for this lane (nothing changes);
//   ik_update  one DLS update of q and (cq, sq) from that error, S = FK(q) of the updated pose, the update counted in `it`.
template <class C, typename T, int MODE = 0>
AE_DEV bool ik_stop(const IKParams<T> &P, const T (&tgt)[3], const FKState<T> &S, T diff2_prev, int it, T res2, T (&e)[6], T (&pe)[3],
                    T &diff2) {
  using M = Mth<T>;
  static_for<0, 3>([&](auto RI) {
    constexpr int r = RI;
    if constexpr (kTipOf(MODE)) pe[r] = M::fma(S.W[0 + r], P.tip[0], M::fma(S.W[3 + r], P.tip[1], M::fma(S.W[6 + r], P.tip[2], S.p[r])));
    else pe[r] = S.p[r];
  });
  e[0] = tgt[0] - pe[0];
  e[1] = tgt[1] - pe[1];
  e[2] = tgt[2] - pe[2];
  diff2 = M::fma(e[0], e[0], M::fma(e[1], e[1], e[2] * e[2]));
  // exit_mode is wave-uniform: written as a ternary on P.exit_mode hipcc builds a uniform BRANCH around the two compares, and
  // lays the default mode's side out of line -- two taken branches per trip, ~25 ns each for a wave that has its SIMD to itself
  // (tests/tools/exp/fwd_branch_probe.hip).  With the mode in a vector register the choice is two v_cndmask.
  int em = P.exit_mode;
  asm("" : "+v"(em));
  const T dtest = em == 0 ? diff2_prev : diff2;
  return (it >= P.max_iters) | !(dtest > res2);
}
// SMALL: 1 / 0 = the caller has already branched on small_steps (a compile-time constant here: the lockstep loop tests it once per
// step instead of once per trip), -1 = the run-time argument decides.
template <class C, typename T, int MODE = 0, int SMALL = -1>
AE_DEV void ik_update(const ChainDev<T> &ch, const IKParams<T> &P, T (&q)[NJ], FKState<T> &S, T (&cq)[NJ], T (&sq)[NJ], T (&e)[6],
                      const T (&pe)[3], T diff2, T &diff2_prev, int &it, bool small_steps, T &minpiv) {
  T qc[4], eo[3], dth[NJ];
  quat_from_frame<T>(S.W, qc);
  orientation_error<T>(P.tq, qc, P.angle_f32, eo);
  e[3] = eo[0]; e[4] = eo[1]; e[5] = eo[2];
  dls_update<C, T, MODE>(S, pe, e, P, dth, minpiv);
  static_for<0, NJ>([&](auto II) { constexpr int i = II; q[i] += dth[i]; });
  if constexpr (SMALL >= 0) small_steps = SMALL != 0;
#ifdef ARMENV_EXACT_ROTATE
  small_steps = false;
#endif
  if (__builtin_expect(small_steps, 1)) {     // (the other side out of line: the loop body then ends in ONE back edge)
    static_for<0, NJ>([&](auto II) { constexpr int i = II; rotate_small<T>(cq[i], sq[i], dth[i]); });
  } else {
    sincos_all<T>(q, cq, sq);
  }
  diff2_prev = diff2;
  ++it;
  fk<C, T>(ch, cq, sq, S);
}
template <class C, typename T, int MODE = 0, int SMALL = -1>
AE_DEV bool ik_trip(const ChainDev<T> &ch, const IKParams<T> &P, T (&q)[NJ], const T (&tgt)[3], FKState<T> &S, T (&cq)[NJ],
                    T (&sq)[NJ], T &diff2_prev, int &it, T res2, bool small_steps, T &minpiv) {
  T e[6], pe[3], diff2;
  if (ik_stop<C, T, MODE>(P, tgt, S, diff2_prev, it, res2, e, pe, diff2)) return true;
  ik_update<C, T, MODE, SMALL>(ch, P, q, S, cq, sq, e, pe, diff2, diff2_prev, it, small_steps, minpiv);
  return false;
}
// Bullet's loop for the lanes of a wave walking through it together (the lockstep kernels).  Returns the lane's number of updates.
// PEEL: the first PEEL trips are straight-line code, each under the mask of the lanes that have not stopped: nearly every step needs
// them (reach: 2 updates on 8 % of the steps, 3 on 92 %), and a loop's taken back edge costs a lone wave ~47 ns
// (tests/tools/exp/branch_cost_probe.hip) where a not-taken skip costs ~8.  The loop behind them takes
// what is left.  The rollout kernels peel three trips (reach -2.4 %, push -1.3 % per step); the one-step kernel none: its
// instruction fetch starts cold at every launch, and 24 KB more code cost it 2-6 %.
// (A form with the wave-level exit spelled out -- `if (__ballot(!stop) == 0) break; if (!stop) ik_update(...)` -- was tried to save
// the loop's entry jump and one of its two exit branches: hipcc's structuriser folds it back into the same exec-mask loop.)
template <class C, typename T, int MODE = 0, int PEEL = 0>
AE_DEV int ik_lockstep(const ChainDev<T> &ch, const IKParams<T> &P, T (&q)[NJ], const T (&tgt)[3], FKState<T> &S, T (&cq)[NJ],
                       T (&sq)[NJ], T &minpiv) {
  const T res2 = P.residual * P.residual;     // |p - tgt| > residual on squares: no sqrt on the loop-carried critical path
  // every update is bounded by max_dtheta; up to pi/4 (Bullet's 45 degrees) the rotations are advanced incrementally,
  // otherwise cos / sin are recomputed from q
  const bool small_steps = P.max_dtheta <= T(0.7854);
  T diff2_prev = T(1e60);
  int it = 0;
  // (Branching on small_steps HERE, once per step, with two compile-time copies of the trips behind it, was tried: the second copy
  // of the loop raised the register pressure -- 124 -> 181 AGPRs of spill space in the reach rollout -- and cost 8 %.)
  bool stopped = false;
  static_for<0, PEEL>([&](auto) {
    if (!stopped) stopped = ik_trip<C, T, MODE>(ch, P, q, tgt, S, cq, sq, diff2_prev, it, res2, small_steps, minpiv);
  });
  if (!stopped) while (!ik_trip<C, T, MODE>(ch, P, q, tgt, S, cq, sq, diff2_prev, it, res2, small_steps, minpiv)) {}
  return it;
}

// ik_limits: URDF joint limits (/root/reference/envs/bmirobot_joints_info_pybullet.txt:1-7, fields 8-9).  The reference never
// passes them to the IK (rl_reach_env.py:103-107 are dead data, :244-250), so q may leave them; Bullet then pushes the joint
// back inside stepSimulation (:258) through a limit constraint this build does not model.  Returns whether the IK result
// lies outside the limits (the parity fence, counted by the caller).  clamp_limits == 1: the result is projected onto them
// -- the hard-limit idealisation of that constraint; clamp_limits == 2: every joint beyond a limit is moved back by the share
// limit_erp of its violation -- one velocity-level constraint solve with Baumgarte stabilisation, as Bullet's
// btMultiBodyJointLimitConstraint does once per stepSimulation (erp 0.2 by default; a named, unpinned model: the first box
// with pybullet fits one scalar).  Either way the frame is recomputed for the lanes that left the limits only (the others
// keep their bits).
template <class C, typename T, int MODE = 0>
AE_DEV bool ik_limits(const ChainDev<T> &ch, const IKParams<T> &P, T (&q)[NJ], FKState<T> &S, T (&cq)[NJ], T (&sq)[NJ]) {
  using M = Mth<T>;
  constexpr bool FENCE = kFenceOf(MODE);
  bool hit = false;
  if (FENCE || __builtin_expect(P.clamp_limits != 0, 0)) {
    T m = M::fabs(q[0]);
    static_for<1, NJ>([&](auto II) { constexpr int i = II; m = M::fmax(m, M::fabs(q[i])); });
    if (__builtin_expect(m > P.lim_min, 0)) {
      T lo[NJ], hi[NJ];
      static_for<0, NJ>([&](auto II) { constexpr int i = II; lo[i] = P.lim[i]; hi[i] = P.lim[NJ + i]; });
      static_for<0, NJ>([&](auto II) { constexpr int i = II; hit = hit | (q[i] < lo[i]) | (q[i] > hi[i]); });
      if (P.clamp_limits && hit) {
        if (P.clamp_limits == 2) {
          const T erp = P.lim[2 * NJ];
          static_for<0, NJ>([&](auto II) {
            constexpr int i = II;
            q[i] = q[i] < lo[i] ? M::fma(erp, lo[i] - q[i], q[i]) : (q[i] > hi[i] ? M::fma(erp, hi[i] - q[i], q[i]) : q[i]);
          });
        } else {
          static_for<0, NJ>([&](auto II) { constexpr int i = II; q[i] = q[i] < lo[i] ? lo[i] : (q[i] > hi[i] ? hi[i] : q[i]); });
        }
        sincos_all<T>(q, cq, sq);
        fk<C, T>(ch, cq, sq, S);
      }
    }
  }
  return hit;
}

template <class C, typename T, bool FROM_ACTION, bool START_F32 = false, int MODE = 0>
AE_DEV int ik_move(const ChainDev<T> &ch, const IKParams<T> &P, T (&q)[NJ], T (&tgt)[3], const T (&a)[3], T dv,
                   const T (&box_lo)[3], const T (&box_hi)[3], FKState<T> &S, T (*p_start)[3] = nullptr,
                   T (*cq_io)[NJ] = nullptr, T (*sq_io)[NJ] = nullptr, bool *limit_hit = nullptr, bool frame_valid = false) {
  // the residual test |p - tgt| > residual is evaluated on squares (no sqrt on the loop-carried critical path)
  const T res2 = P.residual * P.residual;
  T diff2_prev = T(1e60);
  int it = 0;
  T cq[NJ], sq[NJ];
  if (cq_io) { static_for<0, NJ>([&](auto JI) { constexpr int j = JI; cq[j] = (*cq_io)[j]; sq[j] = (*sq_io)[j]; }); }
  else sincos_all<T>(q, cq, sq);
  // every update is bounded by max_dtheta; up to pi/4 (Bullet's 45 degrees) the rotations are advanced
  // incrementally, otherwise cos/sin are recomputed from q
  const bool small_steps = P.max_dtheta <= T(0.7854);
  // The loop is written rotated -- FK of the start pose and the target ahead of it, the FK of each updated pose at its
  // bottom -- so that the target's inputs (action, dv, the workspace box: 14 scalar registers) are dead across the trips.
  // frame_valid: S already is FK of (cq, sq) -- the exit FK of the caller's previous step (the same inputs give the same bits).
  if (!frame_valid) fk<C, T>(ch, cq, sq, S);
  if constexpr (FROM_ACTION) {
    if (p_start) { (*p_start)[0] = S.p[0]; (*p_start)[1] = S.p[1]; (*p_start)[2] = S.p[2]; }
    ik_target<T, START_F32>(S, a, dv, box_lo, box_hi, tgt);
  }
  T minpiv = T(1e30);
  while (!ik_trip<C, T, MODE>(ch, P, q, tgt, S, cq, sq, diff2_prev, it, res2, small_steps, minpiv)) {}
  const bool hit = ik_limits<C, T, MODE>(ch, P, q, S, cq, sq);
  if (limit_hit) *limit_hit = hit;
  if (cq_io) { static_for<0, NJ>([&](auto JI) { constexpr int j = JI; (*cq_io)[j] = cq[j]; (*sq_io)[j] = sq[j]; }); }
  return it;
}

}  // namespace armenv
