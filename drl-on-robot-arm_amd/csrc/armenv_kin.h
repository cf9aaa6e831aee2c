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
  T lim_lo[NJ];
  T lim_hi[NJ];
};

// World frame of a link: W holds the rotation as three column vectors W[3*col + row].
template <typename T> struct FKState {
  T W[9];
  T p[3];
  T z[NJ][3];   // world joint axes
  T pj[NJ][3];  // world joint pivots
};

// Forward kinematics.  Built-in chains: column j of (W * R_origin) is sgn[j] * W[:, perm[j]]
// (register renaming + sign), the translation touches only the non-zero origin components, and
// the joint rotation is a 2x2 rotation of two columns: 15 VALU ops + one sincos per joint.
// Generic chains take the full 3x3 products.
template <class C, typename T>
AE_DEV void fk(const ChainDev<T> &ch, const T (&q)[NJ], FKState<T> &S) {
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
          static_for<0, 3>([&](auto RI) { constexpr int r = RI; S.p[r] = M::fma(S.W[3 * k + r], T(t), S.p[r]); });
        }
      });
      static_for<0, 3>([&](auto CI) {
        constexpr int c = CI;
        constexpr int k = C::perm[j][c];
        constexpr int s = C::sgn[j][c];
        static_for<0, 3>([&](auto RI) { constexpr int r = RI; A[3 * c + r] = (s > 0) ? S.W[3 * k + r] : -S.W[3 * k + r]; });
      });
    }
    T sn, cs;
    M::sincos(q[j], sn, cs);
    static_for<0, 3>([&](auto RI) {
      constexpr int r = RI;
      S.W[0 + r] = M::fma(cs, A[0 + r], sn * A[3 + r]);
      S.W[3 + r] = M::fma(cs, A[3 + r], -(sn * A[0 + r]));
      S.W[6 + r] = A[6 + r];
      S.z[j][r] = A[6 + r];
      S.pj[j][r] = S.p[r];
    });
  });
}

// btMatrix3x3::getRotation (Bullet src/LinearMath/btMatrix3x3.h): rotation matrix -> quaternion xyzw.
// m(r,c) = W[3*c + r].
template <typename T>
AE_DEV void quat_from_frame(const T (&W)[9], T (&q)[4]) {
  using M = Mth<T>;
  const T m00 = W[0], m10 = W[1], m20 = W[2], m01 = W[3], m11 = W[4], m21 = W[5], m02 = W[6], m12 = W[7], m22 = W[8];
  const T trace = m00 + m11 + m22;
  if (trace > T(0)) {
    T s = M::sqrt(trace + T(1));
    q[3] = s * T(0.5);
    s = T(0.5) / s;
    q[0] = (m21 - m12) * s;
    q[1] = (m02 - m20) * s;
    q[2] = (m10 - m01) * s;
  } else if (m00 < m11 ? (m11 < m22) : (m00 < m22)) {  // i = 2, j = 0, k = 1
    T s = M::sqrt(m22 - m00 - m11 + T(1));
    q[2] = s * T(0.5);
    s = T(0.5) / s;
    q[3] = (m10 - m01) * s;
    q[0] = (m02 + m20) * s;
    q[1] = (m12 + m21) * s;
  } else if (m00 < m11) {  // i = 1, j = 2, k = 0
    T s = M::sqrt(m11 - m22 - m00 + T(1));
    q[1] = s * T(0.5);
    s = T(0.5) / s;
    q[3] = (m02 - m20) * s;
    q[2] = (m21 + m12) * s;
    q[0] = (m01 + m10) * s;
  } else {  // i = 0, j = 1, k = 2
    T s = M::sqrt(m00 - m11 - m22 + T(1));
    q[0] = s * T(0.5);
    s = T(0.5) / s;
    q[3] = (m21 - m12) * s;
    q[1] = (m10 + m01) * s;
    q[2] = (m20 + m02) * s;
  }
}

// IKTrajectoryHelper::computeIK orientation part: deltaQ = endQ * startQ^-1, angle = 2 acos(w) wrapped to
// (-pi, pi], e = angle * normalize(axis)  (axis = (1,0,0) when 1 - w^2 < 10 eps).
template <typename T>
AE_DEV void orientation_error(const T (&tq)[4], const T (&qc)[4], int angle_f32, T (&e)[3]) {
  using M = Mth<T>;
  const T bx = -qc[0], by = -qc[1], bz = -qc[2], bw = qc[3];
  const T ax = tq[0], ay = tq[1], az = tq[2], aw = tq[3];
  const T dx = aw * bx + ax * bw + ay * bz - az * by;
  const T dy = aw * by + ay * bw + az * bx - ax * bz;
  const T dz = aw * bz + az * bw + ax * by - ay * bx;
  const T dw = aw * bw - ax * bx - ay * by - az * bz;
  if constexpr (sizeof(T) == 4) {
    // f32: 2*acos(w) is useless near convergence (w = 1 - k*6e-8 quantises the angle to ~7e-4*sqrt(k) rad),
    // so the f32 engine uses the well-conditioned equivalent angle = 2*atan2(|v|, |w|) on the short arc.
    const T sg = dw < T(0) ? T(-1) : T(1);
    const T n = M::sqrt(dx * dx + dy * dy + dz * dz);
    const T ang = T(2) * ::atan2f(n, sg * dw);
    const T k = n > T(1e-30) ? sg * ang / n : T(0);
    e[0] = k * dx; e[1] = k * dy; e[2] = k * dz;
    return;
  }
  const T wc = dw < T(-1) ? T(-1) : (dw > T(1) ? T(1) : dw);
  T angle = T(2) * M::acos(wc);
  const T s2 = T(1) - dw * dw;
  T a0, a1, a2;
  if (s2 < T(10) * M::eps) {
    a0 = T(1); a1 = T(0); a2 = T(0);
  } else {
    const T s = T(1) / M::sqrt(s2);
    a0 = dx * s; a1 = dy * s; a2 = dz * s;
  }
  if (angle_f32) angle = (T)(float)angle;
  if (angle > M::pi) angle -= T(2) * M::pi;
  if (angle_f32) angle = (T)(float)angle;
  const T n = M::sqrt(a0 * a0 + a1 * a1 + a2 * a2);
  e[0] = angle * (a0 / n);
  e[1] = angle * (a1 / n);
  e[2] = angle * (a2 / n);
}

// One damped-least-squares update in the dual 6x6 form  dtheta = J^T (J J^T + lambda I)^-1 e,
// algebraically identical to BussIK's (J^T J + lambda I) dtheta = J^T e (Jacobian::CalcDeltaThetasDLS2)
// but SPD, pivot-free (LDL^T) and well conditioned in f32.  J columns: [z_i x (p - p_i) ; z_i].
template <typename T>
AE_DEV void dls_update(const FKState<T> &S, const T (&e)[6], const IKParams<T> &P, T (&dth)[NJ]) {
  using M = Mth<T>;
  T Jl[NJ][3];
  static_for<0, NJ>([&](auto II) {
    constexpr int i = II;
    const T r0 = S.p[0] - S.pj[i][0], r1 = S.p[1] - S.pj[i][1], r2 = S.p[2] - S.pj[i][2];
    Jl[i][0] = S.z[i][1] * r2 - S.z[i][2] * r1;
    Jl[i][1] = S.z[i][2] * r0 - S.z[i][0] * r2;
    Jl[i][2] = S.z[i][0] * r1 - S.z[i][1] * r0;
  });
  // A = J J^T + lambda I, lower triangle, A[r][c] with row r of J = (r<3 ? Jl[.][r] : z[.][r-3])
  T A[6][6];
  static_for<0, 6>([&](auto RI) {
    constexpr int r = RI;
    static_for<0, r + 1>([&](auto CI) {
      constexpr int c = CI;
      T acc = (r == c) ? P.lambda : T(0);
      static_for<0, NJ>([&](auto KI) {
        constexpr int k = KI;
        const T a = (r < 3) ? Jl[k][r % 3] : S.z[k][r % 3];
        const T b = (c < 3) ? Jl[k][c % 3] : S.z[k][c % 3];
        acc = M::fma(a, b, acc);
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
    invD[j] = T(1) / d;
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
    T s = Jl[i][0] * y[0];
    s = M::fma(Jl[i][1], y[1], s);
    s = M::fma(Jl[i][2], y[2], s);
    s = M::fma(S.z[i][0], y[3], s);
    s = M::fma(S.z[i][1], y[4], s);
    s = M::fma(S.z[i][2], y[5], s);
    dth[i] = s;
    mx = M::fmax(mx, M::fabs(s));
  });
  if (mx > P.max_dtheta) {
    const T sc = P.max_dtheta / mx;
    static_for<0, NJ>([&](auto II) { constexpr int i = II; dth[i] *= sc; });
  }
}

// The arm move of one env step.  With FROM_ACTION the Cartesian target is built from the first FK:
//   tgt = clip(p(q) + dv * a, box)           /root/reference/envs/rl_reach_env.py:231-242
// then pybullet's IK loop runs (Bullet PhysicsServerCommandProcessor::processCalculateInverseKinematicsCommand):
//   for (i = 0; i < maxIter && currentDiff > residual; ++i) { currentDiff = |p(q) - tgt|; q += dls(q); }
// Each trip of the loop below does exactly one FK; the trip that decides to stop leaves S = FK(q_final),
// which is the post-step FK of _reward() (:271).  Returns the number of updates applied.
template <class C, typename T, bool FROM_ACTION>
AE_DEV int ik_move(const ChainDev<T> &ch, const IKParams<T> &P, T (&q)[NJ], T (&tgt)[3], const T (&a)[3], T dv,
                   const T (&box_lo)[3], const T (&box_hi)[3], FKState<T> &S) {
  using M = Mth<T>;
  T diff_prev = T(1e30);
  int it = 0;
  for (;; ++it) {
    fk<C, T>(ch, q, S);
    if constexpr (FROM_ACTION) {
      if (it == 0) {
        static_for<0, 3>([&](auto KI) {
          constexpr int k = KI;
          T v = M::fma(a[k], dv, S.p[k]);
          v = v < box_lo[k] ? box_lo[k] : v;   // clip_val, rl_reach_env.py:225-230
          v = v > box_hi[k] ? box_hi[k] : v;
          tgt[k] = v;
        });
      }
    }
    T e[6];
    e[0] = tgt[0] - S.p[0];
    e[1] = tgt[1] - S.p[1];
    e[2] = tgt[2] - S.p[2];
    const T diff = M::sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    const bool stop = (it >= P.max_iters) || (P.exit_mode == 0 ? !(diff_prev > P.residual) : !(diff > P.residual));
    if (stop) break;
    T qc[4], eo[3], dth[NJ];
    quat_from_frame<T>(S.W, qc);
    orientation_error<T>(P.tq, qc, P.angle_f32, eo);
    e[3] = eo[0]; e[4] = eo[1]; e[5] = eo[2];
    dls_update<T>(S, e, P, dth);
    static_for<0, NJ>([&](auto II) { constexpr int i = II; q[i] += dth[i]; });
    diff_prev = diff;
  }
  if (P.clamp_limits) {
    static_for<0, NJ>([&](auto II) {
      constexpr int i = II;
      q[i] = q[i] < P.lim_lo[i] ? P.lim_lo[i] : (q[i] > P.lim_hi[i] ? P.lim_hi[i] : q[i]);
    });
    fk<C, T>(ch, q, S);
  }
  return it;
}

}  // namespace armenv
