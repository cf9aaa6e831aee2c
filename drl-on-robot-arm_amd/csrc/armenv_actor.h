// armenv_actor.h -- the TD3 actor (PolicyNet, /root/reference/algo/TD3/net_mlp.py:29-40) evaluated per wavefront
// for the 64 envs a wave owns, exact f32:
//     a = action_bound * tanh(W3 relu(W2 relu(W1 s + b1) + b2) + b3),   6|9 -> 256 -> 256 -> 3
//
// Layer 2 is the only dense contraction on the env path (131 kflop of the 136 kflop per env) and runs on the matrix
// cores with the f32-input MFMA (v_mfma_f32_32x32x2_f32: exact f32, bitwise an fmaf chain).  It is computed
// TRANSPOSED, H2^T[neuron][env] = W2 * H1^T, so that in the accumulator layout a lane holds 128 neurons of ONE env:
// layer 3's reduction over neurons is then in-register plus one lane<->lane+32 exchange, and no activation ever goes
// through LDS.  Per k-pair (k = 2kk, 2kk+1) a lane
//   - computes its B operand on the VALU: h1[env][k] for k = 2kk + (lane>>5) and env = (lane&31) [tile 0] and
//     (lane&31)+32 [tile 1]  (layer 1 is 6|9 FMAs per value -- cheaper to recompute in the operand layout than to move),
//   - fetches its A operand, W2[32 nt + (lane&31)][k] for the eight neuron tiles nt, as two 16-byte loads from a
//     pre-packed copy of W2 (256 KB, L2-resident, shared by every wave; prefetched one k-pair ahead; one 16-byte load
//     per pass),
//   - issues 8 MFMAs (4 neuron tiles x 2 env tiles) into 128 accumulator registers, in two passes over the tiles.
// 2048 MFMAs x 64 cycles per wave per step: the fused-actor configuration is MFMA-bound at the f32 matrix rate.
// No LDS, no barriers: waves of a workgroup stay unsynchronised, as in the rest of the rollout kernel.
#pragma once
#include "armenv_math.h"

namespace armenv {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int ACTOR_HID = 256;

struct ActorParams {
  const float4 *W1P;   // [256][3]  float4: w0..w3 | w4..w7 | w8, 0, 0, b1   (inputs beyond in_dim are zero)
  const float4 *W2P;   // [256 k][2 part][32 lane]: (W2[32*(4 part + c) + lane][k], c = 0..3)
  const float4 *B2W3;  // [256]: (b2[n], W3[0][n], W3[1][n], W3[2][n])
  float b3[3];
  float bound;
  int32_t in_dim;      // 6 (reach obs) or 9 (push obs)
};

// s: this lane's env observation (IN floats).  All 64 lanes of the wave must be active.
template <int IN>
AE_DEV void actor_forward_wave(const ActorParams &A, const float (&s)[IN], float (&out)[3]) {
  const int lane = threadIdx.x & 63;
  const int half = lane >> 5;
  const int l32 = lane & 31;
  // observation of the env in tile 0 (env l32) and tile 1 (env l32 + 32) of this lane's MFMA column
  float sA[IN], sB[IN];
  static_for<0, IN>([&](auto DI) {
    constexpr int d = DI;
    const float other = __shfl_xor(s[d], 32);
    sA[d] = half ? other : s[d];
    sB[d] = half ? s[d] : other;
  });
  // Two passes of four neuron tiles (128 accumulator registers each) keep the whole kernel free of spills; layer 1
  // is recomputed per pass (16 VALU ops per k-pair beside 8 x 64 cycles of MFMA).
  float p[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
  const float4 *w2 = A.W2P + l32;
#pragma unroll 1
  for (int part = 0; part < 2; ++part) {
    f32x16 acc[4][2];
    static_for<0, 4>([&](auto NI) {
      constexpr int nt = NI;
      static_for<0, 16>([&](auto RI) { constexpr int r = RI; acc[nt][0][r] = 0.f; acc[nt][1][r] = 0.f; });
    });
    float4 a0 = w2[(half * 2 + part) * 32];
#pragma unroll 2
    for (int kk = 0; kk < ACTOR_HID / 2; ++kk) {
      const int k = 2 * kk + half;
      // prefetch next k-pair's A operand (the last iteration re-reads the current one; harmless)
      const int kn = (kk + 1 < ACTOR_HID / 2) ? k + 2 : k;
      const float4 n0 = w2[(kn * 2 + part) * 32];
      // layer 1 for this lane's k, both env tiles (B operand)
      const float4 wa = A.W1P[k * 3 + 0], wb = A.W1P[k * 3 + 1], wc = A.W1P[k * 3 + 2];
      const float w[9] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w, wc.x};
      float hA = wc.w, hB = wc.w;
      static_for<0, IN>([&](auto DI) {
        constexpr int d = DI;
        hA = fmaf(w[d], sA[d], hA);
        hB = fmaf(w[d], sB[d], hB);
      });
      hA = fmaxf(hA, 0.f);
      hB = fmaxf(hB, 0.f);
      const float av[4] = {a0.x, a0.y, a0.z, a0.w};
      static_for<0, 4>([&](auto NI) {
        constexpr int nt = NI;
        acc[nt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[nt], hA, acc[nt][0], 0, 0, 0);
        acc[nt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[nt], hB, acc[nt][1], 0, 0, 0);
      });
      a0 = n0;
    }
    // layer 2 bias + relu, layer 3 partial sums over the 64 neurons this lane holds per env tile in this pass
    static_for<0, 4>([&](auto NI) {
      constexpr int nt = NI;
      static_for<0, 16>([&](auto RI) {
        constexpr int r = RI;
        const int n = 32 * (4 * part + nt) + (r & 3) + 8 * (r >> 2) + 4 * half;   // accumulator row -> neuron
        const float4 c = A.B2W3[n];
        const float h0 = fmaxf(acc[nt][0][r] + c.x, 0.f);
        const float h1 = fmaxf(acc[nt][1][r] + c.x, 0.f);
        p[0][0] = fmaf(c.y, h0, p[0][0]); p[0][1] = fmaf(c.z, h0, p[0][1]); p[0][2] = fmaf(c.w, h0, p[0][2]);
        p[1][0] = fmaf(c.y, h1, p[1][0]); p[1][1] = fmaf(c.z, h1, p[1][1]); p[1][2] = fmaf(c.w, h1, p[1][2]);
      });
    });
  }
  static_for<0, 3>([&](auto OI) {
    constexpr int o = OI;
    const float t0 = p[0][o] + __shfl_xor(p[0][o], 32);
    const float t1 = p[1][o] + __shfl_xor(p[1][o], 32);
    const float z = (half ? t1 : t0) + A.b3[o];     // lane e holds env e: tile e>>5, column e&31
    out[o] = tanhf(z) * A.bound;                    // net_mlp.py:40
  });
}

}  // namespace armenv
