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
  const float *W1P;    // [128 kk][2 rows k = 2kk, 2kk+1][12]: w0..w8 (zero beyond in_dim), 0, 0, b1 -- read through the
                       // scalar cache (wave-uniform address), never through the vector memory path
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
    // Software pipeline, two k-pairs deep: while the eight MFMAs of k-pair kk occupy the matrix pipe (8 x 64 cycles),
    // the VALU computes layer 1 for kk+1 from W1 rows fetched during kk-1, and the loads for kk+2 are in flight.
    // W1 rows: both halves of the wave need a different row (k = 2kk + lane/32), i.e. two wave-uniform rows per
    // k-pair.  Fetching them per lane through the vector path (64 lanes x 16 B of identical addresses, three times
    // per k-pair) halved the MFMA rate (57 vs 110 TF); through the scalar cache they cost nothing visible.
    typedef const float __attribute__((address_space(4))) *scalar_ptr;
    const scalar_ptr w1s = (scalar_ptr)A.W1P;
    struct Rows { float r0[12], r1[12]; };
    auto load_rows = [&](int kk, Rows &R) {
      const int kc = kk < ACTOR_HID / 2 ? kk : ACTOR_HID / 2 - 1;   // the tail prefetches re-read the last pair
      static_for<0, 12>([&](auto JI) { constexpr int j = JI; R.r0[j] = w1s[kc * 24 + j]; R.r1[j] = w1s[kc * 24 + 12 + j]; });
    };
    auto load_a = [&](int kk) {
      const int kc = kk < ACTOR_HID / 2 ? kk : ACTOR_HID / 2 - 1;
      return w2[((2 * kc + half) * 2 + part) * 32];
    };
    auto layer1 = [&](const Rows &R, float &hA, float &hB) {
      float a0 = R.r0[11], a1 = R.r1[11], b0 = R.r0[11], b1 = R.r1[11];
      static_for<0, IN>([&](auto DI) {
        constexpr int d = DI;
        a0 = fmaf(R.r0[d], sA[d], a0); a1 = fmaf(R.r1[d], sA[d], a1);
        b0 = fmaf(R.r0[d], sB[d], b0); b1 = fmaf(R.r1[d], sB[d], b1);
      });
      hA = fmaxf(half ? a1 : a0, 0.f);
      hB = fmaxf(half ? b1 : b0, 0.f);
    };
    // Software pipeline: while the eight MFMAs of k-pair kk occupy the matrix pipe (8 x 64 cycles), the VALU computes
    // layer 1 for kk+1 and the loads for kk+2 (A operand) / kk+1 (W1 rows) are in flight.
    Rows R;
    float hA, hB, hA_n, hB_n;
    load_rows(0, R);
    layer1(R, hA, hB);
    float4 a_cur = load_a(0), a_nxt = load_a(1);
    load_rows(1, R);
#pragma unroll 1
    for (int kk = 0; kk < ACTOR_HID / 2; ++kk) {
      const float4 a_nn = load_a(kk + 2);
      // keep the loads above the MFMAs: hipcc otherwise sinks them to their first use and exposes an L2 round trip
      __builtin_amdgcn_sched_barrier(0);
      const float av[4] = {a_cur.x, a_cur.y, a_cur.z, a_cur.w};
      static_for<0, 4>([&](auto NI) {
        constexpr int nt = NI;
        acc[nt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[nt], hA, acc[nt][0], 0, 0, 0);
        acc[nt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[nt], hB, acc[nt][1], 0, 0, 0);
      });
      layer1(R, hA_n, hB_n);                                // layer 1 for kk+1
      load_rows(kk + 2, R);
      __builtin_amdgcn_sched_barrier(0);
      hA = hA_n; hB = hB_n;
      a_cur = a_nxt; a_nxt = a_nn;
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

// ------------------------------------------------------------------------------------------------------------------
// Fast variant: layer 2 on the f16 MFMA (v_mfma_f32_32x32x16_f16, 16x the f32 MFMA rate) with both operands split
// x = hi + lo, hi = f16(x), lo = f16(x - hi), and three passes  hi*hi + hi*lo + lo*hi  accumulated in f32.  f16 products
// are exact in f32, so only the lo*lo term (2^-22 relative) and f32 accumulation order separate the result from the
// exact-f32 path above: within the 1e-5 actor tolerance of the reference's vectors (tests).  Measured: 45 us for 65 536
// envs standalone (196 TF useful = 590 TF of f16 MFMA; a streamed-operand f16 MFMA probe reaches ~1000 TF, fixed
// operands 1570 TF), 52 us per fused env step against 113 us for the exact-f32 actor.
// Same transposed decomposition; per k-step of 16 a lane supplies 8 consecutive k of its env column, so layer 1 is
// evaluated 8 rows at a time with W1 (12 KB) staged ONCE per workgroup in LDS (the only LDS use and the only barrier
// of the kernel, before the step loop).
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
#ifndef ACTOR_F16_NT
#define ACTOR_F16_NT 4
#endif

struct ActorParamsH {
  const half8 *W2H;    // [16 ks][8 tile][64 lane]: W2[32 tile + (lane&31)][16 ks + 8 (lane>>5) + j], j = 0..7 -- hi
  const half8 *W2L;    //   same, lo
};

constexpr int ACTOR_W1_LDS_FLOATS = ACTOR_HID * 12;

// copy W1P (global, [256][12] f32) into LDS; every thread of the block must call this once, then __syncthreads()
AE_DEV void actor_stage_w1(const float *W1P, float4 *lds) {
  for (int i = threadIdx.x; i < ACTOR_W1_LDS_FLOATS / 4; i += blockDim.x) lds[i] = reinterpret_cast<const float4 *>(W1P)[i];
}

template <int IN>
AE_DEV void actor_forward_wave_f16x3(const ActorParams &A, const ActorParamsH &H, const float4 *w1_lds, const float (&s)[IN],
                                     float (&out)[3]) {
  const int lane = threadIdx.x & 63;
  const int half = lane >> 5;
  float sA[IN], sB[IN];
  static_for<0, IN>([&](auto DI) {
    constexpr int d = DI;
    const float other = __shfl_xor(s[d], 32);
    sA[d] = half ? other : s[d];
    sB[d] = half ? s[d] : other;
  });
  float p[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
  constexpr int NT = ACTOR_F16_NT;            // neuron tiles per pass: 8 = single pass (256 accumulators), 4 = two passes
  constexpr int NPASS = 8 / NT;
#pragma unroll 1
  for (int part = 0; part < NPASS; ++part) {
    f32x16 acc[NT][2];
    static_for<0, NT>([&](auto NI) {
      constexpr int nt = NI;
      static_for<0, 16>([&](auto RI) { constexpr int r = RI; acc[nt][0][r] = 0.f; acc[nt][1][r] = 0.f; });
    });
    half8 ah[NT], al[NT], nh[NT], nl[NT];
    auto load_a = [&](int ks, half8 (&h)[NT], half8 (&l)[NT]) {
      const int kc = ks < 16 ? ks : 15;
      // table order is [ks][tile 0..7][lane]; pass `part` covers tiles NT*part .. NT*part + NT-1
      const int base = (kc * 8 + NT * part) * 64 + lane;
      static_for<0, NT>([&](auto NI) { constexpr int nt = NI; h[nt] = H.W2H[base + nt * 64]; l[nt] = H.W2L[base + nt * 64]; });
    };
    // B operands: h1 of the lane's two env columns for k = 16 ks + 8 half + j (j = 0..7), split into f16 hi / lo.
    // Row j of the step is built in three pieces so that the pieces can be placed between MFMAs (below).
    auto row_load = [&](int ks, int j, float4 &ra, float4 &rb, float4 &rc) {
      const int kc = ks < 16 ? ks : 15;
      const float4 *row = w1_lds + (16 * kc + 8 * half + j) * 3;
      ra = row[0]; rb = row[1]; rc = row[2];
    };
    auto row_dot = [&](const float4 &ra, const float4 &rb, const float4 &rc, const float (&sv)[IN]) {
      const float w[9] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w, rc.x};
      float h = rc.w;
      static_for<0, IN>([&](auto DI) { constexpr int d = DI; h = fmaf(w[d], sv[d], h); });
      return fmaxf(h, 0.f);
    };
    half8 bh[2], bl[2], bh_n[2], bl_n[2];
    load_a(0, ah, al);
    static_for<0, 8>([&](auto JI) {
      constexpr int j = JI;
      float4 ra, rb, rc;
      row_load(0, j, ra, rb, rc);
      const float hA = row_dot(ra, rb, rc, sA), hB = row_dot(ra, rb, rc, sB);
      const _Float16 ha = (_Float16)hA, hb = (_Float16)hB;
      bh[0][j] = ha; bl[0][j] = (_Float16)(hA - (float)ha);
      bh[1][j] = hb; bl[1][j] = (_Float16)(hB - (float)hb);
    });
#pragma unroll 1
    for (int ks = 0; ks < 16; ++ks) {
      load_a(ks + 1, nh, nl);                               // A operands of the next k-step, in flight under the MFMAs
      // Software pipeline, pinned by sched_barrier(0): the 6*NT MFMAs of k-step ks are issued one at a time with a
      // slice of the layer-1 VALU / LDS work for k-step ks+1 behind each, so the VALU runs while the matrix pipe is
      // busy (an in-order wave cannot overlap the two phases otherwise; left alone hipcc emits them back to back).
      float4 ra, rb, rc, na, nb, nc;
      row_load(ks + 1, 0, ra, rb, rc);
      static_for<0, 8>([&](auto JI) {
        constexpr int j = JI;
        constexpr int m0 = 3 * j;                            // MFMA slots of this row: m0, m0+1, m0+2 (of 6*NT)
        auto mfma = [&](auto MI) {
          constexpr int m = MI;
          if constexpr (m < 6 * NT) {
            // order: pass (hi*hi, hi*lo, lo*hi) outermost, so consecutive MFMAs never share an accumulator -- three
            // back-to-back MFMAs on one accumulator each wait out the full dependent latency (measured 103 instead
            // of 32 cycles per MFMA)
            constexpr int k3 = m / (2 * NT), nt = (m % (2 * NT)) / 2, t = m % 2;
            acc[nt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k3 == 2 ? al[nt] : ah[nt], k3 == 1 ? bl[t] : bh[t], acc[nt][t], 0, 0, 0);
          }
        };
        __builtin_amdgcn_sched_barrier(0);
        mfma(std::integral_constant<int, m0>{});
        __builtin_amdgcn_sched_barrier(0);
        const float hA = row_dot(ra, rb, rc, sA);
        const _Float16 ha = (_Float16)hA;
        bh_n[0][j] = ha; bl_n[0][j] = (_Float16)(hA - (float)ha);
        __builtin_amdgcn_sched_barrier(0);
        mfma(std::integral_constant<int, m0 + 1>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (j < 7) row_load(ks + 1, j + 1, na, nb, nc);
        const float hB = row_dot(ra, rb, rc, sB);
        const _Float16 hb = (_Float16)hB;
        bh_n[1][j] = hb; bl_n[1][j] = (_Float16)(hB - (float)hb);
        __builtin_amdgcn_sched_barrier(0);
        mfma(std::integral_constant<int, m0 + 2>{});
        ra = na; rb = nb; rc = nc;
      });
      __builtin_amdgcn_sched_barrier(0);
      static_for<24, 6 * NT>([&](auto MI) {                  // NT = 8: the remaining MFMAs (none for NT = 4)
        constexpr int m = MI;
        constexpr int k3 = m / (2 * NT), nt = (m % (2 * NT)) / 2, t = m % 2;
        acc[nt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k3 == 2 ? al[nt] : ah[nt], k3 == 1 ? bl[t] : bh[t], acc[nt][t], 0, 0, 0);
      });
      static_for<0, NT>([&](auto NI) { constexpr int nt = NI; ah[nt] = nh[nt]; al[nt] = nl[nt]; });
      bh[0] = bh_n[0]; bh[1] = bh_n[1]; bl[0] = bl_n[0]; bl[1] = bl_n[1];
    }
    static_for<0, NT>([&](auto NI) {
      constexpr int nt = NI;
      static_for<0, 16>([&](auto RI) {
        constexpr int r = RI;
        const int n = 32 * (NT * part + nt) + (r & 3) + 8 * (r >> 2) + 4 * half;
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
    const float z = (half ? t1 : t0) + A.b3[o];
    out[o] = tanhf(z) * A.bound;
  });
}

}  // namespace armenv
