// armenv_actor.h -- the TD3 actor (PolicyNet, /root/reference/algo/TD3/net_mlp.py:29-40) evaluated for the envs a
// wavefront owns:
//     a = action_bound * tanh(W3 relu(W2 relu(W1 s + b1) + b2) + b3),   6|9 -> 256 -> 256 -> 3
//
// Layer 2 is the only dense contraction on the env path (131 kflop of the 136 kflop per env) and runs on the matrix
// cores, TRANSPOSED: H2^T[neuron][env] = W2 * H1^T, so that in the accumulator layout a lane holds 128 neurons of ONE
// env -- layer 3's reduction over neurons is in-register plus one lane<->lane+32 exchange and no activation ever goes
// through LDS.  Layer 1 also runs on the (f32) MFMA, as W1aug * [obs, 1]: its accumulator layout IS layer 2's
// B-operand layout up to a permutation of k, which the A-operand indexing / packing absorbs, so only a relu (and, for
// the f16 variant, the hi / lo split) separates the two layers.
// Two variants:
//   actor_forward_wave      exact f32: v_mfma_f32_32x32x2_f32 for both layers; the A operand of layer 2 streams from an
//                           L2-resident packed copy of W2 through a ring of four register sets; per wave, no barriers.
//   actor_forward_wg_f16x3  f32 emulated by three f16 MFMA passes (hi*hi + hi*lo + lo*hi); a workgroup phase with the W2
//                           fragments shared through an LDS ring filled by direct-to-LDS loads.
// One wave per SIMD (the env step needs the whole register file): an MFMA holds the matrix pipe for 8 / 16 issue slots and
// the wave issues in order, so everything that is not an MFMA is placed between two MFMAs, never behind a block of them.
#pragma once
#include "armenv_math.h"

namespace armenv {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int ACTOR_HID = 256;
constexpr int ACTOR_W1P_COLS = 16;   // a packed layer-1 row: up to 15 input weights (12 = the critics of the cube tasks: obs 9 + action 3), then the bias

struct ActorParams {
  const float *W1P;    // [256 k][ACTOR_W1P_COLS = 16]: w0..w14 (zero beyond in_dim), b1 -- source of the LDS tables (actor_stage_w1)
  const float4 *W2P;   // [256 k][2 part][32 lane]: (W2[32*(4 part + c) + lane][k], c = 0..3)
  const float4 *B2W3;  // [256]: (b2[n], W3[0][n], W3[1][n], W3[2][n])
  float b3[3];
  float bound;
  int32_t in_dim;      // 6 (reach obs) or 9 (push obs; DATD3: every net staged with 9 inputs, see datd3_forward_wg)
  int32_t raw;         // 0: out = bound * tanh(z + b3) (PolicyNet); 1: out = z + b3 (QValueNet, net_mlp.py:43-58: row 0 of the table is fc3)
  const float4 *lds_image;   // nullable: the LDS tables of actor_stage_w1 + actor_stage_w1h for this net as ONE image in global memory
                             // (ACTOR_W1_LDS_FLOATS_H floats, made once when the net is installed): staging is then a copy
};

// ------------------------------------------------------------------------------------------------------------------
// Fast variant: layer 2 on the f16 MFMA (v_mfma_f32_32x32x16_f16, 16x the f32 MFMA rate) with both operands split
// x = hi + lo, hi = f16(x), lo = f16(x - hi), and three passes  hi*hi + hi*lo + lo*hi  accumulated in f32.  f16 products
// are exact in f32, so only the lo*lo term (2^-22 relative) and f32 accumulation order separate the result from the
// exact-f32 path above: within the 1e-5 actor tolerance of the reference's vectors (tests).
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

struct ActorParamsH {
  const half8 *W2H;    // [16 ks][8 tile][64 lane]: W2[32 tile + (lane&31)][16 ks + 8 (j>>2) + 4 (lane>>5) + (j&3)], j = 0..7 -- hi
                       // (the k order in which layer 1's MFMA accumulators hand a lane its eight values)
  const half8 *W2L;    //   same, lo
};

// LDS tables of the workgroup actor (actor_forward_wg_f16x3): layer 1's A operands for the f32 MFMA, then B2W3, then b2.
//   W1A [8 row tiles][ACTOR_NK k-pairs][64 lanes] f32: lane l of k-pair m holds W1aug[32 R + (l & 31)][2 m + (l >> 5)],
//   W1aug = [W1 | b1 | 0...] (the bias rides on a constant-1 input), ACTOR_NK = 5 covers IN = 6 and 9.
constexpr int ACTOR_NK = 5;
constexpr int ACTOR_W1A_FLOATS = 8 * ACTOR_NK * 64;
constexpr int ACTOR_W1_LDS_FLOATS = ACTOR_W1A_FLOATS + ACTOR_HID * 4 + ACTOR_HID;

// fills the tables from W1P (global, [256][16] f32: w0..w14, b1) and B2W3 ([256] float4); every thread of the
// block must call this once, then __syncthreads()
AE_DEV void actor_stage_w1(const float *W1P, float4 *lds, const float4 *B2W3, int in_dim, int nthreads = 0) {
  if (nthreads == 0) nthreads = blockDim.x;      // (a ragged workgroup whose dead waves have exited passes its live thread count)
  float *w1a = reinterpret_cast<float *>(lds);
  for (int i = threadIdx.x; i < ACTOR_W1A_FLOATS; i += nthreads) {
    const int l = i & 63, m = (i >> 6) % ACTOR_NK, R = (i >> 6) / ACTOR_NK;
    const int row = 32 * R + (l & 31), ka = 2 * m + (l >> 5);
    w1a[i] = ka < in_dim ? W1P[row * ACTOR_W1P_COLS + ka] : (ka == in_dim ? W1P[row * ACTOR_W1P_COLS + ACTOR_W1P_COLS - 1] : 0.f);
  }
  for (int i = threadIdx.x; i < ACTOR_HID; i += nthreads) {
    const float4 c = B2W3[i];
    lds[ACTOR_W1A_FLOATS / 4 + i] = c;
    reinterpret_cast<float *>(lds + ACTOR_W1A_FLOATS / 4 + ACTOR_HID)[i] = c.x;
  }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Layer 3 (3 x 256, /root/reference/algo/TD3/net_mlp.py:35,40) on the matrix pipe.  After layer 2 a lane holds, for ITS env column,
// the 16 neurons {32 nt + 8 (r / 4) + 4 half + (r % 4)} of tile nt in its accumulator registers -- exactly the B operand of
// v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4x1 outer products: lane l supplies B[block l / 4][column l % 4] and
// A[block l / 4][row l % 4]; result register i of lane l = sum over the chained instructions of A[block][row i] * B[block][column l % 4];
// tests/tools/exp/mfma4x4_probe.hip).  With A[row i] = W3[i][neuron] (row 3 unused) every lane ends with the three output sums
// of its own env over the 128 neurons it holds; lane ^ 32 has the other 128.  Per neuron: one 4-byte LDS read (the A value: the
// four lanes of a block are in the same wave half, so they want the same neuron), one integer max (relu) and one 8-cycle MFMA
// issue, instead of a 16-byte LDS read, the relu and three dependent f32 FMAs: the epilogues of the two passes were 3.3 us of
// the f16x3 actor's 32 us per fused env step (ablation).
// w3p: the lane's base into the (b2, W3[0], W3[1], W3[2]) table = table + 16 half + 1 + (lane & 3) floats; row 3 of the A
// operand reads the next neuron's b2 -- finite, and result register 3 is never looked at.
// Two independent accumulator chains per call (a 4x4x1 result is ready two passes after issue): d0 / d1 are the two halves of
// one env column's registers (f16x3 actor), or the two env columns of the exact-f32 actor's accumulator pair.
// Layer 3 over the NT tiles of one env column, with the A values (W3) of tile nt + 1 requested from LDS before tile nt's MFMAs are
// issued: tile by tile the sixteen reads sat between the previous tile's MFMAs and this tile's, and every tile began with an
// exposed LDS round trip (round 4, by the instrumented build's stamps: 4 170 -> 3 490 -> see DESIGN.md cycles per pass).
template <int NT, class Relu>
AE_DEV void layer3_column(const float *w3p, const f32x16 (&acc)[NT], f32x4 &d0, f32x4 &d1, Relu relu) {
  float a[2][16], h[16];
  static_for<0, 16>([&](auto RI) { constexpr int r = RI; a[0][r] = w3p[4 * ((r & 3) + 8 * (r >> 2))]; });
  static_for<0, NT>([&](auto NI) {
    constexpr int nt = NI;
    if constexpr (nt + 1 < NT)
      static_for<0, 16>([&](auto RI) { constexpr int r = RI; a[(nt + 1) & 1][r] = w3p[4 * (32 * (nt + 1) + (r & 3) + 8 * (r >> 2))]; });
    static_for<0, 16>([&](auto RI) { constexpr int r = RI; h[r] = relu(acc[nt][r]); });
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, 8>([&](auto RI) {
      constexpr int r = RI;
      d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[nt & 1][r], h[r], d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[nt & 1][r + 8], h[r + 8], d1, 0, 0, 0);
    });
    __builtin_amdgcn_sched_barrier(0);     // (without it hipcc moves the next tile's LDS reads back behind these MFMAs: 2 760 -> 3 500 cycles)
  });
}
// Exact-f32 actor for the 64 envs of one wave.  s: this lane's env observation (IN floats); all 64 lanes must be active;
// w1_lds: the tables of actor_stage_w1.
//   Layer 1 runs on the f32 MFMA as W1aug . [obs, 1] per 32-neuron row tile and env column tile.  In the accumulator
//   layout lane (half h) holds, in register i = 4 b + c, neuron 32 R + 8 b + 4 h + c of its env column -- which is exactly
//   the B operand of layer 2's k-pair {32 R + 8 b + c, 32 R + 8 b + 4 + c}: relu(register i) feeds the MFMAs directly and
//   the A operand is fetched for that k (no repacking: W2P is indexed by k).
//   Layer 2: per k-pair 8 MFMAs (4 neuron tiles x 2 env tiles) into 128 accumulators, two passes over the neuron tiles.
//   The A operand (one 16-byte load per lane per k-pair) is prefetched three k-pairs ahead into a ring of four register
//   sets inside a 16-k-pair unrolled body, where hipcc counts the loads in flight exactly; in a rolled per-k-pair loop it
//   waits with vmcnt(0) at the loop head.  (Loads issued from inline asm with hand-counted waits were tried and dropped:
//   nothing stops the compiler from copying or spilling a destination register between the load and the wait, and under
//   the register pressure of the push kernels it did.)
template <int IN>
AE_DEV void actor_forward_wave(const ActorParams &A, const float4 *w1_lds, const float (&s)[IN], float (&out)[3]) {
  static_assert(IN + 1 <= 2 * ACTOR_NK, "augmented input does not fit ACTOR_NK k-pairs");
  const int lane = threadIdx.x & 63;
  const int half = lane >> 5;
  const int l32 = lane & 31;
  const float *w1a = reinterpret_cast<const float *>(w1_lds) + lane;
  // layer-1 B operands: k-pair m of [obs, 1, 0..] of the env in tile 0 (env l32) and tile 1 (env l32 + 32)
  float bvA[ACTOR_NK], bvB[ACTOR_NK];
  {
    float svA[2 * ACTOR_NK], svB[2 * ACTOR_NK];
    static_for<0, 2 * ACTOR_NK>([&](auto DI) {
      constexpr int d = DI;
      if constexpr (d < IN) {
        const float other = __shfl_xor(s[d], 32);
        svA[d] = half ? other : s[d];
        svB[d] = half ? s[d] : other;
      } else {
        svA[d] = svB[d] = d == IN ? 1.f : 0.f;
      }
    });
    static_for<0, ACTOR_NK>([&](auto MI) {
      constexpr int m = MI;
      bvA[m] = half ? svA[2 * m + 1] : svA[2 * m];
      bvB[m] = half ? svB[2 * m + 1] : svB[2 * m];
    });
  }
  auto layer1 = [&](int R, f32x16 &a1A, f32x16 &a1B) {
    static_for<0, 16>([&](auto RI) { constexpr int r = RI; a1A[r] = 0.f; a1B[r] = 0.f; });
    const float *w = w1a + (R < 8 ? R : 7) * (ACTOR_NK * 64);
    static_for<0, ACTOR_NK>([&](auto MI) {
      constexpr int m = MI;
      if constexpr (2 * m <= IN) {
        const float wv = w[m * 64];
        a1A = __builtin_amdgcn_mfma_f32_32x32x2f32(wv, bvA[m], a1A, 0, 0, 0);
        a1B = __builtin_amdgcn_mfma_f32_32x32x2f32(wv, bvB[m], a1B, 0, 0, 0);
      }
    });
  };
  // relu as ONE compiler-visible instruction: max(bits, 0) on the float's bit pattern as a signed integer (negative
  // floats, -0 included, are negative integers; NaNs with a clear sign bit pass through).  fmaxf / v_med3 add a
  // canonicalising v_max of the accumulator read in front, and an inline-asm v_max reading an MFMA result is outside
  // hipcc's hazard bookkeeping -- it is not padded against the MFMA -> VALU read hazard, so whether it saw the finished
  // accumulator depended on what the scheduler happened to put in between (wrong actions in the 9-input push rollout
  // after an unrelated change elsewhere in the kernel).
  auto relu = [](float x) { const int b = __float_as_int(x); return __int_as_float(b > 0 ? b : 0); };
  const float4 *w2 = A.W2P + l32;
  const float4 *b2w3 = w1_lds + ACTOR_W1A_FLOATS / 4;     // staged beside the layer-1 table
  const float4 *b2tab = b2w3 + ACTOR_HID;                 // b2 alone, four consecutive neurons per float4
  const float *w3p = reinterpret_cast<const float *>(b2w3) + 16 * half + 1 + (lane & 3);   // layer 3's A values
  f32x4 dA = {0.f, 0.f, 0.f, 0.f}, dB = {0.f, 0.f, 0.f, 0.f};   // layer-3 sums of the env in tile 0 / tile 1
  constexpr int AD = 4;
#pragma unroll 1
  for (int part = 0; part < 2; ++part) {
    f32x16 acc[4][2];   // start from the layer-2 bias: register r <-> neuron 32 (4 part + nt) + 8 (r / 4) + 4 half + (r % 4)
    static_for<0, 4>([&](auto NI) {
      constexpr int nt = NI;
      static_for<0, 4>([&](auto QI) {
        constexpr int q = QI;
        const float4 b = b2tab[8 * (4 * part + nt) + 2 * q + half];
        acc[nt][0][4 * q] = b.x; acc[nt][0][4 * q + 1] = b.y; acc[nt][0][4 * q + 2] = b.z; acc[nt][0][4 * q + 3] = b.w;
        acc[nt][1][4 * q] = b.x; acc[nt][1][4 * q + 1] = b.y; acc[nt][1][4 * q + 2] = b.z; acc[nt][1][4 * q + 3] = b.w;
      });
    });
    float4 aring[AD];
    auto load_a = [&](int kk) {   // k-pair kk = 16 R + i, i = 4 b + c  ->  k = 32 R + 8 b + 4 half + c
      const int kc = kk < ACTOR_HID / 2 ? kk : ACTOR_HID / 2 - 1;   // the tail's prefetches re-read the last pair
      const int k = 32 * (kc >> 4) + 8 * ((kc >> 2) & 3) + 4 * half + (kc & 3);
      return w2[(k * 2 + part) * 32];
    };
    static_for<0, AD - 1>([&](auto I) { constexpr int i = I; aring[i] = load_a(i); });
    f32x16 a1A, a1B, nA, nB;
    layer1(0, a1A, a1B);
    // the row-tile loop unrolled (round 4: a taken back edge costs a lone wave ~47 ns): 76.0 -> 74.5 us per
    // fused env step at 65 536 envs; unrolled by 4 it gained 0.5 % only
#ifndef ARMENV_ACTOR_F32_UNROLL
#define ARMENV_ACTOR_F32_UNROLL 8
#endif
#pragma unroll ARMENV_ACTOR_F32_UNROLL
    for (int R = 0; R < 8; ++R) {
      static_for<0, 16>([&](auto I) {
        constexpr int i = I;
        const int kk = 16 * R + i;
        // refill the register set k-pair kk - 1 used, three k-pairs ahead; inside this unrolled body hipcc counts the
        // loads in flight exactly (vmcnt(2) before the use below), only the R loop's back edge drains the queue
        aring[(i + AD - 1) % AD] = load_a(kk + AD - 1);
        const float hA = relu(a1A[i]), hB = relu(a1B[i]);
        const float av[4] = {aring[i % AD].x, aring[i % AD].y, aring[i % AD].z, aring[i % AD].w};
        static_for<0, 4>([&](auto NI) {
          constexpr int nt = NI;
          acc[nt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[nt], hA, acc[nt][0], 0, 0, 0);
          acc[nt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[nt], hB, acc[nt][1], 0, 0, 0);
        });
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (i == 8) layer1(R + 1, nA, nB);   // next row tile's layer 1 goes into the matrix pipe mid-way
      });
      a1A = nA; a1B = nB;
    }
    // relu of layer 2 (the bias is already in the accumulators) and layer 3 over the 64 neurons this lane holds per env tile in
    // this pass, on the matrix pipe (layer3_column)
    // as in layer3_column: relu'd values in registers of their own ahead of the MFMAs, the next tile's W3 values requested from
    // LDS before this tile's MFMAs (hipcc's own order funnels every value through one temporary register)
    const float *w3part = w3p + 4 * 128 * part;
    {
      float a[2][16], hA[16], hB[16];
      static_for<0, 16>([&](auto RI) { constexpr int r = RI; a[0][r] = w3part[4 * ((r & 3) + 8 * (r >> 2))]; });
      static_for<0, 4>([&](auto NI) {
        constexpr int nt = NI;
        if constexpr (nt + 1 < 4)
          static_for<0, 16>([&](auto RI) { constexpr int r = RI; a[(nt + 1) & 1][r] = w3part[4 * (32 * (nt + 1) + (r & 3) + 8 * (r >> 2))]; });
        static_for<0, 16>([&](auto RI) { constexpr int r = RI; hA[r] = relu(acc[nt][0][r]); hB[r] = relu(acc[nt][1][r]); });
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 16>([&](auto RI) {
          constexpr int r = RI;
          dA = __builtin_amdgcn_mfma_f32_4x4x1f32(a[nt & 1][r], hA[r], dA, 0, 0, 0);
          dB = __builtin_amdgcn_mfma_f32_4x4x1f32(a[nt & 1][r], hB[r], dB, 0, 0, 0);
        });
        __builtin_amdgcn_sched_barrier(0);
      });
    }
  }
  const f32x4 sA = dA, sB = dB;
  static_for<0, 3>([&](auto OI) {
    constexpr int o = OI;
    const float t0 = sA[o] + __shfl_xor(sA[o], 32);
    const float t1 = sB[o] + __shfl_xor(sB[o], 32);
    const float z = (half ? t1 : t0) + A.b3[o];     // lane e holds env e: tile e>>5, column e&31
    out[o] = tanhf(z) * A.bound;                    // net_mlp.py:40
  });
}

// ------------------------------------------------------------------------------------------------------------------
// The f16x3 actor is a WORKGROUP phase (four waves, 256 envs), same transposed decomposition as above:
//   - one env column tile at a time with all eight neuron tiles (128 accumulators), so that nothing spills inside the
//     k-loop (scratch traffic there would break the vmcnt accounting of the ring);
//   - layer 1 on the f16 MFMA too (hi / lo split of W1aug and of [obs, 1], three passes); its accumulator layout is layer 2's
//     B-operand layout up to a permutation of k inside each k-step, which the packing of W2H / W2L absorbs: only relu + the
//     hi / lo split is VALU;
//   - the A operands (W2 hi / lo fragments, 16 KB per k-step) reach the four waves through a ring in LDS that the waves
//     fill cooperatively with direct-to-LDS loads (global_load_lds_dwordx4: 1 KB per wave instruction, scalar base + M0,
//     no staging registers, no VALU).  Per-wave streaming of the same fragments from L2 left the matrix pipe waiting on
//     memory: with one wave per SIMD nothing else hides a 1-2 us L2 round trip, and the registers to keep a dozen
//     k-steps in flight do not exist.
// Measured (65 536 envs, round 3): 30.4 us per fused env step; the 792 MFMAs of a wave-step take 10.9-12.1 us at the rate the chip
// sustains with every CU busy (13.7-15.2 ns each, tests/tools/exp/mfma_rate_probe.hip; 13.3 ns nominal), the LDS-DMA pieces of the
// refill ~6 us of issue, the env step itself 6.9 us.
// LDS holds k-steps 0..3 of W2 permanently (64 KB, filled once per launch) and streams k-steps 4..15 through a ring of
// four 16 KB slots (slot = k-step mod 4; 12 streamed k-steps per pass, so the mapping carries over from pass to pass
// and call to call).  Protocol of k-step ks (the tail of a call refills for the following one):
//   head   s_waitcnt vmcnt(0) (own share of the refill issued one k-step ago, which is k-step ks + 1's data)
//          -> s_barrier (everyone's share has landed)
//   body   the 24 MFMAs of k-step ks from the register set read one k-step ago, with everything else issued in their
//          shadow: slots 1..7 the four quarters of the refill two stream positions ahead (k-step ks + 2's slot was last
//          read during k-step ks - 3), slots 8..23 the LDS reads of k-step ks + 1 into the other register set and the
//          relu / split of layer 1.
// The loads are issued from inline asm so that hipcc's waitcnt insertion does not see them (it would drain the queue
// with vmcnt(0) before every LDS read that might alias them); the waits above are therefore explicit.
// Callers: actor_ring_init() once per kernel after computing `nw` (live waves of this workgroup), actor_ring_drain()
// before the kernel ends (an LDS DMA must not outlive the workgroup's LDS allocation).  The k-loop must stay free of
// compiler-generated VMEM (no spills): hipcc's own vmcnt arithmetic does not see the DMA loads.
constexpr int ACTOR_KRES = 4;          // k-steps 0..3 of W2 stay resident in LDS for the whole launch
constexpr int ACTOR_RING_SLOTS = 4;    // k-steps 4..15 stream through four slots; (16 - KRES) is a multiple of the slot count
constexpr int ACTOR_RING_UINT4 = (ACTOR_KRES + ACTOR_RING_SLOTS) * 16 * 64;   // 16 KB per k-step: 128 KB
// layer 1's f16 operand table, staged behind the f32 tables: [8 row tiles][hi | lo][64 lanes] half8 = 16 KB
constexpr int ACTOR_W1H_FLOATS = 8 * 2 * 64 * 4;
constexpr int ACTOR_W1_LDS_FLOATS_H = ACTOR_W1_LDS_FLOATS + ACTOR_W1H_FLOATS;
// W1aug = [W1 | b1 | 0...] split hi / lo in the A-operand order of v_mfma_f32_32x32x16_f16: lane l of row tile R holds
// W1aug[32 R + (l & 31)][8 (l >> 5) + j], j = 0..7.  Every thread of the block calls this once (before the __syncthreads()
// that follows actor_stage_w1); w1_lds must have ACTOR_W1_LDS_FLOATS_H floats.
AE_DEV void actor_stage_w1h(const float *W1P, float4 *w1_lds, int in_dim, int nthreads = 0) {
  if (nthreads == 0) nthreads = blockDim.x;
  _Float16 *t = reinterpret_cast<_Float16 *>(w1_lds + ACTOR_W1_LDS_FLOATS / 4);
  for (int i = threadIdx.x; i < 8 * 64 * 8; i += nthreads) {
    const int j = i & 7, l = (i >> 3) & 63, R = i >> 9;
    const int row = 32 * R + (l & 31), k = 8 * (l >> 5) + j;
    const float x = k < in_dim ? W1P[row * ACTOR_W1P_COLS + k] : (k == in_dim ? W1P[row * ACTOR_W1P_COLS + ACTOR_W1P_COLS - 1] : 0.f);
    const _Float16 hi = (_Float16)x;
    t[((R * 2 + 0) * 64 + l) * 8 + j] = hi;
    t[((R * 2 + 1) * 64 + l) * 8 + j] = (_Float16)(x - (float)hi);
  }
}
// LDS region (in k-step units) that holds k-step ks
AE_DEV int actor_region(int ks) { return ks < ACTOR_KRES ? ks : ACTOR_KRES + (ks & (ACTOR_RING_SLOTS - 1)); }
// How far ahead of its use a streamed k-step is requested: the refill issued during k-step ks fetches the streamed k-step
// ACTOR_FILL_AHEAD stream positions after ks.
//   2 (rounds 2-4): the data of k-step ks + 2, read from LDS during k-step ks + 1, has from the refill slots of k-step ks to the head
//      of k-step ks + 1 to land: ~20 MFMA slots, ~650 cycles -- about what an LDS DMA from L2 takes with every CU streaming, so the
//      head's vmcnt(0) could catch its tail;
//   3: one k-step more (its ring slot was last read during k-step ks - 2 and every wave has passed the barrier of k-step ks - 1
//      since); the head then waits for the refill issued TWO k-steps ago only -- vmcnt(4): the four DMA instructions of the last
//      k-step stay in flight (full workgroups; the ragged copy issues a varying number and keeps vmcnt(0)).
#ifndef ARMENV_ACTOR_FILL_AHEAD
#define ARMENV_ACTOR_FILL_AHEAD 2
#endif
constexpr int ACTOR_FILL_AHEAD = ARMENV_ACTOR_FILL_AHEAD;
static_assert(ACTOR_FILL_AHEAD == 2 || ACTOR_FILL_AHEAD == 3, "ring protocol: 2 or 3 stream positions ahead");
// the streamed k-step ACTOR_FILL_AHEAD stream positions after ks (ks >= KRES); ahead 2: 4 -> 6, ..., 13 -> 15, 14 -> 4, 15 -> 5
AE_DEV int actor_next2(int ks) { const int t = ks - ACTOR_KRES + ACTOR_FILL_AHEAD; return ACTOR_KRES + (t >= 16 - ACTOR_KRES ? t - (16 - ACTOR_KRES) : t); }

// one 1 KB direct-to-LDS copy: lane l moves 16 bytes from src_base + voff (voff = 16 l) to LDS byte lds_dst + 16 l.
// src_base and lds_dst are wave-uniform (SGPRs): a fill costs scalar adds only, no VALU.
AE_DEV void glds16(const void *src_base, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(src_base), "s"(lds_dst)
               : "memory");
}

// A wave-uniform 64-bit value as an OPAQUE pair of scalar registers.  A pointer read straight from the kernel-argument
// segment is rematerialisable: under scalar-register pressure the compiler re-reads it where it is used, and inside the
// k-loop of actor_forward_wg_f16x3 it did so with a VECTOR load from the argument segment followed by s_waitcnt vmcnt(0)
// -- a compiler-made VMEM wait that drains the ring's DMA queue (found by tests/test_isa_guard.py).  An opaque value can
// only be kept in SGPRs or parked in a VGPR lane (v_readlane: no memory, no wait).
AE_DEV uint64_t scalar_opaque(uint64_t v) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  asm volatile("" : "+s"(lo), "+s"(hi));
  return (uint64_t)lo | ((uint64_t)hi << 32);
}

// fragment f = 2 tile + (0 hi | 1 lo) of k-step ks -> ring slot ks mod R; w2h / w2l: byte addresses of the packed W2 hi / lo
AE_DEV void actor_ring_fill_one(uint64_t w2h, uint64_t w2l, uint4 *ring, int ks, int f) {
  const unsigned voff = (threadIdx.x & 63u) * 16u;
  const unsigned base = (unsigned)(uintptr_t)ring + (unsigned)(actor_region(ks) * 16 * 64 * 16);
  const uint64_t a = ((f & 1) ? w2l : w2h) + (uint64_t)(unsigned)((ks * 8 + (f >> 1)) * 64) * sizeof(half8);
  // wave-uniform by construction; readfirstlane tells the compiler so (the asm wants SGPR operands)
  const uint64_t au = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a) |
                      ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) << 32);
  glds16(reinterpret_cast<const void *>(au), voff, (unsigned)__builtin_amdgcn_readfirstlane((int)(base + (unsigned)(f * 1024))));
}
AE_DEV void actor_ring_fill_one(const ActorParamsH &H, uint4 *ring, int ks, int f) {
  actor_ring_fill_one((uint64_t)(uintptr_t)H.W2H, (uint64_t)(uintptr_t)H.W2L, ring, ks, f);
}
// the whole k-step: wave w takes f = w, w + nw, ... (four each when all four waves of the workgroup are live)
AE_DEV void actor_ring_fill(const ActorParamsH &H, uint4 *ring, int ks, int nw) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (nw == 4) {
    static_for<0, 4>([&](auto FI) { constexpr int fi = FI; actor_ring_fill_one(H, ring, ks, wave + 4 * fi); });
  } else {
    for (int f = wave; f < 16; f += nw) actor_ring_fill_one(H, ring, ks, f);
  }
}
AE_DEV void actor_ring_init(const ActorParamsH &H, uint4 *ring, int nw) {   // the resident k-steps and the first ACTOR_FILL_AHEAD streamed
  static_for<0, ACTOR_KRES + ACTOR_FILL_AHEAD>([&](auto KI) { constexpr int k = KI; actor_ring_fill(H, ring, k, nw); });
}
AE_DEV void actor_ring_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// The LDS tables of one net from their pre-built image: 1 KB pieces by direct-to-LDS loads, piece f taken by live wave f mod nw
// (31 pieces: eight instructions per wave and no VALU, against ~4 us of element-wise address arithmetic and f16 splits when the
// tables are computed in place -- DATD3 switches nets four times per env step).  The caller drains and meets afterwards.
AE_DEV void actor_stage_image(const float4 *image, float4 *w1_lds, int nw) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned voff = (threadIdx.x & 63u) * 16u;
  const uint64_t src = scalar_opaque((uint64_t)(uintptr_t)image);
  const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)w1_lds);
  constexpr int pieces = ACTOR_W1_LDS_FLOATS_H * 4 / 1024;
  static_assert(ACTOR_W1_LDS_FLOATS_H * 4 % 1024 == 0, "the table image is a whole number of 1 KB pieces");
  for (int f = wave; f < pieces; f += nw) glds16(reinterpret_cast<const void *>(src + (uint64_t)f * 1024u), voff, dst + (unsigned)f * 1024u);
}

// Instrumented build only (make timeline): shader-clock time of workgroup 0's four waves by phase of the f16x3 pass, accumulated in
// scalar registers and written once per pass to the buffer tests/tools/exp/run_actor_timeline.py installs (row = wave, 64 slots each:
// 8 t + class; classes: 0 prologue, 1 bodies of the resident k-steps, 2 / 3 of the streamed even / odd k-steps, 4 layer 3 + epilogue,
// 5 the sixteen heads' s_waitcnt vmcnt, 6 their s_barrier).
#ifdef ARMENV_TIMELINE
static __device__ unsigned long long *g_actor_tl;
// accumulate the shader-clock time since the previous stamp into class k (scalar registers only: a store inside the k-loops would be
// waited for by the next k-step's vmcnt(0)); ATL_FLUSH writes the classes once per pass
#define ATL(k)                                                                \
  do {                                                                        \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();               \
    atl_acc[k] += t_ - atl_prev;                                              \
    atl_prev = t_;                                                            \
  } while (0)
#define ATL_FLUSH(t)                                                                                                        \
  do {                                                                                                                      \
    if (blockIdx.x == 0 && g_actor_tl && (threadIdx.x & 63) == 0)                                                           \
      for (int k_ = 0; k_ < 7; ++k_) g_actor_tl[(threadIdx.x >> 6) * 64 + 8 * (t) + k_] = atl_acc[k_];                      \
  } while (0)
#else
#define ATL(k)
#define ATL_FLUSH(t)
#endif
// FULL: all four waves of the workgroup are live (nw == 4: every workgroup but the last one of a batch that is not a multiple of
// 256 envs).  That copy has its sixteen k-steps UNROLLED: the k-step index is then a compile-time constant -- which LDS region a
// k-step reads, whether it refills a ring slot and which one fold away, and with them the wave-uniform BRANCHES the rolled loop
// needs around every refill slot (`streamed`, `nw == 4`: six per k-step) and the row-tile loop's eight back edges per env tile.
// A branch costs a wave that has its SIMD to itself ~8 ns when it falls through and 20-50 ns when it is taken
// (tests/tools/exp/fwd_branch_probe.hip, branch_cost_probe.hip); the k-loops have 32 k-steps per env step.
// Measured in one session: 29.3 -> 28.0 us per fused env step.  The ragged copy (FULL = false) keeps the rolled loop and its
// run-time tests.
// (Also tried on this form, not kept: ONE vmcnt(0) + s_barrier per PAIR of k-steps with the refills three positions ahead --
// correct, 16 rendezvous per env step instead of 32, and no faster (28.1-28.6 against 27.9 us): the four waves are not skewed.)
template <int IN, bool FULL>
AE_DEV void actor_forward_wg_f16x3_impl(const ActorParams &A, const ActorParamsH &H, const float4 *w1_lds, uint4 *ring, int nw,
                                        const float (&s)[IN], float (&out)[3]) {
  static_assert(IN + 1 <= 16, "augmented input does not fit one f16 MFMA k-step");
  const int lane = threadIdx.x & 63;
  const int half = lane >> 5;
  constexpr int NT = 8;
  const half8 *w1h = reinterpret_cast<const half8 *>(w1_lds + ACTOR_W1_LDS_FLOATS / 4) + lane;   // [8 R][hi | lo][64]
  const float4 *b2w3 = w1_lds + ACTOR_W1A_FLOATS / 4;
  const float4 *b2tab = b2w3 + ACTOR_HID;                  // b2 alone, four consecutive neurons per float4
  float z[3] = {0.f, 0.f, 0.f};
  // this wave's four fragments of a fill (f = wave + 4 fi): wave-uniform source bases and LDS destinations, once per call
  const unsigned voff16 = (threadIdx.x & 63u) * 16u;
  uint64_t fill_src[4];
  unsigned fill_dst[4];
  // bases for the ragged-workgroup refill inside the k-loop (nw < 4: the last workgroup of a batch that is not a multiple of 256)
  const uint64_t w2h_base = scalar_opaque((uint64_t)(uintptr_t)H.W2H), w2l_base = scalar_opaque((uint64_t)(uintptr_t)H.W2L);
  {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    static_for<0, 4>([&](auto FI) {
      constexpr int fi = FI;
      const int f = wave + 4 * fi;
      const uint64_t a = (uint64_t)(uintptr_t)(((f & 1) ? H.W2L : H.W2H) + (f >> 1) * 64);
      fill_src[fi] = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a) |
                     ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) << 32);
      fill_dst[fi] = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(uintptr_t)ring + (unsigned)(f * 1024)));
    });
  }
  // everything this wave has in flight (ring slots from the previous call's tail or actor_ring_init, and whatever the
  // env step left behind) has landed
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // The two env column tiles of the wave (envs 0..31 and 32..63) are processed one after the other: 128 accumulator
  // registers instead of 256 (the full set plus the operands does not fit without spills into the k-loop, and scratch
  // traffic inside the loop would also break the vmcnt accounting of the ring).  W2 streams through the ring twice.
#pragma unroll 1
  for (int t = 0; t < 2; ++t) {
    // Layer 1 on the f16 MFMA as well, three passes like layer 2: H1^T[32 R + row][env] = W1aug[32 R + row][:] . saug[:][env],
    // saug = [obs, 1, 0..].  B operand: lane l holds saug[8 (l >> 5) + j], j = 0..7, of env (l & 31) + 32 t, split hi / lo.  (As
    // f32 MFMAs, four k-pairs of 64 cycles each, layer 1 held the matrix pipe for 1.7 us of every env step; as 3 x 32 cycles 0.6 us.)
#ifdef ARMENV_TIMELINE
    // prologue | resident k-steps' bodies | streamed even | streamed odd | layer 3 | the sixteen heads' vmcnt waits | their barriers
    unsigned long long atl_acc[7] = {0, 0, 0, 0, 0, 0, 0}, atl_prev = __builtin_amdgcn_s_memtime();
#endif
    half8 oh, ol;
    static_for<0, 8>([&](auto JI) {
      constexpr int j = JI;
      float v0 = 0.f, v1 = 0.f;   // k = j (lanes 0..31) and k = 8 + j (lanes 32..63)
      if constexpr (j < IN) { const float other = __shfl_xor(s[j], 32); v0 = (half == t) ? s[j] : other; }
      else if constexpr (j == IN) v0 = 1.f;
      if constexpr (8 + j < IN) { const float other = __shfl_xor(s[8 + j], 32); v1 = (half == t) ? s[8 + j] : other; }
      else if constexpr (8 + j == IN) v1 = 1.f;
      const float v = half ? v1 : v0;
      const _Float16 h = (_Float16)v;
      oh[j] = h;
      ol[j] = (_Float16)(v - (float)h);
    });
    auto layer1 = [&](int R) {   // raw layer-1 sums of row tile R: register i <-> neuron 32 R + 8 (i / 4) + 4 half + (i % 4)
      const half8 *w = w1h + (R < 8 ? R : 7) * 128;
      const half8 wh = w[0], wl = w[64];
      f32x16 a1;
      static_for<0, 16>([&](auto RI) { constexpr int r = RI; a1[r] = 0.f; });
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, oh, a1, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, ol, a1, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, oh, a1, 0, 0, 0);
      return a1;
    };
    // relu + f16 hi / lo split of registers 8 u + 2 c, 8 u + 2 c + 1 of a1 -> halfs 2 c, 2 c + 1 of k-step u's B operand.
    // K order inside a k-step: (half, j) <-> neuron 16 ks + 8 (j / 4) + 4 half + (j % 4); W2H / W2L are packed to match.
    auto relu = [](float x) { const int b = __float_as_int(x); return __int_as_float(b > 0 ? b : 0); };   // one v_max_i32, see actor_forward_wave
    auto split2 = [&](const f32x16 &a1, auto UI, auto CI, half8 (&h)[2], half8 (&l)[2]) {
      constexpr int u = UI, c = CI;
      const float x0 = relu(a1[8 * u + 2 * c]), x1 = relu(a1[8 * u + 2 * c + 1]);
      const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
      h[u][2 * c] = h0; h[u][2 * c + 1] = h1;
      l[u][2 * c] = (_Float16)(x0 - (float)h0); l[u][2 * c + 1] = (_Float16)(x1 - (float)h1);
    };
    // the same in two halves that fit the shadow of one MFMA each: (A) relu + hi, (B) lo
    // hi as ONE packed conversion (v_cvt_pk_f16_f32) whose two halves are converted back for the lo parts: written on scalar halves
    // hipcc converted each value to f16 on its own for the subtraction and packed the pair a second time for the operand
    typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
    typedef float float2_t __attribute__((ext_vector_type(2)));
    float sx0 = 0.f, sx1 = 0.f;
    half2_t shp = {(_Float16)0.f, (_Float16)0.f};
    auto split2a = [&](const f32x16 &a1, auto UI, auto CI, half8 (&h)[2]) {
      constexpr int u = UI, c = CI;
      sx0 = relu(a1[8 * u + 2 * c]); sx1 = relu(a1[8 * u + 2 * c + 1]);
      const float2_t xp = {sx0, sx1};
      shp = __builtin_convertvector(xp, half2_t);
      h[u][2 * c] = shp[0]; h[u][2 * c + 1] = shp[1];
    };
    auto split2b = [&](auto UI, auto CI, const half8 (&h)[2], half8 (&l)[2]) {
      constexpr int u = UI, c = CI;
      // two scalar subtractions, kept apart: hipcc's SLP pass pairs them into one v_pk_add_f32, and a packed f32 instruction in
      // an MFMA's shadow costs ~13 cycles more than the two scalar ones it replaces (MI355X_MICROARCH.md: an anti-lever beside
      // MFMAs); in-session 28.4 -> 27.9 us per fused env step
      float d0 = sx0 - (float)shp[0];
      asm("" : "+v"(d0));
      const float d1 = sx1 - (float)shp[1];
      const float2_t dp = {d0, d1};
      const half2_t lp = __builtin_convertvector(dp, half2_t);
      l[u][2 * c] = lp[0]; l[u][2 * c + 1] = lp[1];
    };
    f32x16 acc[NT];   // start from the layer-2 bias: register r <-> neuron 32 nt + 8 (r / 4) + 4 half + (r % 4)
    static_for<0, NT>([&](auto NI) {
      constexpr int nt = NI;
      static_for<0, 4>([&](auto QI) {
        constexpr int q = QI;
        const float4 b = b2tab[8 * nt + 2 * q + half];
        acc[nt][4 * q] = b.x; acc[nt][4 * q + 1] = b.y; acc[nt][4 * q + 2] = b.z; acc[nt][4 * q + 3] = b.w;
      });
    });
    half8 bh[2], bl[2], bh_n[2], bl_n[2];
    {
      const f32x16 a1 = layer1(0);
      static_for<0, 2>([&](auto UI) { static_for<0, 4>([&](auto CI) { split2(a1, UI, CI, bh, bl); }); });
    }
    // One k-step.  Head: make k-step ks + 1 visible (own share landed, then everyone's).  Body: the 24 MFMAs of k-step
    // ks from (ch, cl), read one k-step ago, each followed by one slice of the other work so that it issues while the
    // matrix pipe is busy (an MFMA holds the pipe for 8 issue slots and the wave issues in order: work placed behind a
    // block of MFMAs waits for all of them):
    //   slots 1..7    every other slot one quarter of the ring refill (scalar address arithmetic + one LDS DMA),
    //   slots 8..23   one LDS read each of k-step ks + 1's fragments into (nh, nl) and, on odd k-steps, half of the
    //                 relu + hi / lo split of a pair of layer-1 values of the next row tile.
    auto kstep = [&](auto ODD, int ks, const half8 (&ch)[NT], const half8 (&cl)[NT], half8 (&nh)[NT], half8 (&nl)[NT],
                     const f32x16 &a1n, auto &&inject) __attribute__((always_inline)) {
      constexpr int u = ODD;
      // own share of the fill of k-step ks + 1 has landed: issued one k-step ago (ahead 2) / two k-steps ago, with the last k-step's
      // four DMA instructions still in flight behind it (ahead 3, full workgroups: loads return in order)
      if constexpr (FULL && ACTOR_FILL_AHEAD == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ATL(5);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      ATL(6);
      const half8 *slot = reinterpret_cast<const half8 *>(ring) + actor_region((ks + 1) & 15) * 16 * 64 + lane;
      const bool streamed = ks >= ACTOR_KRES;            // resident k-steps consume no ring slot: nothing to refill
      const int kf = actor_next2(streamed ? ks : ACTOR_KRES);
      static_for<0, 3 * NT>([&](auto MI) {
        constexpr int m = MI;
        // order: pass (hi*hi, hi*lo, lo*hi) outermost, so consecutive MFMAs never share an accumulator
        constexpr int k3 = m / NT, nt = m % NT;
        __builtin_amdgcn_sched_barrier(0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k3 == 2 ? cl[nt] : ch[nt], k3 == 1 ? bl[u] : bh[u], acc[nt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        inject(MI);      // the caller's own matrix-pipe work for this slot (even k-steps: layer 1 of the next row tile)
        __builtin_amdgcn_sched_barrier(0);
        // slots 1, 3, 5, 7: one quarter each of the refill two stream positions ahead (early, so that it has most of a
        // k-step to land before the head of the next one waits for it)
        if constexpr (m < 8 && m % 2 == 1) {
          constexpr int fi = m / 2;
          if (streamed) {
            if (FULL || nw == 4)
              glds16(reinterpret_cast<const void *>(fill_src[fi] + (uint64_t)(unsigned)kf * (8u * 64u * 16u)), voff16,
                     fill_dst[fi] + (unsigned)actor_region(kf) * (16u * 64u * 16u));
            else if (m == 1) {
              const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
              for (int f = wave; f < 16; f += nw) actor_ring_fill_one(w2h_base, w2l_base, ring, kf, f);
            }
          }
        }
        // slots 8..23: one LDS read each of k-step ks + 1's fragments; odd k-steps: half of the relu / split of a pair
        if constexpr (m >= 8) {
          constexpr int r = m - 8;
          if constexpr (r % 2 == 0) nh[r / 2] = slot[r * 64];
          else nl[r / 2] = slot[r * 64];
          if constexpr (u == 1) {
            if constexpr (r % 2 == 0) split2a(a1n, std::integral_constant<int, (r / 2) / 4>{}, std::integral_constant<int, (r / 2) % 4>{}, bh_n);
            else split2b(std::integral_constant<int, (r / 2) / 4>{}, std::integral_constant<int, (r / 2) % 4>{}, bh_n, bl_n);
          }
        }
      });
      __builtin_amdgcn_sched_barrier(0);
    };
    half8 ah0[NT], al0[NT], ah1[NT], al1[NT];   // A fragments of even / odd k-steps
    // k-step 0 of this pass.  First pass: every wave's share of it has landed after the vmcnt(0) above and this barrier;
    // second pass: it was made visible by the last k-step of the first pass (its own read there is dropped so that no
    // A fragments stay live across the pass epilogue).
    if (t == 0) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
      const half8 *slot = reinterpret_cast<const half8 *>(ring) + lane;        // k-step 0: region 0
      static_for<0, NT>([&](auto NI) { constexpr int nt = NI; ah0[nt] = slot[(2 * nt) * 64]; al0[nt] = slot[(2 * nt + 1) * 64]; });
    }
    f32x16 a1n = {};
    if constexpr (FULL) ATL(0);
    half8 l1wh, l1wl;
    auto row_tile = [&](int R) __attribute__((always_inline)) {
      // layer 1 of row tile R + 1 rides inside the even k-step: its three MFMAs are a dependent chain on one accumulator tile and
      // need their two operands from LDS; behind the k-step (rounds 2-3) that was ~190 cycles of every row tile with the matrix
      // pipe mostly idle (instrumented build: odd k-steps 1 068 cycles against 882 for even ones); eight slots apart nothing waits
      kstep(std::integral_constant<int, 0>{}, 2 * R, ah0, al0, ah1, al1, a1n, [&](auto MI) __attribute__((always_inline)) {
        constexpr int m = MI;
        if constexpr (m == 2) { const half8 *w = w1h + (R + 1 < 8 ? R + 1 : 7) * 128; l1wh = w[0]; l1wl = w[64]; }
        if constexpr (m == 7) {
          f32x16 z0;
          static_for<0, 16>([&](auto RI) { constexpr int r = RI; z0[r] = 0.f; });
          a1n = __builtin_amdgcn_mfma_f32_32x32x16_f16(l1wh, oh, z0, 0, 0, 0);
        }
        if constexpr (m == 15) a1n = __builtin_amdgcn_mfma_f32_32x32x16_f16(l1wh, ol, a1n, 0, 0, 0);
        if constexpr (m == 23) a1n = __builtin_amdgcn_mfma_f32_32x32x16_f16(l1wl, oh, a1n, 0, 0, 0);
      });
      if constexpr (FULL) ATL(2 * R < ACTOR_KRES ? 1 : 2);
      kstep(std::integral_constant<int, 1>{}, 2 * R + 1, ah1, al1, ah0, al0, a1n, [](auto) {});
      if constexpr (FULL) ATL(2 * R + 1 < ACTOR_KRES ? 1 : 3);
      bh[0] = bh_n[0]; bh[1] = bh_n[1]; bl[0] = bl_n[0]; bl[1] = bl_n[1];
    };
    if constexpr (FULL) {
      static_for<0, 8>([&](auto RI) { constexpr int R = RI; row_tile(R); });
    } else {
#pragma unroll 1
      for (int R = 0; R < 8; ++R) row_tile(R);
    }
    // relu of layer 2 (the bias is already in the accumulators) and layer 3 over the 128 neurons this lane holds for its env
    // column (the other 128 are in lane ^ 32), on the matrix pipe (layer3_column)
    f32x4 d30 = {0.f, 0.f, 0.f, 0.f}, d31 = {0.f, 0.f, 0.f, 0.f};
    const float *w3p = reinterpret_cast<const float *>(b2w3) + 16 * half + 1 + (lane & 3);
    layer3_column<NT>(w3p, acc, d30, d31, relu);
    const f32x4 s3 = d30 + d31;
    static_for<0, 3>([&](auto OI) {
      constexpr int o = OI;
      const float tot = s3[o] + __shfl_xor(s3[o], 32);       // lane e holds env e: tile e >> 5, column e & 31
      z[o] = (half == t) ? tot : z[o];
    });
    if constexpr (FULL) { ATL(4); ATL_FLUSH(t); }
  }
  if (A.raw) static_for<0, 3>([&](auto OI) { constexpr int o = OI; out[o] = z[o] + A.b3[o]; });          // QValueNet: net_mlp.py:57
  else static_for<0, 3>([&](auto OI) { constexpr int o = OI; out[o] = tanhf(z[o] + A.b3[o]) * A.bound; });   // net_mlp.py:40
}
template <int IN>
AE_DEV void actor_forward_wg_f16x3(const ActorParams &A, const ActorParamsH &H, const float4 *w1_lds, uint4 *ring, int nw,
                                   const float (&s)[IN], float (&out)[3]) {
  if (__builtin_expect(nw == 4, 1)) actor_forward_wg_f16x3_impl<IN, true>(A, H, w1_lds, ring, nw, s, out);
  else actor_forward_wg_f16x3_impl<IN, false>(A, H, w1_lds, ring, nw, s, out);
}

// ------------------------------------------------------------------------------------------------------------------
// DATD3_MLP.take_action (/root/reference/algo/DATD3/DATD3_mlp.py:88-109; DARC_MLP's, algo/DARC/DARC_mlp.py:92-113, is the same; DADDPG_MLP's,
// algo/DADDPG/DADDPG_mlp.py:77-97, is the same with critic2 = critic1) for the 256 envs of a workgroup:
//     a1 = actor1(s), a2 = actor2(s), q1 = critic1(s, a1), q2 = critic2(s, a2), action = a1 if q1 >= q2 else a2
// as four passes of the f16x3 workgroup actor above, one network after the other through the same LDS tables and the same W2 ring.
// All four nets run as (OBS + 3)-input nets: the critics take cat(s, a) (net_mlp.py:55; reach 6 + 3, push / pick 9 + 3), the actors
// see s padded with three zeros against zero weight columns (their packed W1 rows carry zeros beyond in_dim: exact), so that ONE copy
// of the pass serves the four calls (a pass is 2 400 instructions; the loop over the nets keeps it at that).  Between two nets every
// wave drains its own ring DMA, the workgroup meets (everyone has finished reading the previous net's tables and ring), the next net's
// tables arrive as one pre-built image and its resident + first streamed k-steps are requested, and the workgroup meets again.
// nets / nets_h: device arrays [4] = actor1, actor2, critic1, critic2 (in_dim OBS + 3; the critics raw = 1 with fc3 in row 0 of their
// B2W3 table).  s: the lane's observation.  nw: live waves (a ragged last workgroup stages with its live threads).
// picked: 0 actor1 / 1 actor2.
template <int OBS>
AE_DEV void datd3_forward_wg(const ActorParams *nets, const ActorParamsH *nets_h, float4 *w1_lds, uint4 *ring, int nw, const float (&s)[OBS],
                             float (&out)[3], float &q1, float &q2, int &picked) {
  constexpr int IN = OBS + 3;
  float x[IN], a1[3] = {0.f, 0.f, 0.f}, a2[3] = {0.f, 0.f, 0.f};   // (no array indexed by `net`: a run-time subscript would put it in scratch)
  q1 = q2 = 0.f;
  static_for<0, OBS>([&](auto DI) { constexpr int d = DI; x[d] = s[d]; });
  x[OBS] = x[OBS + 1] = x[OBS + 2] = 0.f;
  const int live = nw * 64;
  uint64_t staged = 0;      // the W2 fragments whose tables and ring are in LDS
#pragma unroll 1
  for (int net = 0; net < 4; ++net) {
    const ActorParams A = nets[net];
    const ActorParamsH H = nets_h[net];
    // DADDPG (DADDPG_mlp.py:93-94: ONE critic values both proposals): the fourth entry of the table is the third.  Its tables are in LDS
    // and the third pass's tail has refilled the ring for the pass that follows (as between two env steps of the single fused actor):
    // nothing to drain, meet for or stage.
    const uint64_t want = scalar_opaque((uint64_t)(uintptr_t)H.W2H);
    if (want != staged) {
      actor_ring_drain();
      __syncthreads();
      if (A.lds_image) actor_stage_image(A.lds_image, w1_lds, nw);
      else { actor_stage_w1(A.W1P, w1_lds, A.B2W3, IN, live); actor_stage_w1h(A.W1P, w1_lds, IN, live); }
      actor_ring_init(H, ring, nw);
      actor_ring_drain();       // (the table image arrives by DMA like the ring: it must have landed before anyone reads a table)
      __syncthreads();
      staged = want;
    }
    if (net >= 2) static_for<0, 3>([&](auto KI) { constexpr int k = KI; x[OBS + k] = net == 2 ? a1[k] : a2[k]; });      // cat(s, a_i), net_mlp.py:55
    float o[3];
    actor_forward_wg_f16x3<IN>(A, H, w1_lds, ring, nw, x, o);
    static_for<0, 3>([&](auto KI) { constexpr int k = KI; a1[k] = net == 0 ? o[k] : a1[k]; a2[k] = net == 1 ? o[k] : a2[k]; });
    q1 = net == 2 ? o[0] : q1;
    q2 = net == 3 ? o[0] : q2;
  }
  picked = q1 >= q2 ? 0 : 1;                                                                               // DATD3_mlp.py:107
  static_for<0, 3>([&](auto KI) { constexpr int k = KI; out[k] = picked ? a2[k] : a1[k]; });
}

}  // namespace armenv
