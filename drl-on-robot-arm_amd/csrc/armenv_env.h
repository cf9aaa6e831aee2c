// armenv_env.h -- device side of the env engine: per-env state layout (EnvParams), the per-lane env bodies (ReachLane,
// PushLane, PickLane), and the kernels built from them: reset, one-step, T-step rollout with optional fused policy, FK / IK
// entry kernels, state exchange, actor entry + weight packing.  Host side and C ABI: armenv.hip.
#pragma once
#include "../../include/armenv.h"
#include "armenv_kin.h"
#include "armenv_actor.h"

using namespace armenv;


// Handle constants that only the rare paths of a step read -- the in-place reset of a finished env and the joint-limit
// handling -- kept in device memory behind EnvParams::cold instead of the kernel-argument segment: as arguments they are
// loaded into scalar registers at kernel entry and stay live across the IK loop, whose own f64 constants then have to be
// rematerialised (two s_mov each) or spilled to VGPR lanes on every trip.
template <typename T> struct EnvCold {
  T q_init[NJ];
  T p_init[3];          // FK(q_init), computed on the device at create time (init_consts_kernel)
  T trig_init[2 * NJ];  // cos(q_init)[7], sin(q_init)[7] from the device's own sincos_all
  T lim[2 * NJ + 1];    // URDF joint limits: lower[7], upper[7]; then limit_erp (ArmEnvConfig.limit_erp, read by clamp mode 2 only)
  double goal_lo[3], goal_hi[3];
  double push_rest_z, push_place_min, push_place_max, push_place_z;
  uint64_t seed, env_id0;
  // push task, ArmEnvConfig.push_contact_model = 1: the cube under stepSimulation (CubeLane::cube_fall, contact_dyn).  Read once per
  // step, behind the IK loop.
  int32_t push_model;       // contact: 0 rounds 1-4 (tool sphere, full push-out; always 0 for the pick task), 1 velocity-level contact
  int32_t fall_on;          // 1: the cube falls from its spawn height (push and pick: the same body in the same scene); 0: at rest from reset on
  int32_t fall_land;        // the stepSimulation call (counted from the spawn) in which the falling cube reaches the table
  T fall_c;                 // g dt^2 / 2: the cube is fall_c k (k + 1) below its spawn height after k free steps
  T fall_keep;              // 1 - push_drop_relax
  T place_z;                // spawn height
  T tool_radius, tool_below, erp_dt, split, fric_dv, dt;   // erp_dt = push_contact_erp / push_dt
};

template <typename T> struct EnvParams {
  // state
  T *q;
  T *ep_return;
  T *last_return;
  float *goal;
  int32_t *step;
  uint32_t *episode;
  int32_t *last_len;
  uint8_t *last_success;
  unsigned long long *counters;   // [ceil(N / 32)][16]: one row per wave (half-filled waves carry 32 envs), summed by counters_sum_kernel on read
  T *aux;  // push: [9][N] = cube xyz, target xyz, d_last, cube velocity xy;  pick: [11][N] = cube xyz, target xyz, d_last, gripper state, hold offset xyz
  T *trig; // [14][N] = cos q[7], sin q[7]: the pair every FK starts from, carried with q (see ReachLane::trig)
  int64_t n;
  // task constants
  T dv;
  T reach_dis;
  int32_t max_steps;
  int32_t auto_reset;
  T box_lo[3];
  T box_hi[3];
  const EnvCold<T> *cold;   // device memory: reset / limit constants (see EnvCold)
  int32_t half_waves;       // 1: a wave carries 32 envs in its lanes 0..31 (lanes 32..63 exit at once): a batch of at most
                            // 32 x #SIMDs envs then spreads over twice as many SIMDs, one wave each, and every wave's per-step
                            // maximum of IK trips is taken over 32 lanes instead of 64 (ArmEnvConfig.rollout_lanes_per_wave)
  uint64_t seed;            // Philox key and global index of env 0: the fused policy's noise reads them every step
  uint64_t env_id0;
  // push task (rl_push_env.py): simplified pusher model + reward constants
  T push_success_dis, push_cube_half, push_eef_radius;
  T push_rest_z;            // pick: a held cube never sinks below its rest height (placement constants: EnvCold)
  // pick task (rl_pick_env.py): gripper model
  T pick_gripper_length, pick_trigger_dis, pick_jaw_half;
  T fence_z;   // parity fence: steps that end with the flange below this height are counted (ArmEnvConfig.fence_z)
  IKParams<T> ik;
  ChainDev<T> chain;
};

// Launch-end flush of a lane's event counts into the handle's counters.  Every wave owns one 64-byte row
// counters[w][16] = {episodes, successes, env steps (row 0 only), non-finite, IK updates, joint-limit steps, low-flange steps,
// steps whose IK ran to the iteration cap, steps whose IK passed through an ill-conditioned system, 0...}; the wave sums its lanes
// and ONE lane does a plain read-modify-write of the row -- launches on a stream are serialised, nobody else touches it.
// Why not atomicAdd on one address: same-address atomics execute one at a time at the memory side of the fabric,
// 12 ns each from anywhere on the chip (tests/tools/exp/launch_probe.hip: 4096 waves x 1 atomic = 50 us; per-wave rows =
// nothing), and the kernel cannot complete before they have.  One atomic per wave was 12 us of every 24 us
// armenv_step launch at 65536 envs.
// wave_sum: a count v of b significant bits costs b ballots + popcounts on the scalar unit, no cross-lane data moves.
constexpr int kCounterCols = 16;
// wave index of the calling lane in the launch: the row of the handle's per-wave counters it flushes into
AE_DEV int64_t wave_row() { return ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; }
AE_DEV uint32_t wave_sum(uint32_t v) {
  uint32_t s = 0;
  for (int b = 0; b < 32 && __ballot((v >> b) != 0u) != 0ull; ++b) s += (uint32_t)__popcll(__ballot((v >> b) & 1u)) << b;
  return s;
}
// Env index of the calling lane, or -1 for a lane that has none.
template <typename T>
AE_DEV int64_t lane_env(const EnvParams<T> &P) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (!P.half_waves) return t;
  return (t & 32) ? (int64_t)-1 : ((t >> 6) << 5) + (t & 31);
}
template <typename T>
AE_DEV void flush_counts(const EnvParams<T> &P, int64_t i, uint32_t n_done, uint32_t n_succ, uint32_t n_bad, uint32_t n_upd,
                         uint32_t n_lim, uint32_t n_low, uint32_t n_cap, uint32_t n_cond) {
  (void)i;
  unsigned long long *row = P.counters + kCounterCols * wave_row();
  const uint32_t d = wave_sum(n_done), s = wave_sum(n_succ), b = wave_sum(n_bad), u = wave_sum(n_upd);
  const uint32_t l = wave_sum(n_lim), z = wave_sum(n_low), c = wave_sum(n_cap), k = wave_sum(n_cond);
  if ((threadIdx.x & 63) == 0) {   // lane 0 of a launched wave is always a live env (i < N is a prefix)
    if (d) row[0] += d;
    if (s) row[1] += s;
    if (b) row[3] += b;
    row[4] += u;
    if (l) row[5] += l;
    if (z) row[6] += z;
    if (c) row[7] += c;
    if (k) row[8] += k;
  }
}
// largest v of the wave's active lanes (v < 256): one ballot per bit, scalar unit only
AE_DEV uint32_t wave_max8(uint32_t v) {
  uint32_t m = 0;
  for (int b = 7; b >= 0; --b) { if (__ballot(v >= (m | (1u << b))) != 0ull) m |= 1u << b; }
  return m;
}
// Bookkeeping builds: what the schedule cost this wave -- loop iterations that ran an IK trip (row[9]) and step tails (row[10]).
AE_DEV void flush_schedule(unsigned long long *counters, uint32_t wave_trips, uint32_t wave_rounds) {
  // every lane of the wave holds the same two counts; the lowest live lane writes them
  if (__ffsll((unsigned long long)__ballot(1)) - 1 == (int)(threadIdx.x & 63)) {
    unsigned long long *row = counters + kCounterCols * wave_row();
    row[9] += wave_trips;
    row[10] += wave_rounds;
  }
}
AE_DEV void flush_env_steps(unsigned long long *counters, int64_t i, unsigned long long env_steps) {
  if (i == 0) counters[2] += env_steps;
}

// armenv_counters: totals[16] = column sums of the per-wave rows.
static __global__ __launch_bounds__(256) void counters_sum_kernel(const unsigned long long *rows, int64_t n_rows,
                                                                   unsigned long long *totals) {
  __shared__ unsigned long long part[4][kCounterCols];
  const int col = threadIdx.x & (kCounterCols - 1);
  unsigned long long acc = 0;
  for (int64_t r = threadIdx.x / kCounterCols; r < n_rows; r += 256 / kCounterCols) acc += rows[kCounterCols * r + col];
  for (int o = kCounterCols; o < 64; o <<= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) < kCounterCols) part[threadIdx.x >> 6][col] = acc;
  __syncthreads();
  if (threadIdx.x < kCounterCols) totals[threadIdx.x] = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
}

// The rollout's next action, loaded while the current step runs.  The load goes STRAIGHT INTO ACCUMULATION REGISTERS from
// inline asm and is waited for explicitly after the IK: as an ordinary load hipcc parks the three values in AGPRs to
// relieve the IK's VGPR pressure, which needs the data -- so it waited (vmcnt(0)) a few hundred instructions after issuing
// the load, i.e. for most of an HBM round trip, every step.
// Hazard of the pattern: the compiler believes the registers are defined at prefetch_issue; should it ever copy or spill
// them before prefetch_settle it would copy stale data.  To give it no reason to, the three AGPRs are used by nothing
// else and prefetch_settle itself moves them into VGPRs AFTER its wait, inside the same asm block: what is carried round
// the loop is those VGPRs (a first version carried the AGPRs and hipcc renamed them with v_accvgpr_mov in FRONT of the
// wait -- harmless only because the load had been in flight for a whole IK).  Guards: tests/test_isa_guard.py walks the
// built code object from the issue to the settle and fails if anything reachable in between touches the three AGPRs (CPU,
// every build); the rollout == step-launches bitwise tests (reach / push / pick, f64 / f32) catch a wrong value on the GPU.
struct ActionPrefetch { float x, y, z; };
AE_DEV void prefetch_issue(const float *src, ActionPrefetch &d) {
  asm volatile("global_load_dword %0, %3, off\n\tglobal_load_dword %1, %3, off offset:4\n\tglobal_load_dword %2, %3, off offset:8"
               : "=a"(d.x), "=a"(d.y), "=a"(d.z) : "v"(src) : "memory");
}
AE_DEV void prefetch_settle(ActionPrefetch &d, float (&out)[3]) {
  asm volatile("s_waitcnt vmcnt(0)\n\tv_accvgpr_read_b32 %0, %3\n\tv_accvgpr_read_b32 %1, %4\n\tv_accvgpr_read_b32 %2, %5"
               : "=v"(out[0]), "=v"(out[1]), "=v"(out[2]) : "a"(d.x), "a"(d.y), "a"(d.z) : "memory");
}

// The carried (cos q, sin q) pair is re-derived from q when an env's own step counter reaches a multiple of this (a power
// of two; 0 = never).  See ReachLane::trig.
#ifndef ARMENV_TRIG_REDERIVE
#define ARMENV_TRIG_REDERIVE 512
#endif
constexpr int kTrigRederive = ARMENV_TRIG_REDERIVE;
#ifndef ARMENV_ROLLOUT_PEEL
#define ARMENV_ROLLOUT_PEEL 3
#endif
#ifndef ARMENV_STEP_PEEL
#define ARMENV_STEP_PEEL 0
#endif

struct StepIO {
  const float *action;
  float *obs;
  float *reward;
  uint8_t *done;
  uint8_t *success;
  float *terminal_obs;
  uint8_t *updates;   // nullable: DLS updates the step's IK call applied (saturated at 255): the per-step view of counters[4] / [7]
  double *diag;       // nullable, f64 [N][4]: the step's exit frame position (the numbers its distance, reward and flags were computed
                      // from) and its reward before the f32 store.  An output of the MODE 2 bookkeeping build only.
};

// goal ~ U(box): a + (b - a) * u per axis as random.uniform does (rl_reach_env.py:180-182), then the
// f32 cast of :213-215.  Always f64 arithmetic so that both precisions draw identical goals.
template <typename T>
AE_DEV void sample_goal(const EnvParams<T> &P, int64_t i, uint32_t episode, float (&g)[3]) {
  double u0, u1, u2, u3;
  const EnvCold<T> &K = *P.cold;
  philox_pair(K.seed, K.env_id0 + (uint64_t)i, episode, 0u, u0, u1);
  philox_pair(K.seed, K.env_id0 + (uint64_t)i, episode, 1u, u2, u3);
  g[0] = (float)(K.goal_lo[0] + (K.goal_hi[0] - K.goal_lo[0]) * u0);
  g[1] = (float)(K.goal_lo[1] + (K.goal_hi[1] - K.goal_lo[1]) * u1);
  g[2] = (float)(K.goal_lo[2] + (K.goal_hi[2] - K.goal_lo[2]) * u2);
}

template <typename T>
AE_DEV void store_obs6(float *obs, int64_t i, const T (&p)[3], const float (&g)[3]) {
  float2 *o = reinterpret_cast<float2 *>(obs + 6 * i);  // 24 B rows: 8-byte aligned
  o[0] = make_float2((float)p[0], (float)p[1]);
  o[1] = make_float2((float)p[2], g[0]);
  o[2] = make_float2(g[1], g[2]);
}

template <typename T>
AE_DEV void store_obs9(float *obs, int64_t i, const T (&p)[3], const T (&c)[3], const T (&t)[3]) {
  float *o = obs + 9 * i;
  static_for<0, 3>([&](auto KI) { constexpr int k = KI; o[k] = (float)p[k]; o[3 + k] = (float)c[k]; o[6 + k] = (float)t[k]; });
}

// FK(q_init) and (cos, sin)(q_init) once per handle, with the same device code the step uses; written into the handle's EnvCold.
template <class C, typename T>
__global__ void init_consts_kernel(EnvParams<T> P, EnvCold<T> *cold) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  T q[NJ];
  static_for<0, NJ>([&](auto II) { constexpr int i = II; q[i] = cold->q_init[i]; });
  FKState<T> S;
  T cq[NJ], sq[NJ];
  sincos_all<T>(q, cq, sq);
  fk<C, T>(P.chain, cq, sq, S);
  cold->p_init[0] = S.p[0]; cold->p_init[1] = S.p[1]; cold->p_init[2] = S.p[2];
  static_for<0, NJ>([&](auto II) { constexpr int i = II; cold->trig_init[i] = cq[i]; cold->trig_init[NJ + i] = sq[i]; });
}

// A fresh handle's (cos q, sin q) = (1, 0) for q = 0 (the pool is zero-filled): a step issued before the first reset then
// starts from a valid pose instead of degenerate all-zero rotation frames.
template <typename T>
__global__ __launch_bounds__(256) void trig_identity_kernel(EnvParams<T> P) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n) return;
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; P.trig[(int64_t)j * P.n + i] = T(1); });
}

// Optional per-wave timeline (make timeline; tests/tools/exp/run_timeline.py): wall-clock stamps at kernel entry, after
// the state loads, after the IK loop and at exit, plus IK update count and placement.  Off in the product build.
#ifdef ARMENV_TIMELINE
static __device__ unsigned long long *g_timeline;   // [16 launches][waves][8]
static __device__ unsigned g_tl_launch;             // launch counter: read at wave entry, bumped by env 0 at its exit
#define TL_STAMP(name) const unsigned long long name = wall_clock64()
#else
#define TL_STAMP(name)
#endif

// Fused exploration policy of the rollout loop (/root/reference/main.py:116-117):
//   a = clip(actor(obs) + N(0, sigma), +-clip);  kind RANDOM = zero actor.
struct PolicyParams {
  int32_t kind;      // ARMENV_POLICY_*
  float sigma;       // action_bound * opt.gamma = 0.686 in run()
  float clip;        // action_bound = 0.7
  float bound;       // actor output scale
  ActorParams actor; // ARMENV_POLICY_ACTOR / _F16X3
  ActorParamsH actor_h;  // ARMENV_POLICY_ACTOR_F16X3 only
  const ActorParams *datd3;      // ARMENV_POLICY_DATD3: device arrays [4] = actor1, actor2, critic1, critic2 (armenv_actor.h datd3_forward_wg)
  const ActorParamsH *datd3_h;
};

// TD3_MLP.take_action (/root/reference/algo/TD3/TD3_mlp.py:82-97) for n states; MODE 0 exact f32, 1 f16x3.
template <int IN, int MODE>
__global__ __launch_bounds__(256) void actor_kernel(ActorParams A, ActorParamsH H, int64_t n, const float *states, float *actions) {
  __shared__ float4 w1_lds[(MODE == 1 ? ACTOR_W1_LDS_FLOATS_H : ACTOR_W1_LDS_FLOATS) / 4];
  __shared__ uint4 w2_ring[MODE == 1 ? ACTOR_RING_UINT4 : 1];
  actor_stage_w1(A.W1P, w1_lds, A.B2W3, IN);
  if constexpr (MODE == 1) { actor_stage_w1h(A.W1P, w1_lds, IN); actor_ring_init(H, w2_ring, 4); }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // grid covers n rounded up to 256
  const int64_t ic = i < n ? i : n - 1;
  float s[IN], a[3];
  static_for<0, IN>([&](auto DI) { constexpr int d = DI; s[d] = states[ic * IN + d]; });
  if constexpr (MODE == 1) {
    actor_forward_wg_f16x3<IN>(A, H, w1_lds, w2_ring, 4, s, a);
    actor_ring_drain();
  } else {
    actor_forward_wave<IN>(A, w1_lds, s, a);
  }
  if (i < n) { actions[3 * i] = a[0]; actions[3 * i + 1] = a[1]; actions[3 * i + 2] = a[2]; }
}

// torch Linear layouts ([out][in]) -> the operand layouts of armenv_actor.h
// out_dim: rows of W3 (3: PolicyNet; 1: QValueNet, whose fc3 becomes row 0 of the table and rows 1, 2 are zero)
static __global__ void actor_pack_kernel(const float *W1, const float *b1, const float *W2, const float *b2, const float *W3,
                                  int in_dim, float *W1P, float *W2P, float *B2W3, _Float16 *W2H, _Float16 *W2L, int out_dim) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < ACTOR_HID * ACTOR_HID) {   // f16 hi/lo split of W2 in the 32x32x16 A-operand order
    const int j = t & 7, lane = (t >> 3) & 63, nt = (t >> 9) & 3, part = (t >> 11) & 1, ks = t >> 12;
    const float x = W2[(32 * (4 * part + nt) + (lane & 31)) * ACTOR_HID + 16 * ks + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3)];
    const _Float16 hi = (_Float16)x;
    W2H[t] = hi;
    W2L[t] = (_Float16)(x - (float)hi);
  }
  if (t < ACTOR_HID * ACTOR_W1P_COLS) {   // [k][16]: the row's input weights, zeros, the bias last
    const int k = t / ACTOR_W1P_COLS, j = t % ACTOR_W1P_COLS;
    W1P[t] = j < in_dim ? W1[k * in_dim + j] : (j == ACTOR_W1P_COLS - 1 ? b1[k] : 0.f);
  }
  if (t < ACTOR_HID * ACTOR_HID) {
    const int c = t & 3, l32 = (t >> 2) & 31, part = (t >> 7) & 1, k = t >> 8;
    W2P[t] = W2[(32 * (4 * part + c) + l32) * ACTOR_HID + k];
  }
  if (t < ACTOR_HID * 4) {
    const int n = t >> 2, c = t & 3;
    B2W3[t] = c == 0 ? b2[n] : (c - 1 < out_dim ? W3[(c - 1) * ACTOR_HID + n] : 0.f);
  }
}

// sin and cos of 2 pi u for u in [0, 1): the quadrant k = rint(4 u) is taken off exactly (u - k / 4 is exact in f32), the remainder
// |x| <= pi / 4 goes through the Cephes single-precision kernels (< 1 ulp there), the quadrant is put back by swaps and sign flips.
// ~25 instructions for the pair; sincosf on 2 pi u carries a large-argument reduction path (v_alignbit / v_ffbh_u32 chains) that this
// argument never needs: the two calls were ~230 of the ~570 instructions the in-kernel policy added to a step.
struct SinCos { float s, c; };
AE_DEV SinCos sincos_2pi(float u) {
  const int k = (int)fmaf(u, 4.0f, 0.5f);               // the nearest quadrant boundary (ties either way: |x| <= pi / 4 both sides)
  const float kf = (float)k;
  const float x = 6.283185307179586f * fmaf(-0.25f, kf, u);
  const float z = x * x;
  const float sx = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, x, x);
  const float cx = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z, fmaf(-0.5f, z, 1.0f));
  const float ss = (k & 1) ? cx : sx, cc = (k & 1) ? sx : cx;
  return SinCos{(k & 2) ? -ss : ss, ((k + 1) & 2) ? -cc : cc};
}

// Three N(0,1) draws for (env, episode, step): Philox block 0x80000000|step of the env's stream (reset draws use
// blocks < 2^31), Box-Muller in f32 on u = (w + 1) * 2^-32 in (0, 1]: r = sqrt(-2 ln u) through v_log_f32 / v_sqrt_f32 (1 ulp each),
// the angle through sincos_2pi -- within 3e-7 relative of the libm forms (logf, sqrtf, cosf / sinf of 2 pi u) the parity tests hold
// the stream against.
AE_DEV void policy_noise(uint64_t seed, uint64_t env_id, uint32_t episode, uint32_t step, float (&nz)[3]) {
  uint32_t c[4] = {(uint32_t)env_id, (uint32_t)(env_id >> 32), episode, 0x80000000u | step};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  const float k = 2.3283064365386963e-10f;  // 2^-32
  const float u0 = ((float)c[0] + 1.0f) * k, u1 = (float)c[1] * k, u2 = ((float)c[2] + 1.0f) * k, u3 = (float)c[3] * k;
  // -2 ln u = -2 ln 2 * log2 u; u >= 2^-32 is a normal number (v_log_f32 needs no denormal scaling)
  const float r0 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u0));
  const float r1 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u2));
  const SinCos a0 = sincos_2pi(u1), a1 = sincos_2pi(u3);
  nz[0] = r0 * a0.c; nz[1] = r0 * a0.s; nz[2] = r1 * a1.c;
}

// Per-lane state of one reach env and the body of one env step.  The single-step kernel and the T-step rollout
// kernel both run this code: a rollout of any length is bit-identical to the same steps as separate launches (see `cq` below).
template <class C, typename T, int MODE = 0> struct ReachLane {
  static constexpr int kMode = MODE;                  // build mode (armenv_kin.h): 0 default, 1 bookkeeping, 2 bookkeeping + IK tip offset
  static constexpr bool kFence = kFenceOf(MODE);     // the bookkeeping build of the lane: parity-fence counts (ArmEnvConfig.fence_counters)
  using M = Mth<T>;
  using Chain = C;
  static constexpr int kTask = ARMENV_TASK_REACH;
  static constexpr int kObs = 6;
  static constexpr int kAuxRows = 0, kAuxDim = 0;
  static constexpr const char *kName = "reach";

  // RLReachEnv.reset (rl_reach_env.py:132-217) of env i; goal_in (nullable) f32 [N][3].
  static AE_DEV void reset_env(const EnvParams<T> &P, int64_t i, const float *goal_in, float *obs) {
    float g[3];
    // episode[i] counts the resets of env i, whoever chose the goal: it indexes the env's Philox goal draws AND keys the
    // fused policy's exploration noise (policy_noise), so an env reset with caller goals must not replay its noise sequence
    const uint32_t ep = P.episode[i];
    P.episode[i] = ep + 1u;
    if (goal_in) {
      g[0] = goal_in[3 * i]; g[1] = goal_in[3 * i + 1]; g[2] = goal_in[3 * i + 2];
    } else {
      sample_goal(P, i, ep, g);
    }
    static_for<0, NJ>([&](auto JI) { constexpr int j = JI; P.q[(int64_t)j * P.n + i] = P.cold->q_init[j]; });
    static_for<0, 2 * NJ>([&](auto JI) { constexpr int j = JI; P.trig[(int64_t)j * P.n + i] = P.cold->trig_init[j]; });
    static_for<0, 3>([&](auto KI) { constexpr int k = KI; P.goal[(int64_t)k * P.n + i] = g[k]; });
    P.step[i] = 0;
    P.ep_return[i] = T(0);
    if (obs) store_obs6<T>(obs, i, P.cold->p_init, g);
  }

  // distance the logging summary reports: |FK(q) - goal|
  static AE_DEV double summary_distance(const EnvParams<T> &P, int64_t i) {
    T q[NJ], cq[NJ], sq[NJ];
    static_for<0, NJ>([&](auto JI) { constexpr int j = JI; q[j] = P.q[(int64_t)j * P.n + i]; });
    sincos_all<T>(q, cq, sq);
    FKState<T> S;
    fk<C, T>(P.chain, cq, sq, S);
    const T x = S.p[0] - (T)P.goal[0 * P.n + i], y = S.p[1] - (T)P.goal[1 * P.n + i], z = S.p[2] - (T)P.goal[2 * P.n + i];
    return (double)M::sqrt(M::fma(x, x, M::fma(y, y, z * z)));
  }

  T q[NJ];
  // (cos q, sin q) of the seven joints, part of the env's state (EnvParams::trig, [14][N] in HBM): set from q_init at a
  // reset (EnvParams::trig_init) or from q by armenv_set_state, then only ever advanced by the IK's angle-addition updates
  // (rotate_small), so a step's first FK starts from the previous step's last -- a full sincos of seven joints was 328 of a
  // step's 3 263 instructions.  Because the pair travels with q through load / store, a trajectory is a pure function of
  // (state, actions): armenv_rollout(T) and T armenv_step launches produce the same bits for any T (tested at T = 500).
  // Round 1 re-derived the pair at every launch start instead; T-step launches and step launches then differed in the last
  // bit of the pair, which Bullet's 2 acos(w) orientation error (quantised at 3e-8 sqrt(k) rad near convergence) amplified
  // to ~1e-7 rad and, rarely, into a flipped IK update count.  The pair is re-derived from q whenever the env's own step
  // counter reaches a multiple of 512 (never inside the reference's 501-step episodes), which bounds the accumulated
  // rounding of the incremental rotations (~1e-16 each) for callers that run unbounded episodes.
  T cq[NJ], sq[NJ];
  // The link frames of the current pose.  A step's IK leaves FK(q) of the pose it ends with (the frame _reward reads);
  // that is also the frame the NEXT step starts from, so a kernel that keeps the lane in registers across steps (CARRY: the
  // rollout kernels without a fused actor) carries it over instead of recomputing it from the same (cos q, sin q): 120 of a
  // step's ~3 300 instructions, the same bits.  Since round 4 the frame is made EAGERLY wherever the pose is set anew -- at the
  // launch's start (make_frame), inside the in-place reset and the trig re-derivation (both rare, out of line) -- so that
  // step_begin tests nothing: the lazy "if (!have_S) fk" of rounds 1-3 was a taken branch on every step.
  FKState<T> S;
  AE_DEV void make_frame(const EnvParams<T> &P) { fk<C, T>(P.chain, cq, sq, S); }
  float g[3];
  int32_t step;
  T ep_ret;
  uint32_t n_done = 0, n_succ = 0, n_bad = 0, n_upd = 0;   // flushed to the handle's counters once per launch
  uint32_t n_lim = 0, n_low = 0, n_cap = 0, n_cond = 0;    // parity fence: steps the IK left the URDF limits / ended with the flange below fence_z / ran to the iteration cap / was ill-conditioned
  T minpiv = T(1e30);                                      // smallest LDL^T pivot of the running step's IK call (fence bookkeeping)

  AE_DEV void load(const EnvParams<T> &P, int64_t i) {
    const int64_t n = P.n;
    static_for<0, NJ>([&](auto JI) { constexpr int j = JI; q[j] = P.q[(int64_t)j * n + i]; });
    static_for<0, 3>([&](auto KI) { constexpr int k = KI; g[k] = P.goal[(int64_t)k * n + i]; });
    step = P.step[i];
    ep_ret = P.ep_return[i];
    static_for<0, NJ>([&](auto JI) { constexpr int j = JI; cq[j] = P.trig[(int64_t)j * n + i]; sq[j] = P.trig[(int64_t)(NJ + j) * n + i]; });
  }
  AE_DEV void derive_trig() { sincos_all<T>(q, cq, sq); }

  AE_DEV void store(const EnvParams<T> &P, int64_t i) {
    const int64_t n = P.n;
    static_for<0, NJ>([&](auto JI) { constexpr int j = JI; P.q[(int64_t)j * n + i] = q[j]; });
    static_for<0, NJ>([&](auto JI) { constexpr int j = JI; P.trig[(int64_t)j * n + i] = cq[j]; P.trig[(int64_t)(NJ + j) * n + i] = sq[j]; });
    P.step[i] = step;
    P.ep_return[i] = ep_ret;
    flush_counts(P, i, n_done, n_succ, n_bad, n_upd, n_lim, n_low, n_cap, n_cond);
  }

  // RLReachEnv.step + _reward (rl_reach_env.py:219-319) in three pieces -- step_begin, the IK trips (ik_trip), step_tail --
  // so that the lockstep kernels (env_step below: all lanes of a wave walk through a step together) and the
  // lane-asynchronous rollout kernel (env_rollout_async_kernel: every lane at its own step and trip) run the same code.
  //
  // step_begin: everything ahead of the IK trips -- the frame of the start pose (carried over or recomputed) and the
  // clipped Cartesian target (:237-242).
  // CARRY: S already is the frame of (cq, sq) (see S above); otherwise it is computed here.
  template <bool CARRY>
  AE_DEV void step_begin(const EnvParams<T> &P, const T (&a)[3], T (&tgt)[3]) {
    if constexpr (!CARRY) make_frame(P);
    ik_target<T, false>(S, a, P.dv, P.box_lo, P.box_hi, tgt);
    minpiv = T(1e30);
  }
  // step_tail: everything after the IK (q, cq / sq and S = FK(q) hold its result; `updates` trips applied an update; lim_hit:
  // ik_limits): step counter, distance, reward / done / success (:264-309), observation (:319), episode accounting,
  // in-place reset.  Writes row i of the caller's buffers.
  template <bool CARRY>
  AE_DEV void step_tail(const EnvParams<T> &P, int64_t i, const StepIO &io, int updates, bool lim_hit) {
    const int64_t n = P.n;
    n_upd += (uint32_t)updates;
    if constexpr (kFence) {
      n_lim += lim_hit ? 1u : 0u; n_low += (S.p[2] < P.fence_z) ? 1u : 0u; n_cap += (updates >= P.ik.max_iters) ? 1u : 0u;
      n_cond += (minpiv < P.ik.fence_pivot) ? 1u : 0u;
      // the per-step view of the update / cap counts.  Only in the bookkeeping builds: as a nullable pointer of the default
      // kernels it stayed live in two scalar registers across the IK loop, whose f64 constants were then rematerialised on
      // every trip (+7 scalar instructions per trip, +1.3 % on the headline; A/B against the round-2 tree)
      if (io.updates) io.updates[i] = (uint8_t)(updates > 255 ? 255 : updates);
    }
    step += 1;                                                                    // :264
    const T dx = S.p[0] - (T)g[0], dy = S.p[1] - (T)g[1], dz = S.p[2] - (T)g[2];
    const T dist = M::sqrt(M::fma(dx, dx, M::fma(dy, dy, dz * dz)));              // :281
    // :299-309 as selects (the if / else-if chain is three exec-mask regions and a branch per step; the same values):
    //   step > max_steps -> (-10 d, done);  else d < reach_dis -> (0, done, success);  else (-10 d, not done)
    const bool over = step > P.max_steps;
    const bool succ = !over & (dist < P.reach_dis);
    const bool done = over | succ;
    const T reward = succ ? T(0) : -dist * T(10);
    ep_ret += reward;
    if constexpr (kTipOf(kMode)) {
      if (io.diag) { io.diag[4 * i] = (double)S.p[0]; io.diag[4 * i + 1] = (double)S.p[1]; io.diag[4 * i + 2] = (double)S.p[2]; io.diag[4 * i + 3] = (double)reward; }
    }

    bool finite = true;
    {   // one test on the sum: a NaN or an infinity among the seven angles makes it non-finite (inf - inf = NaN), and finite
      // joint angles cannot overflow it
      const T sum = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + q[6]);
      finite = M::finite(sum);
    }
    if (!finite) n_bad += 1;

    io.reward[i] = (float)reward;
    io.done[i] = done ? 1 : 0;
    io.success[i] = succ ? 1 : 0;
    if (__builtin_expect(io.terminal_obs != nullptr, 0)) store_obs6<T>(io.terminal_obs, i, S.p, g);

    // The pair (cos q, sin q) is re-derived from q when the env's own step counter reaches a multiple of kTrigRederive (see cq
    // above; rounds 1-3 did this at the top of the NEXT step: the same values, since nothing touches the pose in between).
    // One rare region for "done" and "re-derive": the common path is a fall-through to the observation store.
    const bool rederive = kTrigRederive > 0 && (step & (kTrigRederive - 1)) == 0;     // step >= 1 here
    store_obs6<T>(io.obs, i, S.p, g);                                                // :319 (a reset overwrites it below)
    cur_obs[0] = (float)S.p[0]; cur_obs[1] = (float)S.p[1]; cur_obs[2] = (float)S.p[2];
    if (__builtin_expect(done | rederive, 0)) {
      if (done) {
        P.last_return[i] = ep_ret;
        P.last_len[i] = step;
        P.last_success[i] = succ ? 1 : 0;
        n_done += 1;
        if (succ) n_succ += 1;
      }
      if (done && P.auto_reset) {
        const uint32_t ep = P.episode[i];
        sample_goal(P, i, ep, g);
        P.episode[i] = ep + 1u;
        static_for<0, 3>([&](auto KI) { constexpr int k = KI; P.goal[(int64_t)k * n + i] = g[k]; });
        const EnvCold<T> &K = *P.cold;
        static_for<0, NJ>([&](auto JI) { constexpr int j = JI; q[j] = K.q_init[j]; });
        static_for<0, NJ>([&](auto JI) { constexpr int j = JI; cq[j] = K.trig_init[j]; sq[j] = K.trig_init[NJ + j]; });
        step = 0;
        ep_ret = T(0);
        store_obs6<T>(io.obs, i, K.p_init, g);      // the same lane's second store to these addresses: program order holds
        cur_obs[0] = (float)K.p_init[0]; cur_obs[1] = (float)K.p_init[1]; cur_obs[2] = (float)K.p_init[2];
        if constexpr (CARRY) make_frame(P);
      } else if (rederive) {
        derive_trig();
        if constexpr (CARRY) make_frame(P);
      }
    }
  }
  // The lockstep step.  Returns the number of IK updates.
  // `prefetched` (nullable): registers a caller is loading during this step (the rollout's next action).  They are
  // consumed here, right after the IK and BEFORE this step's stores are issued: the s_waitcnt the consumption needs
  // then covers only that old load; placed at the next step's top it would also wait for this step's stores
  // (vmcnt is in-order) -- measured 15 % of the wave's cycles.
  template <bool CARRY = false, int PEEL = 0>
  AE_DEV int env_step(const EnvParams<T> &P, int64_t i, const T (&a)[3], const StepIO &io, ActionPrefetch *prefetched = nullptr,
                    float (*next_action)[3] = nullptr) {
    T tgt[3];
    step_begin<CARRY>(P, a, tgt);
    const int updates = ik_lockstep<C, T, kMode, PEEL>(P.chain, P.ik, q, tgt, S, cq, sq, minpiv);                               // :244-257
    const bool lim_hit = ik_limits<C, T, kMode>(P.chain, P.ik, q, S, cq, sq);
    if (prefetched) prefetch_settle(*prefetched, *next_action);
    step_tail<CARRY>(P, i, io, updates, lim_hit);
    return updates;
  }

  float cur_obs[3];   // eef part of the observation the policy sees next (goal part is g)
  // eef of the current state, for the first policy call of a launch (later ones reuse the step's exit FK)
  AE_DEV void refresh_obs(const EnvParams<T> &P) {
    FKState<T> F;
    fk<C, T>(P.chain, cq, sq, F);
    cur_obs[0] = (float)F.p[0]; cur_obs[1] = (float)F.p[1]; cur_obs[2] = (float)F.p[2];
  }
  AE_DEV void policy_obs(float (&s)[6]) const {
    s[0] = cur_obs[0]; s[1] = cur_obs[1]; s[2] = cur_obs[2]; s[3] = g[0]; s[4] = g[1]; s[5] = g[2];
  }
};

// ---- push and pick tasks (/root/reference/envs/rl_push_env.py, envs/rl_pick_env.py) -------------------------------
// Placement of cube and target: rejection sampling, <= 1000 tries (push :195-214, pick :190-208); f64 always.  Both bodies are
// SPAWNED at z = push_place_z (0.01) and the distance test sees them there; the target is a fixed body and stays, the cube is
// dynamic and comes to rest on the table at push_rest_z (fitted to the reference's recorded push run: 14.74 mm lower).
//   push: six draws per try (x, y, yaw, x_t, y_t, yaw_t), both bodies at the rest height, planar distance test;
//   pick: seven draws per try (x, y, yaw, x_t, y_t, z_t, yaw_t), target anywhere in the workspace box, 3-D distance.
template <bool PICK, typename T>
AE_DEV void cube_sample(const EnvParams<T> &P, int64_t i, uint32_t episode, T (&cube)[3], T (&target)[3]) {
  const EnvCold<T> &K = *P.cold;
  double cx = 0, cy = 0, tx = 0, ty = 0, tz = K.push_place_z;    // push: the target is a fixed body at its spawn height (:206, :221-224)
  constexpr uint32_t kBlocks = PICK ? 4u : 3u;
  for (uint32_t t = 0; t < 1000u; ++t) {
    double u0, u1, u2, u3, u4, u5;
    philox_pair(K.seed, K.env_id0 + (uint64_t)i, episode, kBlocks * t + 0u, u0, u1);
    philox_pair(K.seed, K.env_id0 + (uint64_t)i, episode, kBlocks * t + 1u, u2, u3);
    philox_pair(K.seed, K.env_id0 + (uint64_t)i, episode, kBlocks * t + 2u, u4, u5);
    cx = K.goal_lo[0] + (K.goal_hi[0] - K.goal_lo[0]) * u0;
    cy = K.goal_lo[1] + (K.goal_hi[1] - K.goal_lo[1]) * u1;
    tx = K.goal_lo[0] + (K.goal_hi[0] - K.goal_lo[0]) * u3;
    ty = K.goal_lo[1] + (K.goal_hi[1] - K.goal_lo[1]) * u4;
    const double dx = cx - tx, dy = cy - ty;
    double d;
    if constexpr (PICK) {
      tz = K.goal_lo[2] + (K.goal_hi[2] - K.goal_lo[2]) * u5;      // 7th draw (u6, the target yaw) is unused
      const double dz = K.push_place_z - tz;                      // the test sees the cube at its spawn height (:193-207)
      d = ::sqrt(::fma(dx, dx, ::fma(dy, dy, dz * dz)));
    } else {
      d = ::sqrt(::fma(dx, dx, dy * dy));   // both are spawned at the same z (:199, :206)
      (void)u5;
    }
    (void)u2;
    if (d >= K.push_place_min && d <= K.push_place_max) break;
  }
  // push_contact_model 1: spawned at push_place_z, the cube has begun to fall in reset()'s own stepSimulation (:241); otherwise it
  // is already at rest on the table
  cube[0] = (T)cx; cube[1] = (T)cy; cube[2] = K.fall_on ? K.place_z - K.fall_c * T(1) * T(2) : (T)K.push_rest_z;
  target[0] = (T)tx; target[1] = (T)ty; target[2] = (T)tz;
}

// Per-lane state and step body of the two cube tasks.
//   PICK = false: RLPushEnv (rl_push_env.py:310-440).
//   PICK = true:  RLPickEnv (rl_pick_env.py:310-440): same reward / done logic; the arm differs in three ways --
//     the start position is rounded through float (:328), z is clipped to [0, 0.55 + gripper_length] (:313), and only
//     joints 0..5 receive the IK result (:343 `range(self.end_effector_index)`), so joint 7 keeps its reset value and
//     the tool's yaw error is never corrected; the gripper (closed by getClosestPoints, :412-416) is the build's own
//     model, see grip().
template <class C, typename T, bool PICK, int MODE = 0> struct CubeLane {
  static constexpr int kMode = MODE;
  static constexpr bool kFence = kFenceOf(MODE);
  using M = Mth<T>;
  using Chain = C;
  static constexpr int kTask = PICK ? ARMENV_TASK_PICK : ARMENV_TASK_PUSH;
  static constexpr int kObs = 9;
  static constexpr int kAuxRows = PICK ? 11 : 9, kAuxDim = PICK ? 12 : 10;
  static constexpr const char *kName = PICK ? "pick" : "push";
  T q[NJ];
  T cq[NJ], sq[NJ];         // (cos q, sin q), carried with q in the env's state: see ReachLane::cq
  T cube[3], target[3], d_last;
  T vel[2] = {T(0), T(0)};  // push: the cube's planar velocity (push_contact_model 1)
  FKState<T> S;             // link frames of the current pose, carried from step to step (push only): see ReachLane::S
  AE_DEV void make_frame(const EnvParams<T> &P) { fk<C, T>(P.chain, cq, sq, S); }
  T grip = T(0);            // pick: 0 open, 1 closed, 2 closed and holding the cube
  T off[3] = {T(0), T(0), T(0)};   // pick: cube - tip while held
  int32_t step;
  T ep_ret;
  uint32_t n_done = 0, n_succ = 0, n_bad = 0, n_upd = 0, n_lim = 0, n_low = 0, n_cap = 0, n_cond = 0;
  T minpiv = T(1e30);       // smallest LDL^T pivot of the running step's IK call (fence bookkeeping)
  float cur_obs[3];
  AE_DEV void refresh_obs(const EnvParams<T> &P) {
    FKState<T> F;
    fk<C, T>(P.chain, cq, sq, F);
    cur_obs[0] = (float)F.p[0]; cur_obs[1] = (float)F.p[1]; cur_obs[2] = (float)F.p[2];
  }
  AE_DEV void policy_obs(float (&s)[9]) const {
    static_for<0, 3>([&](auto KI) { constexpr int k = KI; s[k] = cur_obs[k]; s[3 + k] = (float)cube[k]; s[6 + k] = (float)target[k]; });
  }

  AE_DEV T dist_ct() const {
    const T x = cube[0] - target[0], y = cube[1] - target[1], z = cube[2] - target[2];
    return M::sqrt(M::fma(x, x, M::fma(y, y, z * z)));
  }

  static AE_DEV double summary_distance(const EnvParams<T> &P, int64_t i) {
    const T x = P.aux[0 * P.n + i] - P.aux[3 * P.n + i], y = P.aux[1 * P.n + i] - P.aux[4 * P.n + i],
            z = P.aux[2 * P.n + i] - P.aux[5 * P.n + i];
    return (double)M::sqrt(M::fma(x, x, M::fma(y, y, z * z)));
  }

  // RLPushEnv.reset (rl_push_env.py:145-256) / RLPickEnv.reset (rl_pick_env.py:140-252) of env i; goal_in (nullable)
  // f32 [N][6] = cube xyz, target xyz.
  static AE_DEV void reset_env(const EnvParams<T> &P, int64_t i, const float *goal_in, float *obs) {
    const int64_t n = P.n;
    T cube[3], target[3];
    const uint32_t ep = P.episode[i];          // counts every reset of the env (see ReachLane::reset_env)
    P.episode[i] = ep + 1u;
    if (goal_in) {
      static_for<0, 3>([&](auto KI) { constexpr int k = KI; cube[k] = (T)goal_in[6 * i + k]; target[k] = (T)goal_in[6 * i + 3 + k]; });
      // the caller places the cube in the plane; its height is the engine's (one step into its fall, as in cube_sample)
      if (P.cold->fall_on) cube[2] = P.cold->place_z - P.cold->fall_c * T(1) * T(2);
    } else {
      cube_sample<PICK, T>(P, i, ep, cube, target);
    }
    static_for<0, NJ>([&](auto JI) { constexpr int j = JI; P.q[(int64_t)j * n + i] = P.cold->q_init[j]; });
    static_for<0, 2 * NJ>([&](auto JI) { constexpr int j = JI; P.trig[(int64_t)j * n + i] = P.cold->trig_init[j]; });
    static_for<0, 3>([&](auto KI) { constexpr int k = KI; P.aux[(int64_t)k * n + i] = cube[k]; P.aux[(int64_t)(3 + k) * n + i] = target[k]; });
    const T x = cube[0] - target[0], y = cube[1] - target[1], z = cube[2] - target[2];
    P.aux[(int64_t)6 * n + i] = M::sqrt(M::fma(x, x, M::fma(y, y, z * z)));
    if constexpr (PICK) static_for<7, 11>([&](auto KI) { constexpr int k = KI; P.aux[(int64_t)k * n + i] = T(0); });   // gripper open (:235-238)
    else static_for<7, 9>([&](auto KI) { constexpr int k = KI; P.aux[(int64_t)k * n + i] = T(0); });                    // cube at rest in the plane
    P.step[i] = 0;
    P.ep_return[i] = T(0);
    if (obs) store_obs9<T>(obs, i, P.cold->p_init, cube, target);
  }

  AE_DEV void load(const EnvParams<T> &P, int64_t i) {
    const int64_t n = P.n;
    static_for<0, NJ>([&](auto JI) { constexpr int j = JI; q[j] = P.q[(int64_t)j * n + i]; });
    static_for<0, 3>([&](auto KI) { constexpr int k = KI; cube[k] = P.aux[(int64_t)k * n + i]; target[k] = P.aux[(int64_t)(3 + k) * n + i]; });
    d_last = P.aux[(int64_t)6 * n + i];
    if constexpr (PICK) {
      grip = P.aux[(int64_t)7 * n + i];
      static_for<0, 3>([&](auto KI) { constexpr int k = KI; off[k] = P.aux[(int64_t)(8 + k) * n + i]; });
    } else {
      vel[0] = P.aux[(int64_t)7 * n + i]; vel[1] = P.aux[(int64_t)8 * n + i];
    }
    step = P.step[i];
    ep_ret = P.ep_return[i];
    static_for<0, NJ>([&](auto JI) { constexpr int j = JI; cq[j] = P.trig[(int64_t)j * n + i]; sq[j] = P.trig[(int64_t)(NJ + j) * n + i]; });
  }
  AE_DEV void derive_trig() { sincos_all<T>(q, cq, sq); }

  AE_DEV void store(const EnvParams<T> &P, int64_t i) {
    const int64_t n = P.n;
    static_for<0, NJ>([&](auto JI) { constexpr int j = JI; P.q[(int64_t)j * n + i] = q[j]; });
    static_for<0, NJ>([&](auto JI) { constexpr int j = JI; P.trig[(int64_t)j * n + i] = cq[j]; P.trig[(int64_t)(NJ + j) * n + i] = sq[j]; });
    static_for<0, 3>([&](auto KI) { constexpr int k = KI; P.aux[(int64_t)k * n + i] = cube[k]; P.aux[(int64_t)(3 + k) * n + i] = target[k]; });
    P.aux[(int64_t)6 * n + i] = d_last;
    if constexpr (PICK) {
      P.aux[(int64_t)7 * n + i] = grip;
      static_for<0, 3>([&](auto KI) { constexpr int k = KI; P.aux[(int64_t)(8 + k) * n + i] = off[k]; });
    } else {
      P.aux[(int64_t)7 * n + i] = vel[0]; P.aux[(int64_t)8 * n + i] = vel[1];
    }
    P.step[i] = step;
    P.ep_return[i] = ep_ret;
    flush_counts(P, i, n_done, n_succ, n_bad, n_upd, n_lim, n_low, n_cap, n_cond);
  }

  // stepSimulation (:349), simplified: sphere (tool, radius r, centre p) vs axis-aligned box (cube, half-size h)
  // overlap test; on overlap the cube is displaced horizontally -- along the contact normal by the penetration
  // depth when the tool centre is outside the footprint, ahead of the tool along its travel p0 -> p when inside.
  AE_DEV void contact(const EnvParams<T> &P, const T (&p0)[3], const T (&p)[3]) {
    const T h = P.push_cube_half, r = P.push_eef_radius;
    if (M::fabs(p[2] - cube[2]) >= h + r) return;
    const T lx = cube[0] - h, hx = cube[0] + h, ly = cube[1] - h, hy = cube[1] + h;
    const T qx = p[0] < lx ? lx : (p[0] > hx ? hx : p[0]);
    const T qy = p[1] < ly ? ly : (p[1] > hy ? hy : p[1]);
    const T gx = p[0] - qx, gy = p[1] - qy;
    const T gap = M::sqrt(M::fma(gx, gx, gy * gy));
    if (gap >= r) return;
    if (gap > T(1e-9)) {
      const T depth = r - gap;
      cube[0] -= depth * (gx / gap);
      cube[1] -= depth * (gy / gap);
    } else {
      T mx = p[0] - p0[0], my = p[1] - p0[1];
      const T mn = M::sqrt(M::fma(mx, mx, my * my));
      if (mn < T(1e-3)) return;   // < 1 mm of horizontal travel: the tool presses down on the cube, no sweep
      mx /= mn; my /= mn;
      const T s = (r + h) - M::fma(cube[0] - p[0], mx, (cube[1] - p[1]) * my);
      if (s > T(0)) { cube[0] = M::fma(s, mx, cube[0]); cube[1] = M::fma(s, my, cube[1]); }
    }
  }

  // stepSimulation (:349), push_contact_model 1 (include/armenv.h).
  // k = number of stepSimulation calls since the cube was spawned, this one included (reset() made the first, :241).
  // The cube's height: free fall through the step in which it reaches the table, then the overshoot decays towards the rest height.
  AE_DEV void cube_fall(const EnvCold<T> &K, int k) {
    const T zf = K.place_z - K.fall_c * (T)k * (T)(k + 1);
    const T zr = (T)K.push_rest_z + (cube[2] - (T)K.push_rest_z) * K.fall_keep;
    cube[2] = k <= K.fall_land ? zf : zr;         // a select: a wave that has its SIMD to itself pays 20-50 ns per taken branch
  }
  // The cube in the plane: collision detection at the positions the step starts from (tool = vertical cylinder about the link-7
  // frame's xy, reaching tool_below under it), velocity-level contact along the horizontal normal unless the tool sits less deep in the
  // cube from above than from the side (then it presses the cube onto the table), Coulomb friction once the cube has landed, integration.
  // Two exec-mask regions at most -- lanes in contact, lanes whose cube moves -- with selects inside (a wave that has its SIMD to itself
  // pays 20-50 ns per taken branch); square roots and quotients through v_rsq + Newton (fast_rsqrt: a few ulp), the overlap test on
  // squared lengths.
  AE_DEV void contact_dyn(const EnvParams<T> &P, const EnvCold<T> &K, const T (&p)[3], int k) {
    const T h = P.push_cube_half, r = K.tool_radius, dt = K.dt;
    const T lo = p[2] - K.tool_below;
    const T lx = cube[0] - h, hx = cube[0] + h, ly = cube[1] - h, hy = cube[1] + h;
    const T qx = p[0] < lx ? lx : (p[0] > hx ? hx : p[0]);
    const T qy = p[1] < ly ? ly : (p[1] > hy ? hy : p[1]);
    const T gx = qx - p[0], gy = qy - p[1];
    const T g2 = gx * gx + gy * gy;
    if ((lo < cube[2] + h) & (g2 < r * r)) {
      const bool outside = g2 > T(1e-18);
      const T rs = fast_rsqrt<T>(outside ? g2 : T(1));
      // tool axis inside the footprint: out through the nearest face (-x, +x, -y, +y in this order on ties)
      const T e0 = hx - p[0], e1 = p[0] - lx, e2 = hy - p[1], e3 = p[1] - ly;
      T eb = e0, ix = T(-1), iy = T(0);
      { const bool c = e1 < eb; eb = c ? e1 : eb; ix = c ? T(1) : ix; }
      { const bool c = e2 < eb; eb = c ? e2 : eb; ix = c ? T(0) : ix; iy = c ? T(-1) : iy; }
      { const bool c = e3 < eb; eb = c ? e3 : eb; ix = c ? T(0) : ix; iy = c ? T(1) : iy; }
      const T pen = outside ? r - g2 * rs : eb + r;
      const T nx = outside ? gx * rs : ix, ny = outside ? gy * rs : iy;
      const T pen_v = (cube[2] + h) - lo;
      const T vn = vel[0] * nx + vel[1] * ny;
      const T tgt = pen < K.split ? K.erp_dt * pen : T(0);
      const T dv = (!(pen_v < pen) & (vn < tgt)) ? tgt - vn : T(0);
      vel[0] += dv * nx; vel[1] += dv * ny;
    }
    const T v2 = vel[0] * vel[0] + vel[1] * vel[1];
    if (v2 > T(0)) {
      const T f = T(1) - K.fric_dv * fast_rsqrt<T>(v2);       // (|v| - dec) / |v|
      const T fc = k >= K.fall_land ? (f > T(0) ? f : T(0)) : T(1);
      vel[0] *= fc; vel[1] *= fc;
      cube[0] += vel[0] * dt; cube[1] += vel[1] * dt;
    }
  }

  // Pick: gripper and cube after the arm's teleport (rl_pick_env.py:349 stepSimulation, :412-417 getClosestPoints ->
  // close the fingers -> stepSimulation).  BUILD-DEFINED MODEL (Bullet's finger / cube contact dynamics are not
  // restated): the gripper tip is the point gripper_length along the tool axis from the link-7
  // frame, a sphere of radius push_eef_radius.
  //   held cube:   rides with the tip (cube = tip + off), never below its rest height;
  //   open gripper: closes for the rest of the episode once the tip sphere is within trigger_dis of the cube box
  //                 (:412); it holds the cube iff the tip is then above the cube centre and the cube centre lies within
  //                 jaw_half of the tool axis horizontally;
  //   otherwise:   the tip pushes the cube like the push task's tool (contact()).
  AE_DEV void grip_step(const EnvParams<T> &P, const T (&p0)[3], const FKState<T> &S) {
    const T L = P.pick_gripper_length;
    T tip[3], tip0[3];
    static_for<0, 3>([&](auto KI) {
      constexpr int k = KI;
      tip[k] = M::fma(L, S.W[6 + k], S.p[k]);        // third column of the link-7 rotation = tool axis
      tip0[k] = tip[k] - (S.p[k] - p0[k]);           // previous tip under an unchanged tool orientation
    });
    if (grip == T(2)) {
      cube[0] = tip[0] + off[0];
      cube[1] = tip[1] + off[1];
      const T z = tip[2] + off[2];
      cube[2] = z < P.push_rest_z ? P.push_rest_z : z;
      return;
    }
    if (grip == T(0)) {
      const T h = P.push_cube_half;
      T g2 = T(0);
      static_for<0, 3>([&](auto KI) {
        constexpr int k = KI;
        const T lo = cube[k] - h, hi = cube[k] + h;
        const T c = tip[k] < lo ? lo : (tip[k] > hi ? hi : tip[k]);
        const T g = tip[k] - c;
        g2 = M::fma(g, g, g2);
      });
      if (M::sqrt(g2) - P.push_eef_radius < P.pick_trigger_dis) {
        const T hx = cube[0] - tip[0], hy = cube[1] - tip[1];
        const bool hold = tip[2] >= cube[2] && M::sqrt(M::fma(hx, hx, hy * hy)) <= P.pick_jaw_half;
        grip = hold ? T(2) : T(1);
        if (hold) {
          off[0] = hx; off[1] = hy; off[2] = cube[2] - tip[2];
          return;
        }
      }
    }
    contact(P, tip0, tip);
  }

  // The step in three pieces (see ReachLane): step_begin (frame of the start pose, clipped target :322-338; pick: the
  // float32-rounded start position :328), the IK trips, step_tail.
  T p0[3];                  // eef position at the start of the running step (the contact model's sweep origin)
  T q7s, c7s, s7s;          // pick: joint 7 and its (cos, sin) before the IK (rl_pick_env.py:343 applies joints 0..5 only)
  // CARRY: the caller keeps the lane in registers across steps (ReachLane::S).  Pick restores joint 7 in step_tail, so its exit
  // frame's orientation is not the next step's: only push carries the frame over.
  template <bool CARRY>
  AE_DEV void step_begin(const EnvParams<T> &P, const T (&a)[3], T (&tgt)[3]) {
    if constexpr (PICK) { q7s = q[NJ - 1]; c7s = cq[NJ - 1]; s7s = sq[NJ - 1]; }
    if constexpr (!(CARRY && !PICK)) make_frame(P);
    p0[0] = S.p[0]; p0[1] = S.p[1]; p0[2] = S.p[2];
    ik_target<T, PICK>(S, a, P.dv, P.box_lo, P.box_hi, tgt);
    minpiv = T(1e30);
  }
  template <bool CARRY>
  AE_DEV void step_tail(const EnvParams<T> &P, int64_t i, const StepIO &io, int updates, bool lim_hit) {
    constexpr bool kCarry = CARRY && !PICK;
    n_upd += (uint32_t)updates;
    if constexpr (kFence) {
      n_lim += lim_hit ? 1u : 0u; n_low += (S.p[2] < P.fence_z) ? 1u : 0u; n_cap += (updates >= P.ik.max_iters) ? 1u : 0u;
      n_cond += (minpiv < P.ik.fence_pivot) ? 1u : 0u;
      // the per-step view of the update / cap counts.  Only in the bookkeeping builds: as a nullable pointer of the default
      // kernels it stayed live in two scalar registers across the IK loop, whose f64 constants were then rematerialised on
      // every trip (+7 scalar instructions per trip, +1.3 % on the headline; A/B against the round-2 tree)
      if (io.updates) io.updates[i] = (uint8_t)(updates > 255 ? 255 : updates);
    }
    if constexpr (PICK) {
      q[NJ - 1] = q7s;              // rl_pick_env.py:343: joints 0..5 only; link-7 position and tool axis do not depend on q7
      cq[NJ - 1] = c7s; sq[NJ - 1] = s7s;
      // RLPickEnv calls stepSimulation twice per env step -- step() :348 and, behind the observation, _reward() :417 -- so env step j
      // observes the cube after call 2 j of its fall (reset() made call 1).  The state between two steps is the OBSERVED one: the call
      // that follows an observation is made up for here, ahead of call 2 j.  A held cube rides the gripper instead.
      const EnvCold<T> *Kp = P.cold;
      asm volatile("" : "+s"(Kp));
      if (Kp->fall_on && grip != T(2)) {
        if (step > 0) cube_fall(*Kp, 2 * step + 1);
        cube_fall(*Kp, 2 * step + 2);
      }
      grip_step(P, p0, S);          // :349, :412-417
    } else {
      // (the constants are read HERE: behind an opaque copy of the pointer hipcc cannot hoist their scalar loads above the IK loop, where
      // a dozen more live scalars cost the loop its f64 constants -- section 4b of DESIGN.md)
      const EnvCold<T> *Kp = P.cold;
      asm volatile("" : "+s"(Kp));
      const EnvCold<T> &K = *Kp;
      if (K.push_model == 1) { cube_fall(K, step + 2); contact_dyn(P, K, S.p, step + 2); }      // :349
      else contact(P, p0, S.p);                                                                // :349, rounds 1-4
    }
    step += 1;                                                                    // :355
    const T d_cur = dist_ct();                                                    // :388
    T test = d_cur - d_last;                                                      // :390-392
    if (M::fabs(test) < T(1e-5)) test = T(0.01);                                  // :393-394
    d_last = d_cur;                                                               // :396-397
    const float fx = (float)cube[0] - (float)target[0], fy = (float)cube[1] - (float)target[1],
                fz = (float)cube[2] - (float)target[2];                           // :378-384 float32 states
    const float dt32 = sqrtf(fmaf(fx, fx, fmaf(fy, fy, fz * fz)));                 // :400
    T reward;
    bool done;
    if (step > P.max_steps) { reward = (T)(-dt32 * 50.0f); done = true; }                       // :418-420
    else if ((double)dt32 < (double)P.push_success_dis) { reward = T(100); done = true; }       // :422-424
    else { reward = -test * T(100); done = false; }                                             // :427-428
    const bool succ = d_cur < P.push_success_dis;                                               // :430-432
    ep_ret += reward;
    if constexpr (kTipOf(kMode)) {
      if (io.diag) { io.diag[4 * i] = (double)S.p[0]; io.diag[4 * i + 1] = (double)S.p[1]; io.diag[4 * i + 2] = (double)S.p[2]; io.diag[4 * i + 3] = (double)reward; }
    }

    bool finite = true;
    {   // one test on the sum: a NaN or an infinity among the seven angles makes it non-finite (inf - inf = NaN), and finite
      // joint angles cannot overflow it
      const T sum = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + q[6]);
      finite = M::finite(sum);
    }
    if (!finite) n_bad += 1;

    io.reward[i] = (float)reward;
    io.done[i] = done ? 1 : 0;
    io.success[i] = succ ? 1 : 0;
    if (__builtin_expect(io.terminal_obs != nullptr, 0)) store_obs9<T>(io.terminal_obs, i, S.p, cube, target);
    // one rare region for "done" and the trig re-derivation (see ReachLane::step_tail)
    const bool rederive = kTrigRederive > 0 && (step & (kTrigRederive - 1)) == 0;     // step >= 1 here
    store_obs9<T>(io.obs, i, S.p, cube, target);                                     // :308 (a reset overwrites it below)
    cur_obs[0] = (float)S.p[0]; cur_obs[1] = (float)S.p[1]; cur_obs[2] = (float)S.p[2];
    if (__builtin_expect(done | rederive, 0)) {
      if (done) {
        P.last_return[i] = ep_ret;
        P.last_len[i] = step;
        P.last_success[i] = succ ? 1 : 0;
        n_done += 1;
        if (succ) n_succ += 1;
      }
      if (done && P.auto_reset) {
        const uint32_t ep = P.episode[i];
        cube_sample<PICK, T>(P, i, ep, cube, target);
        P.episode[i] = ep + 1u;
        d_last = dist_ct();                                                         // :243-245
        if constexpr (PICK) { grip = T(0); off[0] = off[1] = off[2] = T(0); }
        else { vel[0] = vel[1] = T(0); }
        const EnvCold<T> &K = *P.cold;
        static_for<0, NJ>([&](auto JI) { constexpr int j = JI; q[j] = K.q_init[j]; });
        static_for<0, NJ>([&](auto JI) { constexpr int j = JI; cq[j] = K.trig_init[j]; sq[j] = K.trig_init[NJ + j]; });
        step = 0;
        ep_ret = T(0);
        store_obs9<T>(io.obs, i, K.p_init, cube, target);      // the same lane's second store to these addresses
        cur_obs[0] = (float)K.p_init[0]; cur_obs[1] = (float)K.p_init[1]; cur_obs[2] = (float)K.p_init[2];
        if constexpr (kCarry) make_frame(P);
      } else if (rederive) {
        derive_trig();
        if constexpr (kCarry) make_frame(P);
      }
    }
  }
  template <bool CARRY = false, int PEEL = 0>
  AE_DEV int env_step(const EnvParams<T> &P, int64_t i, const T (&a)[3], const StepIO &io, ActionPrefetch *prefetched = nullptr,
                    float (*next_action)[3] = nullptr) {
    T tgt[3];
    step_begin<CARRY>(P, a, tgt);
    const int updates = ik_lockstep<C, T, kMode, PEEL>(P.chain, P.ik, q, tgt, S, cq, sq, minpiv);                               // :339-347
    const bool lim_hit = ik_limits<C, T, kMode>(P.chain, P.ik, q, S, cq, sq);
    if (prefetched) prefetch_settle(*prefetched, *next_action);
    step_tail<CARRY>(P, i, io, updates, lim_hit);
    return updates;
  }
};
template <class C, typename T, int MODE = 0> using PushLane = CubeLane<C, T, false, MODE>;
template <class C, typename T, int MODE = 0> using PickLane = CubeLane<C, T, true, MODE>;

// reset() of the envs whose mask byte is set (mask == NULL: all).
template <class Lane, typename T>
__global__ __launch_bounds__(256) void env_reset_kernel(EnvParams<T> P, const uint8_t *mask, const float *goal_in, float *obs) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n) return;
  if (mask && !mask[i]) return;
  Lane::reset_env(P, i, goal_in, obs);
}

// One env step per launch: load state -> FK -> target = clip(p + dv a) -> DLS IK loop -> FK -> (push: contact) ->
// reward / done -> obs pack -> episode accounting -> optional in-place reset -> store state.
// Lane = ReachLane (rl_reach_env.py:219-319), PushLane (rl_push_env.py:310-440) or PickLane (rl_pick_env.py:310-440).
// WAVES: register budget as for env_rollout_kernel (2: <= 256 registers per lane, for batches with more waves than SIMDs).
template <class Lane, typename T, int WAVES = 1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void env_step_kernel(EnvParams<T> P, StepIO io) {
  TL_STAMP(tl0);
#ifdef ARMENV_TIMELINE
  const unsigned tl_launch = __hip_atomic_load(&g_tl_launch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
  const int64_t i = lane_env(P);   // env of this lane (full or half-filled waves, EnvParams::half_waves)
  if (i < 0) return;
  if (i >= P.n) return;
  Lane L;
  L.load(P, i);
  T a[3];
  static_for<0, 3>([&](auto KI) { constexpr int k = KI; a[k] = (T)io.action[3 * i + k]; });
#ifdef ARMENV_TIMELINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  TL_STAMP(tl1);
  const int updates = L.template env_step<false, ARMENV_STEP_PEEL>(P, i, a, io);
  (void)updates;
  TL_STAMP(tl2);
  L.store(P, i);
  flush_env_steps(P.counters, i, (unsigned long long)P.n);
#ifdef ARMENV_TIMELINE
  {
    TL_STAMP(tl3);
    int mx = updates;
    for (int o = 32; o; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
    const unsigned long long dn = __ballot(L.n_done != 0);
    if ((threadIdx.x & 63) == 0 && g_timeline) {
      unsigned long long *r = g_timeline + 8 * ((int64_t)(tl_launch & 15) * (P.n >> 6) + (i >> 6));
      unsigned xcc, hw;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      r[0] = tl0; r[1] = tl1; r[2] = tl2; r[3] = tl3; r[4] = mx; r[5] = __popcll(dn); r[6] = xcc; r[7] = hw;
      if (i == 0) __hip_atomic_fetch_add(&g_tl_launch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#endif
}

// The rollout inner loop of /root/reference/main.py:108-128 for `steps` consecutive env steps in ONE launch: the env
// state stays in registers, every step's outputs go to row t of [steps][N][...] buffers, and the action of step t
// is either read from actions[t] (external policy: the trajectory of `steps` calls of env_step_kernel) or produced
// in-kernel by the fused exploration policy.  Because lanes never synchronise, a lane that needs extra IK updates
// in one step does not hold the other envs back for the rest of the launch: per-step cost approaches the MEAN
// update count instead of the per-launch MAX.
// POLICY is a compile-time copy of pol.kind: the external-action variant carries no actor / noise code, which keeps it
// free of the register spills the fused-actor variant's 128 MFMA accumulators would otherwise force on it.
// WAVES: waves per SIMD the kernel is built for.
//   1  the whole 512-register file for one wave (f64: 256 VGPRs + AGPRs as spill space, no scratch).  A batch of up to
//      64 x #SIMDs envs (65 536 on MI355X: BASELINE configs 2 and 5) is exactly one wave per SIMD, so nothing else could
//      run beside it anyway; the next action is prefetched through AGPRs (prefetch_issue / prefetch_settle).
//   2  at most 256 registers per lane, so that TWO waves share a SIMD: a larger batch no longer runs as consecutive rounds
//      of single waves -- the second wave issues into the 22 % of the first one's cycles that are dependent-f64 waits and
//      scalar instructions (measured: 1 048 576 envs 102 -> 87 us per step, 10.3 -> 12.0e9 env-steps/s; at one wave per
//      SIMD the same code is no slower).  The compiler spills ~60 dwords per lane to scratch to get there, which the other
//      wave's issue slots cover; the action is an ordinary load at the top of the step (the other wave covers that too).
//      Same arithmetic, same bits.  The engine picks it when the batch has more waves than SIMDs and no fused actor.
template <class Lane, typename T, int POLICY, int WAVES = 1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
void env_rollout_kernel(EnvParams<T> P, PolicyParams pol, int32_t steps, const float *actions, StepIO io0, float *actions_out) {
  constexpr bool kDatd3 = POLICY == ARMENV_POLICY_DATD3;     // four f16x3 passes per step; tables and ring re-staged per net (datd3_forward_wg)
  constexpr bool kActor = POLICY == ARMENV_POLICY_ACTOR || POLICY == ARMENV_POLICY_ACTOR_F16X3 || kDatd3;
  constexpr bool kRing = POLICY == ARMENV_POLICY_ACTOR_F16X3 || kDatd3;
  static_assert(WAVES == 1 || (WAVES == 2 && !kActor), "the fused actors need the whole register file");
  constexpr bool kPrefetch = POLICY == ARMENV_POLICY_EXTERNAL && WAVES == 1;
  __shared__ float4 w1_lds[kActor ? (kRing ? ACTOR_W1_LDS_FLOATS_H : ACTOR_W1_LDS_FLOATS) / 4 : 1];
  __shared__ uint4 w2_ring[kRing ? ACTOR_RING_UINT4 : 1];
  int nw = 4;   // live waves of this workgroup (the last one may be ragged; num_envs is a multiple of 64)
  if constexpr (kDatd3) {
    const int64_t left = P.n - (int64_t)blockIdx.x * blockDim.x;
    nw = (int)(((left < (int64_t)blockDim.x ? left : (int64_t)blockDim.x) + 63) >> 6);
  } else if constexpr (kActor) {   // layer-1 / layer-3 tables staged once per launch
    actor_stage_w1(pol.actor.W1P, w1_lds, pol.actor.B2W3, Lane::kObs);
    if constexpr (POLICY == ARMENV_POLICY_ACTOR_F16X3) {   // W2 streams through the ring every step
      actor_stage_w1h(pol.actor.W1P, w1_lds, Lane::kObs);
      const int64_t left = P.n - (int64_t)blockIdx.x * blockDim.x;
      nw = (int)(((left < (int64_t)blockDim.x ? left : (int64_t)blockDim.x) + 63) >> 6);
      if ((int)(threadIdx.x >> 6) < nw) actor_ring_init(pol.actor_h, w2_ring, nw);
    }
    __syncthreads();
  }
  // env of this lane (full or half-filled waves, EnvParams::half_waves; the fused actors' MFMA phases need full waves)
  const int64_t i = kActor ? (int64_t)blockIdx.x * blockDim.x + threadIdx.x : lane_env(P);
  if (i < 0) return;
  if (i >= P.n) return;
  const int64_t n = P.n;
  constexpr int kObs = Lane::kObs;
  Lane L;
  L.load(P, i);
  uint32_t episode = (POLICY != ARMENV_POLICY_EXTERNAL) ? P.episode[i] : 0u;
  if constexpr (kActor) L.refresh_obs(P);
  float an[3] = {0.f, 0.f, 0.f};
  if constexpr (kPrefetch) {
    an[0] = actions[3 * i]; an[1] = actions[3 * i + 1]; an[2] = actions[3 * i + 2];
    // settle this load before the loop: otherwise the loop header inherits a pending load on these registers from the
    // entry edge and hipcc puts an in-order vmcnt wait at the top of EVERY step, which also waits for the previous
    // step's stores
    asm volatile("" ::"v"(an[0]), "v"(an[1]), "v"(an[2]));
  }
  ActionPrefetch an_next{0.f, 0.f, 0.f};
  [[maybe_unused]] uint32_t w_trips = 0;
  // the link frames travel from step to step in registers -- except across a fused actor, which needs the whole register file
  // between two env steps (the lanes then recompute the frame at the top of every step)
  constexpr bool kCarry = !kActor;
  // straight-line IK trips ahead of the trip loop (armenv_kin.h ik_lockstep); none beside a fused actor (register pressure)
  constexpr int kPeel = kActor ? 0 : ARMENV_ROLLOUT_PEEL;
  if constexpr (kCarry) L.make_frame(P);
  for (int32_t t = 0; t < steps; ++t) {
    T a[3];
    if constexpr (kPrefetch) {
      a[0] = (T)an[0]; a[1] = (T)an[1]; a[2] = (T)an[2];
      // prefetch the next step's action (the last step re-reads its own); settled inside env_step, after the IK
      prefetch_issue(actions + ((int64_t)(t + 1 < steps ? t + 1 : t) * n + i) * 3, an_next);
    } else if constexpr (POLICY == ARMENV_POLICY_EXTERNAL) {
      const float *ap = actions + ((int64_t)t * n + i) * 3;
      a[0] = (T)ap[0]; a[1] = (T)ap[1]; a[2] = (T)ap[2];
    } else {
      float mu[3] = {0.f, 0.f, 0.f};
      if constexpr (POLICY == ARMENV_POLICY_ACTOR) {
        float s[kObs];
        L.policy_obs(s);
        actor_forward_wave<kObs>(pol.actor, w1_lds, s, mu);              // take_action, TD3_mlp.py:82-97
      } else if constexpr (POLICY == ARMENV_POLICY_ACTOR_F16X3) {
        float s[kObs];
        L.policy_obs(s);
        actor_forward_wg_f16x3<kObs>(pol.actor, pol.actor_h, w1_lds, w2_ring, nw, s, mu);
      } else if constexpr (kDatd3) {
        float s[kObs], q1, q2;
        int picked;
        L.policy_obs(s);
        datd3_forward_wg<kObs>(pol.datd3, pol.datd3_h, w1_lds, w2_ring, nw, s, mu, q1, q2, picked);    // take_action, DATD3_mlp.py:88-109
      }
      float nz[3];
      // the episode index of the stream is the number of resets so far minus one (the running episode)
      policy_noise(P.seed, P.env_id0 + (uint64_t)i, episode - 1u, (uint32_t)L.step, nz);
      static_for<0, 3>([&](auto KI) {
        constexpr int k = KI;
        float v = fmaf(nz[k], pol.sigma, mu[k]);                          // + N(0, sigma), main.py:116
        v = fminf(fmaxf(v, -pol.clip), pol.clip);                         // .clip(-bound, bound), main.py:117
        a[k] = (T)v;
        an[k] = v;
      });
    }
    StepIO io;
    io.action = nullptr;
    io.obs = io0.obs + (int64_t)t * n * kObs;
    io.reward = io0.reward + (int64_t)t * n;
    io.done = io0.done + (int64_t)t * n;
    io.success = io0.success + (int64_t)t * n;
    io.terminal_obs = io0.terminal_obs ? io0.terminal_obs + (int64_t)t * n * kObs : nullptr;
    if constexpr (Lane::kFence) io.updates = io0.updates ? io0.updates + (int64_t)t * n : nullptr; else io.updates = nullptr;
    if constexpr (kTipOf(Lane::kMode)) io.diag = io0.diag ? io0.diag + (int64_t)t * n * 4 : nullptr; else io.diag = nullptr;
    if (__builtin_expect(actions_out != nullptr, 0)) {
      float *ao = actions_out + ((int64_t)t * n + i) * 3;
      if constexpr (POLICY == ARMENV_POLICY_EXTERNAL) { ao[0] = (float)a[0]; ao[1] = (float)a[1]; ao[2] = (float)a[2]; }
      else { ao[0] = an[0]; ao[1] = an[1]; ao[2] = an[2]; }
    }
    const uint32_t before = L.n_done;
    int upd;
    if constexpr (kPrefetch) {
      upd = L.template env_step<kCarry, kPeel>(P, i, a, io, &an_next, &an);
    } else {
      upd = L.template env_step<kCarry, kPeel>(P, i, a, io);
    }
    if constexpr (Lane::kFence) w_trips += wave_max8((uint32_t)upd + 1u);   // a lane's trips: its updates + the exit trip
    else (void)upd;
    if constexpr (POLICY != ARMENV_POLICY_EXTERNAL) {
      if (L.n_done != before && P.auto_reset) episode += 1u;
    }
  }
  L.store(P, i);
  if constexpr (kRing) actor_ring_drain();
  if constexpr (Lane::kFence) flush_schedule(P.counters, w_trips, (uint32_t)steps);
  flush_env_steps(P.counters, i, (unsigned long long)n * (unsigned long long)steps);
}

// Lane-asynchronous rollout: the same `steps` env steps per env as env_rollout_kernel, but the lanes of a wave do not walk
// through a step together.  The wave's loop body is ONE IK TRIP; every lane carries its own step index t and trip state.
// A lane whose IK has stopped waits (masked off) until `ready_lanes` lanes of the wave are in that state or nobody is
// iterating any more; then all waiting lanes run their step's tail (contact / gripper, reward, stores to row t, in-place
// reset), take the next action and start the next step's trips, while the laggards carry on with theirs.
// Why: a lockstep wave pays sum_t max_lane trips(lane, t).  Where a few lanes need many more trips than the rest -- pick
// under random exploration: 0.85 % of the env-steps run Bullet's loop to its 20-iteration cap, so nearly every wave-step
// contains one (12.2 trips per wave-step for 4.45 per env-step, tests/tools/trip_stats.py); push: a 5th trip in one lane
// of most waves (5.5 for 4.1) -- the wave is held to its slowest lane at EVERY step.  Here a slow lane only delays itself:
// the wave pays ~max_lane sum_t trips (pick 6.5, push 4.4) plus one tail block per transition round.  ready_lanes trades
// the two: 64 is lockstep (one tail per step), 1 runs a tail block on nearly every trip.  `straggler_trips` > 0 replaces the
// count by a rule that knows what the slow lanes are (ArmEnvConfig.rollout_straggler_trips; tests/tools/async_policy_sim.py).
// Per-lane arithmetic, its order and every store are those of env_step: trajectories are bit-identical to the lockstep
// kernels and to armenv_step launches (tested).  Not used with the fused actors (their MFMA phases are wave-synchronous).
// WAVES: register budget as for env_rollout_kernel (2: <= 256 registers per lane, for batches with more waves than SIMDs).
template <class Lane, typename T, int POLICY, int WAVES = 1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
void env_rollout_async_kernel(EnvParams<T> P, PolicyParams pol, int32_t steps, const float *actions, StepIO io0,
                              float *actions_out, int32_t ready_lanes, int32_t straggler_trips) {
  static_assert(POLICY == ARMENV_POLICY_EXTERNAL || POLICY == ARMENV_POLICY_RANDOM, "no wave-synchronous policy phases here");
  const int64_t i = lane_env(P);   // env of this lane (full or half-filled waves, EnvParams::half_waves)
  if (i < 0) return;
  if (i >= P.n) return;
  const int64_t n = P.n;
  constexpr int kObs = Lane::kObs;
  using C = typename Lane::Chain;
  Lane L;
  L.load(P, i);
  uint32_t episode = (POLICY != ARMENV_POLICY_EXTERNAL) ? P.episode[i] : 0u;
  const T res2 = P.ik.residual * P.ik.residual;
  const bool small_steps = P.ik.max_dtheta <= T(0.7854);
  int32_t t = 0;            // this lane's step inside the launch
  bool ready = false;       // the step's IK has stopped; the lane waits for a transition round
  T tgt[3], diff2_prev = T(1e60);
  int updates = 0;
  float af[3] = {0.f, 0.f, 0.f};     // the action of step t (external policy: loaded by load_action ahead of its use)
  auto load_action = [&](int32_t tt) {
    if constexpr (POLICY == ARMENV_POLICY_EXTERNAL) {
      const float *ap = actions + ((int64_t)tt * n + i) * 3;
      af[0] = ap[0]; af[1] = ap[1]; af[2] = ap[2];
    }
  };
  auto begin_step = [&]() {
    T a[3];
    if constexpr (POLICY != ARMENV_POLICY_EXTERNAL) {
      float nz[3];
      policy_noise(P.seed, P.env_id0 + (uint64_t)i, episode - 1u, (uint32_t)L.step, nz);
      static_for<0, 3>([&](auto KI) {
        constexpr int k = KI;
        float v = nz[k] * pol.sigma;                                     // zero actor + N(0, sigma), main.py:116
        af[k] = fminf(fmaxf(v, -pol.clip), pol.clip);                     // .clip(-bound, bound), main.py:117
      });
    }
    a[0] = (T)af[0]; a[1] = (T)af[1]; a[2] = (T)af[2];
    if (actions_out) {
      float *ao = actions_out + ((int64_t)t * n + i) * 3;
      ao[0] = af[0]; ao[1] = af[1]; ao[2] = af[2];
    }
    L.template step_begin<true>(P, a, tgt);
    diff2_prev = T(1e60);
    updates = 0;
    ready = false;
  };
  load_action(0);
  L.make_frame(P);          // carried from step to step in registers from here on (ReachLane::S; pick recomputes it per step)
  begin_step();
  [[maybe_unused]] uint32_t w_trips = 0, w_rounds = 0;
  for (;;) {
    if constexpr (Lane::kFence) w_trips += __ballot(!ready && t < steps) != 0ull ? 1u : 0u;
    if (!ready && t < steps) ready = ik_trip<C, T, Lane::kMode>(P.chain, P.ik, L.q, tgt, L.S, L.cq, L.sq, diff2_prev, updates, res2, small_steps, L.minpiv);
    const unsigned long long rb = __ballot(ready), ib = __ballot(!ready && t < steps);
    // wave-uniform: a transition round.  Count rule: `ready_lanes` lanes wait.  Straggler rule (straggler_trips > 0): every lane
    // that has spent fewer than that many trips on its step waits -- lanes on their way to the iteration cap carry on, however
    // many they are, and the others stay in phase with each other
    const bool round = straggler_trips > 0 ? __ballot(!ready && t < steps && updates < straggler_trips) == 0ull : __popcll(rb) >= ready_lanes;
    if (round || ib == 0ull) {
      if constexpr (Lane::kFence) w_rounds += rb != 0ull ? 1u : 0u;
      if (ready) {
        // the next step's action is requested first: the step's tail (a few hundred instructions) covers most of the
        // load's latency before begin_step consumes it
        if (t + 1 < steps) load_action(t + 1);
        const bool lim_hit = ik_limits<C, T, Lane::kMode>(P.chain, P.ik, L.q, L.S, L.cq, L.sq);
        StepIO io;
        io.action = nullptr;
        io.obs = io0.obs + (int64_t)t * n * kObs;
        io.reward = io0.reward + (int64_t)t * n;
        io.done = io0.done + (int64_t)t * n;
        io.success = io0.success + (int64_t)t * n;
        io.terminal_obs = io0.terminal_obs ? io0.terminal_obs + (int64_t)t * n * kObs : nullptr;
        if constexpr (Lane::kFence) io.updates = io0.updates ? io0.updates + (int64_t)t * n : nullptr; else io.updates = nullptr;
    if constexpr (kTipOf(Lane::kMode)) io.diag = io0.diag ? io0.diag + (int64_t)t * n * 4 : nullptr; else io.diag = nullptr;
        const uint32_t before = L.n_done;
        L.template step_tail<true>(P, i, io, updates, lim_hit);
        if constexpr (POLICY != ARMENV_POLICY_EXTERNAL) {
          if (L.n_done != before && P.auto_reset) episode += 1u;
        }
        ready = false;
        ++t;
        if (t < steps) begin_step();
      }
    }
    if (__ballot(t < steps) == 0ull) break;
  }
  L.store(P, i);
  if constexpr (Lane::kFence) flush_schedule(P.counters, w_trips, w_rounds);
  flush_env_steps(P.counters, i, (unsigned long long)n * (unsigned long long)steps);
}

// p.getLinkState(body, 6)[4], [5]
template <class C, typename T>
__global__ __launch_bounds__(256) void fk_kernel(EnvParams<T> P, int64_t n, const double *q_in, double *pos, double *quat) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  T q[NJ];
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; q[j] = (T)q_in[7 * i + j]; });
  FKState<T> S;
  T cq[NJ], sq[NJ];
  sincos_all<T>(q, cq, sq);
  fk<C, T>(P.chain, cq, sq, S);
  static_for<0, 3>([&](auto KI) { constexpr int k = KI; pos[3 * i + k] = (double)S.p[k]; });
  if (quat) {
    T qc[4];
    quat_from_frame<T>(S.W, qc);
    static_for<0, 4>([&](auto KI) { constexpr int k = KI; quat[4 * i + k] = (double)qc[k]; });
  }
}

// p.calculateInverseKinematics(body, 6, pos, orn, jointDamping)
template <class C, typename T, int MODE = 0>
__global__ __launch_bounds__(256) void ik_kernel(EnvParams<T> P, int64_t n, const double *q_in, const double *tgt_in,
                                                 double *q_out, int32_t *iters) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  T q[NJ], tgt[3], a[3] = {T(0), T(0), T(0)};
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; q[j] = (T)q_in[7 * i + j]; });
  static_for<0, 3>([&](auto KI) { constexpr int k = KI; tgt[k] = (T)tgt_in[3 * i + k]; });
  FKState<T> S;
  const int it = ik_move<C, T, false, false, MODE>(P.chain, P.ik, q, tgt, a, P.dv, P.box_lo, P.box_hi, S);
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; q_out[7 * i + j] = (double)q[j]; });
  if (iters) iters[i] = it;
}

// Logging summary without a host round trip: per-lane reach distance |FK(q) - goal| (reach) or cube-target distance
// (push, pick) and the last finished episode's return / length / success, reduced across the wavefront with lane shuffles
// into one row per wave, rows[i >> 6][8] =
//   [sum distance, max distance, sum last_return, sum last_len, sum last_success, envs counted, 0, 0];
// summary_reduce_kernel folds the rows into out[8] in a fixed order (no atomics: same-address atomics serialise at 12 ns
// each across the chip, and a sum of doubles accumulated by atomics depends on the order the waves happen to finish in).
template <class Lane, typename T>
__global__ __launch_bounds__(256) void env_summary_kernel(EnvParams<T> P, double *rows) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if ((i & ~(int64_t)63) >= P.n) return;   // whole wave past the end
  const bool live = i < P.n;
  const int64_t ic = live ? i : P.n - 1;
  const double dist = Lane::summary_distance(P, ic);
  double v[5] = {live ? dist : 0.0, live ? (double)P.last_return[ic] : 0.0, live ? (double)P.last_len[ic] : 0.0,
                 live ? (double)P.last_success[ic] : 0.0, live ? 1.0 : 0.0};
  double mx = live ? dist : 0.0;
  for (int o = 32; o; o >>= 1) {
    static_for<0, 5>([&](auto KI) { constexpr int k = KI; v[k] += __shfl_xor(v[k], o); });
    mx = fmax(mx, __shfl_xor(mx, o));
  }
  if ((threadIdx.x & 63) == 0) {
    double *r = rows + 8 * (i >> 6);
    r[0] = v[0]; r[1] = mx; r[2] = v[1]; r[3] = v[2]; r[4] = v[3]; r[5] = v[4]; r[6] = 0.0; r[7] = 0.0;
  }
}

static __global__ __launch_bounds__(256) void summary_reduce_kernel(const double *rows, int64_t n_rows, double *out) {
  __shared__ double part[32][8];
  const int col = threadIdx.x & 7, slot = threadIdx.x >> 3;
  double acc = 0.0;
  for (int64_t r = slot; r < n_rows; r += 32) acc = (col == 1) ? fmax(acc, rows[8 * r + col]) : acc + rows[8 * r + col];
  part[slot][col] = acc;
  __syncthreads();
  if (threadIdx.x < 8) {
    double a = part[0][col];
    for (int k = 1; k < 32; ++k) a = (col == 1) ? fmax(a, part[k][col]) : a + part[k][col];
    out[col] = a;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void get_state_kernel(EnvParams<T> P, double *q, float *goal, int32_t *step,
                                                        uint32_t *episode, double *ep_return, double *aux, double *trig,
                                                        int aux_rows, int aux_dim) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n) return;
  if (trig) static_for<0, 2 * NJ>([&](auto JI) { constexpr int j = JI; trig[2 * NJ * i + j] = (double)P.trig[(int64_t)j * P.n + i]; });
  if (aux && P.aux) {
    for (int k = 0; k < aux_rows; ++k) aux[(int64_t)aux_dim * i + k] = (double)P.aux[(int64_t)k * P.n + i];
    for (int k = aux_rows; k < aux_dim; ++k) aux[(int64_t)aux_dim * i + k] = 0.0;
  }
  if (q) static_for<0, NJ>([&](auto JI) { constexpr int j = JI; q[7 * i + j] = (double)P.q[(int64_t)j * P.n + i]; });
  if (goal) static_for<0, 3>([&](auto KI) { constexpr int k = KI; goal[3 * i + k] = P.goal[(int64_t)k * P.n + i]; });
  if (step) step[i] = P.step[i];
  if (episode) episode[i] = P.episode[i];
  if (ep_return) ep_return[i] = (double)P.ep_return[i];
}

template <typename T>
__global__ __launch_bounds__(256) void set_state_kernel(EnvParams<T> P, const double *q, const float *goal,
                                                        const int32_t *step, const uint32_t *episode,
                                                        const double *ep_return, const double *aux, const double *trig,
                                                        int aux_rows, int aux_dim) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n) return;
  if (aux && P.aux)
    for (int k = 0; k < aux_rows; ++k) P.aux[(int64_t)k * P.n + i] = (T)aux[(int64_t)aux_dim * i + k];
  if (q) {   // resetJointState: without `trig` the carried (cos q, sin q) restart from the new angles
    T qq[NJ], c_[NJ], s_[NJ];
    static_for<0, NJ>([&](auto JI) { constexpr int j = JI; qq[j] = (T)q[7 * i + j]; P.q[(int64_t)j * P.n + i] = qq[j]; });
    if (!trig) {
      sincos_all<T>(qq, c_, s_);
      static_for<0, NJ>([&](auto JI) { constexpr int j = JI; P.trig[(int64_t)j * P.n + i] = c_[j]; P.trig[(int64_t)(NJ + j) * P.n + i] = s_[j]; });
    }
  }
  // a checkpoint's own pair (armenv_get_state): the restored env continues the uninterrupted trajectory bit for bit
  if (trig) static_for<0, 2 * NJ>([&](auto JI) { constexpr int j = JI; P.trig[(int64_t)j * P.n + i] = (T)trig[2 * NJ * i + j]; });
  if (goal) static_for<0, 3>([&](auto KI) { constexpr int k = KI; P.goal[(int64_t)k * P.n + i] = goal[3 * i + k]; });
  if (step) P.step[i] = step[i];
  if (episode) P.episode[i] = episode[i];
  if (ep_return) P.ep_return[i] = (T)ep_return[i];
}

template <typename T>
__global__ __launch_bounds__(256) void episode_returns_f32_kernel(EnvParams<T> P, float *last_return) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < P.n) last_return[i] = (float)P.last_return[i];
}

template <typename T>
__global__ __launch_bounds__(256) void episode_stats_kernel(EnvParams<T> P, double *last_return, int32_t *last_len,
                                                            uint8_t *last_success) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n) return;
  if (last_return) last_return[i] = (double)P.last_return[i];
  if (last_len) last_len[i] = P.last_len[i];
  if (last_success) last_success[i] = P.last_success[i];
}

