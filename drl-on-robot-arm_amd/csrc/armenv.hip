// armenv.hip -- gfx950 kernels and the C ABI (include/armenv.h) of the batched arm-env engine.
//
// Data layout in HBM (N envs, T = f64 or f32 by ArmEnvConfig.precision), struct-of-arrays with the
// env index fastest so that a wave's 64 lanes touch 64 consecutive elements of every array:
//   q[7][N] T | ep_return[N] T | last_return[N] T | goal[3][N] f32 | step[N] i32 | episode[N] u32 |
//   last_len[N] i32 | last_success[N] u8 | counters[4] u64
// Caller-facing buffers keep the reference's array-of-struct shapes (action [N][3], obs [N][6]);
// a wave still reads/writes one contiguous 768 B / 1536 B span of them.
//
// One env per lane, no LDS, no cross-lane traffic on the step path; the chain constants are
// compile-time (built-in arms) or wave-uniform kernel arguments (generic chains).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <string>

#include "../../include/armenv.h"
#include "armenv_kin.h"
#include "armenv_actor.h"
#include "armenv_replay.h"

using namespace armenv;

// ------------------------------------------------------------------------------------------------
// device side
// ------------------------------------------------------------------------------------------------

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
  unsigned long long *counters;
  T *aux;  // push only: [7][N] = cube xyz, target xyz, d_last
  int64_t n;
  // task constants
  T dv;
  T reach_dis;
  int32_t max_steps;
  int32_t auto_reset;
  T box_lo[3];
  T box_hi[3];
  double goal_lo[3];
  double goal_hi[3];
  T q_init[NJ];
  T p_init[3];  // FK(q_init), computed on the device at create time
  uint64_t seed;
  uint64_t env_id0;
  // push task (rl_push_env.py): simplified pusher model + reward constants
  T push_success_dis, push_cube_half, push_eef_radius;
  double push_rest_z, push_place_min, push_place_max;
  IKParams<T> ik;
  ChainDev<T> chain;
};

struct StepIO {
  const float *action;
  float *obs;
  float *reward;
  uint8_t *done;
  uint8_t *success;
  float *terminal_obs;
};

// goal ~ U(box): a + (b - a) * u per axis as random.uniform does (rl_reach_env.py:180-182), then the
// f32 cast of :213-215.  Always f64 arithmetic so that both precisions draw identical goals.
template <typename T>
AE_DEV void sample_goal(const EnvParams<T> &P, int64_t i, uint32_t episode, float (&g)[3]) {
  double u0, u1, u2, u3;
  philox_pair(P.seed, P.env_id0 + (uint64_t)i, episode, 0u, u0, u1);
  philox_pair(P.seed, P.env_id0 + (uint64_t)i, episode, 1u, u2, u3);
  g[0] = (float)(P.goal_lo[0] + (P.goal_hi[0] - P.goal_lo[0]) * u0);
  g[1] = (float)(P.goal_lo[1] + (P.goal_hi[1] - P.goal_lo[1]) * u1);
  g[2] = (float)(P.goal_lo[2] + (P.goal_hi[2] - P.goal_lo[2]) * u2);
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

// FK(q_init) once per handle, with the same device code the step uses.
template <class C, typename T>
__global__ void init_consts_kernel(EnvParams<T> P, T *out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  T q[NJ];
  static_for<0, NJ>([&](auto II) { constexpr int i = II; q[i] = P.q_init[i]; });
  FKState<T> S;
  T cq[NJ], sq[NJ];
  sincos_all<T>(q, cq, sq);
  fk<C, T>(P.chain, cq, sq, S);
  out[0] = S.p[0]; out[1] = S.p[1]; out[2] = S.p[2];
}

// RLReachEnv.reset (rl_reach_env.py:132-217) for masked envs.
template <typename T>
__global__ __launch_bounds__(256) void reach_reset_kernel(EnvParams<T> P, const uint8_t *mask, const float *goal_in,
                                                          float *obs) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n) return;
  if (mask && !mask[i]) return;
  float g[3];
  if (goal_in) {
    g[0] = goal_in[3 * i]; g[1] = goal_in[3 * i + 1]; g[2] = goal_in[3 * i + 2];
  } else {
    const uint32_t ep = P.episode[i];
    sample_goal(P, i, ep, g);
    P.episode[i] = ep + 1u;
  }
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; P.q[(int64_t)j * P.n + i] = P.q_init[j]; });
  static_for<0, 3>([&](auto KI) { constexpr int k = KI; P.goal[(int64_t)k * P.n + i] = g[k]; });
  P.step[i] = 0;
  P.ep_return[i] = T(0);
  if (obs) store_obs6<T>(obs, i, P.p_init, g);
}

// Optional per-wave timeline (make timeline; csrc/exp/run_timeline.py): wall-clock stamps at kernel entry, after
// the state loads, after the IK loop and at exit, plus IK update count and placement.  Off in the product build.
#ifdef ARMENV_TIMELINE
__device__ unsigned long long *g_timeline;
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
};

// TD3_MLP.take_action (/root/reference/algo/TD3/TD3_mlp.py:82-97) for n states; MODE 0 exact f32, 1 f16x3.
template <int IN, int MODE>
__global__ __launch_bounds__(256) void actor_kernel(ActorParams A, ActorParamsH H, int64_t n, const float *states, float *actions) {
  __shared__ float4 w1_lds[MODE == 1 ? ACTOR_W1_LDS_FLOATS / 4 : 1];
  if constexpr (MODE == 1) {
    actor_stage_w1(A.W1P, w1_lds);
    __syncthreads();
  }
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // grid covers n rounded up to 256
  const int64_t ic = i < n ? i : n - 1;
  float s[IN], a[3];
  static_for<0, IN>([&](auto DI) { constexpr int d = DI; s[d] = states[ic * IN + d]; });
  if constexpr (MODE == 1) actor_forward_wave_f16x3<IN>(A, H, w1_lds, s, a);
  else actor_forward_wave<IN>(A, s, a);
  if (i < n) { actions[3 * i] = a[0]; actions[3 * i + 1] = a[1]; actions[3 * i + 2] = a[2]; }
}

// torch Linear layouts ([out][in]) -> the operand layouts of armenv_actor.h
__global__ void actor_pack_kernel(const float *W1, const float *b1, const float *W2, const float *b2, const float *W3,
                                  int in_dim, float *W1P, float *W2P, float *B2W3, _Float16 *W2H, _Float16 *W2L) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < ACTOR_HID * ACTOR_HID) {   // f16 hi/lo split of W2 in the 32x32x16 A-operand order
    const int j = t & 7, lane = (t >> 3) & 63, nt = (t >> 9) & 3, part = (t >> 11) & 1, ks = t >> 12;
    const float x = W2[(32 * (4 * part + nt) + (lane & 31)) * ACTOR_HID + 16 * ks + 8 * (lane >> 5) + j];
    const _Float16 hi = (_Float16)x;
    W2H[t] = hi;
    W2L[t] = (_Float16)(x - (float)hi);
  }
  if (t < ACTOR_HID * 12) {   // [kk][row k = 2kk + r][12] == [k][12]
    const int k = t / 12, j = t % 12;
    W1P[t] = j < in_dim ? W1[k * in_dim + j] : (j == 11 ? b1[k] : 0.f);
  }
  if (t < ACTOR_HID * ACTOR_HID) {
    const int c = t & 3, l32 = (t >> 2) & 31, part = (t >> 7) & 1, k = t >> 8;
    W2P[t] = W2[(32 * (4 * part + c) + l32) * ACTOR_HID + k];
  }
  if (t < ACTOR_HID * 4) {
    const int n = t >> 2, c = t & 3;
    B2W3[t] = c == 0 ? b2[n] : W3[(c - 1) * ACTOR_HID + n];
  }
}

// Three N(0,1) draws for (env, episode, step): Philox block 0x80000000|step of the env's stream (reset draws use
// blocks < 2^31), Box-Muller in f32 on u = (w + 1) * 2^-32 in (0, 1].
AE_DEV void policy_noise(uint64_t seed, uint64_t env_id, uint32_t episode, uint32_t step, float (&nz)[3]) {
  uint32_t c[4] = {(uint32_t)env_id, (uint32_t)(env_id >> 32), episode, 0x80000000u | step};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  const float k = 2.3283064365386963e-10f;  // 2^-32
  const float u0 = ((float)c[0] + 1.0f) * k, u1 = (float)c[1] * k, u2 = ((float)c[2] + 1.0f) * k, u3 = (float)c[3] * k;
  const float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
  float s0, c0, s1, c1;
  sincosf(6.283185307179586f * u1, &s0, &c0);
  sincosf(6.283185307179586f * u3, &s1, &c1);
  nz[0] = r0 * c0; nz[1] = r0 * s0; nz[2] = r1 * c1;
  (void)s1;
}

// Per-lane state of one reach env and the body of one env step.  The single-step kernel and the T-step rollout
// kernel both run this code, so a rollout is bit-identical to T step launches.
template <class C, typename T> struct ReachLane {
  using M = Mth<T>;
  static constexpr int kObs = 6;
  T q[NJ];
  float g[3];
  int32_t step;
  T ep_ret;
  uint32_t n_done = 0, n_succ = 0, n_bad = 0, n_upd = 0;   // flushed to the handle's counters once per launch

  AE_DEV void load(const EnvParams<T> &P, int64_t i) {
    const int64_t n = P.n;
    static_for<0, NJ>([&](auto JI) { constexpr int j = JI; q[j] = P.q[(int64_t)j * n + i]; });
    static_for<0, 3>([&](auto KI) { constexpr int k = KI; g[k] = P.goal[(int64_t)k * n + i]; });
    step = P.step[i];
    ep_ret = P.ep_return[i];
  }

  AE_DEV void store(const EnvParams<T> &P, int64_t i) {
    const int64_t n = P.n;
    static_for<0, NJ>([&](auto JI) { constexpr int j = JI; P.q[(int64_t)j * n + i] = q[j]; });
    P.step[i] = step;
    P.ep_return[i] = ep_ret;
    if (n_done) atomicAdd(&P.counters[0], (unsigned long long)n_done);
    if (n_succ) atomicAdd(&P.counters[1], (unsigned long long)n_succ);
    if (n_bad) atomicAdd(&P.counters[3], (unsigned long long)n_bad);
    // one add per wave for the IK-update total (wave reduction first)
    if (__ballot(1) == ~0ull) {
      unsigned u = n_upd;
      for (int o = 32; o; o >>= 1) u += __shfl_xor(u, o);
      if ((threadIdx.x & 63) == 0) atomicAdd(&P.counters[4], (unsigned long long)u);
    } else if (n_upd) {   // ragged last wave
      atomicAdd(&P.counters[4], (unsigned long long)n_upd);
    }
  }

  // RLReachEnv.step + _reward (rl_reach_env.py:219-319) with action a; writes row i of the caller's buffers.
  // Returns the number of IK updates.
  AE_DEV int env_step(const EnvParams<T> &P, int64_t i, const T (&a)[3], const StepIO &io) {
    const int64_t n = P.n;
    FKState<T> S;
    T tgt[3];
    const int updates = ik_move<C, T, true>(P.chain, P.ik, q, tgt, a, P.dv, P.box_lo, P.box_hi, S);  // :237-257

    n_upd += (uint32_t)updates;
    step += 1;                                                                    // :264
    const T dx = S.p[0] - (T)g[0], dy = S.p[1] - (T)g[1], dz = S.p[2] - (T)g[2];
    const T dist = M::sqrt(M::fma(dx, dx, M::fma(dy, dy, dz * dz)));              // :281
    T reward;
    bool done, succ;
    if (step > P.max_steps) { reward = -dist * T(10); done = true; succ = false; }          // :299-301
    else if (dist < P.reach_dis) { reward = T(0); done = true; succ = true; }               // :303-306
    else { reward = -dist * T(10); done = false; succ = false; }                            // :307-309
    ep_ret += reward;

    bool finite = true;
    static_for<0, NJ>([&](auto JI) { constexpr int j = JI; finite = finite && M::finite(q[j]); });
    if (!finite) n_bad += 1;

    io.reward[i] = (float)reward;
    io.done[i] = done ? 1 : 0;
    io.success[i] = succ ? 1 : 0;
    if (io.terminal_obs) store_obs6<T>(io.terminal_obs, i, S.p, g);

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
      static_for<0, NJ>([&](auto JI) { constexpr int j = JI; q[j] = P.q_init[j]; });
      step = 0;
      ep_ret = T(0);
      store_obs6<T>(io.obs, i, P.p_init, g);
      cur_obs[0] = (float)P.p_init[0]; cur_obs[1] = (float)P.p_init[1]; cur_obs[2] = (float)P.p_init[2];
    } else {
      store_obs6<T>(io.obs, i, S.p, g);                                           // :319
      cur_obs[0] = (float)S.p[0]; cur_obs[1] = (float)S.p[1]; cur_obs[2] = (float)S.p[2];
    }
    return updates;
  }

  float cur_obs[3];   // eef part of the observation the policy sees next (goal part is g)
  // eef of the current state, for the first policy call of a launch (later ones reuse the step's exit FK)
  AE_DEV void refresh_obs(const EnvParams<T> &P) {
    FKState<T> S;
    T cq[NJ], sq[NJ];
    sincos_all<T>(q, cq, sq);
    fk<C, T>(P.chain, cq, sq, S);
    cur_obs[0] = (float)S.p[0]; cur_obs[1] = (float)S.p[1]; cur_obs[2] = (float)S.p[2];
  }
  AE_DEV void policy_obs(float (&s)[6]) const {
    s[0] = cur_obs[0]; s[1] = cur_obs[1]; s[2] = cur_obs[2]; s[3] = g[0]; s[4] = g[1]; s[5] = g[2];
  }
};

// ---- push task (/root/reference/envs/rl_push_env.py) ---------------------------------------------------------------
// Placement of cube and target: rejection sampling, <= 1000 tries, six draws per try (:195-214); f64 always.
template <typename T>
AE_DEV void push_sample(const EnvParams<T> &P, int64_t i, uint32_t episode, T (&cube)[3], T (&target)[3]) {
  double cx = 0, cy = 0, tx = 0, ty = 0;
  for (uint32_t t = 0; t < 1000u; ++t) {
    double u0, u1, u2, u3, u4, u5;
    philox_pair(P.seed, P.env_id0 + (uint64_t)i, episode, 3u * t + 0u, u0, u1);
    philox_pair(P.seed, P.env_id0 + (uint64_t)i, episode, 3u * t + 1u, u2, u3);
    philox_pair(P.seed, P.env_id0 + (uint64_t)i, episode, 3u * t + 2u, u4, u5);
    cx = P.goal_lo[0] + (P.goal_hi[0] - P.goal_lo[0]) * u0;
    cy = P.goal_lo[1] + (P.goal_hi[1] - P.goal_lo[1]) * u1;
    tx = P.goal_lo[0] + (P.goal_hi[0] - P.goal_lo[0]) * u3;
    ty = P.goal_lo[1] + (P.goal_hi[1] - P.goal_lo[1]) * u4;
    const double dx = cx - tx, dy = cy - ty;
    const double d = ::sqrt(::fma(dx, dx, dy * dy));   // both rest at the same z
    if (d >= P.push_place_min && d <= P.push_place_max) break;
    (void)u2; (void)u5;
  }
  cube[0] = (T)cx; cube[1] = (T)cy; cube[2] = (T)P.push_rest_z;
  target[0] = (T)tx; target[1] = (T)ty; target[2] = (T)P.push_rest_z;
}

template <class C, typename T> struct PushLane {
  using M = Mth<T>;
  static constexpr int kObs = 9;
  T q[NJ];
  T cube[3], target[3], d_last;
  int32_t step;
  T ep_ret;
  uint32_t n_done = 0, n_succ = 0, n_bad = 0, n_upd = 0;
  float cur_obs[3];
  AE_DEV void refresh_obs(const EnvParams<T> &P) {
    FKState<T> S;
    T cq[NJ], sq[NJ];
    sincos_all<T>(q, cq, sq);
    fk<C, T>(P.chain, cq, sq, S);
    cur_obs[0] = (float)S.p[0]; cur_obs[1] = (float)S.p[1]; cur_obs[2] = (float)S.p[2];
  }
  AE_DEV void policy_obs(float (&s)[9]) const {
    static_for<0, 3>([&](auto KI) { constexpr int k = KI; s[k] = cur_obs[k]; s[3 + k] = (float)cube[k]; s[6 + k] = (float)target[k]; });
  }

  AE_DEV T dist_ct() const {
    const T x = cube[0] - target[0], y = cube[1] - target[1], z = cube[2] - target[2];
    return M::sqrt(M::fma(x, x, M::fma(y, y, z * z)));
  }

  AE_DEV void load(const EnvParams<T> &P, int64_t i) {
    const int64_t n = P.n;
    static_for<0, NJ>([&](auto JI) { constexpr int j = JI; q[j] = P.q[(int64_t)j * n + i]; });
    static_for<0, 3>([&](auto KI) { constexpr int k = KI; cube[k] = P.aux[(int64_t)k * n + i]; target[k] = P.aux[(int64_t)(3 + k) * n + i]; });
    d_last = P.aux[(int64_t)6 * n + i];
    step = P.step[i];
    ep_ret = P.ep_return[i];
  }

  AE_DEV void store(const EnvParams<T> &P, int64_t i) {
    const int64_t n = P.n;
    static_for<0, NJ>([&](auto JI) { constexpr int j = JI; P.q[(int64_t)j * n + i] = q[j]; });
    static_for<0, 3>([&](auto KI) { constexpr int k = KI; P.aux[(int64_t)k * n + i] = cube[k]; P.aux[(int64_t)(3 + k) * n + i] = target[k]; });
    P.aux[(int64_t)6 * n + i] = d_last;
    P.step[i] = step;
    P.ep_return[i] = ep_ret;
    if (n_done) atomicAdd(&P.counters[0], (unsigned long long)n_done);
    if (n_succ) atomicAdd(&P.counters[1], (unsigned long long)n_succ);
    if (n_bad) atomicAdd(&P.counters[3], (unsigned long long)n_bad);
    if (n_upd) atomicAdd(&P.counters[4], (unsigned long long)n_upd);
  }

  // stepSimulation (:349), simplified: sphere (tool, radius r at the eef) vs axis-aligned box (cube, half-size h)
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

  // RLPushEnv.step + _reward (rl_push_env.py:310-440)
  AE_DEV int env_step(const EnvParams<T> &P, int64_t i, const T (&a)[3], const StepIO &io) {
    FKState<T> S;
    T tgt[3];
    T p0[3];
    const int updates = ik_move<C, T, true>(P.chain, P.ik, q, tgt, a, P.dv, P.box_lo, P.box_hi, S, &p0);  // :322-347
    n_upd += (uint32_t)updates;
    contact(P, p0, S.p);                                                          // :349
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

    bool finite = true;
    static_for<0, NJ>([&](auto JI) { constexpr int j = JI; finite = finite && M::finite(q[j]); });
    if (!finite) n_bad += 1;

    io.reward[i] = (float)reward;
    io.done[i] = done ? 1 : 0;
    io.success[i] = succ ? 1 : 0;
    if (io.terminal_obs) store_obs9<T>(io.terminal_obs, i, S.p, cube, target);
    if (done) {
      P.last_return[i] = ep_ret;
      P.last_len[i] = step;
      P.last_success[i] = succ ? 1 : 0;
      n_done += 1;
      if (succ) n_succ += 1;
    }
    if (done && P.auto_reset) {
      const uint32_t ep = P.episode[i];
      push_sample(P, i, ep, cube, target);
      P.episode[i] = ep + 1u;
      d_last = dist_ct();                                                         // :243-245
      static_for<0, NJ>([&](auto JI) { constexpr int j = JI; q[j] = P.q_init[j]; });
      step = 0;
      ep_ret = T(0);
      store_obs9<T>(io.obs, i, P.p_init, cube, target);
      cur_obs[0] = (float)P.p_init[0]; cur_obs[1] = (float)P.p_init[1]; cur_obs[2] = (float)P.p_init[2];
    } else {
      store_obs9<T>(io.obs, i, S.p, cube, target);                               // :308
      cur_obs[0] = (float)S.p[0]; cur_obs[1] = (float)S.p[1]; cur_obs[2] = (float)S.p[2];
    }
    return updates;
  }
};

// RLPushEnv.reset (rl_push_env.py:145-256) for masked envs; goal_in f32 [N][6] = cube xyz, target xyz.
template <typename T>
__global__ __launch_bounds__(256) void push_reset_kernel(EnvParams<T> P, const uint8_t *mask, const float *goal_in, float *obs) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n) return;
  if (mask && !mask[i]) return;
  const int64_t n = P.n;
  T cube[3], target[3];
  if (goal_in) {
    static_for<0, 3>([&](auto KI) { constexpr int k = KI; cube[k] = (T)goal_in[6 * i + k]; target[k] = (T)goal_in[6 * i + 3 + k]; });
  } else {
    const uint32_t ep = P.episode[i];
    push_sample(P, i, ep, cube, target);
    P.episode[i] = ep + 1u;
  }
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; P.q[(int64_t)j * n + i] = P.q_init[j]; });
  static_for<0, 3>([&](auto KI) { constexpr int k = KI; P.aux[(int64_t)k * n + i] = cube[k]; P.aux[(int64_t)(3 + k) * n + i] = target[k]; });
  const T x = cube[0] - target[0], y = cube[1] - target[1], z = cube[2] - target[2];
  P.aux[(int64_t)6 * n + i] = Mth<T>::sqrt(Mth<T>::fma(x, x, Mth<T>::fma(y, y, z * z)));
  P.step[i] = 0;
  P.ep_return[i] = T(0);
  if (obs) store_obs9<T>(obs, i, P.p_init, cube, target);
}

// One env step per launch: load state -> FK -> target = clip(p + dv a) -> DLS IK loop -> FK -> (push: contact) ->
// reward / done -> obs pack -> episode accounting -> optional in-place reset -> store state.
// Lane = ReachLane (rl_reach_env.py:219-319) or PushLane (rl_push_env.py:310-440).
template <class Lane, typename T>
__global__ __launch_bounds__(256) void env_step_kernel(EnvParams<T> P, StepIO io) {
  TL_STAMP(tl0);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n) return;
  Lane L;
  L.load(P, i);
  T a[3];
  static_for<0, 3>([&](auto KI) { constexpr int k = KI; a[k] = (T)io.action[3 * i + k]; });
#ifdef ARMENV_TIMELINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  TL_STAMP(tl1);
  const int updates = L.env_step(P, i, a, io);
  (void)updates;
  TL_STAMP(tl2);
  L.store(P, i);
  if (i == 0) atomicAdd(&P.counters[2], (unsigned long long)P.n);
#ifdef ARMENV_TIMELINE
  {
    TL_STAMP(tl3);
    int mx = updates;
    for (int o = 32; o; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
    const unsigned long long dn = __ballot(L.n_done != 0);
    if ((threadIdx.x & 63) == 0 && g_timeline) {
      unsigned long long *r = g_timeline + 8 * (i >> 6);
      unsigned xcc, hw;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      r[0] = tl0; r[1] = tl1; r[2] = tl2; r[3] = tl3; r[4] = mx; r[5] = __popcll(dn); r[6] = xcc; r[7] = hw;
    }
  }
#endif
}

// The rollout inner loop of /root/reference/main.py:108-128 for `steps` consecutive env steps in ONE launch: the env
// state stays in registers, every step's outputs go to row t of [steps][N][...] buffers, and the action of step t
// is either read from actions[t] (external policy, identical to `steps` calls of env_step_kernel) or produced
// in-kernel by the fused exploration policy.  Because lanes never synchronise, a lane that needs extra IK updates
// in one step does not hold the other envs back for the rest of the launch: per-step cost approaches the MEAN
// update count instead of the per-launch MAX.
// POLICY is a compile-time copy of pol.kind: the external-action variant carries no actor / noise code, which keeps it
// free of the register spills the fused-actor variant's 128 MFMA accumulators would otherwise force on it.
template <class Lane, typename T, int POLICY>
__global__ __launch_bounds__(256) void env_rollout_kernel(EnvParams<T> P, PolicyParams pol, int32_t steps,
                                                          const float *actions, StepIO io0, float *actions_out) {
  __shared__ float4 w1_lds[POLICY == ARMENV_POLICY_ACTOR_F16X3 ? ACTOR_W1_LDS_FLOATS / 4 : 1];
  if constexpr (POLICY == ARMENV_POLICY_ACTOR_F16X3) {   // the kernel's only LDS use and only barrier: W1, once
    actor_stage_w1(pol.actor.W1P, w1_lds);
    __syncthreads();
  }
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n) return;
  const int64_t n = P.n;
  constexpr int kObs = Lane::kObs;
  Lane L;
  L.load(P, i);
  uint32_t episode = (POLICY != ARMENV_POLICY_EXTERNAL) ? P.episode[i] : 0u;
  if constexpr (POLICY == ARMENV_POLICY_ACTOR || POLICY == ARMENV_POLICY_ACTOR_F16X3) L.refresh_obs(P);
  float an[3] = {0.f, 0.f, 0.f};
  if constexpr (POLICY == ARMENV_POLICY_EXTERNAL) { an[0] = actions[3 * i]; an[1] = actions[3 * i + 1]; an[2] = actions[3 * i + 2]; }
  for (int32_t t = 0; t < steps; ++t) {
    T a[3];
    if constexpr (POLICY == ARMENV_POLICY_EXTERNAL) {
      a[0] = (T)an[0]; a[1] = (T)an[1]; a[2] = (T)an[2];
      if (t + 1 < steps) {   // prefetch the next step's action; its latency hides under this step's IK
        const float *nx = actions + ((int64_t)(t + 1) * n + i) * 3;
        an[0] = nx[0]; an[1] = nx[1]; an[2] = nx[2];
      }
    } else {
      float mu[3] = {0.f, 0.f, 0.f};
      if constexpr (POLICY == ARMENV_POLICY_ACTOR) {
        float s[kObs];
        L.policy_obs(s);
        actor_forward_wave<kObs>(pol.actor, s, mu);                      // take_action, TD3_mlp.py:82-97
      } else if constexpr (POLICY == ARMENV_POLICY_ACTOR_F16X3) {
        float s[kObs];
        L.policy_obs(s);
        actor_forward_wave_f16x3<kObs>(pol.actor, pol.actor_h, w1_lds, s, mu);
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
    if (actions_out) {
      float *ao = actions_out + ((int64_t)t * n + i) * 3;
      if constexpr (POLICY == ARMENV_POLICY_EXTERNAL) { ao[0] = (float)a[0]; ao[1] = (float)a[1]; ao[2] = (float)a[2]; }
      else { ao[0] = an[0]; ao[1] = an[1]; ao[2] = an[2]; }
    }
    const uint32_t before = L.n_done;
    L.env_step(P, i, a, io);
    if constexpr (POLICY != ARMENV_POLICY_EXTERNAL) {
      if (L.n_done != before && P.auto_reset) episode += 1u;
    }
  }
  L.store(P, i);
  if (i == 0) atomicAdd(&P.counters[2], (unsigned long long)n * (unsigned long long)steps);
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
template <class C, typename T>
__global__ __launch_bounds__(256) void ik_kernel(EnvParams<T> P, int64_t n, const double *q_in, const double *tgt_in,
                                                 double *q_out, int32_t *iters) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  T q[NJ], tgt[3], a[3] = {T(0), T(0), T(0)};
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; q[j] = (T)q_in[7 * i + j]; });
  static_for<0, 3>([&](auto KI) { constexpr int k = KI; tgt[k] = (T)tgt_in[3 * i + k]; });
  FKState<T> S;
  const int it = ik_move<C, T, false>(P.chain, P.ik, q, tgt, a, P.dv, P.box_lo, P.box_hi, S);
  static_for<0, NJ>([&](auto JI) { constexpr int j = JI; q_out[7 * i + j] = (double)q[j]; });
  if (iters) iters[i] = it;
}

template <typename T>
__global__ __launch_bounds__(256) void get_state_kernel(EnvParams<T> P, double *q, float *goal, int32_t *step,
                                                        uint32_t *episode, double *ep_return, double *aux) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n) return;
  if (aux && P.aux) {
    static_for<0, 7>([&](auto KI) { constexpr int k = KI; aux[8 * i + k] = (double)P.aux[(int64_t)k * P.n + i]; });
    aux[8 * i + 7] = 0.0;
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
                                                        const double *ep_return, const double *aux) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n) return;
  if (aux && P.aux) static_for<0, 7>([&](auto KI) { constexpr int k = KI; P.aux[(int64_t)k * P.n + i] = (T)aux[8 * i + k]; });
  if (q) static_for<0, NJ>([&](auto JI) { constexpr int j = JI; P.q[(int64_t)j * P.n + i] = (T)q[7 * i + j]; });
  if (goal) static_for<0, 3>([&](auto KI) { constexpr int k = KI; P.goal[(int64_t)k * P.n + i] = goal[3 * i + k]; });
  if (step) P.step[i] = step[i];
  if (episode) P.episode[i] = episode[i];
  if (ep_return) P.ep_return[i] = (T)ep_return[i];
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

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------

static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                            \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess) return fail(ARMENV_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = (prev == dev) || (hipSetDevice(dev) == hipSuccess);
  }
  ~DeviceGuard() {
    int cur = -1;
    if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
  }
};

static void rpy_to_mat(const double rpy[3], double R[9]) {  // row-major, Rz(yaw) Ry(pitch) Rx(roll)
  const double cr = std::cos(rpy[0]), sr = std::sin(rpy[0]);
  const double cp = std::cos(rpy[1]), sp = std::sin(rpy[1]);
  const double cy = std::cos(rpy[2]), sy = std::sin(rpy[2]);
  R[0] = cy * cp; R[1] = cy * sp * sr - sy * cr; R[2] = cy * sp * cr + sy * sr;
  R[3] = sy * cp; R[4] = sy * sp * sr + cy * cr; R[5] = sy * sp * cr - cy * sr;
  R[6] = -sp;     R[7] = cp * sr;                R[8] = cp * cr;
}

template <class C> static bool chain_matches(const ArmEnvChain &ch) {
  for (int k = 0; k < 3; ++k)
    if (ch.base_xyz[k] != 0.0 || ch.base_rpy[k] != 0.0) return false;
  for (int j = 0; j < NJ; ++j) {
    double R[9];
    rpy_to_mat(ch.origin_rpy[j], R);
    for (int c = 0; c < 3; ++c) {
      if (std::fabs(ch.origin_xyz[j][c] - C::xyz[j][c]) > 1e-12) return false;
      for (int r = 0; r < 3; ++r) {
        const double want = (r == C::perm[j][c]) ? (double)C::sgn[j][c] : 0.0;
        if (std::fabs(R[3 * r + c] - want) > 1e-9) return false;
      }
    }
  }
  return true;
}

struct EngineBase {
  virtual ~EngineBase() {
    if (actor_buf) (void)hipFree(actor_buf);
  }
  virtual int init(const ArmEnvConfig &cfg) = 0;
  virtual int reset(const uint8_t *mask, const float *goal, float *obs, hipStream_t s) = 0;
  virtual int step(const StepIO &io, hipStream_t s) = 0;
  virtual int rollout(int32_t steps, const float *actions, const StepIO &io0, float *actions_out, hipStream_t s) = 0;
  PolicyParams pol{};
  float *actor_buf = nullptr;   // packed W1P | W2P | B2W3 on the handle's device
  int set_actor(const float *W1, const float *b1, const float *W2, const float *b2, const float *W3, const float *b3,
                int in_dim, float bound, hipStream_t s);
  int actor_forward(int64_t n, const float *states, float *actions, hipStream_t s);
  virtual int fk(int64_t n, const double *q, double *pos, double *quat, hipStream_t s) = 0;
  virtual int ik(int64_t n, const double *q, const double *tgt, double *q_out, int32_t *iters, hipStream_t s) = 0;
  virtual int get_state(double *q, float *goal, int32_t *step, uint32_t *episode, double *ep_return, double *aux,
                        hipStream_t s) = 0;
  virtual int set_state(const double *q, const float *goal, const int32_t *step, const uint32_t *episode,
                        const double *ep_return, const double *aux, hipStream_t s) = 0;
  virtual int episode_stats(double *last_return, int32_t *last_len, uint8_t *last_success, hipStream_t s) = 0;
  virtual int counters(uint64_t out[8], hipStream_t s) = 0;
  virtual const char *name() const = 0;
};

static inline unsigned grid_for(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }

int EngineBase::set_actor(const float *W1, const float *b1, const float *W2, const float *b2, const float *W3,
                          const float *b3, int in_dim, float bound, hipStream_t s) {
  const size_t n1 = ACTOR_HID * 12, n2 = (size_t)ACTOR_HID * ACTOR_HID, n3 = ACTOR_HID * 4;
  // f32 tables, then the two f16 tables (n2 halfs each = n2 floats together)
  if (!actor_buf && hipMalloc(reinterpret_cast<void **>(&actor_buf), (n1 + n2 + n3 + n2) * sizeof(float)) != hipSuccess)
    return fail(ARMENV_ENOMEM, "armenv_set_policy: hipMalloc failed");
  float *W1P = actor_buf, *W2P = actor_buf + n1, *B2W3 = actor_buf + n1 + n2;
  _Float16 *W2H = reinterpret_cast<_Float16 *>(actor_buf + n1 + n2 + n3), *W2L = W2H + n2;
  hipLaunchKernelGGL(actor_pack_kernel, dim3((unsigned)(n2 / 256)), dim3(256), 0, s, W1, b1, W2, b2, W3, in_dim, W1P, W2P, B2W3,
                     W2H, W2L);
  pol.actor_h.W2H = reinterpret_cast<const half8 *>(W2H);
  pol.actor_h.W2L = reinterpret_cast<const half8 *>(W2L);
  HIP_TRY(hipGetLastError());
  float hb3[3];
  HIP_TRY(hipMemcpyAsync(hb3, b3, sizeof hb3, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  pol.actor.W1P = W1P;
  pol.actor.W2P = reinterpret_cast<const float4 *>(W2P);
  pol.actor.B2W3 = reinterpret_cast<const float4 *>(B2W3);
  for (int k = 0; k < 3; ++k) pol.actor.b3[k] = hb3[k];
  pol.actor.bound = bound;
  pol.actor.in_dim = in_dim;
  return ARMENV_OK;
}

int EngineBase::actor_forward(int64_t n, const float *states, float *actions, hipStream_t s) {
  if (!pol.actor.W2P) return fail(ARMENV_ESTATE, "armenv_actor_forward: no actor installed (armenv_set_policy)");
  const unsigned grid = (unsigned)((n + 255) / 256);
  const bool fast = pol.kind == ARMENV_POLICY_ACTOR_F16X3;
  if (pol.actor.in_dim == 6) {
    if (fast) hipLaunchKernelGGL((actor_kernel<6, 1>), dim3(grid), dim3(256), 0, s, pol.actor, pol.actor_h, n, states, actions);
    else hipLaunchKernelGGL((actor_kernel<6, 0>), dim3(grid), dim3(256), 0, s, pol.actor, pol.actor_h, n, states, actions);
  } else {
    if (fast) hipLaunchKernelGGL((actor_kernel<9, 1>), dim3(grid), dim3(256), 0, s, pol.actor, pol.actor_h, n, states, actions);
    else hipLaunchKernelGGL((actor_kernel<9, 0>), dim3(grid), dim3(256), 0, s, pol.actor, pol.actor_h, n, states, actions);
  }
  HIP_TRY(hipGetLastError());
  return ARMENV_OK;
}

template <class C, typename T> struct Engine final : EngineBase {
  EnvParams<T> P{};
  void *pool = nullptr;
  int block = 256;
  int task = ARMENV_TASK_REACH;
  std::string kname;

  ~Engine() override {
    if (pool) (void)hipFree(pool);
  }

  int init(const ArmEnvConfig &cfg) override {
    const int64_t n = cfg.num_envs;
    P.n = n;
    // carve one allocation, 256-byte aligned sections
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    const size_t o_q = take(sizeof(T) * NJ * n), o_er = take(sizeof(T) * n), o_lr = take(sizeof(T) * n);
    const size_t o_goal = take(sizeof(float) * 3 * n), o_step = take(4 * n), o_ep = take(4 * n), o_ll = take(4 * n);
    const size_t o_ls = take(n), o_cnt = take(8 * 8), o_tmp = take(sizeof(T) * 4);
    task = cfg.task;
    const size_t o_aux = take(task == ARMENV_TASK_PUSH ? sizeof(T) * 7 * n : 0);
    if (hipMalloc(&pool, off) != hipSuccess) return fail(ARMENV_ENOMEM, "hipMalloc(%zu bytes) failed", off);
    HIP_TRY(hipMemset(pool, 0, off));
    char *b = static_cast<char *>(pool);
    P.q = reinterpret_cast<T *>(b + o_q);
    P.ep_return = reinterpret_cast<T *>(b + o_er);
    P.last_return = reinterpret_cast<T *>(b + o_lr);
    P.goal = reinterpret_cast<float *>(b + o_goal);
    P.step = reinterpret_cast<int32_t *>(b + o_step);
    P.episode = reinterpret_cast<uint32_t *>(b + o_ep);
    P.last_len = reinterpret_cast<int32_t *>(b + o_ll);
    P.last_success = reinterpret_cast<uint8_t *>(b + o_ls);
    P.counters = reinterpret_cast<unsigned long long *>(b + o_cnt);
    T *tmp = reinterpret_cast<T *>(b + o_tmp);
    P.aux = task == ARMENV_TASK_PUSH ? reinterpret_cast<T *>(b + o_aux) : nullptr;
    P.push_success_dis = (T)cfg.push_success_dis;
    P.push_cube_half = (T)cfg.push_cube_half;
    P.push_eef_radius = (T)cfg.push_eef_radius;
    P.push_rest_z = cfg.push_rest_z;
    P.push_place_min = cfg.push_place_min;
    P.push_place_max = cfg.push_place_max;

    P.dv = (T)cfg.dv;
    P.reach_dis = (T)cfg.reach_dis;
    P.max_steps = cfg.max_steps;
    P.auto_reset = cfg.auto_reset;
    P.seed = cfg.seed;
    P.env_id0 = cfg.env_id_offset;
    for (int k = 0; k < 3; ++k) {
      P.box_lo[k] = (T)cfg.box_lo[k]; P.box_hi[k] = (T)cfg.box_hi[k];
      P.goal_lo[k] = cfg.goal_lo[k]; P.goal_hi[k] = cfg.goal_hi[k];
    }
    for (int j = 0; j < NJ; ++j) {
      P.q_init[j] = (T)cfg.q_init[j];
      P.ik.lim_lo[j] = (T)cfg.chain.limit_lo[j];
      P.ik.lim_hi[j] = (T)cfg.chain.limit_hi[j];
    }
    for (int k = 0; k < 4; ++k) P.ik.tq[k] = (T)cfg.target_quat[k];
    P.ik.lambda = (T)cfg.ik_lambda;
    P.ik.residual = (T)cfg.ik_residual;
    P.ik.max_dtheta = (T)cfg.ik_max_dtheta;
    P.ik.max_iters = cfg.ik_max_iters;
    P.ik.exit_mode = cfg.ik_exit_mode;
    P.ik.angle_f32 = cfg.ik_angle_f32;
    P.ik.clamp_limits = cfg.clamp_joint_limits;
    for (int j = 0; j < NJ; ++j) {
      double R[9];
      rpy_to_mat(cfg.chain.origin_rpy[j], R);
      for (int k = 0; k < 9; ++k) P.chain.R[j][k] = (T)R[k];
      for (int k = 0; k < 3; ++k) P.chain.xyz[j][k] = (T)cfg.chain.origin_xyz[j][k];
    }
    {
      double R[9];
      rpy_to_mat(cfg.chain.base_rpy, R);
      for (int k = 0; k < 9; ++k) P.chain.base_R[k] = (T)R[k];
      for (int k = 0; k < 3; ++k) P.chain.base_p[k] = (T)cfg.chain.base_xyz[k];
    }
    hipLaunchKernelGGL((init_consts_kernel<C, T>), dim3(1), dim3(64), 0, 0, P, tmp);
    HIP_TRY(hipGetLastError());
    T host_p[3];
    HIP_TRY(hipMemcpy(host_p, tmp, sizeof host_p, hipMemcpyDeviceToHost));
    for (int k = 0; k < 3; ++k) P.p_init[k] = host_p[k];
    if (const char *bs = getenv("ARMENV_BLOCK")) {
      const int v = atoi(bs);
      if (v == 64 || v == 128 || v == 256) block = v;
    }
    kname = std::string(task == ARMENV_TASK_PUSH ? "push_step<" : "reach_step<") + (sizeof(T) == 8 ? "f64" : "f32") + "," + C::kName + ">";
    return ARMENV_OK;
  }

  int reset(const uint8_t *mask, const float *goal, float *obs, hipStream_t s) override {
    if (task == ARMENV_TASK_PUSH)
      hipLaunchKernelGGL((push_reset_kernel<T>), dim3(grid_for(P.n, block)), dim3(block), 0, s, P, mask, goal, obs);
    else
      hipLaunchKernelGGL((reach_reset_kernel<T>), dim3(grid_for(P.n, block)), dim3(block), 0, s, P, mask, goal, obs);
    HIP_TRY(hipGetLastError());
    return ARMENV_OK;
  }
  int step(const StepIO &io, hipStream_t s) override {
    if (task == ARMENV_TASK_PUSH)
      hipLaunchKernelGGL((env_step_kernel<PushLane<C, T>, T>), dim3(grid_for(P.n, block)), dim3(block), 0, s, P, io);
    else
      hipLaunchKernelGGL((env_step_kernel<ReachLane<C, T>, T>), dim3(grid_for(P.n, block)), dim3(block), 0, s, P, io);
    HIP_TRY(hipGetLastError());
    return ARMENV_OK;
  }
  template <class Lane, int POLICY>
  void launch_rollout(int32_t steps, const float *actions, const StepIO &io0, float *actions_out, hipStream_t s) {
    hipLaunchKernelGGL((env_rollout_kernel<Lane, T, POLICY>), dim3(grid_for(P.n, block)), dim3(block), 0, s, P, pol, steps,
                       actions, io0, actions_out);
  }
  template <class Lane>
  void launch_rollout_policy(int32_t steps, const float *actions, const StepIO &io0, float *actions_out, hipStream_t s) {
    if (actions) launch_rollout<Lane, ARMENV_POLICY_EXTERNAL>(steps, actions, io0, actions_out, s);
    else if (pol.kind == ARMENV_POLICY_ACTOR) launch_rollout<Lane, ARMENV_POLICY_ACTOR>(steps, actions, io0, actions_out, s);
    else if (pol.kind == ARMENV_POLICY_ACTOR_F16X3) launch_rollout<Lane, ARMENV_POLICY_ACTOR_F16X3>(steps, actions, io0, actions_out, s);
    else launch_rollout<Lane, ARMENV_POLICY_RANDOM>(steps, actions, io0, actions_out, s);
  }
  int rollout(int32_t steps, const float *actions, const StepIO &io0, float *actions_out, hipStream_t s) override {
    if (task == ARMENV_TASK_PUSH) launch_rollout_policy<PushLane<C, T>>(steps, actions, io0, actions_out, s);
    else launch_rollout_policy<ReachLane<C, T>>(steps, actions, io0, actions_out, s);
    HIP_TRY(hipGetLastError());
    return ARMENV_OK;
  }
  int fk(int64_t n, const double *q, double *pos, double *quat, hipStream_t s) override {
    hipLaunchKernelGGL((fk_kernel<C, T>), dim3(grid_for(n, block)), dim3(block), 0, s, P, n, q, pos, quat);
    HIP_TRY(hipGetLastError());
    return ARMENV_OK;
  }
  int ik(int64_t n, const double *q, const double *tgt, double *q_out, int32_t *iters, hipStream_t s) override {
    hipLaunchKernelGGL((ik_kernel<C, T>), dim3(grid_for(n, block)), dim3(block), 0, s, P, n, q, tgt, q_out, iters);
    HIP_TRY(hipGetLastError());
    return ARMENV_OK;
  }
  int get_state(double *q, float *goal, int32_t *step, uint32_t *episode, double *ep_return, double *aux,
                hipStream_t s) override {
    hipLaunchKernelGGL((get_state_kernel<T>), dim3(grid_for(P.n, block)), dim3(block), 0, s, P, q, goal, step, episode,
                       ep_return, aux);
    HIP_TRY(hipGetLastError());
    return ARMENV_OK;
  }
  int set_state(const double *q, const float *goal, const int32_t *step, const uint32_t *episode,
                const double *ep_return, const double *aux, hipStream_t s) override {
    hipLaunchKernelGGL((set_state_kernel<T>), dim3(grid_for(P.n, block)), dim3(block), 0, s, P, q, goal, step, episode,
                       ep_return, aux);
    HIP_TRY(hipGetLastError());
    return ARMENV_OK;
  }
  int episode_stats(double *last_return, int32_t *last_len, uint8_t *last_success, hipStream_t s) override {
    hipLaunchKernelGGL((episode_stats_kernel<T>), dim3(grid_for(P.n, block)), dim3(block), 0, s, P, last_return,
                       last_len, last_success);
    HIP_TRY(hipGetLastError());
    return ARMENV_OK;
  }
  int counters(uint64_t out[8], hipStream_t s) override {
    HIP_TRY(hipMemcpyAsync(out, P.counters, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return ARMENV_OK;
  }
  const char *name() const override { return kname.c_str(); }
};

struct ArmEnv {
  ArmEnvConfig cfg;
  std::unique_ptr<EngineBase> eng;
};

template <typename T> static EngineBase *make_engine(const ArmEnvConfig &cfg) {
  if (cfg.fk_path == ARMENV_FK_AUTO) {
    if (chain_matches<KukaChain>(cfg.chain)) return new (std::nothrow) Engine<KukaChain, T>();
    if (chain_matches<DianaChain>(cfg.chain)) return new (std::nothrow) Engine<DianaChain, T>();
  }
  return new (std::nothrow) Engine<GenericChain, T>();
}

static void fill_chain(ArmEnvChain *out, const double (*xyz)[3], const double (*rpy)[3], const double *lo,
                       const double *hi) {
  std::memset(out, 0, sizeof *out);
  for (int j = 0; j < NJ; ++j) {
    for (int k = 0; k < 3; ++k) { out->origin_xyz[j][k] = xyz[j][k]; out->origin_rpy[j][k] = rpy[j][k]; }
    out->limit_lo[j] = lo[j];
    out->limit_hi[j] = hi[j];
  }
}

extern "C" {

#ifdef ARMENV_TIMELINE
int armenv_dbg_set_timeline(unsigned long long *buf_dev) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_timeline), &buf_dev, sizeof buf_dev);
}
int armenv_dbg_sections(unsigned long long out[8], int reset) {
  int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(armenv::g_sections), 8 * sizeof(unsigned long long));
  if (reset) {
    unsigned long long z[8] = {0};
    rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(armenv::g_sections), z, sizeof z);
  }
  return rc;
}
#endif

int32_t armenv_abi_version(void) { return ARMENV_ABI_VERSION; }
const char *armenv_last_error(void) { return g_err.c_str(); }

int armenv_builtin_chain(int32_t robot, ArmEnvChain *out) {
  if (!out) return fail(ARMENV_EINVAL, "armenv_builtin_chain: out is NULL");
  if (robot == ARMENV_ROBOT_KUKA) fill_chain(out, KukaChain::xyz, KukaChain::rpy, KukaChain::limit_lo, KukaChain::limit_hi);
  else if (robot == ARMENV_ROBOT_DIANA) fill_chain(out, DianaChain::xyz, DianaChain::rpy, DianaChain::limit_lo, DianaChain::limit_hi);
  else return fail(ARMENV_EINVAL, "armenv_builtin_chain: unknown robot %d", robot);
  return ARMENV_OK;
}

int armenv_default_config(int32_t task, ArmEnvConfig *c) {
  if (!c) return fail(ARMENV_EINVAL, "armenv_default_config: cfg is NULL");
  if (task != ARMENV_TASK_REACH && task != ARMENV_TASK_PUSH) return fail(ARMENV_EINVAL, "unknown task %d", task);
  std::memset(c, 0, sizeof *c);
  c->abi_version = ARMENV_ABI_VERSION;
  c->device = 0;
  c->num_envs = 1;
  c->task = task;
  c->precision = 64;
  c->fk_path = ARMENV_FK_AUTO;
  c->auto_reset = 1;
  c->seed = 0;
  c->env_id_offset = 0;
  c->dv = task == ARMENV_TASK_REACH ? 0.02 : 0.08;
  c->reach_dis = 0.01;
  c->max_steps = 500;
  c->clamp_joint_limits = 0;
  const double lo[3] = {0.2, -0.3, 0.0}, hi[3] = {0.7, 0.3, 0.55};
  for (int k = 0; k < 3; ++k) { c->box_lo[k] = lo[k]; c->box_hi[k] = hi[k]; c->goal_lo[k] = lo[k]; c->goal_hi[k] = hi[k]; }
  if (task == ARMENV_TASK_PUSH) c->box_hi[2] = 0.1;
  // p.getQuaternionFromEuler([0, -pi, pi/2]) (Bullet setEulerZYX)
  {
    const double pi = 3.14159265358979323846;
    const double hr = 0.0, hp = -pi * 0.5, hy = pi * 0.25;
    const double cr = std::cos(hr), sr = std::sin(hr), cp = std::cos(hp), sp = std::sin(hp), cy = std::cos(hy), sy = std::sin(hy);
    c->target_quat[0] = sr * cp * cy - cr * sp * sy;
    c->target_quat[1] = cr * sp * cy + sr * cp * sy;
    c->target_quat[2] = cr * cp * sy - sr * sp * cy;
    c->target_quat[3] = cr * cp * cy + sr * sp * sy;
  }
  const double qi[NJ] = {0.006418, 0.413184, -0.011401, -1.589317, 0.005379, 1.137684, -0.006539};
  for (int j = 0; j < NJ; ++j) c->q_init[j] = qi[j];
  c->ik_lambda = 1e-5;
  c->ik_residual = 1e-4;
  c->ik_max_dtheta = 45.0 * 3.14159265358979323846 / 180.0;
  c->ik_max_iters = 20;
  c->ik_exit_mode = 0;
  c->ik_angle_f32 = 1;
  c->push_success_dis = 0.05;
  c->push_cube_half = 0.02;
  c->push_eef_radius = 0.03;
  c->push_rest_z = 0.01;
  c->push_place_min = 0.22;
  c->push_place_max = 0.25;
  return armenv_builtin_chain(ARMENV_ROBOT_KUKA, &c->chain);
}

int armenv_create(const ArmEnvConfig *cfg, ArmEnv **out) {
  if (!cfg || !out) return fail(ARMENV_EINVAL, "armenv_create: NULL argument");
  *out = nullptr;
  if (cfg->abi_version != ARMENV_ABI_VERSION)
    return fail(ARMENV_EINVAL, "armenv_create: abi_version %d, library is %d", cfg->abi_version, ARMENV_ABI_VERSION);
  if (cfg->num_envs < 1) return fail(ARMENV_EINVAL, "armenv_create: num_envs must be >= 1");
  if (cfg->precision != 64 && cfg->precision != 32) return fail(ARMENV_EINVAL, "armenv_create: precision must be 32 or 64");
  if (cfg->task != ARMENV_TASK_REACH && cfg->task != ARMENV_TASK_PUSH) return fail(ARMENV_EINVAL, "armenv_create: unknown task %d", cfg->task);
  if (cfg->ik_max_iters < 0 || cfg->ik_max_iters > 1000) return fail(ARMENV_EINVAL, "armenv_create: ik_max_iters out of range");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(ARMENV_ENODEV, "armenv_create: no HIP device is visible (this library has no CPU fallback)");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(ARMENV_ENODEV, "armenv_create: device %d of %d", cfg->device, ndev);
  DeviceGuard guard(cfg->device);
  if (!guard.ok) return fail(ARMENV_ENODEV, "armenv_create: hipSetDevice(%d) failed", cfg->device);
  std::unique_ptr<ArmEnv> env(new (std::nothrow) ArmEnv());
  if (!env) return fail(ARMENV_ENOMEM, "armenv_create: host allocation failed");
  env->cfg = *cfg;
  env->eng.reset(cfg->precision == 64 ? make_engine<double>(*cfg) : make_engine<float>(*cfg));
  if (!env->eng) return fail(ARMENV_ENOMEM, "armenv_create: host allocation failed");
  const int rc = env->eng->init(*cfg);
  if (rc != ARMENV_OK) return rc;
  *out = env.release();
  return ARMENV_OK;
}

void armenv_destroy(ArmEnv *env) {
  if (!env) return;
  DeviceGuard guard(env->cfg.device);
  delete env;
}

#define ENV_ENTER(env)                                                                  \
  if (!(env)) return fail(ARMENV_EINVAL, "%s: env is NULL", __func__);                  \
  DeviceGuard guard_((env)->cfg.device);                                                \
  if (!guard_.ok) return fail(ARMENV_ENODEV, "%s: hipSetDevice failed", __func__)

int armenv_reset(ArmEnv *env, const uint8_t *mask_dev, float *obs_dev, void *stream) {
  ENV_ENTER(env);
  return env->eng->reset(mask_dev, nullptr, obs_dev, static_cast<hipStream_t>(stream));
}

int armenv_reset_with_goal(ArmEnv *env, const uint8_t *mask_dev, const float *goal_dev, float *obs_dev, void *stream) {
  ENV_ENTER(env);
  if (!goal_dev) return fail(ARMENV_EINVAL, "armenv_reset_with_goal: goal_dev is NULL");
  return env->eng->reset(mask_dev, goal_dev, obs_dev, static_cast<hipStream_t>(stream));
}

int armenv_step(ArmEnv *env, const float *action_dev, float *obs_dev, float *reward_dev, uint8_t *done_dev,
                uint8_t *success_dev, float *terminal_obs_dev, void *stream) {
  ENV_ENTER(env);
  if (!obs_dev || !reward_dev || !done_dev || !success_dev) return fail(ARMENV_EINVAL, "armenv_step: NULL output buffer");
  StepIO io{action_dev, obs_dev, reward_dev, done_dev, success_dev, terminal_obs_dev};
  if (!action_dev) {   // fused policy: a one-step rollout
    if (env->eng->pol.kind == ARMENV_POLICY_EXTERNAL)
      return fail(ARMENV_ESTATE, "armenv_step: action_dev is NULL and no fused policy is installed");
    return env->eng->rollout(1, nullptr, io, nullptr, static_cast<hipStream_t>(stream));
  }
  return env->eng->step(io, static_cast<hipStream_t>(stream));
}

int armenv_fk(ArmEnv *env, int64_t n, const double *q_dev, double *pos_dev, double *quat_dev, void *stream) {
  ENV_ENTER(env);
  if (n < 0 || (n > 0 && (!q_dev || !pos_dev))) return fail(ARMENV_EINVAL, "armenv_fk: bad arguments");
  if (n == 0) return ARMENV_OK;
  return env->eng->fk(n, q_dev, pos_dev, quat_dev, static_cast<hipStream_t>(stream));
}

int armenv_ik(ArmEnv *env, int64_t n, const double *q_dev, const double *target_pos_dev, double *q_out_dev,
              int32_t *iters_dev, void *stream) {
  ENV_ENTER(env);
  if (n < 0 || (n > 0 && (!q_dev || !target_pos_dev || !q_out_dev))) return fail(ARMENV_EINVAL, "armenv_ik: bad arguments");
  if (n == 0) return ARMENV_OK;
  return env->eng->ik(n, q_dev, target_pos_dev, q_out_dev, iters_dev, static_cast<hipStream_t>(stream));
}

int armenv_get_state(ArmEnv *env, double *q_dev, float *goal_dev, int32_t *step_dev, uint32_t *episode_dev,
                     double *ep_return_dev, double *aux_dev, void *stream) {
  ENV_ENTER(env);
  if (aux_dev && env->cfg.task != ARMENV_TASK_PUSH) return fail(ARMENV_EINVAL, "armenv_get_state: aux is only defined for the push task");
  if (goal_dev && env->cfg.task == ARMENV_TASK_PUSH) return fail(ARMENV_EINVAL, "armenv_get_state: push keeps cube/target in aux, not goal");
  return env->eng->get_state(q_dev, goal_dev, step_dev, episode_dev, ep_return_dev, aux_dev, static_cast<hipStream_t>(stream));
}

int armenv_set_state(ArmEnv *env, const double *q_dev, const float *goal_dev, const int32_t *step_dev,
                     const uint32_t *episode_dev, const double *ep_return_dev, const double *aux_dev, void *stream) {
  ENV_ENTER(env);
  if (aux_dev && env->cfg.task != ARMENV_TASK_PUSH) return fail(ARMENV_EINVAL, "armenv_set_state: aux is only defined for the push task");
  if (goal_dev && env->cfg.task == ARMENV_TASK_PUSH) return fail(ARMENV_EINVAL, "armenv_set_state: push keeps cube/target in aux, not goal");
  return env->eng->set_state(q_dev, goal_dev, step_dev, episode_dev, ep_return_dev, aux_dev, static_cast<hipStream_t>(stream));
}

int armenv_episode_stats(ArmEnv *env, double *last_return_dev, int32_t *last_len_dev, uint8_t *last_success_dev,
                         void *stream) {
  ENV_ENTER(env);
  return env->eng->episode_stats(last_return_dev, last_len_dev, last_success_dev, static_cast<hipStream_t>(stream));
}

int armenv_counters(ArmEnv *env, uint64_t out[8], void *stream) {
  ENV_ENTER(env);
  if (!out) return fail(ARMENV_EINVAL, "armenv_counters: out is NULL");
  return env->eng->counters(out, static_cast<hipStream_t>(stream));
}

int armenv_set_policy(ArmEnv *env, int32_t policy, const float *W1_dev, const float *b1_dev, const float *W2_dev,
                      const float *b2_dev, const float *W3_dev, const float *b3_dev, int32_t hidden_dim, float action_bound,
                      float noise_sigma, float noise_clip, void *stream) {
  ENV_ENTER(env);
  if (policy < ARMENV_POLICY_EXTERNAL || policy > ARMENV_POLICY_ACTOR_F16X3)
    return fail(ARMENV_EINVAL, "armenv_set_policy: unknown policy %d", policy);
  if (policy != ARMENV_POLICY_EXTERNAL && !(noise_sigma >= 0.f && noise_clip > 0.f))
    return fail(ARMENV_EINVAL, "armenv_set_policy: need noise_sigma >= 0 and noise_clip > 0");
  if (policy == ARMENV_POLICY_ACTOR || policy == ARMENV_POLICY_ACTOR_F16X3) {
    if (!W1_dev || !b1_dev || !W2_dev || !b2_dev || !W3_dev || !b3_dev) return fail(ARMENV_EINVAL, "armenv_set_policy: NULL weight pointer");
    if (hidden_dim != ACTOR_HID)
      return fail(ARMENV_EINVAL, "armenv_set_policy: hidden_dim %d; the fused actor is built for %d (config.py:56)", hidden_dim, ACTOR_HID);
    if (env->cfg.num_envs % 64 != 0)
      return fail(ARMENV_EINVAL, "armenv_set_policy: the fused actor needs num_envs to be a multiple of 64 (full wavefronts)");
    const int rc = env->eng->set_actor(W1_dev, b1_dev, W2_dev, b2_dev, W3_dev, b3_dev, armenv_obs_dim(env), action_bound,
                                       static_cast<hipStream_t>(stream));
    if (rc != ARMENV_OK) return rc;
  }
  env->eng->pol.kind = policy;
  env->eng->pol.sigma = noise_sigma;
  env->eng->pol.clip = noise_clip;
  env->eng->pol.bound = action_bound;
  return ARMENV_OK;
}

int armenv_actor_forward(ArmEnv *env, int64_t n, const float *states_dev, float *actions_dev, void *stream) {
  ENV_ENTER(env);
  if (n < 0 || (n > 0 && (!states_dev || !actions_dev))) return fail(ARMENV_EINVAL, "armenv_actor_forward: bad arguments");
  if (n == 0) return ARMENV_OK;
  return env->eng->actor_forward(n, states_dev, actions_dev, static_cast<hipStream_t>(stream));
}

int armenv_rollout(ArmEnv *env, int32_t steps, const float *actions_dev, float *obs_dev, float *reward_dev,
                   uint8_t *done_dev, uint8_t *success_dev, float *actions_out_dev, float *terminal_obs_dev, void *stream) {
  ENV_ENTER(env);
  if (steps < 0) return fail(ARMENV_EINVAL, "armenv_rollout: steps < 0");
  if (steps == 0) return ARMENV_OK;
  if (!obs_dev || !reward_dev || !done_dev || !success_dev) return fail(ARMENV_EINVAL, "armenv_rollout: NULL output buffer");
  if (!actions_dev && env->eng->pol.kind == ARMENV_POLICY_EXTERNAL)
    return fail(ARMENV_ESTATE, "armenv_rollout: actions_dev is NULL and no fused policy is installed");
  if (steps == 0) return ARMENV_OK;
  StepIO io{nullptr, obs_dev, reward_dev, done_dev, success_dev, terminal_obs_dev};
  return env->eng->rollout(steps, actions_dev, io, actions_out_dev, static_cast<hipStream_t>(stream));
}

#define DEV_ENTER(device)                                                              \
  DeviceGuard guard_(device);                                                          \
  if (!guard_.ok) return fail(ARMENV_ENODEV, "%s: hipSetDevice(%d) failed", __func__, (int)(device))

int armenv_count_episodes(int32_t device, int64_t T, int64_t N, int64_t ring_base, int64_t ring_cap, const uint8_t *done_dev,
                          int32_t starts_at_reset, int32_t *counts_dev, void *stream) {
  DEV_ENTER(device);
  if (T < 0 || N < 1 || ring_cap < T || ring_cap < 1 || ring_base < 0 || !done_dev || !counts_dev)
    return fail(ARMENV_EINVAL, "armenv_count_episodes: bad arguments");
  hipLaunchKernelGGL(index_episodes_kernel, dim3(grid_for(N, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), T, N,
                     ring_base, ring_cap, done_dev, starts_at_reset, counts_dev, (const int64_t *)nullptr, (int32_t *)nullptr);
  HIP_TRY(hipGetLastError());
  return ARMENV_OK;
}

int armenv_write_episodes(int32_t device, int64_t T, int64_t N, int64_t ring_base, int64_t ring_cap, const uint8_t *done_dev,
                          int32_t starts_at_reset, const int32_t *counts_dev, const int64_t *offsets_dev,
                          int32_t *episodes_dev, void *stream) {
  DEV_ENTER(device);
  if (T < 0 || N < 1 || ring_cap < T || ring_cap < 1 || ring_base < 0 || !done_dev || !counts_dev || !offsets_dev || !episodes_dev)
    return fail(ARMENV_EINVAL, "armenv_write_episodes: bad arguments");
  hipLaunchKernelGGL(index_episodes_kernel, dim3(grid_for(N, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), T, N,
                     ring_base, ring_cap, done_dev, starts_at_reset, const_cast<int32_t *>(counts_dev), offsets_dev, episodes_dev);
  HIP_TRY(hipGetLastError());
  return ARMENV_OK;
}

int armenv_her_sample(int32_t device, const ArmEnvHerArgs *a, void *stream) {
  DEV_ENTER(device);
  if (!a) return fail(ARMENV_EINVAL, "armenv_her_sample: args is NULL");
  if (a->obs_dim != 6 && a->obs_dim != 9) return fail(ARMENV_EINVAL, "armenv_her_sample: obs_dim must be 6 or 9");
  if (a->batch < 0 || a->T < 1 || a->N < 1 || a->ring_cap < a->T || a->ring_base < 0)
    return fail(ARMENV_EINVAL, "armenv_her_sample: bad sizes");
  if (!a->obs0_dev || !a->obs_after_dev || !a->next_obs_dev || !a->action_dev || !a->reward_dev || !a->done_dev ||
      !a->episodes_dev || !a->num_episodes_dev || !a->states_dev || !a->actions_dev || !a->next_states_dev ||
      !a->rewards_dev || !a->dones_dev)
    return fail(ARMENV_EINVAL, "armenv_her_sample: NULL buffer");
  if (!(a->her_ratio >= 0.f && a->her_ratio <= 1.f)) return fail(ARMENV_EINVAL, "armenv_her_sample: her_ratio outside [0,1]");
  if (a->batch == 0) return ARMENV_OK;
  HerArgs h;
  h.T = a->T; h.N = a->N; h.ring_base = a->ring_base; h.ring_cap = a->ring_cap; h.D = a->obs_dim;
  h.obs0 = a->obs0_dev; h.obs_after = a->obs_after_dev; h.next_obs = a->next_obs_dev; h.action = a->action_dev;
  h.reward = a->reward_dev; h.done = a->done_dev; h.episodes = a->episodes_dev; h.num_episodes = a->num_episodes_dev;
  h.B = a->batch; h.picks_in = a->picks_dev; h.seed = a->seed; h.draw = a->draw; h.use_her = a->use_her;
  h.her_ratio = a->her_ratio; h.dis_threshold = a->dis_threshold;
  h.states = a->states_dev; h.actions = a->actions_dev; h.next_states = a->next_states_dev; h.rewards = a->rewards_dev;
  h.dones = a->dones_dev; h.picks_out = a->picks_out_dev;
  const dim3 grid(grid_for(a->batch, 256)), block(256);
  if (a->obs_dim == 6) hipLaunchKernelGGL((her_sample_kernel<6>), grid, block, 0, static_cast<hipStream_t>(stream), h);
  else hipLaunchKernelGGL((her_sample_kernel<9>), grid, block, 0, static_cast<hipStream_t>(stream), h);
  HIP_TRY(hipGetLastError());
  return ARMENV_OK;
}

int64_t armenv_num_envs(const ArmEnv *env) { return env ? env->cfg.num_envs : 0; }
int32_t armenv_obs_dim(const ArmEnv *env) { return env ? (env->cfg.task == ARMENV_TASK_PUSH ? 9 : 6) : 0; }
int32_t armenv_action_dim(const ArmEnv *) { return 3; }
const char *armenv_kernel_name(const ArmEnv *env) { return env ? env->eng->name() : ""; }

}  // extern "C"
