// armenv_replay.h -- device-resident trajectory store indexing and the HER-"future" sampler, the immediate consumer of
// the step outputs.  Replaces the per-sample Python loops of
// /root/reference/utils/rl_utils.py:108-152 (ReplayBuffer_Trajectory_reach.sample) and :154-199 (..._push.sample)
// for rollout buffers that never leave HBM.
//
// Storage convention (what armenv_rollout writes, time-major):
//   obs0      f32 [N][D]      observation before step 0 of the chunk
//   obs_after f32 [T][N][D]   row t = observation returned by step t (after an auto-reset: the new episode's first obs)
//   next_obs  f32 [T][N][D]   row t = terminal_obs of step t = the true next state of the transition
//   action f32 [T][N][3], reward f32 [T][N], done u8 [T][N]
// A trajectory (rl_utils.py:91-105) is one episode of one env: states[0..L], actions/rewards/dones[0..L-1];
//   states[0]   = the observation before its first step   (obs0 or obs_after[t_start-1])
//   states[j>0] = next_obs[t_start + j - 1]
// Only complete episodes (start and terminating done inside the chunk) are indexed, as the reference only stores
// finished trajectories (main.py:129).
#pragma once
#include "armenv_math.h"

namespace armenv {

// pass 1: number of complete episodes per env column; pass 2 (write != nullptr): emit (env, t_start, length).
// `starts_at_reset`: the chunk began right after a reset of every env, so the first episode's start is inside it.
// The buffers may be a ring: logical step t lives in physical row (ring_base + t) % ring_cap (ring_cap >= T).
__global__ __launch_bounds__(256) void index_episodes_kernel(int64_t T, int64_t N, int64_t ring_base, int64_t ring_cap,
                                                            const uint8_t *done, int32_t starts_at_reset,
                                                            int32_t *counts, const int64_t *offsets, int32_t *episodes) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  int32_t cnt = 0;
  int64_t start = starts_at_reset ? 0 : -1;
  int64_t w = episodes ? (offsets[n] - (int64_t)counts[n]) : 0;   // offsets = inclusive cumsum of counts
  for (int64_t t = 0; t < T; ++t) {
    if (done[((ring_base + t) % ring_cap) * N + n]) {
      if (start >= 0) {
        if (episodes) {
          int32_t *e = episodes + 3 * w;
          e[0] = (int32_t)n; e[1] = (int32_t)start; e[2] = (int32_t)(t - start + 1);
          ++w;
        }
        ++cnt;
      }
      start = t + 1;
    }
  }
  if (!episodes) counts[n] = cnt;
}

struct HerArgs {
  int64_t T, N, ring_base, ring_cap;
  int32_t D;                 // 6 reach, 9 push
  const float *obs0, *obs_after, *next_obs, *action, *reward;
  const uint8_t *done;
  const int32_t *episodes;   // [E][3]
  const int64_t *num_episodes;  // device scalar E
  int64_t B;
  const int32_t *picks_in;   // nullable [B][4] = (episode, step_state, use_her, step_goal): teacher-forced draws
  uint64_t seed, draw;
  int32_t use_her;
  float her_ratio, dis_threshold;
  float *states, *actions, *next_states, *rewards;
  uint8_t *dones;
  int32_t *picks_out;        // nullable
};

template <int D>
__global__ __launch_bounds__(256) void her_sample_kernel(HerArgs A) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= A.B) return;
  const int64_t E = *A.num_episodes;
  int32_t ep, st, her, sg;
  if (A.picks_in) {
    ep = A.picks_in[4 * b]; st = A.picks_in[4 * b + 1]; her = A.picks_in[4 * b + 2]; sg = A.picks_in[4 * b + 3];
  } else {
    if (E <= 0) {   // nothing to sample from: a defined, inert batch
      for (int k = 0; k < D; ++k) { A.states[b * D + k] = 0.f; A.next_states[b * D + k] = 0.f; }
      A.actions[3 * b] = A.actions[3 * b + 1] = A.actions[3 * b + 2] = 0.f;
      A.rewards[b] = 0.f;
      A.dones[b] = 1;
      if (A.picks_out) { A.picks_out[4 * b] = -1; A.picks_out[4 * b + 1] = A.picks_out[4 * b + 2] = A.picks_out[4 * b + 3] = 0; }
      return;
    }
    double u0, u1, u2, u3;
    // the sampler's own Philox key: with the env's key, (batch slot b, draw d) would read exactly the uniforms that chose the
    // goal of env b in episode d whenever a caller seeds both with the same number (armenv.train does)
    const uint64_t key = A.seed ^ 0x5DEECE66D1CE4E5Bull;
    philox_pair(key, (uint64_t)b, (uint32_t)A.draw, 0u, u0, u1);
    philox_pair(key, (uint64_t)b, (uint32_t)A.draw, 1u, u2, u3);
    ep = (int32_t)(u0 * (double)E);                         // random.sample(buffer, 1)      rl_utils.py:126
    const int32_t L = A.episodes[3 * ep + 2];
    st = (int32_t)(u1 * (double)L);                         // np.random.randint(length)     :127
    her = (A.use_her && u2 <= (double)A.her_ratio) ? 1 : 0; // np.random.uniform() <= ratio  :134
    sg = st + 1 + (int32_t)(u3 * (double)(L - st));         // randint(step+1, length+1)     :135
  }
  const int32_t *e = A.episodes + 3 * ep;
  const int64_t n = e[0], t0 = e[1];
  auto row = [&](int64_t lt) { return (A.ring_base + lt) % A.ring_cap; };   // logical step -> physical row
  const int64_t t = row(t0 + st);                           // physical row of the transition
  auto state_ptr = [&](int64_t j) -> const float * {        // traj.states[j]
    const int64_t tt = t0 + j;                              // states[j] is the observation before logical step tt
    if (j == 0) return tt == 0 ? A.obs0 + n * D : A.obs_after + (row(tt - 1) * A.N + n) * D;
    return A.next_obs + (row(tt - 1) * A.N + n) * D;
  };
  const float *s = state_ptr(st), *s2 = state_ptr(st + 1);
  float sv[D], nv[D];
#pragma unroll
  for (int k = 0; k < D; ++k) { sv[k] = s[k]; nv[k] = s2[k]; }
  float r = A.reward[t * A.N + n];
  uint8_t dn = A.done[t * A.N + n];
  if (her) {
    const float *gsrc = state_ptr(sg);                      // goal = traj.states[step_goal][:3]   :136
    const float g0 = gsrc[0], g1 = gsrc[1], g2 = gsrc[2];
    bool far;
    if (D == 9) {   // push observations are float64 in the reference (rl_push_env.py:308): f64 arithmetic
      const double d0 = (double)nv[0] - (double)g0, d1 = (double)nv[1] - (double)g1, d2 = (double)nv[2] - (double)g2;
      far = ::sqrt((d0 * d0 + d1 * d1) + d2 * d2) > (double)A.dis_threshold;
    } else {        // reach observations are float32: np.sqrt(np.sum(np.square(f32[3])))  :137
      const float d0 = nv[0] - g0, d1 = nv[1] - g1, d2 = nv[2] - g2;
      far = (double)sqrtf((d0 * d0 + d1 * d1) + d2 * d2) > (double)A.dis_threshold;
    }
    r = far ? -0.1f : 1.0f;                                 // :138
    dn = far ? 0 : 1;                                       // :139
    if (D == 9) {                                           // push: state[6:10] kept from `state` for BOTH rows, :187-188
      nv[6] = sv[6]; nv[7] = sv[7]; nv[8] = sv[8];
    }
    sv[3] = g0; sv[4] = g1; sv[5] = g2;                     // :140-141 / :187-188
    nv[3] = g0; nv[4] = g1; nv[5] = g2;
  }
#pragma unroll
  for (int k = 0; k < D; ++k) { A.states[b * D + k] = sv[k]; A.next_states[b * D + k] = nv[k]; }
  const float *a = A.action + (t * A.N + n) * 3;
  A.actions[3 * b] = a[0]; A.actions[3 * b + 1] = a[1]; A.actions[3 * b + 2] = a[2];
  A.rewards[b] = r;
  A.dones[b] = dn;
  if (A.picks_out) { A.picks_out[4 * b] = ep; A.picks_out[4 * b + 1] = st; A.picks_out[4 * b + 2] = her; A.picks_out[4 * b + 3] = sg; }
}

}  // namespace armenv
