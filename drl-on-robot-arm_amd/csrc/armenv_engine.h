// armenv_engine.h -- host-side engine shared by the translation units of libarmenv.so: error plumbing, device guard,
// chain classification, the EngineBase interface the C ABI (armenv.hip) talks to, and the Engine template that owns
// one handle's HBM state and launches the kernels of armenv_env.h.
//
// Data layout in HBM (N envs, T = f64 or f32 by ArmEnvConfig.precision), struct-of-arrays with the env index fastest
// so that a wave's 64 lanes touch 64 consecutive elements of every array:
//   q[7][N] T | trig[14][N] T (cos q, sin q) | ep_return[N] T | last_return[N] T | goal[3][N] f32 | step[N] i32 | episode[N] u32 |
//   last_len[N] i32 | last_success[N] u8 | counters[N/64][16] u64 (one row per wave) | totals[16] u64 | summary rows[N/64][8] f64 | EnvCold | push: aux[7][N] T | pick: aux[11][N] T
// Caller-facing buffers keep the reference's array-of-struct shapes (action [N][3], obs [N][6|9]); a wave still
// reads/writes one contiguous span of them.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <type_traits>

#include "armenv_env.h"

// sets the thread-local message behind armenv_last_error() and returns `code` (armenv.hip)
int armenv_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3), visibility("hidden")));
#define fail armenv_fail

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

static inline void rpy_to_mat(const double rpy[3], double R[9]) {  // row-major, Rz(yaw) Ry(pitch) Rx(roll)
  const double cr = std::cos(rpy[0]), sr = std::sin(rpy[0]);
  const double cp = std::cos(rpy[1]), sp = std::sin(rpy[1]);
  const double cy = std::cos(rpy[2]), sy = std::sin(rpy[2]);
  R[0] = cy * cp; R[1] = cy * sp * sr - sy * cr; R[2] = cy * sp * cr + sy * sr;
  R[3] = sy * cp; R[4] = sy * sp * sr + cy * cr; R[5] = sy * sp * cr - cy * sr;
  R[6] = -sp;     R[7] = cp * sr;                R[8] = cp * cr;
}

template <class C> static inline bool chain_matches(const ArmEnvChain &ch) {
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
    if (datd3_buf) (void)hipFree(datd3_buf);
  }
  virtual int init(const ArmEnvConfig &cfg) = 0;
  virtual int reset(const uint8_t *mask, const float *goal, float *obs, hipStream_t s) = 0;
  virtual int step(const StepIO &io, hipStream_t s) = 0;
  virtual int rollout(int32_t steps, const float *actions, const StepIO &io0, float *actions_out, hipStream_t s) = 0;
  PolicyParams pol{};
  float *actor_buf = nullptr;   // packed W1P | W2P | B2W3 on the handle's device
  int set_actor(const float *W1, const float *b1, const float *W2, const float *b2, const float *W3, const float *b3,
                int in_dim, float bound, hipStream_t s);
  float *datd3_buf = nullptr;   // four packed nets + the device arrays of their ActorParams / ActorParamsH (armenv_set_policy_datd3)
  int set_datd3(const ArmEnvMlp *const nets[4], int obs_dim, float bound, hipStream_t s);
  int datd3_obs = 0;            // observation width the installed DATD3 nets were packed for (6 | 9)
  int datd3_forward(int64_t n, const float *states, float *actions, float *q1, float *q2, uint8_t *picked, hipStream_t s);
  int actor_forward(int64_t n, const float *states, float *actions, hipStream_t s);
  virtual int fk(int64_t n, const double *q, double *pos, double *quat, hipStream_t s) = 0;
  virtual int ik(int64_t n, const double *q, const double *tgt, double *q_out, int32_t *iters, hipStream_t s) = 0;
  virtual int get_state(double *q, float *goal, int32_t *step, uint32_t *episode, double *ep_return, double *aux,
                        double *trig, hipStream_t s) = 0;
  virtual int set_state(const double *q, const float *goal, const int32_t *step, const uint32_t *episode,
                        const double *ep_return, const double *aux, const double *trig, hipStream_t s) = 0;
  virtual int episode_stats(double *last_return, int32_t *last_len, uint8_t *last_success, hipStream_t s) = 0;
  virtual int episode_returns_f32(float *last_return, hipStream_t s) = 0;
  virtual int counters(uint64_t out[16], hipStream_t s) = 0;
  virtual int summary(double *out_dev, hipStream_t s) = 0;
  virtual const char *name() const = 0;
};

static inline unsigned grid_for(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }


// One engine per (task, chain, precision); each task x precision pair is compiled in its own translation unit
// (armenv_task.hip, see the Makefile) so that the kernel variants build in parallel.
template <template <class, class, int> class LaneT, class C, typename T> struct Engine final : EngineBase {
  using Lane = LaneT<C, T, 0>;
  using LaneF = LaneT<C, T, 1>;   // the bookkeeping build of the same lane (parity-fence counters)
  using LaneTip = LaneT<C, T, 2>; // the bookkeeping build with the IK evaluated at ArmEnvConfig.ik_tip_offset + the f64 step diagnostics
  bool fence_on = false;          // bookkeeping kernels (fence_counters, or implied by a tip offset)
  bool tip_on = false;
  // run f with the lane type of the handle's bookkeeping build, passed as a null pointer tag
  template <class F> void with_book_lane(F &&f) { if (tip_on) f((LaneTip *)nullptr); else f((LaneF *)nullptr); }
  EnvParams<T> P{};
  void *pool = nullptr;
  unsigned long long *counter_totals = nullptr;
  double *summary_rows = nullptr;
  static constexpr int block = 256;   // four waves: one per SIMD of a CU; the f16x3 actor phase is a 4-wave workgroup
  int cus = 256;
  int ready_lanes = 0;   // > 0: rollouts run lane-asynchronously (env_rollout_async_kernel)
  int straggler_trips = 0;   // > 0: its transition rule (ArmEnvConfig.rollout_straggler_trips) instead of the count
  int waves_cfg = 0;     // ArmEnvConfig.rollout_waves_per_simd: 0 auto, 1, 2
  // Workgroup size of the env kernels that have no workgroup phase (step, rollout without a fused actor): the IK keeps
  // one wave per SIMD, so a batch that does not fill the chip is launched as smaller workgroups -- the dispatcher then
  // spreads its waves over all CUs (1 or 2 per CU) instead of packing four onto a quarter or half of them, and a wave
  // that shares its CU's instruction fetch with fewer neighbours runs 5-10 % faster (65 536 envs: 9.9 us per step,
  // 16 384: 9.0).
  int lanes_cfg = 0;     // ArmEnvConfig.rollout_lanes_per_wave: 0 auto, 64, 32
  // Half-filled waves (32 envs in lanes 0..31) for the kernels without a workgroup phase: when the batch leaves at least half
  // of the SIMDs without a wave anyway, spreading it over twice as many waves costs nothing in issue slots and every wave's
  // per-step maximum of IK trips is taken over 32 lanes instead of 64 (push at 32 768 envs: 5.67 -> 5.36 trips per wave-step,
  // pick lane-asynchronous 6.51 -> 6.14; tests/tools/regroup_sim.py).  Same per-env arithmetic, same bits.
  bool half_waves() const {
    if (lanes_cfg == 64 || two_waves()) return false;
    if (lanes_cfg == 32) return true;
    return task != ARMENV_TASK_REACH && (P.n + 31) / 32 <= (int64_t)4 * cus;
  }
  int lane_block() const {
    const int64_t waves = half_waves() ? (P.n + 31) / 32 : (P.n + 63) / 64;
    const int64_t per_cu = (waves + cus - 1) / cus;
    return 64 * (int)(per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu));
  }
  static constexpr int task = Lane::kTask;
  std::string kname;

  ~Engine() override {
    if (pool) (void)hipFree(pool);
  }

  int init(const ArmEnvConfig &cfg) override {
    const int64_t n = cfg.num_envs;
    P.n = n;
    {
      int dev = 0, v = 0;
      if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    // carve one allocation, 256-byte aligned sections
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    const size_t o_q = take(sizeof(T) * NJ * n), o_er = take(sizeof(T) * n), o_lr = take(sizeof(T) * n);
    const size_t o_goal = take(sizeof(float) * 3 * n), o_step = take(4 * n), o_ep = take(4 * n), o_ll = take(4 * n);
    const size_t o_ls = take(n), o_cnt = take(8 * kCounterCols * (size_t)((n + 31) / 32)), o_tot = take(8 * kCounterCols), o_sum = take(64 * (size_t)((n + 63) / 64)), o_cold = take(sizeof(EnvCold<T>));
    const size_t o_aux = take(sizeof(T) * Lane::kAuxRows * n), o_trig = take(sizeof(T) * 2 * NJ * n);
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
    counter_totals = reinterpret_cast<unsigned long long *>(b + o_tot);
    summary_rows = reinterpret_cast<double *>(b + o_sum);
    EnvCold<T> *cold_dev = reinterpret_cast<EnvCold<T> *>(b + o_cold);
    P.cold = cold_dev;
    P.aux = Lane::kAuxRows ? reinterpret_cast<T *>(b + o_aux) : nullptr;
    P.trig = reinterpret_cast<T *>(b + o_trig);
    P.push_success_dis = (T)cfg.push_success_dis;
    P.push_cube_half = (T)cfg.push_cube_half;
    P.push_eef_radius = (T)cfg.push_eef_radius;
    P.push_rest_z = (T)cfg.push_rest_z;
    EnvCold<T> K{};
    K.push_rest_z = cfg.push_rest_z;
    K.push_place_min = cfg.push_place_min;
    K.push_place_max = cfg.push_place_max;
    K.push_place_z = cfg.push_place_z;
    {   // the cube under stepSimulation (CubeLane::cube_fall / contact_dyn): constants folded on the host in f64
      K.push_model = cfg.task == ARMENV_TASK_PUSH ? cfg.push_contact_model : 0;
      K.fall_on = cfg.task != ARMENV_TASK_REACH && cfg.push_contact_model == 1 ? 1 : 0;
      const double c = 0.5 * cfg.push_gravity * cfg.push_dt * cfg.push_dt;
      int kl = 1;
      while (kl < 100000 && c * (double)kl * (double)(kl + 1) < cfg.push_drop_contact) ++kl;
      K.fall_land = kl;
      K.fall_c = (T)c;
      K.fall_keep = (T)(1.0 - cfg.push_drop_relax);
      K.place_z = (T)cfg.push_place_z;
      K.tool_radius = (T)cfg.push_tool_radius;
      K.tool_below = (T)cfg.push_tool_below;
      K.erp_dt = (T)(cfg.push_contact_erp / cfg.push_dt);
      K.split = (T)cfg.push_contact_split;
      K.fric_dv = (T)(cfg.push_friction * cfg.push_gravity * cfg.push_dt);
      K.dt = (T)cfg.push_dt;
    }
    K.seed = cfg.seed;
    K.env_id0 = cfg.env_id_offset;
    P.pick_gripper_length = (T)cfg.pick_gripper_length;
    P.pick_trigger_dis = (T)cfg.pick_trigger_dis;
    P.pick_jaw_half = (T)cfg.pick_jaw_half;

    P.dv = (T)cfg.dv;
    P.reach_dis = (T)cfg.reach_dis;
    P.max_steps = cfg.max_steps;
    P.auto_reset = cfg.auto_reset;
    P.seed = cfg.seed;
    P.env_id0 = cfg.env_id_offset;
    for (int k = 0; k < 3; ++k) {
      P.box_lo[k] = (T)cfg.box_lo[k]; P.box_hi[k] = (T)cfg.box_hi[k];
      K.goal_lo[k] = cfg.goal_lo[k]; K.goal_hi[k] = cfg.goal_hi[k];
    }
    double lim_min = 1e300;
    for (int j = 0; j < NJ; ++j) {
      K.q_init[j] = (T)cfg.q_init[j];
      K.lim[j] = (T)cfg.chain.limit_lo[j];
      K.lim[NJ + j] = (T)cfg.chain.limit_hi[j];
      K.lim[2 * NJ] = (T)cfg.limit_erp;
      // no joint can be outside [lower, upper] while max |q| <= min(-lower, upper) (as T: the values the kernel compares with)
      const double a = -(double)K.lim[j], b2 = (double)K.lim[NJ + j];
      lim_min = std::fmin(lim_min, std::fmin(a, b2));
    }
    P.ik.lim = cold_dev->lim;
    P.ik.lim_min = (T)(lim_min > 0.0 ? lim_min : -1.0);
    if ((double)P.ik.lim_min > lim_min) P.ik.lim_min = std::nextafter(P.ik.lim_min, (T)0);   // the cast must not round up
    for (int k = 0; k < 4; ++k) P.ik.tq[k] = (T)cfg.target_quat[k];
    P.ik.lambda = (T)cfg.ik_lambda;
    P.ik.residual = (T)cfg.ik_residual;
    P.ik.max_dtheta = (T)cfg.ik_max_dtheta;
    P.ik.max_iters = cfg.ik_max_iters;
    P.ik.exit_mode = cfg.ik_exit_mode;
    P.ik.angle_f32 = cfg.ik_angle_f32;
    P.ik.clamp_limits = cfg.clamp_joint_limits;
    waves_cfg = cfg.rollout_waves_per_simd;
    lanes_cfg = cfg.rollout_lanes_per_wave;
    ready_lanes = cfg.rollout_ready_lanes < 0 ? 0 : (cfg.rollout_ready_lanes > 64 ? 64 : cfg.rollout_ready_lanes);
    straggler_trips = cfg.rollout_straggler_trips < 0 ? 0 : cfg.rollout_straggler_trips;
    // MODE 2 of the lanes: a tip offset, or the f64 step diagnostics (fence_counters = 2)
    tip_on = cfg.ik_tip_offset[0] != 0.0 || cfg.ik_tip_offset[1] != 0.0 || cfg.ik_tip_offset[2] != 0.0 || cfg.fence_counters == 2;
    fence_on = cfg.fence_counters != 0 || tip_on;
    for (int k = 0; k < 3; ++k) P.ik.tip[k] = (T)cfg.ik_tip_offset[k];
    P.ik.fence_pivot = (T)cfg.fence_pivot;
    P.fence_z = (T)cfg.fence_z;
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
    HIP_TRY(hipMemcpy(cold_dev, &K, sizeof K, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((trig_identity_kernel<T>), dim3(grid_for(n, block)), dim3(block), 0, 0, P);
    hipLaunchKernelGGL((init_consts_kernel<C, T>), dim3(1), dim3(64), 0, 0, P, cold_dev);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    kname = std::string(Lane::kName) + "_step<" + (sizeof(T) == 8 ? "f64" : "f32") + "," + C::kName + ">";
    return ARMENV_OK;
  }

  int reset(const uint8_t *mask, const float *goal, float *obs, hipStream_t s) override {
    hipLaunchKernelGGL((env_reset_kernel<Lane, T>), dim3(grid_for(P.n, block)), dim3(block), 0, s, P, mask, goal, obs);
    HIP_TRY(hipGetLastError());
    return ARMENV_OK;
  }
  // launch geometry of the kernels without a workgroup phase: threads to cover every env, and the params with the lane mapping
  int64_t lane_threads(bool half) const { return half ? 64 * ((P.n + 31) / 32) : P.n; }
  EnvParams<T> params(bool half) const { EnvParams<T> Q = P; Q.half_waves = half ? 1 : 0; return Q; }
  int step(const StepIO &io, hipStream_t s) override {
    // the one-launch-per-step kernel keeps full waves: a launch ends with its slowest wave whatever the wave count, and twice
    // as many waves share the CUs' instruction fetch (push, 32 768 envs: 22.9 us per launch with full waves, 23.4 with half)
    const bool h = false;
    const int b = 64 * (int)std::min<int64_t>(4, std::max<int64_t>(1, ((P.n + 63) / 64 + cus - 1) / cus));
    if (fence_on) {
      with_book_lane([&](auto *tag) {
        using LX = std::remove_pointer_t<decltype(tag)>;
        if (two_waves()) hipLaunchKernelGGL((env_step_kernel<LX, T, 2>), dim3(grid_for(P.n, block)), dim3(block), 0, s, P, io);
        else hipLaunchKernelGGL((env_step_kernel<LX, T>), dim3(grid_for(lane_threads(h), b)), dim3(b), 0, s, params(h), io);
      });
    } else {
      if (two_waves()) hipLaunchKernelGGL((env_step_kernel<Lane, T, 2>), dim3(grid_for(P.n, block)), dim3(block), 0, s, P, io);
      else hipLaunchKernelGGL((env_step_kernel<Lane, T>), dim3(grid_for(lane_threads(h), b)), dim3(b), 0, s, params(h), io);
    }
    HIP_TRY(hipGetLastError());
    return ARMENV_OK;
  }
  // Two waves per SIMD (env_rollout_kernel<..., 2>) when the batch has more waves than the chip has SIMDs, or when the
  // caller asks for it (ArmEnvConfig.rollout_waves_per_simd); never with a fused actor.
  bool two_waves() const {
    if (waves_cfg == 1) return false;
    if (waves_cfg == 2) return true;
    return (P.n + 63) / 64 > (int64_t)4 * cus;
  }
  template <int POLICY, class LaneX = Lane>
  void launch_rollout(int32_t steps, const float *actions, const StepIO &io0, float *actions_out, hipStream_t s) {
    constexpr bool kActor = POLICY == ARMENV_POLICY_ACTOR || POLICY == ARMENV_POLICY_ACTOR_F16X3 || POLICY == ARMENV_POLICY_DATD3;
    if constexpr (!kActor) {
      if (two_waves()) {
        hipLaunchKernelGGL((env_rollout_kernel<LaneX, T, POLICY, 2>), dim3(grid_for(P.n, block)), dim3(block), 0, s, P, pol,
                           steps, actions, io0, actions_out);
        return;
      }
    }
    const int b = kActor ? block : lane_block();
    const bool h = !kActor && half_waves();
    hipLaunchKernelGGL((env_rollout_kernel<LaneX, T, POLICY>), dim3(grid_for(lane_threads(h), b)), dim3(b), 0, s, params(h), pol, steps,
                       actions, io0, actions_out);
  }
  template <int POLICY, class LaneX = Lane>
  void launch_rollout_async(int32_t steps, const float *actions, const StepIO &io0, float *actions_out, hipStream_t s) {
    if (two_waves()) {
      hipLaunchKernelGGL((env_rollout_async_kernel<LaneX, T, POLICY, 2>), dim3(grid_for(P.n, block)), dim3(block), 0, s, P, pol,
                         steps, actions, io0, actions_out, (int32_t)ready_lanes, (int32_t)straggler_trips);
      return;
    }
    const int b = lane_block();
    const bool h = half_waves();
    // count rule under half-filled waves: the same share of the live lanes (62 of 64 -> 31 of 32; 22.1 us vs 23.2 at 30)
    hipLaunchKernelGGL((env_rollout_async_kernel<LaneX, T, POLICY>), dim3(grid_for(lane_threads(h), b)), dim3(b), 0, s, params(h), pol, steps,
                       actions, io0, actions_out, (int32_t)(h ? (ready_lanes + 1) / 2 : ready_lanes), (int32_t)straggler_trips);
  }
  void launch_rollout_policy(int32_t steps, const float *actions, const StepIO &io0, float *actions_out, hipStream_t s) {
    // lane-asynchronous form (ArmEnvConfig.rollout_ready_lanes > 0): external actions or the in-kernel random policy
    const bool fused_actor = !actions && (pol.kind == ARMENV_POLICY_ACTOR || pol.kind == ARMENV_POLICY_ACTOR_F16X3 || pol.kind == ARMENV_POLICY_DATD3 ||
                                         pol.kind == ARMENV_POLICY_DADDPG);
    if (ready_lanes > 0 && steps > 1 && !fused_actor) {
      if (fence_on) {
        with_book_lane([&](auto *tag) {
          using LX = std::remove_pointer_t<decltype(tag)>;
          if (actions) launch_rollout_async<ARMENV_POLICY_EXTERNAL, LX>(steps, actions, io0, actions_out, s);
          else launch_rollout_async<ARMENV_POLICY_RANDOM, LX>(steps, actions, io0, actions_out, s);
        });
      } else {
        if (actions) launch_rollout_async<ARMENV_POLICY_EXTERNAL>(steps, actions, io0, actions_out, s);
        else launch_rollout_async<ARMENV_POLICY_RANDOM>(steps, actions, io0, actions_out, s);
      }
      return;
    }
    if (fence_on && !fused_actor) {   // the bookkeeping builds exist for external actions and the in-kernel random policy
      with_book_lane([&](auto *tag) {
        using LX = std::remove_pointer_t<decltype(tag)>;
        if (actions) launch_rollout<ARMENV_POLICY_EXTERNAL, LX>(steps, actions, io0, actions_out, s);
        else launch_rollout<ARMENV_POLICY_RANDOM, LX>(steps, actions, io0, actions_out, s);
      });
      return;
    }
    if (actions) launch_rollout<ARMENV_POLICY_EXTERNAL>(steps, actions, io0, actions_out, s);
    else if (pol.kind == ARMENV_POLICY_ACTOR) launch_rollout<ARMENV_POLICY_ACTOR>(steps, actions, io0, actions_out, s);
    else if (pol.kind == ARMENV_POLICY_ACTOR_F16X3) launch_rollout<ARMENV_POLICY_ACTOR_F16X3>(steps, actions, io0, actions_out, s);
    else if (pol.kind == ARMENV_POLICY_DATD3 || pol.kind == ARMENV_POLICY_DADDPG)   // DADDPG: the same kernel over a net table whose two critics are one
      launch_rollout<ARMENV_POLICY_DATD3>(steps, actions, io0, actions_out, s);
    else launch_rollout<ARMENV_POLICY_RANDOM>(steps, actions, io0, actions_out, s);
  }
  int rollout(int32_t steps, const float *actions, const StepIO &io0, float *actions_out, hipStream_t s) override {
    launch_rollout_policy(steps, actions, io0, actions_out, s);
    HIP_TRY(hipGetLastError());
    return ARMENV_OK;
  }
  int fk(int64_t n, const double *q, double *pos, double *quat, hipStream_t s) override {
    hipLaunchKernelGGL((fk_kernel<C, T>), dim3(grid_for(n, block)), dim3(block), 0, s, P, n, q, pos, quat);
    HIP_TRY(hipGetLastError());
    return ARMENV_OK;
  }
  int ik(int64_t n, const double *q, const double *tgt, double *q_out, int32_t *iters, hipStream_t s) override {
    if (tip_on) hipLaunchKernelGGL((ik_kernel<C, T, 2>), dim3(grid_for(n, block)), dim3(block), 0, s, P, n, q, tgt, q_out, iters);
    else hipLaunchKernelGGL((ik_kernel<C, T>), dim3(grid_for(n, block)), dim3(block), 0, s, P, n, q, tgt, q_out, iters);
    HIP_TRY(hipGetLastError());
    return ARMENV_OK;
  }
  int get_state(double *q, float *goal, int32_t *step, uint32_t *episode, double *ep_return, double *aux, double *trig,
                hipStream_t s) override {
    hipLaunchKernelGGL((get_state_kernel<T>), dim3(grid_for(P.n, block)), dim3(block), 0, s, P, q, goal, step, episode,
                       ep_return, aux, trig, (int)Lane::kAuxRows, (int)Lane::kAuxDim);
    HIP_TRY(hipGetLastError());
    return ARMENV_OK;
  }
  int set_state(const double *q, const float *goal, const int32_t *step, const uint32_t *episode,
                const double *ep_return, const double *aux, const double *trig, hipStream_t s) override {
    hipLaunchKernelGGL((set_state_kernel<T>), dim3(grid_for(P.n, block)), dim3(block), 0, s, P, q, goal, step, episode,
                       ep_return, aux, trig, (int)Lane::kAuxRows, (int)Lane::kAuxDim);
    HIP_TRY(hipGetLastError());
    return ARMENV_OK;
  }
  int episode_stats(double *last_return, int32_t *last_len, uint8_t *last_success, hipStream_t s) override {
    hipLaunchKernelGGL((episode_stats_kernel<T>), dim3(grid_for(P.n, block)), dim3(block), 0, s, P, last_return,
                       last_len, last_success);
    HIP_TRY(hipGetLastError());
    return ARMENV_OK;
  }
  int episode_returns_f32(float *last_return, hipStream_t s) override {
    hipLaunchKernelGGL((episode_returns_f32_kernel<T>), dim3(grid_for(P.n, block)), dim3(block), 0, s, P, last_return);
    HIP_TRY(hipGetLastError());
    return ARMENV_OK;
  }
  int counters(uint64_t out[16], hipStream_t s) override {
    hipLaunchKernelGGL(counters_sum_kernel, dim3(1), dim3(256), 0, s, P.counters, (P.n + 31) / 32, counter_totals);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, counter_totals, kCounterCols * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return ARMENV_OK;
  }
  int summary(double *out_dev, hipStream_t s) override {
    hipLaunchKernelGGL((env_summary_kernel<Lane, T>), dim3(grid_for(P.n, block)), dim3(block), 0, s, P, summary_rows);
    hipLaunchKernelGGL(summary_reduce_kernel, dim3(1), dim3(256), 0, s, summary_rows, (P.n + 63) / 64, out_dev);
    HIP_TRY(hipGetLastError());
    return ARMENV_OK;
  }
  const char *name() const override { return kname.c_str(); }
};


template <template <class, class, int> class LaneT, typename T> static EngineBase *make_task_engine(const ArmEnvConfig &cfg) {
  if (cfg.fk_path == ARMENV_FK_AUTO) {
    if (chain_matches<KukaChain>(cfg.chain)) return new (std::nothrow) Engine<LaneT, KukaChain, T>();
    if (chain_matches<DianaChain>(cfg.chain)) return new (std::nothrow) Engine<LaneT, DianaChain, T>();
  }
  return new (std::nothrow) Engine<LaneT, GenericChain, T>();
}

// the six translation units of armenv_task.hip
EngineBase *armenv_make_engine_reach_f64(const ArmEnvConfig &cfg) __attribute__((visibility("hidden")));
EngineBase *armenv_make_engine_reach_f32(const ArmEnvConfig &cfg) __attribute__((visibility("hidden")));
EngineBase *armenv_make_engine_push_f64(const ArmEnvConfig &cfg) __attribute__((visibility("hidden")));
EngineBase *armenv_make_engine_push_f32(const ArmEnvConfig &cfg) __attribute__((visibility("hidden")));
EngineBase *armenv_make_engine_pick_f64(const ArmEnvConfig &cfg) __attribute__((visibility("hidden")));
EngineBase *armenv_make_engine_pick_f32(const ArmEnvConfig &cfg) __attribute__((visibility("hidden")));
