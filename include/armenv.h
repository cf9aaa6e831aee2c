/*
 * armenv.h -- C ABI of the MI355X-native batched robot-arm environment engine (libarmenv.so).
 *
 * Drop-in boundary for the env hot path of Shimly-2/DRL-on-robot-arm.  The reference has no FFI of
 * its own (it is Python over the third-party pybullet C++ extension); the entry points below are
 * what a binding for that path replaces, cited as /root/reference file:line on each function.
 * A maintainer's ctypes stub is shown in INTEGRATION.md.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / HIP types in any signature.
 *   - `*_dev` pointers are DEVICE pointers owned by the caller (e.g. torch tensor.data_ptr()),
 *     valid until the work enqueued on `stream` has completed.  `stream` is a hipStream_t passed
 *     as void* (NULL = the default stream).  No call allocates, frees or synchronises unless it
 *     says so; step/reset only enqueue kernels.
 *   - Every function returns ARMENV_OK (0) or a negative error code; the message is available
 *     from armenv_last_error() (thread-local).  Nothing aborts.
 *   - A handle is not thread-safe; distinct handles (one per GPU / per stream) are independent.
 *   - There is no CPU fallback: creating a handle without a usable HIP device fails with
 *     ARMENV_ENODEV.
 *
 * Layouts (N = num_envs):
 *   action  f32 [N][3]      obs  f32 [N][obs_dim]  (reach: 6 = [eef xyz, goal xyz];
 *                                                    push / pick: 9 = [eef, cube, target])
 *   reward  f32 [N]         done / success  u8 [N]
 *   state exchange (get/set_state): q f64 [N][7], goal f32 [N][3], step i32 [N],
 *                                   episode u32 [N], ep_return f64 [N], trig f64 [N][14] = cos q[7], sin q[7]
 */
#ifndef ARMENV_H
#define ARMENV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ARMENV_NJ 7
#define ARMENV_ABI_VERSION 6   /* 6: + armenv_set_policy_daddpg, armenv_episode_returns_f32, ARMENV_POLICY_DADDPG; ArmEnvConfig unchanged since 5 */

enum {
  ARMENV_OK = 0,
  ARMENV_EINVAL = -1,   /* bad argument / config */
  ARMENV_ENODEV = -2,   /* no HIP device or wrong device index */
  ARMENV_ENOMEM = -3,   /* device allocation failed */
  ARMENV_EHIP = -4,     /* a HIP runtime call failed (message has the HIP error string) */
  ARMENV_ESTATE = -5    /* call not valid in the handle's current state (e.g. actor not set) */
};

enum { ARMENV_TASK_REACH = 0, ARMENV_TASK_PUSH = 1, ARMENV_TASK_PICK = 2 };
enum { ARMENV_ROBOT_KUKA = 0, ARMENV_ROBOT_DIANA = 1 };
enum { ARMENV_FK_AUTO = 0, ARMENV_FK_GENERIC = 1 };
enum {
  ARMENV_POLICY_EXTERNAL = 0,
  ARMENV_POLICY_RANDOM = 1,
  ARMENV_POLICY_ACTOR = 2,        /* TD3 actor, exact f32 (layers 1 and 2 on the f32-input MFMA) */
  ARMENV_POLICY_ACTOR_F16X3 = 3,  /* same actor, layer 2 on the f16 MFMA with 3-pass hi/lo operand splitting (~1e-6) */
  ARMENV_POLICY_DATD3 = 4,        /* DATD3_MLP / DARC_MLP.take_action: two actors, two critics, the better-valued action (armenv_set_policy_datd3) */
  ARMENV_POLICY_DADDPG = 5        /* DADDPG_MLP.take_action: two actors, ONE critic on both proposals (armenv_set_policy_daddpg) */
};

typedef struct ArmEnv ArmEnv;

/* 7-revolute serial chain, every joint axis +z of its child frame (both robots the reference
 * uses: /root/reference/envs/bmirobot_joints_info_pybullet.txt field 14).  URDF semantics:
 * child frame = parent frame * Trans(origin_xyz) * Rz(yaw)Ry(pitch)Rx(roll) * Rz(q). */
typedef struct ArmEnvChain {
  double origin_xyz[ARMENV_NJ][3];
  double origin_rpy[ARMENV_NJ][3];
  double limit_lo[ARMENV_NJ];
  double limit_hi[ARMENV_NJ];
  double base_xyz[3];
  double base_rpy[3];
} ArmEnvChain;

typedef struct ArmEnvConfig {
  int32_t abi_version;     /* must be ARMENV_ABI_VERSION */
  int32_t device;          /* HIP device ordinal */
  int64_t num_envs;        /* N */
  int32_t task;            /* ARMENV_TASK_* */
  int32_t precision;       /* 64: state + arithmetic in f64 (the reference's number type); 32: f32 */
  int32_t fk_path;         /* ARMENV_FK_AUTO picks the signed-permutation fast path for KUKA / Diana */
  int32_t auto_reset;      /* 1: an env that finishes is reset inside the same step call */
  uint64_t seed;           /* Philox key */
  uint64_t env_id_offset;  /* global index of env 0 (rank * N when sharding across GPUs) */

  /* task constants; armenv_default_config() fills the reference's values */
  double dv;               /* /root/reference/config.py:41 (reach 0.02); envs/rl_push_env.py:322 (push 0.08) */
  double reach_dis;        /* config.py:42 */
  int32_t max_steps;       /* config.py:51 ; done when step_counter > max_steps (rl_reach_env.py:299) */
  int32_t clamp_joint_limits; /* what stepSimulation (:258) does to a joint the IK left outside chain.limit_lo/hi (the URDF limits
                                 of envs/bmirobot_joints_info_pybullet.txt:1-7 fields 8-9; the reference never passes them to
                                 the IK: rl_reach_env.py:103-107 are dead data, :244-250).
                                 0 = nothing (this build's kinematic stepSimulation; default) -- what real PyBullet does, by the
                                     reference's own recorded run: its first five episodes contain 276 steps with a joint beyond
                                     its limit and are reproduced to 4e-7 relative at 0, missed by 1e1..1e2 at 1 or 2
                                     (tests/reference_run.py, tests/tools/fit_bullet.py);
                                 1 = project onto the limits -- the hard-limit idealisation of Bullet's joint-limit constraint;
                                 2 = move the joint back by the share limit_erp of its violation per step -- one Baumgarte-
                                     stabilised constraint solve per stepSimulation, as btMultiBodyJointLimitConstraint does.
                                 Modes 1 and 2 are named models of a hard / soft limit for callers who want one; they are NOT what the
                                 reference's engine does. */
  double box_lo[3];        /* Cartesian clip, rl_reach_env.py:221-223 */
  double box_hi[3];
  double goal_lo[3];       /* target sampling box, rl_reach_env.py:65-70,180-182 */
  double goal_hi[3];
  double target_quat[4];   /* xyzw, rl_reach_env.py:121-122 */
  double q_init[ARMENV_NJ];/* rl_reach_env.py:116-119 */

  /* IK: pybullet.calculateInverseKinematics defaults for the call at rl_reach_env.py:244-250 */
  double ik_lambda;        /* jointDamping 1e-5, rl_reach_env.py:111-113 */
  double ik_residual;      /* 1e-4 */
  double ik_max_dtheta;    /* 45 deg */
  int32_t ik_max_iters;    /* 20 */
  int32_t ik_exit_mode;    /* 0 Bullet loop, 1 test-before-update */
  int32_t ik_angle_f32;    /* 1: orientation-error angle rounded through f32 as Bullet does */
  int32_t fence_counters;  /* 1: count the env steps on which this build's kinematic stepSimulation is known to differ from
                              Bullet's -- the IK result lies outside the URDF joint limits (Bullet's limit constraint pushes
                              back), or the step ends with the flange below fence_z (arm-table contact) -- and the env steps on
                              which no two implementations of the reference's algorithm can be expected to agree, Bullet's own
                              included: the IK call ran to ik_max_iters without converging (the update oscillates), or one of
                              its damped systems was ill-conditioned (fence_pivot) -- in armenv_counters out[5] / out[6] / out[7] /
                              out[8].  Every parity claim is fenced by these four rates (bench.py reports them for its
                              workloads).  0 (default): no bookkeeping in the step.  2: the same plus the f64 step diagnostics
                              (armenv_step / armenv_rollout diag_dev) -- a third build of the kernels, kept apart from 1 because one
                              more output pointer costs the bookkeeping build 5-10 %.  Needs ik_max_iters <= 254 (the per-step update
                              count is reported in a u8, the wave's trip maximum folded over 8 bits). */

  /* push task, /root/reference/envs/rl_push_env.py (the pick task, envs/rl_pick_env.py, shares all six) */
  double push_success_dis; /* 0.05  :422 (pick :425) */
  double push_cube_half;   /* 0.02  models/cube_small_push.urdf */
  double push_eef_radius;  /* pusher radius of the simplified contact model (pick: radius of the gripper tip) */
  double push_rest_z;      /* z at which the cube rests on the table: -0.00474 = 14.74 mm below its spawn height, fitted to the
                              reference's recorded push run */
  double push_place_min;   /* 0.22  :213 (pick :207) */
  double push_place_max;   /* 0.25  :213 (pick :207) */
  double push_place_z;     /* 0.01  :199,206 (pick :194): height at which cube and (push) target are spawned; the placement test sees
                              both there, the fixed target stays, the cube settles to push_rest_z */

  /* pick task, /root/reference/envs/rl_pick_env.py: gripper model (build-defined) */
  double pick_gripper_length; /* 0.257 :79 -- the gripper tip sits this far along the tool axis from the link-7 frame */
  double pick_trigger_dis;    /* 0.006 :412 -- p.getClosestPoints distance that closes the gripper */
  double pick_jaw_half;       /* the closing gripper holds the cube when the cube centre is within this horizontal
                                 distance of the tool axis (default: push_cube_half) */
  double fence_z;             /* 0.05 */
  double fence_pivot;         /* 1e-2: an IK call is counted as ill-conditioned (armenv_counters out[8]) when an LDL^T pivot of
                                 one of its damped systems J J^T + ik_lambda I fell below this -- the arm passed through a
                                 near-singular pose (stretched elbow at the edge of its reach, aligned wrist), where the damped
                                 solve amplifies rounding differences by ~1 / pivot and the trajectories of any two
                                 implementations start to part */
  double limit_erp;           /* clamp_joint_limits == 2: share of a joint-limit violation removed per step (Bullet's default
                                 constraint error-reduction parameter, 0.2) */
  double ik_tip_offset[3];    /* The point of link 7, in the link-7 frame, at which calculateInverseKinematics (rl_reach_env.py:244-250)
                                 takes its position error and its linear Jacobian.  Default (0,0,0): the URDF link frame, the point
                                 p.getLinkState(body, 6)[4] reports (:237).  (0,0,0.02) is the KUKA link-7 inertial origin: Bullet's
                                 multibody link frames sit at the centre of mass, and whether pybullet 3.0.6 runs the IK there was
                                 the one unknown of the restatement that had no switch before ABI 4; the reference's recorded run
                                 decides it: link frame (the inertial setting misses the recorded returns by 1e3).  The
                                 target clip (:231-242), the reward and the observation always use the link frame.  A non-zero
                                 offset selects the MODE 2 bookkeeping build of the kernels (as fence_counters = 2 does; it is a fitting
                                 switch for tests/tools/fit_bullet.py, not a tuned path) and excludes the fused actors. */

  /* push task: the cube under stepSimulation (/root/reference/envs/rl_push_env.py:349; models/cube_small_push.urdf: 4 cm box, 1 kg,
   * lateral friction 5, pushed by link meshes that resetJointState teleports with zero velocity, :339-347).  Bullet's rigid-body step
   * over the KUKA meshes is not restated; push_contact_model names what stands in for it:
   *   1 (default, ABI 5): the cube as a planar point mass under Bullet's step order -- collision detection at the positions the
   *     step starts from, velocity-level contact (the cube's velocity along the contact normal is raised to push_contact_erp x
   *     penetration / push_dt; penetrations beyond push_contact_split get no positional correction), Coulomb friction against the
   *     table (push_friction x push_gravity), integration -- against a vertical tool cylinder (push_tool_radius about the link-7
   *     frame, reaching push_tool_below under it); a tool that is less deep in the cube from above than from the side presses it onto
   *     the table and moves nothing.  The cube's HEIGHT follows Bullet exactly as far as the reference's recorded runs can tell: spawned
   *     at push_place_z it is in free fall (semi-implicit Euler, push_gravity, push_dt) from reset()'s own stepSimulation (:241)
   *     through the step in which its drop passes push_drop_contact, then recovers the overshoot towards push_rest_z by the share
   *     push_drop_relax per step.  The fall has no fitted number but the rest height and reproduces the untouched episodes of both
   *     recorded runs (visdata/push/origin_TD3 and updata_TD3: tests/reference_run.py) to 3e-4.  The tool geometry is the KUKA flange's
   *     nominal one; push_contact_erp and push_friction are FITTED to the touched episodes (tests/tools/fit_bullet.py part C, DESIGN.md
   *     section 2): counts of steps on which the cube moves 149 / 192 / 90 / 43 against Bullet's 149 / 192 / 84 / 32, returns under the
   *     shipped reward within 12.4 (rounds 1-4: 25-195 off), final cube-target distances within 4.3 cm.  Effective values of this planar
   *     stand-in, not Bullet's contact parameters, and PROVISIONAL: checked in-sample only.  (Round 6 tested the alternative the data
   *     suggests -- a cube that can tip, Bullet's own ERP 0.2 and friction 2.5, nothing fitted: it explains the moving-step counts the
   *     planar model needs an 80 x too slippery table for, and misses the returns by 38 or more, chaotically in the contact height:
   *     profiles/r06_push_rocking_model.txt.  Not adopted.)
   *   0: rounds 1-4 -- tool sphere of push_eef_radius at the link-7 frame, the whole penetration removed in one step, the cube
   *     already at rest at push_rest_z after reset().
   * The pick task loads the SAME body into the same scene (rl_pick_env.py:210): with push_contact_model = 1 its cube falls the same way
   * -- RLPickEnv calls stepSimulation twice per env step (:348, and :417 behind the observation), so it lands within seven env steps;
   * the state between two steps holds the cube as the last observation showed it --
   * unless the gripper holds it; its gripper tip keeps the model-0 contact (build-defined). */
  double push_tool_radius;    /* 0.045: the flange's radius */
  double push_tool_below;     /* 0.045: the flange face below the link-7 frame */
  double push_contact_erp;    /* 0.01, fitted (Bullet's contact ERP, 0.2, is the share against an immovable body; link 7 weighs 0.3 kg) */
  double push_contact_split;  /* 0.04: Bullet's m_splitImpulsePenetrationThreshold */
  double push_friction;       /* 0.03, fitted */
  double push_gravity;        /* 10: p.setGravity(0, 0, -10), :155 */
  double push_dt;             /* 1 / 240: Bullet's default fixed time step */
  double push_drop_contact;   /* 0.015 = spawn height 0.01 - cube half 0.02 - table top -0.025 (pybullet_data table.urdf based at z = -0.65, :189) */
  double push_drop_relax;     /* 0.1; the recorded run bounds it to (0, 0.14): no recovery step moves the cube-target distance by 1e-5 */
  int32_t push_contact_model; /* 1 */
  int32_t reserved0;          /* must be 0 */

  /* armenv_rollout scheduling.  0: lockstep -- the lanes of a wavefront walk through every step together (a step costs the
   * wave its slowest lane's IK trips).  k in 1..64: lane-asynchronous -- a lane whose IK has stopped waits until k lanes of
   * its wave are waiting (or nobody iterates), then they finish their step and start the next one while the others carry on;
   * same trajectories bit for bit.  Defaults: reach and push 0 (their lanes mostly agree: 4.00 / 5.5 trips per wave-step for
   * 3.92 / 4.1 per env-step, and every transition round of the asynchronous form costs a tail block), pick 62 (0.85 % of its
   * env-steps run Bullet's loop to the 20-iteration cap: 12.2 trips per wave-step for 4.45 per env-step in lockstep;
   * DESIGN.md section 4).  Ignored with a fused actor. */
  int32_t rollout_ready_lanes;
  /* Register budget of the step and rollout kernels.  1: one wave per SIMD with the whole register file (the shape of a batch
   * of up to 64 x #SIMDs envs: 65 536 on MI355X).  2: at most 256 registers per lane so that two waves share a SIMD -- the
   * right shape for larger batches (+17 % at 1 048 576 envs), same bits.  0 (default): chosen from num_envs and the device's
   * CU count.  Ignored with a fused actor (always 1). */
  int32_t rollout_waves_per_simd;
  /* Envs per wavefront of the step and rollout kernels.  64: full waves.  32: half-filled waves (lanes 0..31 carry envs) -- for a
   * batch of at most 32 x #SIMDs envs (32 768 on MI355X: BASELINE config 4) twice as many SIMDs get a wave, at no cost in issue
   * slots, and a wave's per-step maximum of IK trips is taken over 32 lanes (push 5.67 -> 5.36 trips per wave-step); same bits.
   * 0 (default): 32 for push / pick when the half-filled waves still get one SIMD each, 64 otherwise.  Ignored with a fused actor. */
  int32_t rollout_lanes_per_wave;
  /* Transition rule of the lane-asynchronous schedule (rollout_ready_lanes > 0).  0: by count (rollout_ready_lanes lanes wait).
   * K > 0: a round starts when every lane that has spent fewer than K IK trips on its current step is waiting -- the lanes that
   * are on their way to Bullet's iteration cap carry on, however many there are, and all the others stay in phase (a count
   * lets at most 64 - k slow lanes run on, and a lane that merely needs one trip more than its neighbours drops out of phase
   * with them).  Default: pick 6, others 0.  Occupies the int32 that ABI 3 reserved (must-be-zero) at this place: a caller
   * that zero-fills it keeps the count rule.  Same bits under every rule. */
  int32_t rollout_straggler_trips;

  ArmEnvChain chain;
} ArmEnvConfig;

/* Fills `cfg` with the constants of RLReachEnv.__init__ / RLPushEnv.__init__ / RLPickEnv.__init__
 * (/root/reference/envs/rl_reach_env.py:44-125, envs/rl_push_env.py:49-143, envs/rl_pick_env.py:51-133), Bullet's IK defaults and
 * the KUKA iiwa chain.  num_envs is set to 1, precision to 64, auto_reset to 1. */
int armenv_default_config(int32_t task, ArmEnvConfig *cfg);

/* Built-in chains: KUKA iiwa (the robot reach/push/pick load, rl_reach_env.py:174) and Diana S1
 * (/root/reference/models/diana/DianaS1_robot.urdf, loaded by envs/diana_cam_reach.py:201). */
int armenv_builtin_chain(int32_t robot, ArmEnvChain *out);

/* Replaces RLReachEnv.__init__ minus its implicit reset (rl_reach_env.py:44-125): allocates the
 * per-env state on `cfg->device`.  Until the first armenv_reset every env sits at q = 0 with a zero goal. */
int armenv_create(const ArmEnvConfig *cfg, ArmEnv **out);
void armenv_destroy(ArmEnv *env);

/* Replaces RLReachEnv.reset / RLPushEnv.reset / RLPickEnv.reset (rl_reach_env.py:132-217, rl_push_env.py:145-256,
 * rl_pick_env.py:140-252) for
 * the envs whose mask byte is non-zero (mask_dev == NULL: all).  Goals come from the engine's
 * Philox stream.  obs_dev (nullable) receives the first observation of the reset envs only. */
int armenv_reset(ArmEnv *env, const uint8_t *mask_dev, float *obs_dev, void *stream);

/* Same, with caller-supplied goals f32 [N][3] (reach) -- the N=1 compatibility class uses this to
 * keep the reference's Python `random` stream (rl_reach_env.py:180-183). Push / pick: goal_dev is
 * f32 [N][6] = [cube xyz, target xyz]; push with push_contact_model = 1 takes the cube's xy from it and lets the cube fall from
 * push_place_z itself (its height is a function of the step count). */
int armenv_reset_with_goal(ArmEnv *env, const uint8_t *mask_dev, const float *goal_dev, float *obs_dev,
                           void *stream);

/* Replaces RLReachEnv.step + _reward (rl_reach_env.py:219-319) / RLPushEnv.step + _reward
 * (rl_push_env.py:310-440) / RLPickEnv.step + _reward (rl_pick_env.py:310-440) for all N envs in one fused kernel:
 *   FK -> add dv*action -> clip to the workspace box -> DLS IK -> FK -> (push / pick: cube contact, gripper) ->
 *   distance/reward/done -> obs.
 * action_dev may be NULL when a fused policy was installed with armenv_set_policy.
 * terminal_obs_dev (nullable, f32 [N][obs_dim]) receives the observation of this step before any
 * auto-reset; with auto_reset the obs of a finished env is its next episode's first observation.
 * ik_updates_dev (nullable, u8 [N]) receives the number of DLS updates the step's calculateInverseKinematics call
 * (:244-250) applied: ik_max_iters means the call did not converge (the cap term of the parity fence).
 * diag_dev (nullable, f64 [N][4]) receives what _reward (:267-309) computed in the reference's number type before anything was
 * rounded to f32: [0..2] the end-effector position of getLinkState(...)[4] (:271) -- the very numbers this step's distance,
 * done and success flags come from -- and [3] the reward as a double (the reference returns a Python float).
 * ik_updates_dev and diag_dev are diagnostics: only on a handle created with fence_counters >= 1 (ik_updates_dev) / = 2
 * (diag_dev); ARMENV_ESTATE otherwise. */
int armenv_step(ArmEnv *env, const float *action_dev, float *obs_dev, float *reward_dev, uint8_t *done_dev,
                uint8_t *success_dev, float *terminal_obs_dev, uint8_t *ik_updates_dev, double *diag_dev, void *stream);

/* Replaces the rollout inner loop of /root/reference/main.py:108-128 (take_action -> noise -> step -> store) for
 * `steps` consecutive env steps of all N envs in ONE kernel launch; the env state stays in registers between steps.
 *   actions_dev  f32 [steps][N][3]: external policy -- exactly the trajectory of `steps` armenv_step calls, bit for bit, for
 *                any `steps` and any split of the same steps into several launches ((cos q, sin q) of the joints travel
 *                with q in the handle's state).
 *                NULL: the fused policy installed with armenv_set_policy produces the actions in-kernel.
 *   obs_dev f32 [steps][N][obs_dim], reward_dev f32 [steps][N], done_dev / success_dev u8 [steps][N]: row t holds
 *   what armenv_step would have returned at step t.  actions_out_dev (nullable, f32 [steps][N][3]) receives the
 *   actions taken; terminal_obs_dev (nullable), ik_updates_dev (nullable, u8 [steps][N]) and diag_dev (nullable, f64 [steps][N][4])
 *   as in armenv_step, per step.
 * Waves never synchronise inside the launch, so an env that needs extra IK iterations in one step stalls only its own
 * wavefront for that step (lockstep schedule) or only itself (lane-asynchronous schedule, ArmEnvConfig.rollout_ready_lanes):
 * throughput follows the mean IK cost per step, not the per-launch maximum. */
int armenv_rollout(ArmEnv *env, int32_t steps, const float *actions_dev, float *obs_dev, float *reward_dev,
                   uint8_t *done_dev, uint8_t *success_dev, float *actions_out_dev, float *terminal_obs_dev,
                   uint8_t *ik_updates_dev, double *diag_dev, void *stream);

/* Replaces p.getLinkState(body, 6)[4], [5] (call sites rl_reach_env.py:202,237,271): world position
 * f64 [n][3] and orientation quaternion xyzw f64 [n][4] (nullable) of the link-7 frame for joint
 * vectors q f64 [n][7].  Uses the handle's chain; n is independent of num_envs. */
int armenv_fk(ArmEnv *env, int64_t n, const double *q_dev, double *pos_dev, double *quat_dev, void *stream);

/* Replaces p.calculateInverseKinematics(body, 6, pos, orn, jointDamping) (rl_reach_env.py:244-250):
 * q_out f64 [n][7] from start q f64 [n][7] and target position f64 [n][3]; orientation target is the
 * handle's target_quat.  iters_dev (nullable) receives the number of DLS updates applied. */
int armenv_ik(ArmEnv *env, int64_t n, const double *q_dev, const double *target_pos_dev, double *q_out_dev,
              int32_t *iters_dev, void *stream);

/* State exchange for teacher-forced parity tests and checkpointing (the reference keeps this state
 * inside the PyBullet client: joint angles via resetJointState rl_reach_env.py:252-257, target via
 * loadURDF :186-189, step_counter :264).  Any pointer may be NULL to skip that field. Push and pick keep their
 * cube state in aux f64 [N][armenv_aux_dim()]: push [N][10] = [cube xyz, target xyz, d_last, cube velocity xy, pad]; pick [N][12] =
 * [cube xyz, target xyz, d_last, gripper (0 open, 1 closed, 2 closed and holding the cube), cube - tip offset xyz
 * while held, pad].
 * trig f64 [N][14] = (cos q[7], sin q[7]) is the pair the engine carries with q and advances incrementally (DESIGN.md
 * section 4): a checkpoint that restores q AND trig continues the uninterrupted trajectory bit for bit; set_state with q
 * and trig_dev == NULL re-derives the pair from q (resetJointState semantics: equal to ~1e-16, which the IK's 2 acos(w)
 * orientation error can amplify to ~1e-7 rad over the following steps). */
int armenv_get_state(ArmEnv *env, double *q_dev, float *goal_dev, int32_t *step_dev, uint32_t *episode_dev,
                     double *ep_return_dev, double *aux_dev, double *trig_dev, void *stream);
int armenv_set_state(ArmEnv *env, const double *q_dev, const float *goal_dev, const int32_t *step_dev,
                     const uint32_t *episode_dev, const double *ep_return_dev, const double *aux_dev,
                     const double *trig_dev, void *stream);

/* Per-env statistics of the most recently finished episode (what main.py:125-130 accumulates on
 * the host: episode_return, success).  Any pointer may be NULL. */
int armenv_episode_stats(ArmEnv *env, double *last_return_dev, int32_t *last_len_dev, uint8_t *last_success_dev,
                         void *stream);

/* The same returns as ONE f32 vector [N]: the send buffer of the logging all-gather (what main.py:130,150 plot from a single env; SURVEY.md
 * section 8e: ncclAllGather(episode_return_local[N/R] f32)) written by one kernel on `stream`, no intermediate f64 vector. */
int armenv_episode_returns_f32(ArmEnv *env, float *last_return_dev, void *stream);

/* Totals since creation, copied to host (synchronises `stream`), out[16]: out[0] episodes finished, out[1] successes,
 * out[2] env-steps executed, out[3] non-finite joint states seen, out[4] IK (DLS) updates applied, out[5] env steps whose IK
 * result left the URDF joint limits, out[6] env steps that ended with the flange below fence_z, out[7] env steps whose IK
 * call ran to ik_max_iters, out[8] env steps whose IK call passed through an ill-conditioned system (fence_pivot);
 * out[5..8]: the parity fence, counted only with ArmEnvConfig.fence_counters.  Also only with fence_counters, for
 * armenv_rollout launches without a fused actor: out[9] IK trips as the wavefronts paid them (a trip of a wave counts once,
 * however few of its lanes took part; out[9] / (waves * steps) is the schedule's trips per wave-step, to hold against
 * out[4] / out[2] + 1 per env-step), out[10] step tails as the wavefronts paid them (lockstep: one per step; lane-asynchronous:
 * one per transition round).  out[11..15] 0 (reserved). */
int armenv_counters(ArmEnv *env, uint64_t out[16], void *stream);

/* Logging summary computed on the device (no host sync; what main.py:130-160 prints/plots from one env): out_dev f64 [8] =
 * [sum over envs of the current distance to the goal (reach: |FK(q) - goal|, push: |cube - target|), max of it, sum of
 * the last finished episodes' returns, sum of their lengths, sum of their success flags, number of envs, 0, 0].
 * Wavefront shuffles reduce each wave's 64 envs into one row per wave; a one-workgroup kernel folds the rows in a fixed order
 * (no atomics: bitwise reproducible). */
int armenv_summary(ArmEnv *env, double *out_dev, void *stream);

/* Installs the TD3 actor (PolicyNet, /root/reference/algo/TD3/net_mlp.py:29-40; take_action
 * algo/TD3/TD3_mlp.py:82-97) for fused stepping: a = action_bound * tanh(W3 relu(W2 relu(W1 s + b1) + b2) + b3),
 * then the rollout loop's exploration a = clip(a + N(0, noise_sigma), +-noise_clip) (main.py:116-117).
 * Weights are DEVICE pointers in torch Linear layout ([out][in], f32) and are copied.
 * policy = ARMENV_POLICY_RANDOM ignores the weights (zero actor, noise only); ARMENV_POLICY_EXTERNAL removes the
 * fused policy.  The noise of (env, episode, step) is Philox block 0x80000000|step of that env's stream, Box-Muller
 * in f32, so it does not depend on launch geometry, sharding or how steps are grouped into rollouts. */
int armenv_set_policy(ArmEnv *env, int32_t policy, const float *W1_dev, const float *b1_dev, const float *W2_dev,
                      const float *b2_dev, const float *W3_dev, const float *b3_dev, int32_t hidden_dim,
                      float action_bound, float noise_sigma, float noise_clip, void *stream);

/* One three-layer perceptron of the reference's net_mlp.py (PolicyNet :29-40 or QValueNet :43-58): DEVICE pointers, torch Linear
 * layout ([out][in], f32): W1 [hidden][in], b1 [hidden], W2 [hidden][hidden], b2 [hidden], W3 [out][hidden], b3 [out]. */
typedef struct ArmEnvMlp {
  const float *W1, *b1, *W2, *b2, *W3, *b3;
} ArmEnvMlp;

/* Installs DATD3_MLP.take_action (/root/reference/algo/DATD3/DATD3_mlp.py:88-109) as the fused policy of armenv_rollout /
 * armenv_step(action_dev = NULL):
 *     a1 = actor1(s), a2 = actor2(s), q1 = critic1(cat(s, a1)), q2 = critic2(cat(s, a2)), a = a1 if q1 >= q2 else a2,
 * then the rollout loop's exploration a = clip(a + N(0, noise_sigma), +-noise_clip) as in armenv_set_policy.  The four networks
 * run inside the rollout kernel as four passes of the f16x3 MFMA actor (f32 emulated by three f16 passes, ~1e-6 of f32; the
 * arg-max can differ from an f32 evaluation only where |q1 - q2| is of that order).  actors: in = obs_dim, out = 3, tanh x
 * action_bound; critics: in = obs_dim + 3, out = 1, no output activation (obs_dim 6 reach, 9 push / pick).  The weights are copied.
 * hidden_dim 256; num_envs must be a multiple of 64; not on a bookkeeping handle (fence_counters). */
int armenv_set_policy_datd3(ArmEnv *env, const ArmEnvMlp *actor1, const ArmEnvMlp *actor2, const ArmEnvMlp *critic1,
                            const ArmEnvMlp *critic2, int32_t hidden_dim, float action_bound, float noise_sigma, float noise_clip,
                            void *stream);

/* Installs DADDPG_MLP.take_action (/root/reference/algo/DADDPG/DADDPG_mlp.py:77-97; opt.algo's default, /root/reference/config.py:33)
 * as the fused policy of armenv_rollout / armenv_step(action_dev = NULL):
 *     a1 = actor1(s), a2 = actor2(s), q1 = critic(cat(s, a1)), q2 = critic(cat(s, a2)), a = a1 if q1 >= q2 else a2
 * -- the selection rule of armenv_set_policy_datd3 (`>=`: a tie takes actor 1) with ONE critic valuing both proposals: three networks
 * are packed and staged, and the critic's second pass runs on the tables and the W2 ring its first pass left in LDS.
 * DARC_MLP.take_action (/root/reference/algo/DARC/DARC_mlp.py:92-113) is DATD3's, two critics: install it with
 * armenv_set_policy_datd3.  armenv_datd3_forward serves all three. */
int armenv_set_policy_daddpg(ArmEnv *env, const ArmEnvMlp *actor1, const ArmEnvMlp *actor2, const ArmEnvMlp *critic,
                             int32_t hidden_dim, float action_bound, float noise_sigma, float noise_clip, void *stream);

/* The installed DATD3 / DARC / DADDPG policy alone (take_action without noise) for n states f32 [n][obs_dim]: actions f32 [n][3] and, nullable, the two
 * Q values f32 [n] and the index of the actor whose action was taken u8 [n] (0: actor1, 1: actor2). */
int armenv_datd3_forward(ArmEnv *env, int64_t n, const float *states_dev, float *actions_dev, float *q1_dev, float *q2_dev,
                         uint8_t *picked_dev, void *stream);

/* The installed actor alone (TD3_MLP.take_action without noise, /root/reference/algo/TD3/TD3_mlp.py:82-97):
 * states f32 [n][obs_dim] -> actions f32 [n][3]; both layers on the MFMA, exact f32 or the f16x3 emulation according to
 * the installed policy.  Needs a prior armenv_set_policy(ARMENV_POLICY_ACTOR | ARMENV_POLICY_ACTOR_F16X3, ...). */
int armenv_actor_forward(ArmEnv *env, int64_t n, const float *states_dev, float *actions_dev, void *stream);

/* ---- trajectory store + HER-"future" sampler (consumer of the rollout buffers; replaces the per-sample Python loops of
 * /root/reference/utils/rl_utils.py:108-152 and :154-199).  Buffers are the time-major tensors armenv_rollout writes:
 * obs0 f32 [N][D] (observation before step 0), obs_after f32 [T][N][D] (= obs_dev), next_obs f32 [T][N][D]
 * (= terminal_obs_dev), action f32 [T][N][3], reward f32 [T][N], done u8 [T][N].  A trajectory (rl_utils.py:91-105) is one
 * complete episode of one env inside the chunk. */

/* Episode index, two passes around a caller-side inclusive prefix sum (e.g. torch.cumsum) of counts:
 *   armenv_count_episodes -> counts_dev i32 [N];   offsets = inclusive cumsum(counts) as i64 [N];
 *   armenv_write_episodes -> episodes_dev i32 [E][3] = (env, t_start, length), E = offsets[N-1].
 * starts_at_reset != 0: every env was reset right before step 0 of the chunk.
 * The [T] axis may be a ring of ring_cap >= T physical rows: logical step t (0 = oldest kept) lives in physical row
 * (ring_base + t) % ring_cap; a plain chunk is ring_base = 0, ring_cap = T. */
int armenv_count_episodes(int32_t device, int64_t T, int64_t N, int64_t ring_base, int64_t ring_cap, const uint8_t *done_dev,
                          int32_t starts_at_reset, int32_t *counts_dev, void *stream);
int armenv_write_episodes(int32_t device, int64_t T, int64_t N, int64_t ring_base, int64_t ring_cap, const uint8_t *done_dev,
                          int32_t starts_at_reset, const int32_t *counts_dev, const int64_t *offsets_dev,
                          int32_t *episodes_dev, void *stream);

typedef struct ArmEnvHerArgs {
  int64_t T, N, ring_base, ring_cap;
  int32_t obs_dim;            /* 6: reach relabel rule (rl_utils.py:140-141); 9: push rule (:187-188) */
  int32_t use_her;
  const float *obs0_dev, *obs_after_dev, *next_obs_dev, *action_dev, *reward_dev;
  const uint8_t *done_dev;
  const int32_t *episodes_dev;      /* [E][3] */
  const int64_t *num_episodes_dev;  /* device scalar E (no host sync needed to size the draw) */
  int64_t batch;                    /* B */
  const int32_t *picks_dev;         /* nullable i32 [B][4] = (episode, step_state, use_her, step_goal): caller-supplied
                                       draws (parity tests feed the reference's own draws); NULL: Philox(seed, draw) */
  uint64_t seed, draw;
  float her_ratio, dis_threshold;   /* 0.8 (config.py:80), 0.1 (rl_utils.py:119) */
  float *states_dev;                /* out f32 [B][D] */
  float *actions_dev;               /* out f32 [B][3] */
  float *next_states_dev;           /* out f32 [B][D] */
  float *rewards_dev;               /* out f32 [B] */
  uint8_t *dones_dev;               /* out u8 [B] */
  int32_t *picks_out_dev;           /* nullable out i32 [B][4] */
} ArmEnvHerArgs;

/* ReplayBuffer_Trajectory_{reach,push}.sample(batch, use_her, dis_threshold, her_ratio) for B samples in one launch:
 * uniform episode, uniform step, with probability her_ratio a future state's first three dims become the goal,
 * reward -0.1 / 1.0 and done by the distance threshold. */
int armenv_her_sample(int32_t device, const ArmEnvHerArgs *args, void *stream);

/* Measurement aid (bench.py's roofline.valu.one_wave_per_simd; no reference counterpart): the interval at which SIMDs issue
 * independent 64-lane v_fma_f64 (precision 64) / v_fma_f32 (32) instructions when every SIMD of `device` holds
 * `waves_per_simd` waves that do nothing else -- the env kernels run ONE wave per SIMD, whose f64 issue interval is above the
 * pipe's nominal four cycles.  Out: nanoseconds per wave instruction per SIMD (host double).  Blocking
 * (about 60 ms: the device is brought to its steady clocks first; default stream). */
int armenv_probe_issue_rate(int32_t device, int32_t precision, int32_t waves_per_simd, double *ns_per_instruction);

/* Measurement aid (bench.py's clock_probe samples; no reference counterpart): enqueues on `stream` one wavefront per SIMD of the
 * device (256-thread blocks, one per CU), each running a fixed dependent chain of v_fma_f32 between two readings of the device's
 * constant-rate 100 MHz counter -- the whole chip under the kind of load the env kernels put on it, for about 10 us.
 * *rows (nullable, host) receives the number of rows = 4 x CUs; out_dev == NULL only answers that.  out_dev u64 [rows][4] (device),
 * one row per wavefront: [0] the chain's duration in 10 ns ticks -- inversely proportional to the shader clock the wave's XCD ran
 * at, so the ratio of two samples is the ratio of the clocks; [1] the same in s_memtime ticks; [2] the 100 MHz counter at the
 * chain's start (the device's own time line); [3] bits 0..3 the XCD the wave ran on, bits 8.. the chain length in instructions.
 * Does not synchronise. */
int armenv_probe_clock(int32_t device, uint64_t *out_dev, int32_t *rows, void *stream);

/* Shape / capability queries. */
int64_t armenv_num_envs(const ArmEnv *env);
int32_t armenv_obs_dim(const ArmEnv *env);
int32_t armenv_aux_dim(const ArmEnv *env);   /* row length of get/set_state's aux: 0 reach, 10 push, 12 pick */
int32_t armenv_action_dim(const ArmEnv *env);
/* name of the step kernel variant in use, e.g. "reach_step<f64,kuka>" (for profiles) */
const char *armenv_kernel_name(const ArmEnv *env);

const char *armenv_last_error(void);
int32_t armenv_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ARMENV_H */
