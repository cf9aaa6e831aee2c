"""ctypes front-end of the CPU oracle (oracle/armenv_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- never from the product package.  pybullet is not installable
here; the reach path is pinned by the reference's own recorded run instead (real PyBullet, 2 068 free-running
env steps reproduced to 4e-7 relative: tests/reference_run.py); see the header of armenv_oracle.c.

The chain tables below are entered independently of the product's URDF assets so that the two
can be cross-checked:
  KUKA iiwa : SURVEY.md Appendix A (pybullet_data/kuka_iiwa/model.urdf is not vendored by the
              reference; pinned by /root/reference/envs/bmirobot_joints_info_pybullet.txt:1-7 and the
              FK known answer at /root/reference/main.py:106)
  Diana S1  : /root/reference/models/diana/DianaS1_robot.urdf:30,60,90,120,150,180,210
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def _host_tag():
    """The oracle is compiled -march=native: one object per host CPU model (the build container and the GPU box differ)."""
    import hashlib
    model = flags = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name") and not model:
                model = ln.split(":", 1)[1].strip()
            if ln.startswith("flags") and not flags:
                flags = ln.split(":", 1)[1].strip()
            if model and flags:
                break
    except OSError:
        pass
    return hashlib.sha1((model + "|" + flags).encode()).hexdigest()[:10]


_TAG = _host_tag()
_SO = os.path.join(_HERE, "build", f"liboracle-{_TAG}.so")

NJ = 7
H = 1.57079632679      # the URDF text value, not math.pi/2
PI = 3.14159265359

KUKA = dict(
    xyz=[(0, 0, 0.1575), (0, 0, 0.2025), (0, 0.2045, 0), (0, 0, 0.2155), (0, 0.1845, 0), (0, 0, 0.2155), (0, 0.081, 0)],
    rpy=[(0, 0, 0), (H, 0, PI), (H, 0, PI), (H, 0, 0), (-H, PI, 0), (H, 0, 0), (-H, PI, 0)],
    limit=[2.96705972839, 2.09439510239, 2.96705972839, 2.09439510239, 2.96705972839, 2.09439510239, 3.05432619099],
    inertial=[(-0.1, 0, 0.07), (0, -0.03, 0.12), (0.0003, 0.059, 0.042), (0, 0.03, 0.13), (0, 0.067, 0.034),
              (0.0001, 0.021, 0.076), (0, 0.0006, 0.0004), (0, 0, 0.02)],
)
DIANA = dict(
    xyz=[(0, 0, 0.2723), (0, -0.156, 0), (0, 0.4577, 0), (0, 0.1371, 0), (0, -0.449, 0), (0, -0.1365, 0), (0, 0.0825, 0)],
    rpy=[(PI, 0, 0), (-H, 0, 0), (H, 0, 0), (H, 0, 0), (-H, 0, 0), (-H, 0, 0), (H, 0, 0)],
    limit=[3.12413936107, 2.79252680319, 3.12413936107, 2.79252680319, 3.12413936107, 3.12413936107, 3.12413936107],
    inertial=[(-0.00025974, -0.00026507, 0.024973), (-0.000108043, -0.044600027, 0.068607373),
              (0.00000875698, 0.047132796, 0.006996823), (0.0000183253, 0.03187396, 0.136688085),
              (-0.0000171058, -0.048491536, 0.007150206), (-0.000027456, -0.002120073, 0.13041583),
              (0.0000262506, 0.003206123, 0.027090603), (0.0, 0.0001, 0.03405)],
)
ROBOTS = {"kuka": KUKA, "diana": DIANA}

# /root/reference/envs/rl_reach_env.py:116-119
INIT_Q = [0.006418, 0.413184, -0.011401, -1.589317, 0.005379, 1.137684, -0.006539]


class OrcChain(C.Structure):
    _fields_ = [("xyz", C.c_double * 3 * NJ), ("R", C.c_double * 9 * NJ),
                ("base_p", C.c_double * 3), ("base_R", C.c_double * 9)]


class OrcConfig(C.Structure):
    _fields_ = [
        ("dv", C.c_double), ("reach_dis", C.c_double), ("max_steps", C.c_int32), ("task", C.c_int32),
        ("box_lo", C.c_double * 3), ("box_hi", C.c_double * 3),
        ("goal_lo", C.c_double * 3), ("goal_hi", C.c_double * 3),
        ("target_quat", C.c_double * 4), ("q_init", C.c_double * NJ),
        ("ik_lambda", C.c_double), ("ik_residual", C.c_double), ("ik_max_dtheta", C.c_double),
        ("ik_max_iters", C.c_int32), ("ik_exit_mode", C.c_int32), ("ik_angle_f32", C.c_int32),
        ("ik_form", C.c_int32),
        ("push_success_dis", C.c_double), ("push_cube_half", C.c_double), ("push_eef_radius", C.c_double),
        ("push_rest_z", C.c_double), ("push_place_min", C.c_double), ("push_place_max", C.c_double), ("push_place_z", C.c_double),
        ("pick_gripper_length", C.c_double), ("pick_trigger_dis", C.c_double), ("pick_jaw_half", C.c_double),
        ("clamp_joint_limits", C.c_int32), ("pad0", C.c_int32), ("lim_lo", C.c_double * NJ), ("lim_hi", C.c_double * NJ),
        ("fence_z", C.c_double), ("limit_erp", C.c_double), ("fence_pivot", C.c_double), ("ik_tip_offset", C.c_double * 3),
        ("push_tool_radius", C.c_double), ("push_tool_below", C.c_double), ("push_contact_erp", C.c_double),
        ("push_contact_split", C.c_double), ("push_friction", C.c_double), ("push_gravity", C.c_double), ("push_dt", C.c_double),
        ("push_drop_contact", C.c_double), ("push_drop_relax", C.c_double), ("push_contact_model", C.c_int32), ("pad1", C.c_int32),
    ]


def build(force=False):
    """Compile the oracle with gcc (no GPU, no reference sources involved)."""
    src = os.path.join(_HERE, "armenv_oracle.c")
    stale = lambda: not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src)
    if force or stale():
        # first users race on a fresh box (pytest-xdist workers, torchrun ranks, bench's cpu-leg child beside its parent): one
        # builds under the lock, the others find the object up to date when they get it; the Makefile renames into place
        import fcntl
        os.makedirs(os.path.join(_HERE, "build"), exist_ok=True)
        with open(os.path.join(_HERE, "build", ".lock"), "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            if force or stale():
                subprocess.check_call(["make", "-s", "-C", _HERE, f"TAG={_TAG}"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_num_threads.restype = C.c_int
        _lib.orc_ik.restype = C.c_int
        _lib.orc_dls_delta.restype = C.c_int
    return _lib


def _p(a, t=None):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


def quat_from_euler(rpy):
    out = (C.c_double * 4)()
    lib().orc_quat_from_euler((C.c_double * 3)(*rpy), out)
    return list(out)


def make_chain(robot="kuka", base_xyz=(0, 0, 0), base_rpy=(0, 0, 0)):
    r = ROBOTS[robot] if isinstance(robot, str) else robot
    xyz = np.ascontiguousarray(r["xyz"], dtype=np.float64)
    rpy = np.ascontiguousarray(r["rpy"], dtype=np.float64)
    ch = OrcChain()
    lib().orc_chain_from_rpy(_p(xyz), _p(rpy), (C.c_double * 3)(*base_xyz), (C.c_double * 3)(*base_rpy), C.byref(ch))
    return ch


def default_config(task="reach", robot="kuka"):
    """Constants of RLReachEnv.__init__ (/root/reference/envs/rl_reach_env.py:44-125),
    config.py:41-42,51, and Bullet's IK defaults."""
    c = OrcConfig()
    c.clamp_joint_limits = 0
    c.lim_lo[:] = [-x for x in ROBOTS[robot]["limit"]]      # bmirobot_joints_info_pybullet.txt:1-7 fields 8-9 (symmetric)
    c.lim_hi[:] = list(ROBOTS[robot]["limit"])
    c.fence_z = 0.05
    c.limit_erp = 0.2          # Bullet's default constraint ERP; read by clamp_joint_limits == 2 only
    c.fence_pivot = 1e-2       # conditioning term of the parity fence
    c.ik_tip_offset[:] = [0.0, 0.0, 0.0]      # the URDF link-7 frame (getLinkState(...)[4])
    c.task = {"reach": 0, "push": 1, "pick": 2}[task]
    c.dv = 0.02 if task == "reach" else 0.08
    c.reach_dis = 0.01
    c.max_steps = 500
    c.box_lo[:] = [0.2, -0.3, 0.0]
    c.box_hi[:] = [0.7, 0.3, {"reach": 0.55, "push": 0.1, "pick": 0.55 + 0.257}[task]]   # rl_push_env.py:314, rl_pick_env.py:313
    c.goal_lo[:] = [0.2, -0.3, 0.0]
    c.goal_hi[:] = [0.7, 0.3, 0.55]
    c.target_quat[:] = quat_from_euler([0.0, -math.pi, math.pi / 2.0])
    c.q_init[:] = INIT_Q
    c.ik_lambda = 1e-5
    c.ik_residual = 1e-4
    c.ik_max_dtheta = 45.0 * math.pi / 180.0
    c.ik_max_iters = 20
    c.ik_exit_mode = 0
    c.ik_angle_f32 = 1
    c.ik_form = 0
    c.push_success_dis = 0.05
    c.push_cube_half = 0.02
    c.push_eef_radius = 0.03
    c.push_place_z = 0.01              # rl_push_env.py:199,206: spawn height of cube and target
    c.push_rest_z = 0.01 - 0.01474     # where the cube comes to rest (fitted to the reference's recorded push run)
    c.push_place_min = 0.22
    c.push_place_max = 0.25
    # the cube under stepSimulation (armenv_oracle.c push_contact_dyn / push_cube_z); fitted values: tests/tools/fit_bullet.py part (C)
    c.push_contact_model = 1
    c.push_tool_radius = 0.045         # the KUKA flange: 45 mm radius ...
    c.push_tool_below = 0.045          # ... its face 45 mm below the link-7 frame (nominal geometry, not fitted)
    c.push_contact_erp = 0.01          # fitted
    c.push_contact_split = 0.04        # Bullet: m_splitImpulsePenetrationThreshold = -0.04
    c.push_friction = 0.03             # fitted
    c.push_gravity = 10.0              # rl_push_env.py:155
    c.push_dt = 1.0 / 240.0            # Bullet's default time step
    c.push_drop_contact = 0.015        # spawn 0.01 - half 0.02 - table top -0.025
    c.push_drop_relax = 0.1
    c.pick_gripper_length = 0.257      # rl_pick_env.py:79
    c.pick_trigger_dis = 0.006         # rl_pick_env.py:412
    c.pick_jaw_half = 0.02
    return c


# ------------------------------------------------------------------ thin numpy wrappers

def fk(chain, q):
    q = np.ascontiguousarray(q, dtype=np.float64).reshape(-1, NJ)
    n = q.shape[0]
    pos = np.empty((n, 3)); quat = np.empty((n, 4))
    lib().orc_fk_batch(C.byref(chain), C.c_int64(n), _p(q), _p(pos), _p(quat))
    return pos, quat


def fk_full(chain, q):
    """Single-env FK returning (p, R, zs, ps)."""
    q = np.ascontiguousarray(q, dtype=np.float64)
    p = np.empty(3); R = np.empty(9); zs = np.empty(21); ps = np.empty(21)
    lib().orc_fk(C.byref(chain), _p(q), _p(p), _p(R), _p(zs), _p(ps))
    return p, R.reshape(3, 3), zs.reshape(7, 3), ps.reshape(7, 3)


def jacobian(chain, q):
    p, _, zs, ps = fk_full(chain, q)
    J = np.empty((6, 7))
    lib().orc_jacobian(_p(np.ascontiguousarray(zs)), _p(np.ascontiguousarray(ps)), _p(p), _p(J))
    return J


def dls_delta(J, e, lam, max_dtheta, form):
    J = np.ascontiguousarray(J, dtype=np.float64); e = np.ascontiguousarray(e, dtype=np.float64)
    out = np.empty(7)
    rc = lib().orc_dls_delta(_p(J), _p(e), C.c_double(lam), C.c_double(max_dtheta), C.c_int(form), _p(out))
    assert rc == 0
    return out


def orientation_error(qt, qc, angle_f32=1):
    e = np.empty(3)
    lib().orc_orientation_error(_p(np.ascontiguousarray(qt, dtype=np.float64)),
                                _p(np.ascontiguousarray(qc, dtype=np.float64)), C.c_int(angle_f32), _p(e))
    return e


def ik(chain, cfg, q, tgt):
    q = np.ascontiguousarray(q, dtype=np.float64).reshape(-1, NJ)
    tgt = np.ascontiguousarray(tgt, dtype=np.float64).reshape(-1, 3)
    n = q.shape[0]
    out = np.empty_like(q); iters = np.empty(n, dtype=np.int32)
    lib().orc_ik_batch(C.byref(chain), C.byref(cfg), C.c_int64(n), _p(q), _p(tgt), _p(out), _p(iters))
    return out, iters


def ik_diag(chain, cfg, q, tgt):
    """Conditioning of IK calls: (diag [n,3] = smallest LDL^T pivot of J J^T + lambda I over the call, largest |dtheta|
    before the scale-back, final position residual; updates [n])."""
    q = np.ascontiguousarray(q, dtype=np.float64).reshape(-1, NJ)
    tgt = np.ascontiguousarray(tgt, dtype=np.float64).reshape(-1, 3)
    n = q.shape[0]
    diag = np.empty((n, 3)); iters = np.empty(n, dtype=np.int32)
    lib().orc_ik_diag_batch(C.byref(chain), C.byref(cfg), C.c_int64(n), _p(q), _p(tgt), _p(diag), _p(iters))
    return diag, iters


def pose_min_pivot(chain, cfg, q):
    """Smallest LDL^T pivot of J J^T + lambda I at the poses q [n,7] (the conditioning term of the parity fence)."""
    q = np.ascontiguousarray(q, dtype=np.float64).reshape(-1, NJ)
    out = np.empty(q.shape[0])
    lib().orc_pose_min_pivot_batch(C.byref(chain), C.c_double(cfg.ik_lambda), C.c_int64(q.shape[0]), _p(q), _p(out))
    return out


def fence_flags(chain, cfg, q, tgt):
    """Per env: bit 0 = the (unclamped) IK result leaves the URDF limits, bit 1 = the flange ends below cfg.fence_z,
    bit 2 = the IK call ran to cfg.ik_max_iters, bit 3 = an LDL^T pivot of one of its damped systems fell below cfg.fence_pivot."""
    q = np.ascontiguousarray(q, dtype=np.float64).reshape(-1, NJ)
    tgt = np.ascontiguousarray(tgt, dtype=np.float64).reshape(-1, 3)
    lib().orc_fence_flags.restype = C.c_int
    return np.array([lib().orc_fence_flags(C.byref(chain), C.byref(cfg), _p(q[i]), _p(tgt[i])) for i in range(q.shape[0])], dtype=np.int32)


def philox(ctr, key):
    out = (C.c_uint32 * 4)()
    lib().orc_philox((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out)
    return list(out)


def draw6(seed, env_id, episode, draw):
    u = (C.c_double * 6)()
    lib().orc_draw6(C.c_uint64(seed), C.c_uint64(env_id), C.c_uint32(episode), C.c_uint32(draw), u)
    return list(u)


class ReachState:
    """Host mirror of the engine's per-env state (AoS, the get_state/set_state exchange layout)."""

    def __init__(self, n):
        self.n = n
        self.q = np.zeros((n, NJ)); self.goal = np.zeros((n, 3), dtype=np.float32)
        self.step = np.zeros(n, dtype=np.int32); self.episode = np.zeros(n, dtype=np.uint32)
        self.ep_return = np.zeros(n)
        self.last_return = np.zeros(n); self.last_len = np.zeros(n, dtype=np.int32)
        self.last_success = np.zeros(n, dtype=np.uint8)

    def copy(self):
        o = ReachState(self.n)
        for k, v in self.__dict__.items():
            if isinstance(v, np.ndarray):
                setattr(o, k, v.copy())
        return o


def reach_reset(chain, cfg, st, seed=0, env_id0=0, mask=None):
    obs = np.zeros((st.n, 6), dtype=np.float32)
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    lib().orc_reach_reset(C.byref(chain), C.byref(cfg), C.c_uint64(seed), C.c_uint64(env_id0), C.c_int64(st.n),
                          _p(m), _p(st.q), _p(st.goal), _p(st.step), _p(st.episode), _p(obs))
    if mask is None:
        st.ep_return[:] = 0
    else:
        st.ep_return[np.asarray(mask, dtype=bool)] = 0
    return obs


def reach_reset_with_goal(chain, cfg, st, goal):
    obs = np.zeros((st.n, 6), dtype=np.float32)
    g = np.ascontiguousarray(goal, dtype=np.float32).reshape(st.n, 3)
    lib().orc_reach_reset_with_goal(C.byref(chain), C.byref(cfg), C.c_int64(st.n), _p(g), _p(st.q), _p(st.goal),
                                    _p(st.step), _p(obs))
    st.episode += 1          # episode counts every reset of an env, whoever chose the goal (it keys the exploration noise)
    st.ep_return[:] = 0
    return obs


def reach_step(chain, cfg, st, action, minpiv=None):
    """One step, no auto-reset.  Returns obs f32[N,6], reward f64[N], done, success, ik updates.
    minpiv (optional f64 [N]): receives the smallest LDL^T pivot of each env's IK call (the fence's conditioning measure)."""
    n = st.n
    a = np.ascontiguousarray(action, dtype=np.float32).reshape(n, 3)
    obs = np.zeros((n, 6), dtype=np.float32); rew = np.zeros(n)
    done = np.zeros(n, dtype=np.uint8); succ = np.zeros(n, dtype=np.uint8); iters = np.zeros(n, dtype=np.int32)
    lib().orc_reach_step(C.byref(chain), C.byref(cfg), C.c_int64(n), _p(st.q), _p(st.goal), _p(st.step), _p(a),
                         _p(obs), _p(rew), _p(done), _p(succ), _p(iters), _p(minpiv))
    st.ep_return += rew
    return obs, rew, done, succ, iters


def reach_step_autoreset(chain, cfg, st, action, seed=0, env_id0=0, want_terminal=True, iters=None, minpiv=None):
    n = st.n
    a = np.ascontiguousarray(action, dtype=np.float32).reshape(n, 3)
    obs = np.zeros((n, 6), dtype=np.float32); rew = np.zeros(n)
    done = np.zeros(n, dtype=np.uint8); succ = np.zeros(n, dtype=np.uint8)
    term = np.zeros((n, 6), dtype=np.float32) if want_terminal else None
    lib().orc_reach_step_autoreset(C.byref(chain), C.byref(cfg), C.c_uint64(seed), C.c_uint64(env_id0), C.c_int64(n),
                                   _p(st.q), _p(st.goal), _p(st.step), _p(st.episode), _p(st.ep_return), _p(a),
                                   _p(obs), _p(rew), _p(done), _p(succ), _p(term),
                                   _p(st.last_return), _p(st.last_len), _p(st.last_success), _p(iters), _p(minpiv))
    return obs, rew, done, succ, term


def actor_forward(sd, states, action_bound):
    """sd: dict fc1.weight ... fc3.bias as float32 numpy (torch Linear layout)."""
    s = np.ascontiguousarray(states, dtype=np.float32)
    n, in_dim = s.shape
    W1, b1, W2, b2, W3, b3 = (np.ascontiguousarray(sd[k], dtype=np.float32) for k in
                              ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc3.weight", "fc3.bias"))
    hid, out_dim = W1.shape[0], W3.shape[0]
    a = np.zeros((n, out_dim), dtype=np.float32)
    lib().orc_actor_forward(C.c_int64(n), C.c_int(in_dim), C.c_int(hid), C.c_int(out_dim), _p(W1), _p(b1), _p(W2),
                            _p(b2), _p(W3), _p(b3), C.c_float(action_bound), _p(s), _p(a))
    return a


def qvalue_forward(sd, states, actions):
    """QValueNet.forward(state, action) (/root/reference/algo/DATD3/net_mlp.py:54-58): [n] f32."""
    x = np.ascontiguousarray(np.concatenate([np.asarray(states, dtype=np.float32), np.asarray(actions, dtype=np.float32)], axis=1))
    n, in_dim = x.shape
    W1, b1, W2, b2, W3, b3 = (np.ascontiguousarray(sd[k], dtype=np.float32) for k in
                              ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc3.weight", "fc3.bias"))
    q = np.zeros(n, dtype=np.float32)
    lib().orc_qvalue_forward(C.c_int64(n), C.c_int(in_dim), C.c_int(W1.shape[0]), _p(W1), _p(b1), _p(W2), _p(b2), _p(W3), _p(b3), _p(x), _p(q))
    return q


def datd3_take_action(nets, states, action_bound):
    """DATD3_MLP.take_action (/root/reference/algo/DATD3/DATD3_mlp.py:88-109), batched: nets = (actor1, actor2, critic1, critic2)
    state dicts.  DARC_MLP.take_action (algo/DARC/DARC_mlp.py:92-113) is the same statement; DADDPG_MLP.take_action
    (algo/DADDPG/DADDPG_mlp.py:77-97) is the same with its ONE critic in both critic places.  Returns (action [n,3], q1 [n], q2 [n], picked [n] u8: 0 actor1, 1 actor2)."""
    a1 = actor_forward(nets[0], states, action_bound)
    a2 = actor_forward(nets[1], states, action_bound)
    q1 = qvalue_forward(nets[2], states, a1)
    q2 = qvalue_forward(nets[3], states, a2)
    pick = ~(q1 >= q2)                                     # :107  action1 if q1 >= q2 else action2
    return np.where(pick[:, None], a2, a1), q1, q2, pick.astype(np.uint8)


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    lib().orc_set_num_threads(C.c_int(int(n)))


def usable_cores():
    """Cores this process may actually use: the affinity mask, capped by the cgroup CPU quota when there is one
    (a container with 128 visible CPUs and a quota of 16 runs 128 OpenMP threads at 1/8 speed each)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    eff = n if quota is None else max(1, min(n, int(quota + 0.999)))
    return dict(affinity=n, cgroup_quota=quota, usable=eff)


def reach_outcome(cfg, dist, step_counter):
    r = C.c_double(); d = C.c_uint8(); s = C.c_uint8()
    lib().orc_reach_outcome(C.byref(cfg), C.c_double(dist), C.c_int32(step_counter), C.byref(r), C.byref(d), C.byref(s))
    return r.value, bool(d.value), bool(s.value)


def random_policy_actions(st, seed=0, env_id0=0, sigma=0.7 * 0.98, clip=0.7):
    a = np.zeros((st.n, 3), dtype=np.float32)
    lib().orc_random_policy_actions(C.c_uint64(seed), C.c_uint64(env_id0), C.c_int64(st.n), _p(st.episode), _p(st.step),
                                    C.c_float(sigma), C.c_float(clip), _p(a))
    return a


def policy_noise(st, seed=0, env_id0=0):
    nz = np.zeros((st.n, 3), dtype=np.float32)
    lib().orc_policy_noise_batch(C.c_uint64(seed), C.c_uint64(env_id0), C.c_int64(st.n), _p(st.episode), _p(st.step), _p(nz))
    return nz


def policy_noise_ids(seed, env_ids, episode, step):
    """Exploration noise [n,3] of the envs with global ids `env_ids`, `episode` = resets so far (u32), `step` = step counter."""
    ids = np.ascontiguousarray(env_ids, dtype=np.int64)
    ep = np.ascontiguousarray(episode, dtype=np.uint32); sp = np.ascontiguousarray(step, dtype=np.int32)
    nz = np.zeros((ids.size, 3), dtype=np.float32)
    lib().orc_policy_noise_ids(C.c_uint64(seed), C.c_int64(ids.size), _p(ids), _p(ep), _p(sp), _p(nz))
    return nz


def reach_rollout(chain, cfg, st, steps, actions=None, seed=0, env_id0=0, sigma=0.7 * 0.98, clip=0.7, actor=None,
                  bound=0.7, obs0=None):
    """`steps` auto-reset steps; actions [steps,N,3], or None for the fused policy: zero actor (random) or, with
    actor = state-dict and obs0 = the current observation, a = clip(actor(obs) + sigma * N(0,1), +-clip)
    (/root/reference/main.py:114-117).  Returns dict of [steps, N, ...] arrays like the engine's rollout."""
    out = dict(obs=[], reward=[], done=[], success=[], actions=[], terminal_obs=[])
    obs = obs0
    for t in range(steps):
        if actions is not None:
            a = actions[t]
        elif actor is None:
            a = random_policy_actions(st, seed, env_id0, sigma, clip)
        else:
            mu = actor_forward(actor, obs, bound)
            a = np.clip(mu + np.float32(sigma) * policy_noise(st, seed, env_id0), -np.float32(clip), np.float32(clip)).astype(np.float32)
        o, r, d, s, term = reach_step_autoreset(chain, cfg, st, a, seed=seed, env_id0=env_id0)
        obs = o
        for k, v in zip(("obs", "reward", "done", "success", "actions", "terminal_obs"), (o, r, d, s, a, term)):
            out[k].append(np.array(v, copy=True))
    return {k: np.stack(v) for k, v in out.items()}


# ------------------------------------------------------------------ push env

PUSH_AUX = 10


class PushState(ReachState):
    """adds aux [N,10] = cube xyz, target xyz, d_last, cube velocity xy, pad"""

    def __init__(self, n):
        super().__init__(n)
        self.aux = np.zeros((n, PUSH_AUX))


def push_reset(chain, cfg, st, seed=0, env_id0=0, mask=None):
    obs = np.zeros((st.n, 9), dtype=np.float32)
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    lib().orc_push_reset(C.byref(chain), C.byref(cfg), C.c_uint64(seed), C.c_uint64(env_id0), C.c_int64(st.n), _p(m),
                         _p(st.q), _p(st.aux), _p(st.step), _p(st.episode), _p(obs))
    if mask is None:
        st.ep_return[:] = 0
    else:
        st.ep_return[np.asarray(mask, dtype=bool)] = 0
    return obs


def push_reset_with_goal(chain, cfg, st, goal6):
    obs = np.zeros((st.n, 9), dtype=np.float32)
    g = np.ascontiguousarray(goal6, dtype=np.float32).reshape(st.n, 6)
    lib().orc_push_reset_with_goal(C.byref(chain), C.byref(cfg), C.c_int64(st.n), _p(g), _p(st.q), _p(st.aux), _p(st.step), _p(obs))
    st.episode += 1
    st.ep_return[:] = 0
    return obs


def push_step(chain, cfg, st, action, minpiv=None):
    n = st.n
    a = np.ascontiguousarray(action, dtype=np.float32).reshape(n, 3)
    obs = np.zeros((n, 9), dtype=np.float32); rew = np.zeros(n)
    done = np.zeros(n, dtype=np.uint8); succ = np.zeros(n, dtype=np.uint8); iters = np.zeros(n, dtype=np.int32)
    lib().orc_push_step(C.byref(chain), C.byref(cfg), C.c_int64(n), _p(st.q), _p(st.aux), _p(st.step), _p(a), _p(obs),
                        _p(rew), _p(done), _p(succ), _p(iters), _p(minpiv))
    st.ep_return += rew
    return obs, rew, done, succ, iters


def push_step_autoreset(chain, cfg, st, action, seed=0, env_id0=0, iters=None, minpiv=None):
    n = st.n
    a = np.ascontiguousarray(action, dtype=np.float32).reshape(n, 3)
    obs = np.zeros((n, 9), dtype=np.float32); rew = np.zeros(n)
    done = np.zeros(n, dtype=np.uint8); succ = np.zeros(n, dtype=np.uint8); term = np.zeros((n, 9), dtype=np.float32)
    lib().orc_push_step_autoreset(C.byref(chain), C.byref(cfg), C.c_uint64(seed), C.c_uint64(env_id0), C.c_int64(n),
                                  _p(st.q), _p(st.aux), _p(st.step), _p(st.episode), _p(st.ep_return), _p(a), _p(obs),
                                  _p(rew), _p(done), _p(succ), _p(term), _p(st.last_return), _p(st.last_len), _p(st.last_success), _p(iters), _p(minpiv))
    return obs, rew, done, succ, term


def push_outcome(cfg, cube, target, d_last, step_counter):
    r = C.c_double(); d = C.c_uint8(); s = C.c_uint8(); dl = C.c_double(d_last)
    lib().orc_push_outcome(C.byref(cfg), (C.c_double * 3)(*cube), (C.c_double * 3)(*target), C.byref(dl), C.c_int32(step_counter),
                           C.byref(r), C.byref(d), C.byref(s))
    return r.value, bool(d.value), bool(s.value), dl.value


# ------------------------------------------------------------------ pick env

class PickState(ReachState):
    """adds aux [N,12] = cube xyz, target xyz, d_last, gripper (0 open, 1 closed, 2 holding), hold offset xyz, pad"""

    def __init__(self, n):
        super().__init__(n)
        self.aux = np.zeros((n, 12))


def pick_reset(chain, cfg, st, seed=0, env_id0=0, mask=None):
    obs = np.zeros((st.n, 9), dtype=np.float32)
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    lib().orc_pick_reset(C.byref(chain), C.byref(cfg), C.c_uint64(seed), C.c_uint64(env_id0), C.c_int64(st.n), _p(m),
                         _p(st.q), _p(st.aux), _p(st.step), _p(st.episode), _p(obs))
    if mask is None:
        st.ep_return[:] = 0
    else:
        st.ep_return[np.asarray(mask, dtype=bool)] = 0
    return obs


def pick_reset_with_goal(chain, cfg, st, goal6):
    obs = np.zeros((st.n, 9), dtype=np.float32)
    g = np.ascontiguousarray(goal6, dtype=np.float32).reshape(st.n, 6)
    lib().orc_pick_reset_with_goal(C.byref(chain), C.byref(cfg), C.c_int64(st.n), _p(g), _p(st.q), _p(st.aux), _p(st.step), _p(obs))
    st.episode += 1
    st.ep_return[:] = 0
    return obs


def pick_step(chain, cfg, st, action, minpiv=None):
    n = st.n
    a = np.ascontiguousarray(action, dtype=np.float32).reshape(n, 3)
    obs = np.zeros((n, 9), dtype=np.float32); rew = np.zeros(n)
    done = np.zeros(n, dtype=np.uint8); succ = np.zeros(n, dtype=np.uint8); iters = np.zeros(n, dtype=np.int32)
    lib().orc_pick_step(C.byref(chain), C.byref(cfg), C.c_int64(n), _p(st.q), _p(st.aux), _p(st.step), _p(a), _p(obs),
                        _p(rew), _p(done), _p(succ), _p(iters), _p(minpiv))
    st.ep_return += rew
    return obs, rew, done, succ, iters


def pick_step_autoreset(chain, cfg, st, action, seed=0, env_id0=0, iters=None, minpiv=None):
    n = st.n
    a = np.ascontiguousarray(action, dtype=np.float32).reshape(n, 3)
    obs = np.zeros((n, 9), dtype=np.float32); rew = np.zeros(n)
    done = np.zeros(n, dtype=np.uint8); succ = np.zeros(n, dtype=np.uint8); term = np.zeros((n, 9), dtype=np.float32)
    lib().orc_pick_step_autoreset(C.byref(chain), C.byref(cfg), C.c_uint64(seed), C.c_uint64(env_id0), C.c_int64(n),
                                  _p(st.q), _p(st.aux), _p(st.step), _p(st.episode), _p(st.ep_return), _p(a), _p(obs),
                                  _p(rew), _p(done), _p(succ), _p(term), _p(st.last_return), _p(st.last_len), _p(st.last_success), _p(iters), _p(minpiv))
    return obs, rew, done, succ, term
