"""CPU tests: the oracle against every known answer the reference holds for the env hot path
 and against its own invariants."""
import math

import numpy as np
import pytest

from conftest import golden_json, golden_npz


def test_fk_known_answer_main_py_106(O, kuka):
    g = golden_json("fk_kat.json")
    p, _ = O.fk(kuka, g["q"])
    # main.py:106 holds the float32-rounded getLinkState(kuka,6)[4]
    assert np.abs(p[0] - np.array(g["p_f32"])).max() < 1e-7
    assert np.abs(p[0].astype(np.float32) - np.float32(g["p_f32"])).max() <= 6.0e-8   # <= 1 f32 ulp


def _quat_inv_from_rpy(O, rpy):
    R = np.empty(9)
    import ctypes as C
    O.lib().orc_rpy_to_mat((C.c_double * 3)(*rpy), R.ctypes.data_as(C.c_void_p))
    q = np.empty(4)
    O.lib().orc_quat_from_mat(np.ascontiguousarray(R.reshape(3, 3).T).ctypes.data_as(C.c_void_p), q.ctypes.data_as(C.c_void_p))
    return q


@pytest.mark.parametrize("robot,first", [("kuka", 0), ("diana", 1)])
def test_joint_frames_match_getjointinfo_dump(O, robot, first):
    """parentFramePos = joint origin - parent link inertial origin; parentFrameOrn = inverse of the
    origin rotation; limits -- envs/bmirobot_joints_info_pybullet.txt:1-16."""
    rows = golden_json("joint_info.json")[robot][first:]
    tab = O.ROBOTS[robot]
    assert len(rows) == 7
    for i, row in enumerate(rows):
        want = np.array(tab["xyz"][i]) - np.array(tab["inertial"][i])
        assert np.abs(want - np.array(row["parent_frame_pos"])).max() < 1e-9, (i, want, row["parent_frame_pos"])
        assert row["axis"] == [0.0, 0.0, 1.0]
        assert abs(row["lower"] + tab["limit"][i]) < 1e-12 and abs(row["upper"] - tab["limit"][i]) < 1e-12
        q = _quat_inv_from_rpy(O, tab["rpy"][i])
        w = np.array(row["parent_frame_orn"])
        assert min(np.abs(q - w).max(), np.abs(q + w).max()) < 1e-9, (i, q, w)


def test_product_urdf_assets_equal_oracle_tables(O):
    """The product's URDF assets and the oracle's tables were entered independently."""
    from armenv.urdf import builtin_chain
    for robot in ("kuka", "diana"):
        ch = builtin_chain(robot)
        tab = O.ROBOTS[robot]
        assert np.allclose(ch.origin_xyz, tab["xyz"], atol=0, rtol=0)
        assert np.allclose(ch.origin_rpy, tab["rpy"], atol=0, rtol=0)
        assert np.allclose(ch.limit_hi, tab["limit"], atol=0, rtol=0)
        assert np.allclose(ch.inertial_xyz, tab["inertial"], atol=0, rtol=0)


def test_diana_derived_positions(O, diana):
    """SURVEY.md G1 derived values (numpy, not PyBullet outputs)."""
    p, _ = O.fk(diana, [0.0] * 7)
    assert np.abs(p[0] - [0, 0.1554, 1.2615]).max() < 1e-4
    dy = O.make_chain("diana", base_rpy=(0, 0, math.pi))
    p2, _ = O.fk(dy, [0.0] * 7)
    assert np.abs(p2[0] - [0, -0.1554, 1.2615]).max() < 1e-4


def test_philox_known_answers(O):
    """Random123 kat_vectors, philox4x32 10 rounds."""
    assert O.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert O.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert O.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_reward_truth_table(O):
    g = golden_json("reward_truth.json")
    cfg = O.default_config()
    assert cfg.max_steps == g["max_steps"] and cfg.reach_dis == g["reach_dis"]
    for row in g["rows"]:
        r, d, s = O.reach_outcome(cfg, row["distance"], row["step_counter"])
        assert (r, d, s) == (row["reward"], row["done"], row["success"]), row


def test_target_quaternion(O):
    q = O.default_config().target_quat
    s = math.sqrt(0.5)
    assert np.abs(np.array(q[:]) - [s, -s, 0, 0]).max() < 1e-15


def test_jacobian_matches_finite_differences(O, kuka):
    rng = np.random.default_rng(1)
    for _ in range(5):
        q = rng.uniform(-1.5, 1.5, 7)
        J = O.jacobian(kuka, q)
        h = 1e-6
        for i in range(7):
            dq = np.zeros(7); dq[i] = h
            pp, _ = O.fk(kuka, q + dq); pm, _ = O.fk(kuka, q - dq)
            assert np.abs((pp[0] - pm[0]) / (2 * h) - J[:3, i]).max() < 1e-8


def test_dls_primal_equals_dual(O, kuka):
    """(J^T J + l I)^-1 J^T e == J^T (J J^T + l I)^-1 e to 1e-9 in f64."""
    rng = np.random.default_rng(2)
    worst = 0.0
    for _ in range(50):
        q = rng.uniform(-2, 2, 7)
        J = O.jacobian(kuka, q)
        e = rng.normal(0, 0.05, 6)
        a = O.dls_delta(J, e, 1e-5, 10.0, 0)
        b = O.dls_delta(J, e, 1e-5, 10.0, 1)
        ref = J.T @ np.linalg.solve(J @ J.T + 1e-5 * np.eye(6), e)
        worst = max(worst, np.abs(a - b).max(), np.abs(b - ref).max())
    assert worst < 1e-9


def test_dls_clamp_45_degrees(O, kuka):
    J = O.jacobian(kuka, O.INIT_Q)
    d = O.dls_delta(J, np.array([0, 0, 0, 0, 0, 3.0]), 1e-5, math.pi / 4, 0)
    assert abs(np.abs(d).max() - math.pi / 4) < 1e-12


def test_orientation_error_small_and_wrapped(O):
    # rotation by angle a about z: q = (0,0,sin(a/2),cos(a/2)); error of target vs identity is +a z
    for a in (0.3, -0.3, 3.0, -3.0):
        qt = [0, 0, math.sin(a / 2), math.cos(a / 2)]
        e = O.orientation_error(qt, [0, 0, 0, 1], angle_f32=0)
        assert np.abs(e - [0, 0, a]).max() < 1e-12
    # the same rotation written with the opposite quaternion sign (w<0) wraps to the short way
    a = 0.4
    qt = [0, 0, -math.sin(a / 2), -math.cos(a / 2)]
    e = O.orientation_error(qt, [0, 0, 0, 1], angle_f32=0)
    assert np.abs(e - [0, 0, a]).max() < 1e-12
    # identical orientations: axis falls back to (1,0,0), angle 0
    assert np.abs(O.orientation_error([0, 0, 0, 1], [0, 0, 0, 1])).max() == 0.0


@pytest.mark.parametrize("mode", [0, 1])
def test_ik_postconditions(O, kuka, mode):
    """After calculateInverseKinematics: |p - target| well below the residual threshold, tool
    orientation at the target quaternion, iteration counts as probed in SURVEY.md Appendix E."""
    cfg = O.default_config(); cfg.ik_exit_mode = mode
    rng = np.random.default_rng(3)
    st = O.ReachState(64)
    O.reach_reset(kuka, cfg, st, seed=7)
    hist = []
    for t in range(40):
        a = np.clip(rng.normal(0, 0.686, (64, 3)), -0.7, 0.7)
        q0 = st.q.copy()
        p0, _ = O.fk(kuka, q0)
        obs, rew, done, succ, iters = O.reach_step(kuka, cfg, st, a)
        tgt = np.clip(p0 + 0.02 * a.astype(np.float32).astype(np.float64), cfg.box_lo[:], cfg.box_hi[:])
        p1, quat = O.fk(kuka, st.q)
        assert np.linalg.norm(p1 - tgt, axis=1).max() < 1e-4
        if t > 0:
            qt = np.array(cfg.target_quat[:])
            assert np.minimum(np.abs(quat - qt).max(1), np.abs(quat + qt).max(1)).max() < (1e-4 if mode == 0 else 1e-3)
        assert np.abs(obs[:, :3] - p1.astype(np.float32)).max() == 0
        hist.append(iters)
    hist = np.array(hist)
    assert hist[0].max() == (4 if mode == 0 else 3)       # 90 degree tool-yaw correction after reset
    assert 1.5 + (1 - mode) < hist[1:].mean() < 2.1 + (1 - mode)
    assert hist.max() <= 20


@pytest.mark.parametrize("form", [0, 1])
@pytest.mark.parametrize("off", [(0.0, 0.0, 0.02), (0.01, -0.02, 0.03)])
def test_ik_tip_offset_switch(O, kuka, off, form):
    """OrcConfig.ik_tip_offset: the IK's position error and linear Jacobian are taken at the point p + R * offset of link 7
    ((0,0,0.02) = the KUKA link-7 inertial origin, where Bullet's multibody keeps its link frame; (0,0,0) = the URDF link
    frame getLinkState(...)[4] reports, /root/reference/envs/rl_reach_env.py:237,244-250).  Post-conditions: THAT point ends
    within the residual of the target, the link frame ends offset from it by R * offset, the tool orientation is still the
    target quaternion, a zero offset changes no bit, and the primal and dual solve forms agree."""
    rng = np.random.default_rng(11)
    n = 512
    cfg0 = O.default_config(); cfg0.ik_form = form
    cfg = O.default_config(); cfg.ik_form = form; cfg.ik_tip_offset[:] = list(off)
    st = O.ReachState(n); O.reach_reset(kuka, cfg0, st, seed=2)
    for _ in range(4):
        O.reach_step(kuka, cfg0, st, np.clip(rng.normal(0, 0.686, (n, 3)), -0.7, 0.7))
    q0 = st.q.copy()
    p0, _ = O.fk(kuka, q0)
    tgt = p0 + rng.normal(0, 0.01, (n, 3))
    q1, it1 = O.ik(kuka, cfg, q0, tgt)
    for _ in range(2):                       # settle the one-update-past-the-test exit of Bullet's loop
        q1, _ = O.ik(kuka, cfg, q1, tgt)
    o = np.array(off)
    tip = np.empty((n, 3)); quat = np.empty((n, 4))
    for i in range(n):
        p, R, _, _ = O.fk_full(kuka, q1[i])
        tip[i] = p + R @ o
    _, quat = O.fk(kuka, q1)
    assert np.linalg.norm(tip - tgt, axis=1).max() < 1e-4
    p1, _ = O.fk(kuka, q1)
    assert np.abs(np.linalg.norm(p1 - tgt, axis=1) - np.linalg.norm(o)).max() < 1e-4
    qt = np.array(cfg.target_quat[:])
    assert np.minimum(np.abs(quat - qt).max(1), np.abs(quat + qt).max(1)).max() < 1e-4
    # the switch at zero is the old code path, bit for bit
    z = O.default_config(); z.ik_form = form; z.ik_tip_offset[:] = [0.0, 0.0, 0.0]
    qa, ita = O.ik(kuka, cfg0, q0, tgt); qb, itb = O.ik(kuka, z, q0, tgt)
    assert np.array_equal(qa, qb) and np.array_equal(ita, itb)
    # and the other solve form reaches the same joints
    other = O.default_config(); other.ik_form = 1 - form; other.ik_tip_offset[:] = list(off)
    qo, ito = O.ik(kuka, other, q0, tgt)
    qs, its = O.ik(kuka, cfg, q0, tgt)
    same = ito == its
    assert same.mean() > 0.99 and np.abs(qo - qs)[same].max() < 1e-6
    # (0,0,0.02) lies on joint 7's axis: the IK's working point does not move with q7 either
    if off == (0.0, 0.0, 0.02):
        q2 = q1.copy(); q2[:, 6] += 0.7
        p2 = np.array([O.fk_full(kuka, q2[i])[0] + O.fk_full(kuka, q2[i])[1] @ o for i in range(8)])
        assert np.abs(p2 - tip[:8]).max() < 1e-12


def test_ik_iteration_cap(O, kuka):
    cfg = O.default_config(); cfg.ik_max_iters = 2
    q, it = O.ik(kuka, cfg, O.INIT_Q, [0.3, 0.2, 0.1])
    assert it[0] == 2


def test_reset_goals_in_box_and_f32(O, kuka):
    cfg = O.default_config()
    st = O.ReachState(4096)
    obs = O.reach_reset(kuka, cfg, st, seed=123, env_id0=10)
    lo, hi = np.float32(cfg.goal_lo[:]), np.float32(cfg.goal_hi[:])
    assert (st.goal >= lo).all() and (st.goal <= hi).all()
    assert (st.episode == 1).all() and (st.step == 0).all()
    assert np.abs(obs[:, 3:] - st.goal).max() == 0
    g = golden_json("fk_kat.json")
    assert np.abs(obs[:, :3] - np.float32(g["p_f32"])).max() <= 6.0e-8
    # goals depend on (seed, global env id, episode) only: shifting env_id0 shifts the stream
    st2 = O.ReachState(4096)
    O.reach_reset(kuka, cfg, st2, seed=123, env_id0=11)
    assert np.array_equal(st2.goal[:-1], st.goal[1:])
    # uniformity (coarse)
    u = (st.goal - lo) / (hi - lo)
    assert np.abs(u.mean(0) - 0.5).max() < 0.03


def test_autoreset_bookkeeping(O, kuka):
    cfg = O.default_config(); cfg.max_steps = 5
    st = O.ReachState(8)
    O.reach_reset(kuka, cfg, st, seed=1)
    rets = np.zeros(8)
    for t in range(6):
        obs, rew, done, succ, term = O.reach_step_autoreset(kuka, cfg, st, np.zeros((8, 3)), seed=1)
        rets += rew
        assert done.all() == (t == 5)
    assert (st.step == 0).all() and (st.episode == 2).all() and (st.last_len == 6).all()
    assert np.allclose(st.last_return, rets) and (st.ep_return == 0).all()
    assert np.abs(obs[:, 3:] - st.goal).max() == 0 and np.abs(term[:, 3:] - st.goal).max() > 0


def test_actor_matches_reference_golden(O):
    """G3: vectors produced by importing the reference's TD3_MLP (algo/TD3/net_mlp.py:29-40)."""
    g = golden_npz("td3_actor_seed0.npz")
    sd = {k: g[k.replace(".", "_")] for k in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc3.weight", "fc3.bias")}
    a = O.actor_forward(sd, g["states"], float(g["action_bound"]))
    assert np.abs(a - g["actions"]).max() < 1e-5
    assert np.abs(g["actions"][0] - [0.02861051, -0.01318958, -0.02435399]).max() < 1e-6   # SURVEY.md G3 spot value


# ------------------------------------------------------------------------------ push (P1-P4)

def test_push_reward_truth_table(O):
    g = golden_json("push_reward_truth.json")
    cfg = O.default_config("push")
    for row in g["rows"]:
        r, d, s, dl = O.push_outcome(cfg, row["cube"], row["target"], row["d_last"], row["step_counter"])
        assert d == row["done"] and s == row["success"], row
        assert abs(r - row["reward"]) <= 1e-12 * max(1.0, abs(row["reward"])), (r, row)
        assert dl == row["d_new"]


def test_push_placement_and_reset(O, kuka):
    cfg = O.default_config("push")
    st = O.PushState(4096)
    obs = O.push_reset(kuka, cfg, st, seed=3)
    d = np.linalg.norm(st.aux[:, 0:3] - st.aux[:, 3:6], axis=1)
    planar = np.linalg.norm(st.aux[:, 0:2] - st.aux[:, 3:5], axis=1)      # both bodies are spawned at z = 0.01 (:199,206) ...
    assert (planar >= 0.22).all() and (planar <= 0.25).all()              # ... and the test sees them there (rl_push_env.py:213)
    # the fixed target stays (:221-224); the cube is one stepSimulation into its fall (reset() calls it once, :241): g dt^2 = 0.174 mm
    # below its spawn height, at rest in the plane; it will settle 14.74 mm lower (fitted to the reference's recorded run)
    assert (st.aux[:, 5] == 0.01).all() and np.allclose(st.aux[:, 2], 0.01 - 10.0 / 240.0 ** 2, atol=1e-15) and abs(cfg.push_rest_z - (0.01 - 0.01474)) < 1e-12
    assert not st.aux[:, 7:].any() and st.aux.shape == (4096, 10)
    assert np.allclose(st.aux[:, 6], d) and (st.episode == 1).all()
    legacy = O.default_config("push"); legacy.push_contact_model = 0          # rounds 1-4: already at rest on the table
    st0 = O.PushState(64)
    O.push_reset(kuka, legacy, st0, seed=3)
    assert (st0.aux[:, 2] == legacy.push_rest_z).all() and np.array_equal(st0.aux[:, 0:2], st.aux[:64, 0:2])
    lo, hi = np.array(cfg.goal_lo[:2]), np.array(cfg.goal_hi[:2])
    for k in (0, 3):
        assert (st.aux[:, k:k + 2] >= lo).all() and (st.aux[:, k:k + 2] <= hi).all()
    assert np.abs(obs[:, 3:9] - st.aux[:, :6].astype(np.float32)).max() == 0
    g = golden_json("fk_kat.json")
    assert np.abs(obs[:, :3] - np.float32(g["p_f32"])).max() <= 6e-8


def test_push_arm_pipeline_and_contact_rounds_1_to_4_model(O, kuka):
    """dv = 0.08, z clipped to [0, 0.1] (rl_push_env.py:314,322); idle steps cost -1 (:393-394,427); the cube moves
    only when the tool overlaps it, away from the tool, and the shaped reward is -100 * (d_now - d_last).  The named model
    push_contact_model = 0 (tool sphere, the whole penetration removed in one step; what the pick task's gripper tip uses)."""
    cfg = O.default_config("push")
    cfg.push_contact_model = 0
    st = O.PushState(1)
    O.push_reset_with_goal(kuka, cfg, st, [[0.55, 0.0, 0.01, 0.55, 0.23, 0.01]])
    obs, r, d, s, it = O.push_step(kuka, cfg, st, np.zeros((1, 3)))
    assert abs(obs[0, 2] - 0.1) < 1e-4 and r[0] == -1.0 and not d[0]     # first step: z 0.496 -> clip 0.1
    # go down next to the cube on the far side from the target (y < 0), then sweep in +y through the cube
    for _ in range(40):
        p = obs[0, :3]
        a = np.clip((np.array([0.55, -0.08, 0.01]) - p) / 0.08, -1, 1)
        obs, r, d, s, it = O.push_step(kuka, cfg, st, a[None])
    assert np.abs(obs[0, :3] - [0.55, -0.08, 0.01]).max() < 2e-3 and np.allclose(st.aux[0, :3], [0.55, 0.0, 0.01], atol=1e-7)
    moved = False
    for _ in range(12):
        c0 = st.aux[0, :3].copy(); dl = st.aux[0, 6]
        obs, r, d, s, it = O.push_step(kuka, cfg, st, np.array([[0.0, 0.3, 0.0]]))
        c1 = st.aux[0, :3]
        if c1[1] > c0[1] + 1e-6:
            moved = True
            assert abs(c1[0] - c0[0]) < 1e-3 and c1[2] == c0[2]
            assert r[0] > 0 and (r[0] == 100.0 if d[0] else abs(r[0] - (-(st.aux[0, 6] - dl) * 100)) < 1e-9)   # closer
            # the tool sphere no longer overlaps the box footprint
            gap = np.linalg.norm(np.maximum(np.abs(obs[0, :2] - c1[:2]) - 0.02, 0))
            assert gap >= 0.03 - 1e-3
    assert moved


def test_push_fall_and_contact_dynamics(O, kuka):
    """push_contact_model = 1 (default).  dv = 0.08, z clipped to [0, 0.1] (rl_push_env.py:314,322).  The cube falls from its spawn
    height: semi-implicit Euler under g = 10 at dt = 1 / 240 (z_k = 0.01 - g dt^2 k (k + 1) / 2 after k stepSimulation calls, the first
    of them in reset()) through call 13, in which it reaches the table, then recovers the 0.8 mm overshoot towards its rest height
    14.74 mm down; a step costs -1 when the cube-target distance moves by less than 1e-5 and -100 x the change otherwise (:388-397,427):
    exactly eight of the falling steps are above the threshold.  In the plane the cube stays put until the tool cylinder overlaps it,
    is then given velocity along the contact normal (erp x penetration / dt), keeps sliding after the tool has stopped and is brought
    to rest by friction; a tool that comes down on top of it moves nothing."""
    cfg = O.default_config("push")
    assert cfg.push_contact_model == 1
    st = O.PushState(1)
    O.push_reset_with_goal(kuka, cfg, st, [[0.55, 0.0, 0.01, 0.55, 0.23, 0.01]])
    c = 0.5 * 10.0 / 240.0 ** 2
    assert abs(st.aux[0, 2] - (0.01 - c * 2)) < 1e-15
    zs, rs = [], []
    obs = None
    for j in range(1, 61):
        obs, r, d, s, it = O.push_step(kuka, cfg, st, np.zeros((1, 3)))
        zs.append(st.aux[0, 2]); rs.append(r[0])
    assert abs(obs[0, 2] - 0.1) < 1e-4 and not d[0]                        # the arm: z 0.496 -> clip 0.1, far above the cube
    for j in range(1, 13):                                                   # env step j = stepSimulation call j + 1
        assert abs(zs[j - 1] - (0.01 - c * (j + 1) * (j + 2))) < 1e-15, j
    assert abs(zs[11] - (0.01 - 0.0157986)) < 1e-6 and all(zs[j] > zs[j - 1] for j in range(12, 60))      # 0.8 mm into the table, then back up
    assert abs(zs[59] - cfg.push_rest_z) < 1e-5 and zs[59] < cfg.push_rest_z
    assert sum(1 for x in rs if x != -1.0) == 8 and all(x == -1.0 for x in rs[:4] + rs[12:])              # steps 5..12 move the distance by >= 1e-5
    assert all(-0.02 < x < 0 for x in rs[4:12]) and np.array_equal(st.aux[0, 0:2], [np.float32(0.55), 0.0]) and not st.aux[0, 7:9].any()
    # go down beside the cube on the far side from the target (y < 0): 8 cm away, flange 1 cm above the table plane
    for _ in range(40):
        p = obs[0, :3]
        a = np.clip((np.array([0.55, -0.08, 0.01]) - p) / 0.08, -1, 1)
        obs, r, d, s, it = O.push_step(kuka, cfg, st, a[None])
    assert np.abs(obs[0, :3] - [0.55, -0.08, 0.01]).max() < 2e-3 and np.allclose(st.aux[0, :2], [0.55, 0.0], atol=1e-7) and r[0] == -1.0
    # sweep in +y into the cube: contact gives it velocity towards the target (erp x penetration / dt along the normal); the tool then
    # backs off, the cube keeps sliding and Coulomb friction alone brings it to rest (a tool that stayed would keep recovering the
    # remaining overlap by the erp share per step: the slow push-out the recorded runs' long moving-step counts ask for)
    moved, speeds = 0, []
    for k in range(300):
        c0 = st.aux[0, :3].copy(); dl = st.aux[0, 6]
        act = np.array([[0.0, 0.3, 0.0]]) if k < 2 else (np.array([[0.0, -1.0, 0.0]]) if k < 4 else np.zeros((1, 3)))
        obs, r, d, s, it = O.push_step(kuka, cfg, st, act)
        c1 = st.aux[0, :3]
        speeds.append(float(np.hypot(*st.aux[0, 7:9])))
        if c1[1] > c0[1]:
            moved += 1
            assert abs(c1[0] - c0[0]) < 1e-3 and st.aux[0, 8] > 0
            if abs(st.aux[0, 6] - dl) >= 1e-5 and not d[0]:
                assert r[0] > 0 and abs(r[0] - (-(st.aux[0, 6] - dl) * 100)) < 1e-9                   # closer to the target
        if d[0]:
            break
    assert not d[0] and moved >= 8 and max(speeds) > 0.01 and speeds[-1] == 0.0                         # slid for many steps, at rest at the end
    first = next(k for k, v in enumerate(speeds) if v > 0)
    dec = cfg.push_friction * cfg.push_gravity * cfg.push_dt
    after = [v for v in speeds[4:] if v > 0]
    assert first <= 2 and all(abs((a_ - b_) - dec) < 1e-12 for a_, b_ in zip(after[:-1], after[1:]))      # Coulomb: a constant step down
    # a tool that comes down ON the cube presses it onto the table: nothing moves in the plane
    st2 = O.PushState(1)
    O.push_reset_with_goal(kuka, cfg, st2, [[0.55, 0.0, 0.01, 0.55, 0.23, 0.01]])
    o2 = None
    for _ in range(60):
        p = np.array([0.55, 0.0, 0.1]) if o2 is None else o2[0, :3]
        o2, r2, d2, s2, _ = O.push_step(kuka, cfg, st2, np.clip((np.array([0.55, 0.0, 0.0]) - p) / 0.08, -1, 1)[None])
    assert o2[0, 2] < 0.01 and np.allclose(st2.aux[0, :2], [0.55, 0.0], atol=1e-7) and not st2.aux[0, 7:9].any()


def test_pick_placement_and_reset(O, kuka):
    cfg = O.default_config("pick")
    st = O.PickState(4096)
    obs = O.pick_reset(kuka, cfg, st, seed=5)
    d = np.linalg.norm(st.aux[:, 0:3] - st.aux[:, 3:6], axis=1)
    spawn = st.aux[:, 0:3].copy(); spawn[:, 2] = 0.01                    # :194: the placement test sees the cube at its spawn height
    ds = np.linalg.norm(spawn - st.aux[:, 3:6], axis=1)
    assert (ds >= 0.22).all() and (ds <= 0.25).all()                     # rl_pick_env.py:205-208 (3-D distance)
    assert np.allclose(st.aux[:, 2], 0.01 - 10.0 / 240.0 ** 2, atol=1e-15)  # the push task's cube in the same scene (:210): one step into its fall
    assert st.aux[:, 5].min() >= 0.0 and st.aux[:, 5].max() <= 0.26 and st.aux[:, 5].std() > 0.03   # :200 floating target
    assert np.allclose(st.aux[:, 6], d) and not st.aux[:, 7:].any() and (st.episode == 1).all()
    assert np.abs(obs[:, 3:9] - st.aux[:, :6].astype(np.float32)).max() == 0
    g = golden_json("fk_kat.json")
    assert np.abs(obs[:, :3] - np.float32(g["p_f32"])).max() <= 6e-8     # link-7 frame, not the gripper tip (:264)


def test_pick_arm_pipeline_and_gripper(O, kuka):
    """rl_pick_env.py:310-351: dv = 0.08, z <= 0.55 + 0.257, start position rounded through float32, joint 7 never
    written; build-defined gripper: closes within 6 mm of the cube (:412), holds a cube centred under the tool, the
    held cube rides with the tip, success when it reaches the floating target (:425).  Gripper mechanics on the caller's cube
    height (push_contact_model = 0: no fall, the cube stays where reset_with_goal puts it); the fall has its own test below."""
    cfg = O.default_config("pick")
    cfg.push_contact_model = 0
    L = cfg.pick_gripper_length
    st = O.PickState(1)
    O.pick_reset_with_goal(kuka, cfg, st, [[0.5, 0.0, 0.01, 0.5, 0.1, 0.2]])
    q7 = st.q[0, 6]
    p0 = O.fk(kuka, st.q)[0][0]
    obs, r, d, s, it = O.pick_step(kuka, cfg, st, np.array([[0.0, 0.0, 1.0]]))
    want_z = float(np.float32(p0[2])) + 0.08                             # f32-rounded start (:328) + dv * a
    assert abs(O.fk(kuka, st.q)[0][0][2] - want_z) < 1e-4 and abs(want_z - (p0[2] + 0.08)) < 1e-7
    assert st.q[0, 6] == q7 and r[0] == -1.0 and not d[0] and st.aux[0, 7] == 0
    for _ in range(12):                                                   # straight up: clipped at 0.55 + L (:313)
        obs, r, d, s, it = O.pick_step(kuka, cfg, st, np.array([[0.0, 0.0, 2.0]]))
    assert obs[0, 2] <= 0.55 + L + 1e-4 and st.q[0, 6] == q7
    # scripted grasp and lift
    def act():
        tip = obs[0, :3].astype(np.float64) - [0, 0, L]
        cube, tgt, grip, off = st.aux[0, 0:3], st.aux[0, 3:6], st.aux[0, 7], st.aux[0, 8:11]
        if grip == 2: want = tgt - off
        elif np.linalg.norm(tip[:2] - cube[:2]) > 0.004: want = cube + [0, 0, 0.10]
        else: want = cube + [0, 0, 0.05]
        a = (want - tip) / 0.08
        return (a / max(np.abs(a).max(), 1.0))[None]
    closed_at = None
    for t in range(40):
        obs, r, d, s, it = O.pick_step(kuka, cfg, st, act())
        tip = O.fk(kuka, st.q)[0][0] - [0, 0, L]
        if st.aux[0, 7] == 2 and closed_at is None:
            closed_at = t
            gap = np.linalg.norm(np.maximum(np.abs(tip - st.aux[0, 0:3]) - 0.02, 0)) - 0.03
            assert gap < 0.006 and np.allclose(st.aux[0, 0:3], [0.5, 0.0, 0.01])   # triggered by proximity, cube untouched
        elif st.aux[0, 7] == 2:
            assert np.abs(st.aux[0, 0:3] - (tip + st.aux[0, 8:11])).max() < 1e-6 or abs(st.aux[0, 2] - 0.01) < 1e-8
        if d[0]:
            break
    assert closed_at is not None and d[0] and s[0] and r[0] == 100.0 and st.aux[0, 2] > 0.15 and st.q[0, 6] == q7
    # a side approach closes the gripper without a hold; the closed gripper then only pushes
    st = O.PickState(1)
    O.pick_reset_with_goal(kuka, cfg, st, [[0.5, 0.0, 0.01, 0.5, 0.1, 0.2]])
    obs = np.zeros((1, 9), np.float32); obs[0, :3] = O.fk(kuka, st.q)[0][0]
    for t in range(40):
        tip = obs[0, :3].astype(np.float64) - [0, 0, L]
        way = np.array([0.5, -0.12, 0.02]) if (abs(tip[2] - 0.02) > 0.004 or tip[1] < -0.125) and st.aux[0, 7] == 0 and t < 15 \
            else np.array([0.5, 0.1, 0.02])
        a = (way - tip) / 0.08
        obs, r, d, s, it = O.pick_step(kuka, cfg, st, (a / max(np.abs(a).max(), 2.0))[None])
    assert st.aux[0, 7] == 1 and abs(st.aux[0, 2] - 0.01) < 1e-8 and st.aux[0, 1] > 0.02    # pushed along +y on the table, never lifted


def test_pick_cube_falls_like_the_push_cube_two_calls_per_step(O, kuka):
    """RLPickEnv loads the push task's cube into the same scene (rl_pick_env.py:210: models/cube_small_push.urdf at z = 0.01 over the
    table) and calls stepSimulation in reset() (:242) and TWICE per env step (step() :348, and _reward() :417 behind the observation):
    the cube's fall -- pinned for that body and scene by the reference's recorded push runs -- is observed after call 2 j at env step j:
    free fall through call 13, 0.8 mm into the table, then back up to the rest height; on the table from step 7 on."""
    cfg = O.default_config("pick")
    assert cfg.push_contact_model == 1
    st = O.PickState(1)
    O.pick_reset_with_goal(kuka, cfg, st, [[0.5, 0.0, 0.01, 0.5, 0.1, 0.2]])
    c = 0.5 * 10.0 / 240.0 ** 2
    assert abs(st.aux[0, 2] - (0.01 - c * 2)) < 1e-15
    zs = []
    for j in range(1, 31):
        obs, r, d, s, it = O.pick_step(kuka, cfg, st, np.array([[0.0, 0.0, 1.0]]))       # the arm goes up, away from the cube
        zs.append(float(obs[0, 5]))
    for j in range(1, 7):                                                   # observed after call 2 j <= 12: free fall
        assert abs(zs[j - 1] - (0.01 - c * (2 * j) * (2 * j + 1))) < 1e-7, j
    assert zs[6] < cfg.push_rest_z and all(zs[j] >= zs[j - 1] for j in range(7, 30)) and abs(zs[29] - cfg.push_rest_z) < 1e-5
    assert st.aux[0, 7] == 0 and np.allclose(st.aux[0, 0:2], [0.5, 0.0], atol=1e-7)


# ------------------------------------------------------------------------------ trajectory store + HER (next row 8f.1)

@pytest.mark.parametrize("task", ["reach", "push"])
def test_her_sampler_restatement_matches_reference(task):
    """G6: the reference's own sampler outputs, with the draws it made (utils/rl_utils.py:108-199)."""
    from oracle import her
    g = golden_npz(f"her_{task}_seed0.npz")
    ch = {k: g[k] for k in ("obs0", "obs_after", "next_obs", "action", "reward", "done")}
    eps = her.index_episodes(g["done"])
    assert np.array_equal(eps, g["episodes"])
    assert (g["picks"][:, 2] == 1).sum() > 150 and (g["picks"][:, 2] == 0).sum() > 20
    out = her.sample_with_picks(ch, eps, g["picks"], float(g["dis_threshold"]))
    assert np.array_equal(out["states"].astype(np.float64), g["states"])
    assert np.array_equal(out["next_states"].astype(np.float64), g["next_states"])
    assert np.array_equal(out["actions"], g["actions"]) and np.array_equal(out["dones"], g["dones"])
    assert np.abs(out["rewards"].astype(np.float64) - g["rewards"]).max() < 1e-7      # -0.1 is not exact in f32
    assert 0.1 < g["dones"].mean() < 0.9


def test_oracle_joint_limit_projection(O, kuka):
    """clamp_joint_limits: the oracle's IK result projected onto the URDF limits of
    /root/reference/envs/bmirobot_joints_info_pybullet.txt:1-7 (fields 8-9, pinned by joint_info.json); results inside the
    limits are untouched, and the fence flags mark exactly the calls whose unclamped result leaves them."""
    info = golden_json("joint_info.json")
    lim = np.array(O.KUKA["limit"])
    rows = info["kuka"] if isinstance(info, dict) and "kuka" in info else None
    if rows is not None:
        for j, r in enumerate(rows[:7]):
            lo = r.get("lower", r.get("jointLowerLimit")); hi = r.get("upper", r.get("jointUpperLimit"))
            if lo is not None:
                assert abs(lo + lim[j]) < 1e-9 and abs(hi - lim[j]) < 1e-9
    cfg0, cfg1 = O.default_config(), O.default_config()
    cfg1.clamp_joint_limits = 1
    assert list(cfg0.lim_hi) == list(lim) and list(cfg0.lim_lo) == list(-lim) and cfg0.fence_z == 0.05
    rng = np.random.default_rng(0)
    n = 512
    q = np.tile(np.array(O.INIT_Q), (n, 1)) + rng.uniform(-0.3, 0.3, (n, 7))
    j = rng.integers(0, 7, n // 2)
    q[np.arange(n // 2), j] = rng.choice([-1.0, 1.0], n // 2) * (lim[j] - rng.uniform(0, 0.01, n // 2))
    p0, _ = O.fk(kuka, q)
    tgt = np.clip(p0 + rng.normal(0, 0.014, (n, 3)), [0.2, -0.3, 0.0], [0.7, 0.3, 0.55])
    q0, it0 = O.ik(kuka, cfg0, q, tgt)
    q1, it1 = O.ik(kuka, cfg1, q, tgt)
    out = (np.abs(q0) > lim).any(1)
    assert out.sum() > 20 and np.array_equal(it0, it1)
    assert np.array_equal(q1, np.clip(q0, -lim, lim)) and np.array_equal(q1[~out], q0[~out])
    flags = O.fence_flags(kuka, cfg0, q, tgt)
    assert np.array_equal((flags & 1) != 0, out)
    p1, _ = O.fk(kuka, q0)
    assert np.array_equal((flags & 2) != 0, p1[:, 2] < 0.05)


def test_oracle_reproduces_the_reference_runs_first_episodes(O):
    """The known answer for the WHOLE path -- FK, clipped target, calculateInverseKinematics, resetJointState, stepSimulation,
    _reward, free-running -- that real PyBullet computed: the per-episode returns of the reference's recorded
    train_reach_with_TD3 run (visdata/reach/TD3_0.01/Reach_TD3.json -> tests/golden/visdata_reach_td3.json, main.py:165-231,
    opt.random_seed = 0).  No network update happens before five episodes are stored, so episodes 1-5 depend only on seeded
    streams (goals, actor init = golden G3, exploration noise) and on the env: 2 068 env steps, a success at step 64 and four
    501-step time-outs, 276 steps with a joint beyond its URDF limit and 274 with the flange below z = 0.05.  The oracle at its
    DEFAULT switches reproduces the five returns to 6e-4 (4e-7 relative; ~1e-7 m per step); see tests/reference_run.py."""
    import reference_run as R
    fx = R.fixture_returns()
    out, fence = R.replay_on_oracle(O, 5)
    assert [n for _, n, _ in out] == [64, 501, 501, 501, 501] and [s for _, _, s in out] == [True, False, False, False, False]
    diffs = [abs(r - x) for (r, _, _), x in zip(out, fx)]
    assert max(diffs) < 8e-4, diffs
    assert fence[0] > 200 and fence[1] > 200            # the run does visit the steps the limit / flange counters name


@pytest.mark.parametrize("name,setter,worst", [
    ("ik_exit_mode=1", lambda c: setattr(c, "ik_exit_mode", 1), 1e-2),
    ("ik_tip_offset=inertial", lambda c: c.ik_tip_offset.__setitem__(slice(0, 3), [0.0, 0.0, 0.02]), 100.0),
    ("clamp_joint_limits=1", lambda c: setattr(c, "clamp_joint_limits", 1), 10.0),
    ("clamp_joint_limits=2", lambda c: setattr(c, "clamp_joint_limits", 2), 10.0)])
def test_reference_run_rejects_the_other_switch_settings(O, name, setter, worst):
    """The same data decides the restatement's named unknowns: Bullet's loop form (not test-before-update), the URDF link frame
    (not the inertial frame 2 cm along the tool axis), and NO joint-limit push-back inside stepSimulation -- each alternative
    misses the recorded returns by orders of magnitude more than the default's 6e-4."""
    import reference_run as R
    fx = R.fixture_returns()
    out, _ = R.replay_on_oracle(O, 5, setter)
    assert max(abs(r - x) for (r, _, _), x in zip(out, fx)) > worst, name


def test_push_placement_stream_and_rest_height_against_the_recorded_push_run(O):
    """P1 against real numbers: the reference's recorded train_push_with_TD3 run (visdata/push/origin_TD3/TD3.json ->
    tests/golden/visdata_push_td3.json; main.py:449-515, seed 0).  An episode whose cube is never touched returns
    500 x (-1) - 50 |cube - target| (rl_push_env.py:393-394,418-420,427); |cube - target| = sqrt(planar^2 + dz^2) with the
    placement distance of THAT reset -- a fixed function of random.seed(0), 6 draws per placement try (:197-209) and 3 per step
    (:435-437) while every episode lasts 501 steps (true up to episode 32) -- and dz = how far the dynamic cube settles below
    the fixed target.  Episodes 4, 6, 13 and 24 of the recorded run fit ONE dz = 14.74 mm (ArmEnvConfig.push_rest_z) to 2e-3."""
    import reference_run as R
    fx = R.push_fixture_returns()
    cfg = O.default_config("push")
    dz = cfg.push_place_z - cfg.push_rest_z
    assert abs(dz - 0.01474) < 1e-12
    base = R.push_untouched_returns(32, dz)
    for ep in (4, 6, 13, 24):
        assert abs(fx[ep - 1] - base[ep - 1]) < 2e-3, (ep, fx[ep - 1], base[ep - 1])
    # every recorded return lies near its episode's untouched baseline (contacts move the cube by centimetres: a few units of return)
    assert max(abs(fx[k] - base[k]) for k in range(32)) < 8.0
    # and the fit is sharp: with the cube at the target's height (dz = 0) the four episodes are off by 0.02
    flat = R.push_untouched_returns(32, 0.0)
    assert min(abs(fx[ep - 1] - flat[ep - 1]) for ep in (4, 6, 13, 24)) > 0.015


def test_oracle_push_env_on_the_recorded_runs_first_episodes(O):
    """The first five episodes of the reference's two recorded push runs (before any network update: untrained 9-input TD3 actor of
    torch.manual_seed(0), N(0, 0.392) exploration from np.random.seed(0); same trajectories, two rewards -- tests/reference_run.py)
    on the oracle's push env.
    * The runs are consistent: the count of moving steps M derived from the two returns of an episode is an integer to 1e-2.
    * Episode 4, which Bullet's arm does not touch: BOTH returns to 2e-3 -- the shipped reward's -504.1221 (the cube's free fall:
      eight steps above the 1e-5 threshold, nothing fitted but the rest height) and the earlier reward's -512.0719 (the stand-in's
      tool grazes the cube on three steps, 2e-5 of cube travel).  Rounds 1-4 (push_contact_model = 0: cube at rest from reset on)
      return -512.07 under the SHIPPED reward, 7.95 off.
    * Touched episodes 1, 2, 3, 5 (Bullet's contact dynamics are not restated: a planar stand-in, nominal flange geometry, TWO fitted
      numbers): counts of moving steps 149 / 192 / 90 / 43 against Bullet's 149 / 192 / 84 / 32 (rounds 1-4: 5 / 6 / 5 / 4), returns
      under the SHIPPED reward within 12.4 (rounds 1-4: 25-195 off), final cube-target distances within 4.3 cm (the earlier reward's
      returns within 2.2; rounds 1-4: 3.3).  Row P3 stays partial."""
    import reference_run as R
    org, upd = R.push_fixture_returns("origin"), R.push_fixture_returns("updata")
    rec = R.push_recorded_observables(6)
    assert [round(m) for _, m, _ in rec] == [149, 192, 84, 8, 32, 8] and max(abs(m - round(m)) for _, m, _ in rec) < 1e-2
    out = R.replay_push_on_oracle(O, 5)
    assert [o["n"] for o in out] == [501] * 5
    assert abs(out[3]["ret"] - upd[3]) < 2e-3 and abs(out[3]["ret_origin"] - org[3]) < 2e-3 and out[3]["M"] == 8 and out[3]["moved"] <= 5, out[3]
    assert [o["M"] > 8 for o in out] == [True, True, True, False, True] and [o["moved"] > 20 for o in out] == [True, True, True, False, True]
    for k in (0, 1, 2, 4):
        assert abs(out[k]["M"] - rec[k][1]) <= 12, (k, out[k]["M"], rec[k][1])
        assert abs(out[k]["ret"] - upd[k]) < 13.0, (k, out[k]["ret"], upd[k])
        assert abs(out[k]["d_f"] - rec[k][0]) < 0.045 and abs(out[k]["ret_origin"] - org[k]) < 2.2, (k, out[k], rec[k])
    legacy = R.replay_push_on_oracle(O, 5, lambda c: setattr(c, "push_contact_model", 0))
    assert abs(legacy[3]["ret"] - upd[3]) > 7.9 and max(abs(legacy[k]["ret_origin"] - org[k]) for k in (0, 1, 2, 4)) > 3.0
    assert max(legacy[k]["M"] for k in range(5)) <= 6 and min(abs(legacy[k]["ret"] - upd[k]) for k in (0, 1, 2, 4)) > 24.0


@pytest.mark.parametrize("fixture", ["datd3_take_action_seed0.npz", "datd3_take_action9_seed0.npz"])
def test_oracle_datd3_take_action_matches_reference_golden(O, fixture):
    """G11 (reach, 6-float observations) / G14 (push / pick, 9): the oracle's batched DATD3 take_action (two actors, two critics on cat(s, a_i), the better-valued action) against vectors
    produced by calling the reference's own DATD3_MLP.take_action one state at a time (algo/DATD3/DATD3_mlp.py:88-109): Q values to
    1e-5, the same actor picked wherever the two Q values are not within rounding of each other (both branches occur), actions 1e-6."""
    g = golden_npz(fixture)
    nets = [{k: g["%s_%s" % (n, k.replace(".", "_"))] for k in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc3.weight", "fc3.bias")}
            for n in ("actor1", "actor2", "critic1", "critic2")]
    a, q1, q2, pick = O.datd3_take_action(nets, g["states"], float(g["action_bound"]))
    assert np.abs(q1 - g["q1"]).max() < 1e-5 and np.abs(q2 - g["q2"]).max() < 1e-5
    clear = np.abs(g["q1"] - g["q2"]) > 1e-4
    assert clear.sum() >= 245 and 100 < g["picked_actor"][clear].sum() < 156
    assert np.array_equal(pick[clear], g["picked_actor"][clear].astype(np.uint8)) and np.abs(a - g["actions"])[clear].max() < 1e-6


@pytest.mark.parametrize("fixture,names", [("daddpg_take_action_seed0.npz", ("actor1", "actor2", "critic", "critic")),
                                           ("daddpg_take_action9_seed0.npz", ("actor1", "actor2", "critic", "critic")),
                                           ("darc_take_action_seed0.npz", ("actor1", "actor2", "critic1", "critic2"))])
def test_oracle_daddpg_darc_take_action_match_reference_golden(O, fixture, names):
    """G15: the oracle's batched take_action given DADDPG's net table (two actors and the ONE critic in both critic places,
    algo/DADDPG/DADDPG_mlp.py:77-97 -- the reference's default agent, config.py:33) and DARC's (algo/DARC/DARC_mlp.py:92-113, the same
    selection as DATD3's) against vectors produced by calling the reference's own take_action one state at a time: Q values to 1e-5, the
    same actor picked wherever the two Q values are not within rounding of each other (both branches occur), actions 5e-6 (G15's
    DADDPG actors have a wide first layer -- activations of order 10, where f32 summation order shows at 1e-6; the bar is 1e-5)."""
    g = golden_npz(fixture)
    nets = [{k: g["%s_%s" % (n, k.replace(".", "_"))] for k in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc3.weight", "fc3.bias")}
            for n in names]
    a, q1, q2, pick = O.datd3_take_action(nets, g["states"], float(g["action_bound"]))
    assert np.abs(q1 - g["q1"]).max() < 1e-5 and np.abs(q2 - g["q2"]).max() < 1e-5
    clear = np.abs(g["q1"] - g["q2"]) > 1e-4
    assert clear.sum() >= 240 and 10 < g["picked_actor"][clear].sum() < 246
    assert np.array_equal(pick[clear], g["picked_actor"][clear].astype(np.uint8)) and np.abs(a - g["actions"])[clear].max() < 5e-6


def test_policy_noise_angle_kernel_against_libm(O):
    """The sin / cos pair of the in-kernel policy's Box-Muller (sincos_2pi, engine and oracle the same statements): within 1.2e-7 of
    sin / cos of 2 pi u in f64 over a million u in [0, 1), quadrant boundaries and the ends included; and the stream's normals have
    the moments of N(0, 1)."""
    import ctypes as C
    rng = np.random.default_rng(0)
    u = np.concatenate([rng.random(1 << 20, dtype=np.float32), np.float32([0.0, 0.125, 0.25, 0.375, 0.5, 0.625, 0.75, 0.875, np.nextafter(np.float32(1), np.float32(0))])])
    u = np.ascontiguousarray(u[u < 1.0])
    s, c = np.zeros_like(u), np.zeros_like(u)
    O.lib().orc_sincos_2pi_batch(C.c_int64(u.size), O._p(u), O._p(s), O._p(c))
    a = 2.0 * np.pi * u.astype(np.float64)
    assert np.abs(s - np.sin(a)).max() < 1.2e-7 and np.abs(c - np.cos(a)).max() < 1.2e-7
    st = O.ReachState(1 << 16); st.episode[:] = 1; st.step[:] = 7
    nz = O.policy_noise(st, 3, 0)
    assert abs(float(nz.mean())) < 0.01 and abs(float(nz.std()) - 1.0) < 0.01 and abs(float((nz ** 4).mean()) - 3.0) < 0.1
