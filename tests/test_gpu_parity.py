"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs; against the committed golden fixtures; and -- at BASELINE.json's full size -- through
size-independent properties.

Tolerances (north_star: joint positions and reward within 1e-4 of the reference):
  f64 engine vs f64 oracle   FK 1e-12 m; one IK call / one env step 1e-6 rad (the orientation error's
                             2*acos(w) has ~3e-8 rad of conditioning noise at convergence, in Bullet too),
                             reward 1e-6, obs 1 f32 ulp
  f32 engine vs f64 oracle   one env step 1e-4 rad / 1e-4 reward (the stated tolerance)
"""
import math
import os

import numpy as np
import pytest
import torch

from conftest import golden_json, golden_npz

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def envs():
    from armenv import envs
    return envs


def _np(t):
    return t.detach().cpu().numpy()


def _same_rollout_step(out, t, o, r, d):
    """Row t of a multi-step rollout against the same step run as its own launch: the same bits.  (cos q, sin q) travel
    with q in the env's state, so a trajectory is a pure function of (state, actions) however the steps are grouped into
    launches (round 1 re-derived the pair at every launch start and the two agreed only to the IK's 1e-7 rad noise floor)."""
    assert torch.equal(out["done"][t], d), t
    assert torch.equal(out["obs"][t], o), (t, (out["obs"][t] - o).abs().max().item())
    assert torch.equal(out["reward"][t], r), t


def _actions(rng, n):
    """the run() exploration distribution with a zero actor, main.py:116-117"""
    return np.clip(rng.normal(0.0, 0.7 * 0.98, (n, 3)), -0.7, 0.7).astype(np.float32)


def _mk(envs, n, **kw):
    return envs.BatchedReachEnv(n, device=DEV, **kw)


# ------------------------------------------------------------------------------ FK (R5)

def test_library_is_the_hip_build(envs):
    e = _mk(envs, 64)
    assert e.kernel_name == "reach_step<f64,kuka>"
    e.close()


@pytest.mark.parametrize("precision", [64, 32])
def test_fk_known_answer_on_gpu(envs, precision):
    g = golden_json("fk_kat.json")
    e = _mk(envs, 1, precision=precision)
    pos, quat = e.fk(torch.tensor([g["q"]], dtype=torch.float64))
    assert np.abs(_np(pos)[0] - np.array(g["p_f32"])).max() < (1e-7 if precision == 64 else 5e-7)
    obs = _np(e.reset())
    assert np.abs(obs[0, :3] - np.float32(g["p_f32"])).max() <= (6e-8 if precision == 64 else 5e-7)
    e.close()


@pytest.mark.parametrize("robot", ["kuka", "diana"])
@pytest.mark.parametrize("fk_path", [0, 1])
@pytest.mark.parametrize("precision", [64, 32])
def test_fk_matches_oracle(envs, O, robot, fk_path, precision):
    rng = np.random.default_rng(10)
    q = rng.uniform(-3.0, 3.0, (4096 + 37, 7))       # ragged size: not a multiple of the block
    e = _mk(envs, 8, robot=robot, fk_path=fk_path, precision=precision)
    assert ("generic" in e.kernel_name) == (fk_path == 1)
    pos, quat = e.fk(torch.from_numpy(q))
    p_ref, q_ref = O.fk(O.make_chain(robot), q)
    # the built-in fast path snaps the URDF's 11-digit rpy text to exact signed permutations
    # (|delta R| ~ 5e-12); the generic path uses the same rpy-derived matrices as the oracle
    tol = (5e-11 if fk_path == 0 else 1e-12) if precision == 64 else 5e-6
    assert np.abs(_np(pos) - p_ref).max() < tol
    dq = np.minimum(np.abs(_np(quat) - q_ref).max(1), np.abs(_np(quat) + q_ref).max(1))
    # getRotation switches branch on the trace / largest diagonal; both give the same rotation
    assert np.median(dq) < (tol if precision == 64 else 1e-5)
    assert dq.max() < (1e-6 if precision == 64 else 2e-2)
    e.close()


def test_fk_generic_chain_with_base_transform(envs, O):
    """Diana S1 as diana_cam_reach.py loads it: base yawed by pi (envs/diana_cam_reach.py:202-203)."""
    from armenv.urdf import builtin_chain
    ch = builtin_chain("diana").with_base(xyz=(0.1, -0.2, 0.3), rpy=(0.0, 0.0, math.pi))
    e = _mk(envs, 8, chain=ch)
    assert "generic" in e.kernel_name
    rng = np.random.default_rng(11)
    q = rng.uniform(-3, 3, (513, 7))
    pos, _ = e.fk(torch.from_numpy(q))
    p_ref, _ = O.fk(O.make_chain("diana", base_xyz=(0.1, -0.2, 0.3), base_rpy=(0, 0, math.pi)), q)
    assert np.abs(_np(pos) - p_ref).max() < 1e-12
    e.close()


def test_fk_large_joint_angles(envs, O):
    """The lockstep Cody-Waite sincos has no library fallback: its fused reduction stays exactly rounded far beyond any joint
    angle a live env can hold.  |q| up to 1e7 rad (1.6 million turns) against the oracle's libm sincos; non-finite angles
    must come out non-finite, not as a plausible pose."""
    rng = np.random.default_rng(12)
    q = rng.uniform(-1.0, 1.0, (2048, 7)) * 10.0 ** rng.uniform(0, 7, (2048, 1))
    e = _mk(envs, 8, fk_path=1)
    pos, _ = e.fk(torch.from_numpy(q))
    p_ref, _ = O.fk(O.make_chain("kuka"), q)
    assert np.abs(_np(pos) - p_ref).max() < 1e-9
    bad = np.zeros((2, 7)); bad[0, 3] = np.inf; bad[1, 5] = np.nan
    pos, _ = e.fk(torch.from_numpy(bad))
    assert not np.isfinite(_np(pos)).all(axis=1).any()
    e.close()


def test_fk_empty_batch(envs):
    e = _mk(envs, 4)
    pos, quat = e.fk(torch.empty((0, 7), dtype=torch.float64))
    assert pos.shape == (0, 3)
    e.close()


# ------------------------------------------------------------------------------ IK (R6)

@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("robot,fk_path", [("kuka", 0), ("kuka", 1), ("diana", 0)])
def test_ik_matches_oracle_f64(envs, O, robot, fk_path, mode):
    rng = np.random.default_rng(20)
    n = 2048 + 5
    ch = O.make_chain(robot)
    cfg = O.default_config(); cfg.ik_exit_mode = mode
    if robot == "diana":
        cfg.target_quat[:] = [0.0, 0.0, 0.0, 1.0]
    # start poses near the workspace: init pose plus noise; targets 0..3 cm away like an env step
    q0 = np.array(O.INIT_Q) + rng.normal(0, 0.3, (n, 7))
    p0, _ = O.fk(ch, q0)
    tgt = p0 + rng.normal(0, 0.012, (n, 3))
    e = _mk(envs, 8, robot=robot, fk_path=fk_path, ik_exit_mode=mode, target_quat=list(cfg.target_quat))
    q_gpu, it_gpu = e.ik(torch.from_numpy(q0), torch.from_numpy(tgt))
    q_ref, it_ref = O.ik(ch, cfg, q0, tgt)
    same = _np(it_gpu) == it_ref
    assert same.mean() > 0.999                       # a residual within ~1e-15 of 1e-4 may flip one trip
    # Diana from these arbitrary poses passes near singular configurations, where the damped solve
    # amplifies the 3e-8 rad conditioning noise of 2*acos(w) by up to ~1/(2 sqrt(lambda)) = 158
    assert np.abs(_np(q_gpu) - q_ref)[same].max() < (1e-6 if robot == "kuka" else 1e-5)
    assert it_ref.max() <= 20 and it_ref.min() >= (1 if mode == 0 else 0)
    e.close()


def test_ik_iteration_cap_and_far_targets(envs, O):
    """Unreachable targets run into the 20-iteration cap with 45-degree-clamped updates."""
    rng = np.random.default_rng(21)
    n = 256
    q0 = np.tile(O.INIT_Q, (n, 1))
    tgt = rng.uniform(-2.0, 2.0, (n, 3))
    e = _mk(envs, 8)
    q_gpu, it_gpu = e.ik(torch.from_numpy(q0), torch.from_numpy(tgt))
    q_ref, it_ref = O.ik(O.make_chain("kuka"), O.default_config(), q0, tgt)
    assert it_ref.max() == 20 and np.array_equal(_np(it_gpu), it_ref)
    # long clamped trajectories amplify rounding; compare the well-conditioned majority tightly
    err = np.abs(_np(q_gpu) - q_ref).max(1)
    assert np.median(err) < 1e-8 and np.isfinite(_np(q_gpu)).all()
    e.close()


def test_ik_f32_within_stated_tolerance(envs, O):
    rng = np.random.default_rng(22)
    n = 4096
    ch = O.make_chain("kuka"); cfg = O.default_config()
    st = O.ReachState(n); O.reach_reset(ch, cfg, st, seed=5)
    for _ in range(3):                                # leave the post-reset yaw transient
        O.reach_step(ch, cfg, st, _actions(rng, n))
    p0, _ = O.fk(ch, st.q)
    tgt = np.clip(p0 + 0.02 * _actions(rng, n), cfg.box_lo[:], cfg.box_hi[:])
    e = _mk(envs, 8, precision=32)
    q_gpu, it_gpu = e.ik(torch.from_numpy(st.q), torch.from_numpy(tgt))
    q_ref, it_ref = O.ik(ch, cfg, st.q, tgt)
    same = _np(it_gpu) == it_ref
    assert same.mean() > 0.98
    assert np.abs(_np(q_gpu) - q_ref)[same].max() < 1e-4
    p1, _ = O.fk(ch, _np(q_gpu))
    assert np.linalg.norm(p1 - tgt, axis=1).max() < 2e-4
    e.close()


# ------------------------------------------------------------------------------ reset (R2)

def test_reset_matches_oracle_and_golden(envs, O, kuka):
    n = 4096 + 3
    cfg = O.default_config()
    e = _mk(envs, n, seed=99, env_id_offset=1234)
    obs = _np(e.reset())
    st = O.ReachState(n)
    obs_ref = O.reach_reset(kuka, cfg, st, seed=99, env_id0=1234)
    assert np.array_equal(obs[:, 3:], obs_ref[:, 3:])                      # goals: bit-exact (Philox + f64 affine)
    assert np.abs(obs[:, :3] - obs_ref[:, :3]).max() <= 6e-8
    s = e.get_state()
    assert np.array_equal(_np(s["q"]), st.q) and np.array_equal(_np(s["goal"]), st.goal)
    assert (_np(s["step"]) == 0).all() and (_np(s["episode"]) == 1).all()
    # masked reset touches only the masked envs
    mask = torch.zeros(n, dtype=torch.uint8); mask[::3] = 1
    e.reset(mask=mask)
    s2 = e.get_state()
    ep = _np(s2["episode"])
    assert (ep[::3] == 2).all() and (np.delete(ep, np.arange(0, n, 3)) == 1).all()
    m = _np(mask).astype(bool)
    O.reach_reset(kuka, cfg, st, seed=99, env_id0=1234, mask=_np(mask))
    assert np.array_equal(_np(s2["goal"]), st.goal)
    assert np.array_equal(_np(s2["goal"])[~m], _np(s["goal"])[~m])
    e.close()


def test_reset_with_goal_and_state_roundtrip(envs):
    n = 130
    e = _mk(envs, n)
    goal = torch.rand(n, 3)
    obs = _np(e.reset(goal=goal))
    assert np.array_equal(obs[:, 3:], goal.numpy())
    rng = np.random.default_rng(0)
    q = rng.uniform(-1, 1, (n, 7)); step = rng.integers(0, 500, n).astype(np.int32)
    ret = rng.normal(size=n); epi = rng.integers(0, 1000, n).astype(np.int32)
    e.set_state(q=q, step=step, ep_return=ret, episode=epi)
    s = e.get_state()
    assert np.array_equal(_np(s["q"]), q) and np.array_equal(_np(s["step"]), step)
    assert np.array_equal(_np(s["ep_return"]), ret) and np.array_equal(_np(s["episode"]), epi)
    assert np.array_equal(_np(s["goal"]), goal.numpy())
    e.close()


# ------------------------------------------------------------------------------ step (R3, R4)

@pytest.mark.parametrize("robot,fk_path,mode", [("kuka", 0, 0), ("kuka", 0, 1), ("kuka", 1, 0), ("diana", 0, 0)])
def test_step_teacher_forced_f64(envs, O, robot, fk_path, mode):
    """Every step starts from the oracle's state (set_state), so the comparison is per step."""
    n = 1024 + 9
    rng = np.random.default_rng(30)
    ch = O.make_chain(robot)
    cfg = O.default_config(); cfg.ik_exit_mode = mode
    over = {}
    if robot == "diana":      # a reachable set-up for the second chain (diana_cam_reach.py:102-104,156-157)
        cfg.target_quat[:] = [1.0, 0.0, 0.0, 0.0]
        cfg.q_init[:] = [0.0, 0.5, 0.0, 1.6, 0.0, -1.0, 0.0]
        p_init, _ = O.fk(ch, cfg.q_init[:])
        lo = (p_init[0] - 0.25).tolist(); hi = (p_init[0] + 0.25).tolist()
        cfg.box_lo[:] = lo; cfg.box_hi[:] = hi; cfg.goal_lo[:] = lo; cfg.goal_hi[:] = hi
        over = dict(target_quat=list(cfg.target_quat), q_init=list(cfg.q_init), box_lo=lo, box_hi=hi, goal_lo=lo, goal_hi=hi)
    e = _mk(envs, n, robot=robot, fk_path=fk_path, auto_reset=False, seed=3, ik_exit_mode=mode, **over)
    st = O.ReachState(n)
    O.reach_reset(ch, cfg, st, seed=3)
    e.reset()
    worst_q = worst_r = 0.0
    flips = 0
    for t in range(25):
        a = _actions(rng, n)
        e.set_state(q=st.q, goal=st.goal, step=st.step, ep_return=st.ep_return)
        obs, rew, done, succ = e.step(torch.from_numpy(a).to(DEV))
        obs, rew, done, succ = _np(obs).copy(), _np(rew).copy(), _np(done).copy(), _np(succ).copy()
        obs_r, rew_r, done_r, succ_r, iters = O.reach_step(ch, cfg, st, a)
        s = e.get_state()
        dq = np.abs(_np(s["q"]) - st.q).max(1)
        ok = dq < 1e-6
        flips += int((~ok).sum())
        worst_q = max(worst_q, dq[ok].max())
        worst_r = max(worst_r, np.abs(rew.astype(np.float64) - rew_r)[ok].max())
        assert np.abs(obs - obs_r)[ok].max() <= 1.2e-7
        assert np.array_equal(done[ok], done_r[ok].astype(bool)) and np.array_equal(succ[ok], succ_r[ok].astype(bool))
        assert np.array_equal(_np(s["step"]), st.step)
        assert np.abs(_np(s["ep_return"]) - st.ep_return)[ok].max() < 1e-5
    assert flips <= 1, flips                   # an IK residual within rounding of 1e-4 may flip one trip
    assert worst_q < 1e-6 and worst_r < 1e-5


def test_step_teacher_forced_f32(envs, O, kuka):
    n = 4096
    rng = np.random.default_rng(31)
    cfg = O.default_config()
    e = _mk(envs, n, precision=32, auto_reset=False, seed=4)
    st = O.ReachState(n)
    O.reach_reset(kuka, cfg, st, seed=4)
    e.reset()
    bad = 0
    for t in range(20):
        a = _actions(rng, n)
        e.set_state(q=st.q, goal=st.goal, step=st.step, ep_return=st.ep_return)
        obs, rew, done, succ = e.step(torch.from_numpy(a).to(DEV))
        obs, rew = _np(obs).copy(), _np(rew).copy()
        obs_r, rew_r, done_r, succ_r, iters = O.reach_step(kuka, cfg, st, a)
        dq = np.abs(_np(e.get_state()["q"]) - st.q).max(1)
        bad += int((dq >= 1e-4).sum())
        ok = dq < 1e-4
        assert np.abs(rew.astype(np.float64) - rew_r)[ok].max() < 1e-4
        assert np.abs(obs - obs_r)[ok].max() < 1e-4
    assert bad <= 0.002 * 20 * n, bad         # trip-count flips at the 1e-4 residual gate


def _limit_fence_states(O, kuka, cfg, n, rng):
    """Joint states that make the fence fire: perturbed init poses, a third of them with one joint parked just inside one
    of its URDF limits (bmirobot_joints_info_pybullet.txt:1-7), a tenth driven down to z ~ 0.04 (below fence_z)."""
    q = np.tile(np.array(O.INIT_Q), (n, 1)) + rng.uniform(-0.3, 0.3, (n, 7))
    lim = np.array(O.KUKA["limit"])
    k = n // 3
    j = rng.integers(0, 7, k)
    side = rng.choice([-1.0, 1.0], k)
    q[np.arange(k), j] = side * (lim[j] - rng.uniform(0.0, 0.01, k))
    low = np.arange(n - n // 10, n)
    ql = np.tile(np.array(O.INIT_Q), (low.size, 1))
    tgt = np.column_stack([rng.uniform(0.35, 0.6, low.size), rng.uniform(-0.2, 0.2, low.size), rng.uniform(0.03, 0.06, low.size)])
    for _ in range(4):
        ql, _ = O.ik(kuka, cfg, ql, tgt)
    q[low] = ql
    return q


@pytest.mark.parametrize("precision", [64, 32])
def test_joint_limit_clamp_and_parity_fence(envs, O, kuka, precision):
    """N1 / R7.  clamp_joint_limits = 1 projects the IK result onto the URDF limits
    (/root/reference/envs/bmirobot_joints_info_pybullet.txt:1-7; the reference's own limit arrays, rl_reach_env.py:103-107,
    are dead data) and is checked teacher-forced against the oracle's projection; the fence counters -- steps whose
    IK result leaves the limits, steps that end with the flange below fence_z = 0.05 (rl_reach_env.py:258: Bullet's limit
    constraint / arm-table contact act there, this build's stepSimulation is kinematic) -- against the oracle's flags."""
    n = 4096
    rng = np.random.default_rng(123)
    cfg0, cfg1 = O.default_config(), O.default_config()
    cfg1.clamp_joint_limits = 1
    lim = np.array(O.KUKA["limit"])
    free = _mk(envs, n, auto_reset=False, precision=precision, fence_counters=1)          # reference behaviour + bookkeeping
    clamped = _mk(envs, n, auto_reset=False, precision=precision, clamp_joint_limits=1, fence_counters=1)
    assert np.allclose(np.array(free.cfg.chain.limit_hi[:]), lim, atol=1e-11) and free.cfg.fence_z == 0.05
    tol = 1e-6 if precision == 64 else 1e-4
    lim_tol = 1e-12 if precision == 64 else 3e-7          # the f32 engine projects onto the limits rounded to f32
    n_lim = n_low = n_low1 = flips = 0
    for rep in range(3):
        q = _limit_fence_states(O, kuka, cfg0, n, rng)
        a = _actions(rng, n)
        st0, st1 = O.ReachState(n), O.ReachState(n)
        for st in (st0, st1):
            st.q[:] = q; st.goal[:] = np.float32([0.45, 0.1, 0.3])
        p0, _ = O.fk(kuka, q)
        tgt = np.clip(p0 + 0.02 * a.astype(np.float64), [0.2, -0.3, 0.0], [0.7, 0.3, 0.55])
        flags0 = O.fence_flags(kuka, cfg0, q, tgt)
        flags1 = O.fence_flags(kuka, cfg1, q, tgt)
        for env in (free, clamped):
            env.reset(); env.set_state(q=q, goal=st0.goal, step=st0.step)
        c0, c1 = free.counters(), clamped.counters()
        at = torch.from_numpy(a).to(DEV)
        obs_f = _np(free.step(at)[0]).copy(); obs_c = _np(clamped.step(at)[0]).copy()
        obs_r0, *_ = O.reach_step(kuka, cfg0, st0, a)
        obs_r1, *_ = O.reach_step(kuka, cfg1, st1, a)
        qf, qc = _np(free.get_state()["q"]), _np(clamped.get_state()["q"])
        d0, d1 = np.abs(qf - st0.q).max(1), np.abs(qc - st1.q).max(1)
        ok = (d0 < tol) & (d1 < tol)
        flips += int((~ok).sum())
        assert np.abs(obs_f - obs_r0)[ok].max() < max(tol, 2e-7) and np.abs(obs_c - obs_r1)[ok].max() < max(tol, 2e-7)
        assert (np.abs(qc) <= lim + lim_tol).all()                                 # the projection holds for every env
        hit = (flags0 & 1) != 0
        assert np.array_equal(qf[~hit & ok], qc[~hit & ok])                        # envs inside the limits keep their bits
        assert (np.abs(st0.q[hit]) > lim).any(axis=1).all() and (np.abs(st1.q) <= lim).all()
        d0c, d1c = free.counters(), clamped.counters()
        # a residual / limit comparison within rounding of its threshold may differ between the two implementations
        slack = 2 if precision == 64 else 12
        assert abs((d0c["limit_steps"] - c0["limit_steps"]) - int(hit.sum())) <= slack
        assert abs((d1c["limit_steps"] - c1["limit_steps"]) - int(hit.sum())) <= slack
        assert abs((d0c["low_flange_steps"] - c0["low_flange_steps"]) - int(((flags0 & 2) != 0).sum())) <= slack
        assert abs((d1c["low_flange_steps"] - c1["low_flange_steps"]) - int(((flags1 & 2) != 0).sum())) <= slack
        n_lim += int(hit.sum()); n_low += int(((flags0 & 2) != 0).sum())
    assert n_lim >= 100 and n_low >= 100, (n_lim, n_low)                           # the test has teeth
    assert flips <= (3 if precision == 64 else 0.01 * 3 * n), flips
    # fence_counters = 0 (the default) switches the bookkeeping off
    off = _mk(envs, 256, auto_reset=False)
    assert off.cfg.fence_counters == 0
    off.reset(); off.set_state(q=q[:256], goal=st0.goal[:256], step=st0.step[:256]); off.step(at[:256].contiguous())
    assert off.counters()["limit_steps"] == 0 and off.counters()["low_flange_steps"] == 0
    for env in (free, clamped, off):
        env.close()


def test_clamp_applies_to_rollouts_and_ik_entry(envs, O, kuka):
    """The projection inside the rollout kernel and the standalone armenv_ik entry: free-running 60 steps from states at the
    limits never leaves them, and rollout == step launches bit for bit with the clamp on."""
    n, T = 1024, 60
    rng = np.random.default_rng(5)
    cfg1 = O.default_config(); cfg1.clamp_joint_limits = 1
    lim = np.array(O.KUKA["limit"])
    q = _limit_fence_states(O, kuka, cfg1, n, rng)
    acts = torch.from_numpy(np.stack([_actions(rng, n) for _ in range(T)])).to(DEV)
    a, b = (_mk(envs, n, seed=2, clamp_joint_limits=1, fence_counters=1) for _ in range(2))
    for env in (a, b):
        env.reset(); env.set_state(q=q)
    out = a.rollout(T, acts)
    for t in range(T):
        o, r, d, s = b.step(acts[t])
        _same_rollout_step(out, t, o, r, d)
    qa = _np(a.get_state()["q"])
    assert (np.abs(qa) <= lim + 1e-12).all() and a.counters()["limit_steps"] > 0
    assert torch.equal(a.get_state()["q"], b.get_state()["q"])
    tgt = np.column_stack([rng.uniform(0.2, 0.7, n), rng.uniform(-0.3, 0.3, n), rng.uniform(0.0, 0.55, n)])
    q_gpu, it_gpu = a.ik(torch.from_numpy(q), torch.from_numpy(tgt))
    q_ref, it_ref = O.ik(kuka, cfg1, q, tgt)
    same = _np(it_gpu) == it_ref
    assert same.mean() > 0.995 and np.abs(_np(q_gpu) - q_ref)[same].max() < 1e-6 and (np.abs(_np(q_gpu)) <= lim + 1e-12).all()
    a.close(); b.close()


def test_asymmetric_joint_limits_and_other_chain(envs, O, kuka):
    """The limit handling away from its fast path: a chain whose limits do NOT straddle zero for one joint (the max |q|
    pre-test is then always true: ArmEnv's lim_min = -1) and a tight asymmetric elbow range, on the generic-FK path; the
    projection and the fence counters follow the oracle configured with the same limits."""
    import copy
    from armenv.urdf import builtin_chain
    n = 2048
    rng = np.random.default_rng(9)
    ch = copy.deepcopy(builtin_chain("kuka"))
    ch.limit_lo = list(ch.limit_lo); ch.limit_hi = list(ch.limit_hi)
    ch.limit_lo[3], ch.limit_hi[3] = -1.9, -0.4          # elbow: a range that does not contain 0
    ch.limit_lo[5], ch.limit_hi[5] = 0.2, 1.6            # wrist: asymmetric, positive only
    cfg1 = O.default_config(); cfg1.clamp_joint_limits = 1
    cfg1.lim_lo[:] = ch.limit_lo; cfg1.lim_hi[:] = ch.limit_hi
    cfg0 = O.default_config(); cfg0.lim_lo[:] = ch.limit_lo; cfg0.lim_hi[:] = ch.limit_hi
    lo, hi = np.array(ch.limit_lo), np.array(ch.limit_hi)
    q = np.tile(np.array(O.INIT_Q), (n, 1)) + rng.uniform(-0.25, 0.25, (n, 7))
    a = _actions(rng, n)
    for fk_path in (0, 1):
        e = _mk(envs, n, auto_reset=False, chain=ch, fk_path=fk_path, clamp_joint_limits=1, fence_counters=1)
        e.reset(); e.set_state(q=q)
        c0 = e.counters()
        e.step(torch.from_numpy(a).to(DEV))
        st = O.ReachState(n); st.q[:] = q; st.goal[:] = _np(e.get_state()["goal"])
        p0, _ = O.fk(kuka, q)
        tgt = np.clip(p0 + 0.02 * a.astype(np.float64), [0.2, -0.3, 0.0], [0.7, 0.3, 0.55])
        flags = O.fence_flags(kuka, cfg0, q, tgt)
        O.reach_step(kuka, cfg1, st, a)
        qg = _np(e.get_state()["q"])
        ok = np.abs(qg - st.q).max(1) < 1e-6
        assert ok.mean() > 0.999 and (qg >= lo - 1e-12).all() and (qg <= hi + 1e-12).all()
        hits = int(((flags & 1) != 0).sum())
        assert hits > 50 and abs((e.counters()["limit_steps"] - c0["limit_steps"]) - hits) <= 2
        e.close()


def test_reward_done_success_thresholds(envs, O, kuka):
    """Drive the branch of rl_reach_env.py:299-309 through the kernel: goals placed just inside / outside
    reach_dis of where the arm ends up, and step counters around max_steps (strict > and <)."""
    cfg = O.default_config()
    n = 6
    e = _mk(envs, n, auto_reset=False)
    e.reset()
    st = O.ReachState(n)
    O.reach_reset(kuka, cfg, st)
    a = np.zeros((n, 3), dtype=np.float32)
    # where does a zero action end? (IK still corrects the 90 degree tool yaw)
    probe = st.copy()
    obs_p, *_ = O.reach_step(kuka, cfg, probe, a)
    end = obs_p[0, :3].astype(np.float64)
    offs = np.array([0.0099, 0.0101, 0.0099, 0.0101, 0.3, 0.0])
    goal = np.tile(end, (n, 1)); goal[:, 0] += offs
    step = np.array([10, 10, 500, 500, 499, 500], dtype=np.int32)      # counter before the step
    st.goal[:] = goal.astype(np.float32); st.step[:] = step
    e.set_state(q=st.q, goal=st.goal, step=st.step)
    obs, rew, done, succ = e.step(torch.from_numpy(a).to(DEV))
    obs_r, rew_r, done_r, succ_r, _ = O.reach_step(kuka, cfg, st, a)
    assert np.array_equal(_np(done), done_r.astype(bool)) and np.array_equal(_np(succ), succ_r.astype(bool))
    assert list(done_r) == [1, 0, 1, 1, 0, 1] and list(succ_r) == [1, 0, 0, 0, 0, 0]
    assert _np(rew)[0] == 0.0 and abs(_np(rew)[2] + 0.099) < 2e-3 and abs(_np(rew)[5]) < 1e-3
    assert np.abs(_np(rew) - rew_r).max() < 1e-6
    e.close()


def test_trajectory_with_autoreset_f64(envs, O, kuka):
    """Free-running rollout (no teacher forcing): 560 steps x 96 envs with in-kernel auto-reset, crossing
    the 501-step time-out and any successes.  In f64 the HIP path tracks the oracle over whole episodes."""
    n, T = 96, 560
    rng = np.random.default_rng(40)
    cfg = O.default_config()
    e = _mk(envs, n, seed=11, env_id_offset=77)
    st = O.ReachState(n)
    obs0 = _np(e.reset()).copy()
    obs0_r = O.reach_reset(kuka, cfg, st, seed=11, env_id0=77)
    assert np.abs(obs0 - obs0_r).max() <= 6e-8
    # put a few goals next to the start pose so that successes happen
    g = _np(e.get_state()["goal"]).copy()
    g[:8] = obs0[:8, :3] + np.float32([0.03, 0.0, 0.0])
    st.goal[:] = g; e.set_state(goal=g)
    n_done = n_succ = 0
    for t in range(T):
        a = _actions(rng, n)
        if t < 40:
            a[:8] = np.float32([0.7, 0.0, 0.0])       # walk the first 8 envs toward their goal
        obs, rew, done, succ = e.step(torch.from_numpy(a).to(DEV), want_terminal_obs=True)
        obs, rew, done, succ, term = _np(obs).copy(), _np(rew).copy(), _np(done).copy(), _np(succ).copy(), _np(e.terminal_obs).copy()
        obs_r, rew_r, done_r, succ_r, term_r = O.reach_step_autoreset(kuka, cfg, st, a, seed=11, env_id0=77)
        assert np.array_equal(done, done_r.astype(bool)), t
        assert np.array_equal(succ, succ_r.astype(bool)), t
        assert np.abs(rew - rew_r).max() < 1e-4, t
        assert np.abs(obs - obs_r).max() < 1e-4 and np.abs(term - term_r).max() < 1e-4, t
        assert np.array_equal(obs[:, 3:], obs_r[:, 3:])              # goals (incl. re-sampled ones) bit-exact
        n_done += int(done.sum()); n_succ += int(succ.sum())
    assert n_succ >= 8 and n_done >= n                                # successes and 501-step time-outs both happened
    s = e.get_state()
    assert np.abs(_np(s["q"]) - st.q).max() < 1e-4
    assert np.array_equal(_np(s["step"]), st.step) and np.array_equal(_np(s["episode"]).astype(np.uint32), st.episode)
    ret, ln, su = e.episode_stats()
    assert np.abs(_np(ret) - st.last_return).max() < 1e-3 and np.array_equal(_np(ln), st.last_len)
    assert np.array_equal(_np(su), st.last_success)
    c = e.counters()
    assert c["episodes"] == n_done and c["successes"] == n_succ and c["env_steps"] == n * T and c["nonfinite"] == 0
    e.close()


def test_step_argument_errors(envs):
    from armenv import ArmEnvError
    e = _mk(envs, 8)
    e.reset()
    with pytest.raises(ValueError):
        e.step(torch.zeros(8, 3))                                     # CPU tensor
    with pytest.raises(ValueError):
        e.step(torch.zeros(7, 3, device=DEV))
    with pytest.raises(ArmEnvError):
        envs.BatchedReachEnv(0, device=DEV)
    with pytest.raises(ArmEnvError):
        envs.BatchedReachEnv(8, device=DEV, precision=16)
    with pytest.raises(ArmEnvError):
        envs.BatchedReachEnv(8, device="cuda:63")
    e.close()


# ------------------------------------------------------------------------------ full size: properties

def test_full_size_properties_65536(envs, O, kuka):
    """BASELINE.json config 2 size.  The oracle checks a strided sample; everything else is a
    size-independent property: determinism, shard invariance, box containment, IK residual, counters."""
    n = 65536
    rng = np.random.default_rng(50)
    acts = [torch.from_numpy(_actions(rng, n)).to(DEV) for _ in range(12)]
    cfg = O.default_config()

    def run(env, lo=0, hi=n, keep=False):
        out = []
        env.reset()
        for a in acts:
            o, r, d, s = env.step(a[lo:hi].contiguous())
            out.append((o.clone(), r.clone(), d.clone(), s.clone()))
        return out

    e = _mk(envs, n, seed=5)
    ref = run(e)
    st_full = e.get_state()
    # determinism: a second handle with the same seed reproduces every byte
    e2 = _mk(envs, n, seed=5)
    again = run(e2)
    for (o1, r1, d1, s1), (o2, r2, d2, s2) in zip(ref, again):
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2) and torch.equal(s1, s2)
    e2.close()
    # shard invariance: two half-size handles with env_id_offset reproduce the big one (multi-GPU sharding)
    h0 = _mk(envs, n // 2, seed=5, env_id_offset=0)
    h1 = _mk(envs, n // 2, seed=5, env_id_offset=n // 2)
    r0, r1 = run(h0, 0, n // 2), run(h1, n // 2, n)
    for (o, r, d, s), (oa, ra, da, sa), (ob, rb, db, sb) in zip(ref, r0, r1):
        assert torch.equal(o, torch.cat([oa, ob])) and torch.equal(r, torch.cat([ra, rb]))
        assert torch.equal(d, torch.cat([da, db]))
    h0.close(); h1.close()
    # containment and reward identity
    lo = torch.tensor([0.2, -0.3, 0.0], device=DEV) - 2e-4
    hi = torch.tensor([0.7, 0.3, 0.55], device=DEV) + 2e-4
    for o, r, d, s in ref:
        assert bool(((o[:, :3] >= lo) & (o[:, :3] <= hi)).all())
        assert bool(((o[:, 3:] >= lo + 2e-4) & (o[:, 3:] <= hi - 2e-4)).all())
        dist = (o[:, :3].double() - o[:, 3:].double()).norm(dim=1)
        live = ~d
        assert float((r[live].double() + 10.0 * dist[live]).abs().max()) < 1e-5
        assert bool((s <= d).all())
    # oracle on a strided sample of the final state
    idx = np.arange(0, n, 257)
    st = O.ReachState(len(idx))
    full = O.ReachState(n)
    O.reach_reset(kuka, cfg, full, seed=5)
    st.q[:] = full.q[idx]; st.goal[:] = full.goal[idx]; st.episode[:] = 1
    for a in acts:
        O.reach_step(kuka, cfg, st, _np(a)[idx])
    assert np.abs(_np(st_full["q"])[idx] - st.q).max() < 1e-5
    c = e.counters()
    assert c["env_steps"] == n * len(acts) and c["nonfinite"] == 0
    e.close()


# ------------------------------------------------------------------------------ N=1 compat class (boundary)

def test_rlreachenv_compat_surface(envs, O, kuka):
    """The reference's own call pattern, main.py:83-128, on the drop-in class; goals follow Python's
    `random` stream (golden G4), numbers follow the oracle."""
    import random
    g = golden_json("py_random_targets_seed0.json")
    env = envs.RLReachEnv(is_render=False, is_good_view=False)
    state_dim = env.observation_space.shape[0]
    action_dim = env.action_space.shape[0]
    action_bound = float(env.action_space.high[0]) + 0.3
    assert (state_dim, action_dim) == (6, 3) and abs(action_bound - 0.7) < 1e-7
    random.seed(0); np.random.seed(0)
    cfg = O.default_config()
    for ep in g["episodes"][:2]:
        state = env.reset()
        assert isinstance(state, np.ndarray) and state.dtype == np.float32 and state.shape == (6,)
        assert np.array_equal(state[3:], np.float32(ep["goal"]))
        st = O.ReachState(1)
        O.reach_reset_with_goal(kuka, cfg, st, np.float32([ep["goal"]]))
        for k in range(g["steps_per_episode"]):
            action = (np.zeros(3) + np.random.normal(0, action_bound * 0.98, size=action_dim)).clip(-action_bound, action_bound)
            state, reward, done, is_success = env.step(action)
            obs_r, rew_r, done_r, succ_r, _ = O.reach_step(kuka, cfg, st, action.astype(np.float32)[None])
            assert isinstance(reward, float) and isinstance(done, bool) and isinstance(is_success, bool)
            assert state.dtype == np.float32 and np.abs(state - obs_r[0]).max() < 1e-6
            assert abs(reward - rew_r[0]) < 1e-5 and done == bool(done_r[0]) and is_success == bool(succ_r[0])
            assert np.array_equal(state[-3:], np.float32(ep["goal"]))
    assert env.seed(3) == [3]
    env.close()


# ------------------------------------------------------------------------------ rollout engine (C1)

@pytest.mark.parametrize("precision", [64, 32])
def test_rollout_external_actions_equals_step_calls(envs, precision):
    """armenv_rollout with external actions against T armenv_step launches: the same trajectory bit for bit (state and
    counters included), whatever the launch grouping."""
    n, T = 4096 + 64 + 3, 37
    rng = np.random.default_rng(60)
    acts = torch.from_numpy(np.stack([_actions(rng, n) for _ in range(T)])).to(DEV)
    a = _mk(envs, n, seed=9, precision=precision, max_steps=20)       # short episodes: resets inside the rollout
    b = _mk(envs, n, seed=9, precision=precision, max_steps=20)
    a.reset(); b.reset()
    out = a.rollout(T, acts, want_actions=True, want_terminal_obs=True)
    for t in range(T):
        o, r, d, s = b.step(acts[t], want_terminal_obs=True)
        _same_rollout_step(out, t, o, r, d)
        assert torch.equal(out["success"][t], s), t
        assert torch.equal(out["terminal_obs"][t], b.terminal_obs), t
    assert torch.equal(out["actions"], acts)
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    ca, cb = a.counters(), b.counters()
    assert all(ca[k] == cb[k] for k in ("episodes", "successes", "env_steps", "nonfinite", "ik_updates"))
    # the SAME launch sequence is deterministic, and a one-step rollout is the step kernel's arithmetic bit for bit
    c = _mk(envs, n, seed=9, precision=precision, max_steps=20); c.reset()
    out_c = c.rollout(T, acts, want_terminal_obs=True)
    assert all(torch.equal(out[k], out_c[k]) for k in ("obs", "reward", "done", "success", "terminal_obs"))
    d1 = _mk(envs, n, seed=9, precision=precision, max_steps=20); d1.reset()
    d2 = _mk(envs, n, seed=9, precision=precision, max_steps=20); d2.reset()
    for t in range(5):
        o1 = d1.rollout(1, acts[t:t + 1])
        o, r, d, s = d2.step(acts[t])
        assert torch.equal(o1["obs"][0], o) and torch.equal(o1["reward"][0], r) and torch.equal(o1["done"][0], d), t
    c.close(); d1.close(); d2.close()
    assert a.counters()["episodes"] >= n
    a.close(); b.close()


@pytest.mark.parametrize("task", ["reach", "push", "pick"])
@pytest.mark.parametrize("precision", [64, 32])
def test_rollout_500_equals_500_step_launches_bitwise(envs, precision, task):
    """The benchmarked launch shape against the gym-style path over a whole reference episode and the reset after it
    (501 steps, rl_reach_env.py:299): 6 x rollout(100) == one rollout(600) == 600 armenv_step launches, bit for bit --
    outputs, final state, counters -- for the three tasks (pick through its lane-asynchronous default schedule)."""
    n, T = (2048, 600) if task == "reach" else (1024, 600)
    Env = dict(reach=envs.BatchedReachEnv, push=envs.BatchedPushEnv, pick=envs.BatchedPickEnv)[task]
    gen = torch.Generator(device=DEV); gen.manual_seed(77)
    if task == "reach":
        acts = (torch.randn((T, n, 3), device=DEV, generator=gen) * 0.686).clamp_(-0.7, 0.7).contiguous()
    else:
        acts = (torch.randn((T, n, 3), device=DEV, generator=gen) * 0.392).contiguous()       # main.py:484: unclipped
    a, b, c = (Env(n, device=DEV, seed=31, precision=precision) for _ in range(3))
    for e in (a, b, c):
        e.reset()
    whole = {k: v.clone() for k, v in a.rollout(T, acts).items()}
    parts = [{k: v.clone() for k, v in b.rollout(100, acts[k0:k0 + 100].contiguous()).items()} for k0 in range(0, T, 100)]
    for k in ("obs", "reward", "done", "success"):
        assert torch.equal(torch.cat([p[k] for p in parts]), whole[k]), k
    for t in range(T):
        o, r, d, s = c.step(acts[t])
        assert torch.equal(whole["obs"][t], o) and torch.equal(whole["reward"][t], r) and torch.equal(whole["done"][t], d), t
    sa, sb, sc = a.get_state(), b.get_state(), c.get_state()
    for k in sa:
        assert torch.equal(sa[k], sb[k]) and torch.equal(sa[k], sc[k]), k
    assert a.counters() == b.counters() == c.counters()
    assert a.counters()["episodes"] >= n          # every env finished its 501-step episode (or reached its goal) and was reset
    for e in (a, b, c):
        e.close()


def _soak_vs_oracle(envs, O, kuka, n, precision, launches=6, R=100, seed=5):
    """BASELINE config 2's launch shape -- armenv_rollout(100) launches, 501-step episodes, auto-reset, i.i.d. clipped
    Gaussian actions -- free-running against the CPU oracle's reach_step_autoreset on the same actions (the loop of
    /root/reference/main.py:108-128 with the env of rl_reach_env.py:132-319).  Returns per-step worst |obs difference|,
    the fraction of envs within 1e-4 per step, and the episode / success totals of both sides."""
    cfg = O.default_config()
    e = _mk(envs, n, seed=seed, precision=precision)
    st = O.ReachState(n)
    O.reach_reset(kuka, cfg, st, seed=seed)
    e.reset()
    gen = torch.Generator(device=DEV); gen.manual_seed(9)
    worst, within, rworst = [], [], []
    tot = dict(ep_g=0, ep_o=0, su_g=0, su_o=0, flag_diff=0)
    bufs = {}
    for b in range(launches):
        acts = (torch.randn((R, n, 3), device=DEV, generator=gen) * 0.686).clamp_(-0.7, 0.7).contiguous()
        out = e.rollout(R, acts, out=bufs)
        a_np = _np(acts)
        obs_g, rew_g, done_g, succ_g = _np(out["obs"]), _np(out["reward"]), _np(out["done"]), _np(out["success"])
        for t in range(R):
            obs_o, rew_o, done_o, succ_o, _ = O.reach_step_autoreset(kuka, cfg, st, a_np[t], seed=seed, want_terminal=False)
            d = np.abs(obs_g[t] - obs_o).max(1)
            worst.append(d.max()); within.append((d < 1e-4).mean())
            same = done_g[t] == done_o.astype(bool)
            tot["flag_diff"] += int((~same).sum())
            rworst.append(np.abs(rew_g[t].astype(np.float64) - rew_o)[same].max())
            tot["ep_o"] += int(done_o.sum()); tot["su_o"] += int((done_o.astype(bool) & succ_o.astype(bool)).sum())
        tot["ep_g"] += int(done_g.sum()); tot["su_g"] += int((done_g & succ_g).sum())
    q = _np(e.get_state()["q"])
    cnt = e.counters()
    e.close()
    return np.array(worst), np.array(within), np.array(rworst), tot, np.abs(q - st.q).max(1), cnt


@pytest.mark.parametrize("n", [8192, 65536])
def test_benchmarked_launch_shape_free_running_vs_oracle_f64(envs, O, kuka, n):
    """The headline kernel at the headline shape under the oracle: 6 launches of rollout(100) --
    a whole 501-step episode, its time-limit reset and 99 steps of the next one, plus every goal reached on the way --
    at 8 192 envs and at BASELINE config 2's 65 536.  f64 engine: every env, every step within 1e-4 (north_star's
    tolerance) on the observation and 1e-3 on the reward (= 10 x distance), identical done / success flags, equal episode and
    success totals."""
    worst, within, rworst, tot, dq, cnt = _soak_vs_oracle(envs, O, kuka, n, 64)
    assert tot["flag_diff"] == 0, tot
    assert tot["ep_g"] == tot["ep_o"] >= n and tot["su_g"] == tot["su_o"], tot
    assert cnt["episodes"] == tot["ep_g"] and cnt["successes"] == tot["su_g"] and cnt["nonfinite"] == 0
    assert worst.max() < 1e-4, (worst.max(), int(worst.argmax()))
    assert within.min() == 1.0
    assert rworst.max() < 1e-3, rworst.max()
    assert (dq < 1e-4).mean() > 0.999, (dq < 1e-4).mean()      # joint angles after 600 free-running steps


def test_benchmarked_launch_shape_free_running_vs_oracle_f32(envs, O, kuka):
    """Same shape, f32 engine against the f64 oracle.  The f32 engine's stated tolerance is per step (1e-4 teacher-forced,
    test_step_teacher_forced_f32); free-running over 600 steps it measures 6e-6 at worst with identical flags and totals
    (an IK update count that flips at the 1e-4 residual gate could move an env by up to that residual, hence the margins):
    >= 99.9 % of the envs within 1e-4 at every step, nobody beyond 5e-4, episode totals within 0.1 %."""
    n = 8192
    worst, within, rworst, tot, dq, cnt = _soak_vs_oracle(envs, O, kuka, n, 32)
    assert within.min() >= 0.999, within.min()
    assert worst.max() < 5e-4, worst.max()
    assert abs(tot["ep_g"] - tot["ep_o"]) <= 1e-3 * tot["ep_o"] and abs(tot["su_g"] - tot["su_o"]) <= max(3, 0.05 * tot["su_o"]), tot
    assert cnt["nonfinite"] == 0


@pytest.mark.parametrize("precision", [64, 32])
@pytest.mark.parametrize("task", ["reach", "push", "pick"])
def test_lane_asynchronous_rollout_equals_lockstep(envs, task, precision):
    """armenv_rollout's schedules: lockstep (ArmEnvConfig.rollout_ready_lanes 0) and lane-asynchronous -- by count, with several
    waiting thresholds: 1 (a tail block on nearly every trip), 7, 33, 64 (every lane waits for the whole wave), and by the
    straggler rule (rollout_straggler_trips 1, 3, 6) -- give the SAME bits: every per-step output, the final state, the
    counters.  Ragged batch (not a multiple of 64), 20-step episodes (in-place resets while other lanes of the wave are
    mid-IK), external actions and the in-kernel random policy.  What differs is what the schedule cost the waves
    (counters wave_trips / wave_rounds): lockstep pays one tail per step."""
    n, T = 2048 + 64 + 5, 45
    Env = dict(reach=envs.BatchedReachEnv, push=envs.BatchedPushEnv, pick=envs.BatchedPickEnv)[task]
    rng = np.random.default_rng(77)
    sig = 0.686 if task == "reach" else 0.392
    acts = torch.from_numpy((rng.standard_normal((T, n, 3)) * sig).clip(-0.7, 0.7).astype(np.float32)).to(DEV)
    schedule = ("wave_trips", "wave_rounds")
    waves = (n + 31) // 32 if task != "reach" else (n + 63) // 64      # push / pick: half-filled waves at this batch size
    for policy in ("external", "random"):
        ref = None
        for k, K in ((0, 0), (1, 0), (7, 0), (33, 0), (64, 0), (62, 1), (62, 3), (62, 6)):
            e = Env(n, device=DEV, seed=13, precision=precision, max_steps=20, rollout_ready_lanes=k, rollout_straggler_trips=K,
                    fence_counters=1)
            if policy == "random":
                e.set_policy("random", noise_sigma=sig, noise_clip=0.7)
            e.reset()
            out = e.rollout(T, acts if policy == "external" else None, want_actions=True, want_terminal_obs=True)
            out2 = e.rollout(7, acts[:7].contiguous() if policy == "external" else None)       # a second launch continues the same envs
            got = {kk: out[kk].clone() for kk in ("obs", "reward", "done", "success", "actions", "terminal_obs")}
            got.update({"obs2": out2["obs"].clone(), "done2": out2["done"].clone()})
            got.update({"st_" + kk: v.clone() for kk, v in e.get_state().items()})
            cnt = e.counters()
            sched = {kk: cnt.pop(kk) for kk in schedule}
            e.close()
            if ref is None:
                ref, ref_cnt, ref_sched = got, cnt, sched
                assert cnt["episodes"] >= 2 * n and cnt["env_steps"] == n * (T + 7)
                assert sched["wave_rounds"] == waves * (T + 7), sched
                # a wave pays at least its mean lane's trips (updates + one exit trip per env-step)
                assert sched["wave_trips"] * (n / waves) >= cnt["ik_updates"] + cnt["env_steps"], (sched, cnt)
            else:
                for kk in ref:
                    assert torch.equal(ref[kk], got[kk]), (task, precision, policy, k, K, kk)
                assert cnt == ref_cnt, (task, precision, policy, k, K)
                # (a badly chosen rule can cost more trips than lockstep -- reach with K = 3 calls every lane on its exit trip a
                # straggler and the wave falls out of phase; the bits do not care)
                assert sched["wave_rounds"] >= ref_sched["wave_rounds"] and sched["wave_trips"] > 0, (task, precision, policy, k, K, sched, ref_sched)
                if K == 0 and k in (1, 64):      # fully asynchronous / lockstep by another name: never more trips than lockstep
                    assert sched["wave_trips"] <= ref_sched["wave_trips"], (task, precision, policy, k, sched, ref_sched)
    from armenv import ArmEnvError
    with pytest.raises(ArmEnvError):
        Env(64, device=DEV, rollout_straggler_trips=65)


@pytest.mark.parametrize("precision", [64, 32])
@pytest.mark.parametrize("task", ["reach", "push", "pick"])
def test_two_waves_per_simd_rollout_equals_one(envs, task, precision):
    """ArmEnvConfig.rollout_waves_per_simd: the rollout kernel built for two waves per SIMD (<= 256 registers per lane,
    ordinary action loads, the overflow in scratch) against the one-wave form (whole register file, AGPR action prefetch):
    the same bits for outputs, state and counters, with external actions and the in-kernel random policy; 0 picks by batch
    size (one wave per SIMD up to 64 x #SIMDs envs)."""
    n, T = 4096 + 64 + 3, 50
    Env = dict(reach=envs.BatchedReachEnv, push=envs.BatchedPushEnv, pick=envs.BatchedPickEnv)[task]   # pick: lane-asynchronous
    rng = np.random.default_rng(3)
    sig = 0.686 if task == "reach" else 0.392
    acts = torch.from_numpy((rng.standard_normal((T, n, 3)) * sig).clip(-0.7, 0.7).astype(np.float32)).to(DEV)
    for policy in ("external", "random"):
        ref = None
        for w in (1, 2, 0):
            e = Env(n, device=DEV, seed=21, precision=precision, max_steps=20, rollout_waves_per_simd=w)
            if policy == "random":
                e.set_policy("random", noise_sigma=sig, noise_clip=0.7)
            e.reset()
            out = e.rollout(T, acts if policy == "external" else None, want_actions=True, want_terminal_obs=True)
            got = {kk: out[kk].clone() for kk in ("obs", "reward", "done", "success", "actions", "terminal_obs")}
            for j in range(4):                                   # the step kernel has the same two budgets
                o, r, d, s_ = e.step(acts[j] if policy == "external" else None)
                got.update({f"step{j}_obs": o.clone(), f"step{j}_rew": r.clone(), f"step{j}_done": d.clone()})
            got.update({"st_" + kk: v.clone() for kk, v in e.get_state().items()})
            cnt = e.counters()
            e.close()
            if ref is None:
                ref, ref_cnt = got, cnt
            else:
                for kk in ref:
                    assert torch.equal(ref[kk], got[kk]), (task, precision, policy, w, kk)
                assert cnt == ref_cnt
    from armenv import ArmEnvError
    with pytest.raises(ArmEnvError):
        Env(64, device=DEV, rollout_waves_per_simd=3)


def test_rollout_random_policy_matches_oracle(envs, O, kuka):
    """Fused random policy (zero actor + clipped Gaussian noise, main.py:116-117) inside the rollout kernel."""
    n, T = 512, 60
    cfg = O.default_config(); cfg.max_steps = 25
    e = _mk(envs, n, seed=21, env_id_offset=5, max_steps=25)
    e.set_policy("random", action_bound=0.7, noise_sigma=0.7 * 0.98, noise_clip=0.7)
    st = O.ReachState(n)
    e.reset(); O.reach_reset(kuka, cfg, st, seed=21, env_id0=5)
    out = e.rollout(T, None, want_actions=True, want_terminal_obs=True)
    ref = O.reach_rollout(kuka, cfg, st, T, None, seed=21, env_id0=5)
    acts = _np(out["actions"])
    assert np.abs(acts - ref["actions"]).max() < 2e-6
    # std of N(0, 0.686) clipped at +-0.7 is 0.499
    assert 0.47 < acts.std() < 0.53 and np.abs(acts).max() <= 0.7 and abs(acts.mean()) < 0.01
    assert np.array_equal(_np(out["done"]), ref["done"].astype(bool))
    assert np.array_equal(_np(out["success"]), ref["success"].astype(bool))
    assert np.abs(_np(out["obs"]) - ref["obs"]).max() < 1e-5
    assert np.abs(_np(out["terminal_obs"]) - ref["terminal_obs"]).max() < 1e-5
    assert np.abs(_np(out["reward"]) - ref["reward"]).max() < 1e-4
    assert ref["done"].sum() >= 2 * n
    # splitting the same rollout into two launches gives the same trajectory (noise keyed by episode/step)
    e2 = _mk(envs, n, seed=21, env_id_offset=5, max_steps=25)
    e2.set_policy("random")
    e2.reset()
    o1 = {k: v.clone() for k, v in e2.rollout(23, None, want_actions=True).items()}
    o2 = e2.rollout(T - 23, None, want_actions=True)
    assert torch.equal(torch.cat([o1["obs"], o2["obs"]]), out["obs"])      # launch grouping does not change a bit
    assert torch.equal(torch.cat([o1["actions"], o2["actions"]]), out["actions"])
    e.close(); e2.close()


def test_step_with_fused_policy_equals_one_step_rollouts(envs):
    a = _mk(envs, 256, seed=4); b = _mk(envs, 256, seed=4)
    for e in (a, b):
        e.set_policy("random"); e.reset()
    out = a.rollout(9, None)
    for t in range(9):
        o, r, d, s = b.step(None)
        _same_rollout_step(out, t, o, r, d)
    a.close(); b.close()


def test_rollout_argument_errors(envs):
    from armenv import ArmEnvError
    e = _mk(envs, 64)
    e.reset()
    with pytest.raises(ArmEnvError, match="no fused policy"):
        e.rollout(4, None)
    with pytest.raises(ValueError):
        e.rollout(4, torch.zeros(3, 64, 3, device=DEV))
    out = e.rollout(0, torch.zeros(0, 64, 3, device=DEV))
    assert out["obs"].shape == (0, 64, 6)
    e.close()


# ------------------------------------------------------------------------------ push task (P1-P4, config 4)

def _push_actions(rng, st_aux, eef, n, chase=None):
    """random actions (main.py:484 noise); envs in `chase` steer toward their cube at table height so that
    contacts happen"""
    a = (rng.normal(0.0, 0.4 * 0.98, (n, 3))).astype(np.float32)
    if chase is not None:
        want = st_aux[:, 0:3].copy(); want[:, 2] = 0.015
        c = np.clip((want - eef) / 0.08, -1, 1).astype(np.float32)
        a[chase] = c[chase]
    return a


def test_push_reset_matches_oracle(envs, O, kuka):
    n = 2048 + 7
    cfg = O.default_config("push")
    e = envs.BatchedPushEnv(n, device=DEV, seed=17, env_id_offset=99)
    assert e.kernel_name == "push_step<f64,kuka>"
    obs = _np(e.reset())
    st = O.PushState(n)
    obs_ref = O.push_reset(kuka, cfg, st, seed=17, env_id0=99)
    assert obs.shape == (n, 9) and np.array_equal(obs[:, 3:], obs_ref[:, 3:]) and np.abs(obs - obs_ref).max() <= 6e-8
    s = e.get_state()
    assert np.array_equal(_np(s["aux"])[:, :6], st.aux[:, :6]) and np.abs(_np(s["aux"])[:, 6] - st.aux[:, 6]).max() < 1e-15
    assert _np(s["aux"]).shape == (n, 10) and not _np(s["aux"])[:, 7:].any()          # at rest in the plane, one step into the fall
    assert np.abs(st.aux[:, 2] - (0.01 - 10.0 / 240.0 ** 2)).max() < 1e-15
    assert np.array_equal(_np(s["q"]), st.q)
    e.close()


@pytest.mark.parametrize("precision", [64, 32])
def test_push_step_teacher_forced(envs, O, kuka, precision):
    n = 1024 + 5
    rng = np.random.default_rng(70)
    cfg = O.default_config("push")
    e = envs.BatchedPushEnv(n, device=DEV, seed=2, auto_reset=False, precision=precision)
    st = O.PushState(n)
    obs_r = O.push_reset(kuka, cfg, st, seed=2)
    e.reset()
    chase = np.arange(n) < n // 2
    tight = 1e-6 if precision == 64 else 1e-4
    moved_total = 0
    frac_tight = []
    for t in range(60):
        a = _push_actions(rng, st.aux, obs_r[:, :3].astype(np.float64), n, chase)
        e.set_state(q=st.q, aux=st.aux, step=st.step, ep_return=st.ep_return)
        c0 = st.aux[:, :3].copy()
        obs, rew, done, succ = e.step(torch.from_numpy(a).to(DEV))
        obs, rew, done, succ = _np(obs).copy(), _np(rew).copy(), _np(done).copy(), _np(succ).copy()
        obs_r, rew_r, done_r, succ_r, iters = O.push_step(kuka, cfg, st, a)
        s = e.get_state()
        dq = np.abs(_np(s["q"]) - st.q).max(1)
        # Cubes near the edge of the arm's reach (x ~ 0.7, z ~ 0) make the chasing envs run the IK into its 20-iteration
        # cap at a stretched, near-singular pose, where the damped solve amplifies rounding (primal GE in the oracle vs
        # dual LDL^T here): those stay within the stated 1e-4 only.  Everything else agrees tightly.
        capped = iters >= 20
        ok = dq < tight
        frac_tight.append(ok[~capped].mean())
        assert (dq[~capped] < 1e-4).mean() > 0.995, (t, (dq[~capped] < 1e-4).mean())
        moved_total += int((np.abs(st.aux[:, :3] - c0).max(1) > 1e-9).sum())
        tol_c = 1e-6 if precision == 64 else 2e-4
        # contact and reward are discontinuous in the eef position: compare where the arm agrees
        dc = np.abs(_np(s["aux"])[:, :9] - st.aux[:, :9]).max(1)[ok]      # cube, target, d_last, cube velocity
        assert (dc < tol_c).mean() > 0.999, t
        rdiff = np.abs(rew.astype(np.float64) - rew_r)[ok]
        assert (rdiff > (1e-4 if precision == 64 else 5e-2)).mean() < (1e-3 if precision == 64 else 2e-2), t
        if precision == 64:
            assert (done[ok] == done_r[ok].astype(bool)).mean() > 0.999
            assert np.quantile(np.abs(obs - obs_r).max(1)[ok], 0.999) < 1e-6
    assert np.mean(frac_tight) > (0.99 if precision == 64 else 0.97), np.mean(frac_tight)
    assert moved_total > 50                               # the cube really was pushed around
    e.close()


def test_push_trajectory_autoreset_and_rollout(envs, O, kuka):
    """Free-running push episodes with auto-reset (time-outs at 21 steps here) against the oracle; the first 16 envs
    get a scripted slow push of a cube placed in the well-conditioned middle of the workspace (the tool comes down 6 cm short of it
    and advances 5 mm per step: shaped rewards and the +100 success branch; contact switches at Bullet's own ERP 0.2 so that a
    20-step episode is long enough -- the fitted defaults recover a penetration at 2 % per step); and the rollout kernel against
    step launches bit for bit."""
    n, T = 256, 70
    rng = np.random.default_rng(71)
    cfg = O.default_config("push"); cfg.max_steps = 20; cfg.push_contact_erp = 0.2; cfg.push_friction = 0.5
    mk = lambda: envs.BatchedPushEnv(n, device=DEV, seed=8, max_steps=20, push_contact_erp=0.2, push_friction=0.5)
    e, r_env, e2 = mk(), mk(), mk()
    st = O.PushState(n)
    O.push_reset(kuka, cfg, st, seed=8)
    g = st.aux[:, :6].astype(np.float32)
    g[:16] = np.float32([0.5, 0.0, 0.01, 0.5, 0.07, 0.01])            # cube 7 cm from its target, pushed along +y
    obs_r = O.push_reset_with_goal(kuka, cfg, st, g)            # episode: 1 (Philox reset) -> 2 (reset with goals), as on the device
    for x in (e, r_env, e2):
        x.reset(); x.reset(goal=torch.from_numpy(g))
    acts = []
    n_done = n_succ = 0
    for t in range(T):
        a = _push_actions(rng, st.aux, obs_r[:, :3].astype(np.float64), n)
        k = st.step[:16]
        a[:16] = np.where(k[:, None] < 8, np.clip((np.float32([0.5, -0.06, 0.015]) - obs_r[:16, :3]) / 0.08, -0.5, 0.5), np.float32([0.0, 0.0625, 0.0]))
        acts.append(a)
        obs, rew, done, succ = e.step(torch.from_numpy(a).to(DEV), want_terminal_obs=True)
        obs_r, rew_r, done_r, succ_r, term_r = O.push_step_autoreset(kuka, cfg, st, a, seed=8)
        assert np.array_equal(_np(done), done_r.astype(bool)), t
        assert np.abs(_np(obs) - obs_r).max() < 1e-5 and np.abs(_np(e.terminal_obs) - term_r).max() < 1e-5, t
        assert np.abs(_np(rew) - rew_r).max() < 1e-3, t
        n_done += int(done_r.sum()); n_succ += int((rew_r == 100).sum())
    assert n_done >= 3 * n and n_succ >= 16
    s = e.get_state()
    assert np.abs(_np(s["aux"])[:, :9] - st.aux[:, :9]).max() < 1e-6 and np.array_equal(_np(s["step"]), st.step)
    out = r_env.rollout(T, torch.from_numpy(np.stack(acts)).to(DEV))
    for t in range(T):
        o, r, d, su = e2.step(torch.from_numpy(acts[t]).to(DEV))
        _same_rollout_step(out, t, o, r, d)
    for x in (e, r_env, e2):
        x.close()


def test_push_full_size_properties_32768(envs):
    """BASELINE.json config 4 size: placement distance, workspace clip (z <= 0.1), idle reward -1, counters."""
    n = 32768
    e = envs.BatchedPushEnv(n, device=DEV, seed=1)
    obs = e.reset()
    d = (obs[:, 3:5] - obs[:, 6:8]).double().norm(dim=1)           # the placement test sees both bodies at their spawn height (:199,206,213)
    assert float(d.min()) >= 0.22 - 1e-6 and float(d.max()) <= 0.25 + 1e-6
    rest = float(e.cfg.push_rest_z)                                # the cube one step into its fall (:241), the fixed target where it was spawned
    assert bool((obs[:, 5] == np.float32(0.01 - 10.0 / 240.0 ** 2)).all()) and bool((obs[:, 8] == np.float32(0.01)).all()) and abs(rest + 0.00474) < 1e-12
    e.set_policy("random", action_bound=0.4, noise_sigma=0.4 * 0.98, noise_clip=1e9)      # main.py:457,484
    out = e.rollout(40, None)
    eef = out["obs"][..., :3][~out["done"]]            # (a finished env's row shows its next episode's first observation)
    assert float(eef[..., 2].max()) <= 0.1 + 2e-4 and float(eef[..., 2].min()) >= -2e-4
    assert float(out["done"].float().mean()) < 1e-4      # a lucky sweep can deliver the cube within 40 steps; it is rare
    # past the fall (13 stepSimulation calls, then the overshoot decays below the reward's 1e-5 threshold at once) an untouched cube costs -1 a step
    idle = (out["obs"][15:, :, 3:5] == out["obs"][14:-1, :, 3:5]).all(-1) & ~out["done"][15:] & ~out["done"][14:-1]
    idle &= ~out["done"][:15].any(0)[None]
    assert bool((out["reward"][15:][idle] == -1.0).all()) and float(idle.float().mean()) > 0.75      # (0.85 measured: under the fitted contact model a touched cube creeps for tens of steps)
    z = out["obs"][:, :, 5].double()
    nd = ~out["done"][:13].any(0)
    k = torch.arange(2, 14, device=z.device, dtype=torch.float64)[:, None]           # env step j = stepSimulation call j + 1
    assert float((z[:12, nd] - (0.01 - 0.5 * 10.0 / 240.0 ** 2 * k * (k + 1))).abs().max()) < 1e-7      # free fall, f32 observation
    c = e.counters()
    assert c["env_steps"] == n * 40 and c["nonfinite"] == 0
    assert c["episodes"] == c["successes"] == int(out["done"].sum())      # nothing times out in 40 steps: every finish is a delivery
    e.close()


def test_rlpushenv_compat_surface(envs):
    """train_push_with_TD3's call pattern (main.py:453-487) on the drop-in class."""
    import random
    random.seed(0); np.random.seed(0)
    env = envs.RLPushEnv(is_render=False, is_good_view=False)
    action_bound = float(env.action_space.high[0])
    assert abs(action_bound - 0.4) < 1e-7                                # main.py:457
    state = env.reset()
    assert state.shape == (9,) and state.dtype == np.float64
    d = np.linalg.norm(state[3:5] - state[6:8])                          # planar: both bodies are spawned at z = 0.01 (:199,206)
    assert 0.22 <= d <= 0.25 and abs(state[5] - (0.01 - 10.0 / 240.0 ** 2)) < 1e-15 and state[8] == 0.01   # cube one step into its fall, target fixed
    for _ in range(5):
        action = np.zeros(3) + np.random.normal(0, action_bound * 0.98, size=3)
        state, reward, done, info = env.step(action)
        assert state.shape == (9,) and isinstance(done, bool) and set(info) == {"is_success"}
        assert info["is_success"].dtype == np.float32 and (reward == -1.0 or -0.01 < reward < 0) and not done
    env.close()


# ------------------------------------------------------------------------------ pick task (section 8f row 4)

def _pick_actions(obs, aux, dv=0.08, L=0.257, rng=None, scripted=None, sigma=0.4 * 0.98):
    """scripted grasp-and-lift controller on the gripper tip (eef - (0,0,L)): go above the cube, descend until the
    gripper closes, then carry the held cube to the target; envs outside `scripted` act randomly (main.py:484 noise)"""
    n = len(obs)
    tip = obs[:, 0:3].astype(np.float64).copy(); tip[:, 2] -= L
    cube, tgt, grip, off = aux[:, 0:3], aux[:, 3:6], aux[:, 7], aux[:, 8:11]
    horiz = np.linalg.norm(tip[:, :2] - cube[:, :2], axis=1)
    want = np.where((horiz > 0.004)[:, None], cube + np.array([0, 0, 0.10]), cube + np.array([0, 0, 0.05]))
    want = np.where((grip == 1)[:, None], tip, want)
    want = np.where((grip == 2)[:, None], tgt - off, want)
    a = (want - tip) / dv
    a = a / np.maximum(np.abs(a).max(axis=1, keepdims=True), 1.0)
    if rng is not None:
        r = rng.normal(0.0, sigma, (n, 3))
        a = np.where(scripted[:, None], a, r) if scripted is not None else r
    return a.astype(np.float32)


def test_pick_reset_matches_oracle(envs, O, kuka):
    n = 2048 + 7
    cfg = O.default_config("pick")
    e = envs.BatchedPickEnv(n, device=DEV, seed=23, env_id_offset=5)
    assert e.kernel_name == "pick_step<f64,kuka>"
    obs = _np(e.reset())
    st = O.PickState(n)
    obs_ref = O.pick_reset(kuka, cfg, st, seed=23, env_id0=5)
    assert obs.shape == (n, 9) and np.array_equal(obs[:, 3:], obs_ref[:, 3:]) and np.abs(obs - obs_ref).max() <= 6e-8
    s = e.get_state()
    aux = _np(s["aux"])
    assert aux.shape == (n, 12) and np.array_equal(aux[:, :6], st.aux[:, :6]) and np.abs(aux[:, 6] - st.aux[:, 6]).max() < 1e-15
    assert not aux[:, 7:].any()                                          # gripper open, nothing held
    spawn = aux[:, 0:3].copy(); spawn[:, 2] = 0.01                       # :194: the test sees the cube at its spawn height
    d = np.linalg.norm(spawn - aux[:, 3:6], axis=1)                      # rl_pick_env.py:205-208: 3-D distance
    assert d.min() >= 0.22 and d.max() <= 0.25 and np.abs(aux[:, 2] - (0.01 - 10.0 / 240.0 ** 2)).max() < 1e-15      # one step into its fall
    assert aux[:, 5].min() >= 0.0 and aux[:, 5].max() <= 0.26 and aux[:, 5].std() > 0.03   # target floats above the table
    assert np.array_equal(_np(s["q"]), st.q)
    e.close()


@pytest.mark.parametrize("precision", [64, 32])
def test_pick_step_teacher_forced(envs, O, kuka, precision):
    """Every step starts from the oracle's state: arm (f32-rounded start, joints 0..5 only), gripper trigger / hold
    decision, carried cube, reward.  Half of the envs follow the scripted grasp-and-lift, half act randomly."""
    n = 1024 + 5
    rng = np.random.default_rng(90)
    cfg = O.default_config("pick")
    e = envs.BatchedPickEnv(n, device=DEV, seed=4, auto_reset=False, precision=precision)
    st = O.PickState(n)
    obs_r = O.pick_reset(kuka, cfg, st, seed=4)
    e.reset()
    scripted = np.arange(n) < n // 2
    tight = 1e-6 if precision == 64 else 1e-4
    held_seen = closed_seen = succ_seen = 0
    for t in range(40):
        a = _pick_actions(obs_r, st.aux, rng=rng, scripted=scripted)
        e.set_state(q=st.q, aux=st.aux, step=st.step, ep_return=st.ep_return)
        q7 = st.q[:, 6].copy()
        obs, rew, done, succ = e.step(torch.from_numpy(a).to(DEV))
        obs, rew, done, succ = _np(obs).copy(), _np(rew).copy(), _np(done).copy(), _np(succ).copy()
        obs_r, rew_r, done_r, succ_r, iters = O.pick_step(kuka, cfg, st, a)
        s = e.get_state()
        q, aux = _np(s["q"]), _np(s["aux"])
        assert np.array_equal(q[:, 6], q7.astype(np.float32 if precision == 32 else np.float64))   # joint 7 never moves (:343)
        dq = np.abs(q - st.q).max(1)
        capped = iters >= 20
        ok = (dq < tight) & ~capped
        assert ok.mean() > (0.99 if precision == 64 else 0.95), (t, ok.mean())
        assert (dq[~capped] < 1e-4).mean() > 0.995
        # grip decisions are discontinuous in the tip position: compare where the arm agrees
        same_grip = aux[ok, 7] == st.aux[ok, 7]
        assert same_grip.mean() > 0.999, t
        okg = ok.copy(); okg[ok] = same_grip
        tol_c = 1e-6 if precision == 64 else 2e-4
        assert (np.abs(aux[okg][:, :7] - st.aux[okg][:, :7]).max(1) < tol_c).mean() > 0.999, t
        assert (np.abs(aux[okg][:, 8:11] - st.aux[okg][:, 8:11]).max(1) < tol_c).mean() > 0.999, t
        rdiff = np.abs(rew.astype(np.float64) - rew_r)[okg]
        assert (rdiff > (1e-4 if precision == 64 else 5e-2)).mean() < (1e-3 if precision == 64 else 2e-2), t
        if precision == 64:
            assert (done[okg] == done_r[okg].astype(bool)).mean() > 0.999
            assert np.quantile(np.abs(obs - obs_r).max(1)[okg], 0.999) < 1e-6
        held_seen = max(held_seen, int((st.aux[:, 7] == 2).sum()))
        closed_seen = max(closed_seen, int((st.aux[:, 7] == 1).sum()))
        succ_seen = max(succ_seen, int(succ_r.sum()))
    assert held_seen > 0.9 * (n // 2) and succ_seen > 0.8 * (n // 2)     # the scripted envs grasp, lift and deliver
    e.close()


@pytest.mark.parametrize("sigma", [0.4 * 0.98, 0.1])
def test_pick_trajectory_autoreset_and_rollout(envs, O, kuka, sigma):
    """Free-running pick episodes with auto-reset (time-outs at 25 steps; scripted envs finish with +100) against
    the oracle, and the rollout kernel against step launches bit for bit.  The random envs act with the reference's own
    exploration noise (main.py:552: N(0, 0.392), unclipped); some of them wander to the top of the 0.807 m box within 15
    steps, where the IK runs into its 20-iteration cap -- the cap term of the parity fence (tests/test_gpu_fence.py): an
    env is left out of the comparison from a capped call to its next reset, every other env-step agrees; an env that IS left
    out still stays within a loose bound of its twin while both are in the same episode (a divergence that starts at a
    near-singular pose for any other reason than the fence's -- a pivot-ordering bug, say -- would break that bound).
    sigma = 0.1: the same run with gentle exploration, where no call is capped and NO env is masked."""
    from test_gpu_fence import FenceBook
    n, T = 256, 80
    rng = np.random.default_rng(91)
    cfg = O.default_config("pick"); cfg.max_steps = 24
    mk = lambda **kw: envs.BatchedPickEnv(n, device=DEV, seed=12, max_steps=24, **kw)
    e, r_env, e2 = mk(fence_counters=1), mk(), mk()      # e: the bookkeeping build (per-step IK update counts out)
    st = O.PickState(n)
    obs_r = O.pick_reset(kuka, cfg, st, seed=12)
    for x in (e, r_env, e2):
        x.reset()
    scripted = np.arange(n) < n // 2
    acts = []
    n_done = n_succ = 0
    book = FenceBook(n, 20, cfg.fence_pivot)
    iters, minpiv = np.zeros(n, dtype=np.int32), np.zeros(n)
    loose_worst, n_masked = 0.0, 0
    for t in range(T):
        a = _pick_actions(obs_r, st.aux, rng=rng, scripted=scripted, sigma=sigma)
        acts.append(a)
        obs, rew, done, succ = e.step(torch.from_numpy(a).to(DEV), want_terminal_obs=True, want_ik_updates=True)
        obs_r, rew_r, done_r, succ_r, term_r = O.pick_step_autoreset(kuka, cfg, st, a, seed=12, iters=iters, minpiv=minpiv)
        chk = book.comparable(iters, minpiv)
        assert np.array_equal(_np(done)[chk], done_r.astype(bool)[chk]), t
        assert np.array_equal(_np(e.ik_updates)[chk], iters[chk]), t
        assert np.abs(_np(obs) - obs_r)[chk].max() < 1e-5 and np.abs(_np(e.terminal_obs) - term_r)[chk].max() < 1e-5, t
        assert np.abs(_np(rew) - rew_r)[chk].max() < 1e-3, t
        masked = book.sync & ~chk                     # excluded by the fence, still in the same episode as the twin
        loose_worst = max(loose_worst, float(np.abs(_np(obs) - obs_r)[masked].max(initial=0.0)))
        n_masked += int(masked.sum())
        n_done += int(done_r.sum()); n_succ += int((rew_r == 100).sum())
        book.advance(_np(done), done_r.astype(bool))
    assert n_done >= 3 * n and n_succ >= n // 2
    if sigma < 0.2:       # gentle exploration: nothing capped, (next to) nothing ill-conditioned: (nearly) every env compared at every step
        assert book.cap_calls == 0 and book.checked >= 0.99 * book.total, (book.cap_calls, book.cond_calls, book.checked, book.total)
    else:
        assert book.checked >= 0.9 * book.total and book.cap_calls > 0, (book.checked, book.total, book.cap_calls)
        # a capped call leaves the two sides a few cm apart at most (the arm is at the edge of its reach, twenty clamped
        # updates each): bounded, not skipped
        assert n_masked > 0 and loose_worst < 0.15, (n_masked, loose_worst)
    s = e.get_state()
    ok = book.sync & book.clean
    assert np.abs(_np(s["aux"])[ok, :11] - st.aux[ok, :11]).max() < 1e-6 and np.array_equal(_np(s["step"])[book.sync], st.step[book.sync])
    c = e.counters()
    assert abs(c["episodes"] - n_done) <= int((~book.sync).sum()) and c["nonfinite"] == 0
    out = r_env.rollout(T, torch.from_numpy(np.stack(acts)).to(DEV))
    for t in range(T):
        o, r, d, su = e2.step(torch.from_numpy(acts[t]).to(DEV))
        _same_rollout_step(out, t, o, r, d)
    sa, sb = r_env.get_state(), e2.get_state()
    assert (sa["aux"] - sb["aux"]).abs().max().item() < 2e-6 and (sa["q"] - sb["q"]).abs().max().item() < 2e-5
    for x in (e, r_env, e2):
        x.close()


def test_pick_gripper_model_properties(envs):
    """Size-independent properties of the gripper model at 32 768 envs: a held cube rides rigidly with the tip, a
    closed gripper never reopens within an episode, the cube never sinks below its rest height, the eef honours the
    pick workspace (z <= 0.55 + 0.257), and the fused random policy runs the task."""
    n, T = 32768, 30
    e = envs.BatchedPickEnv(n, device=DEV, seed=3, auto_reset=False)
    obs = e.reset()
    grip_prev = torch.zeros(n, dtype=torch.float64, device=DEV)
    rel_prev = None
    for t in range(T):
        aux = e.get_state()["aux"]
        a = torch.from_numpy(_pick_actions(_np(obs), _np(aux))).to(DEV)
        obs, rew, done, succ = e.step(a)
        aux = e.get_state()["aux"]
        grip = aux[:, 7]
        assert bool((grip >= grip_prev).all())                          # 0 -> 1 / 2, never back
        assert float(aux[:, 2].min()) >= 0.01 - 0.0158 - 1e-9                   # (the landing overshoot: 0.8 mm into the table)
        held = (grip == 2) & (grip_prev == 2)
        if rel_prev is not None and bool(held.any()):
            fk_p, fk_q = e.fk(e.get_state()["q"])
            lifted = held & (aux[:, 2] > float(e.cfg.push_rest_z) + 1e-4)    # above the table: exactly tip + offset
            w, x, y, z = fk_q[:, 3], fk_q[:, 0], fk_q[:, 1], fk_q[:, 2]
            axis = torch.stack([2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)], 1)
            tip = fk_p + 0.257 * axis
            assert float(((aux[:, 0:3] - tip) - aux[:, 8:11])[lifted].abs().max()) < 1e-9
        grip_prev, rel_prev = grip.clone(), True
    assert float((grip_prev == 2).double().mean()) > 0.95 and float(succ.double().mean()) > 0.9
    e.close()
    e = envs.BatchedPickEnv(n, device=DEV, seed=1)
    e.reset()
    e.set_policy("random", action_bound=0.4, noise_sigma=0.4 * 0.98, noise_clip=1e9)
    out = e.rollout(40, None)
    eef = out["obs"][..., :3]
    # the box clips the IK *target* (:313, :330-333); near the top of the pick box the arm cannot reach it with the tool
    # pointing down and Bullet's loop stops after 20 updates wherever it is, a few cm off at most
    assert float(eef[..., 2].max()) <= 0.55 + 0.257 + 0.05 and float(eef[..., 2].min()) >= -0.05
    assert float((eef[..., 2] <= 0.55 + 0.257 + 2e-4).float().mean()) > 0.999
    c = e.counters()
    assert c["env_steps"] == n * 40 and c["nonfinite"] == 0
    e.close()


def test_rlpickenv_compat_surface(envs):
    """The reference's call pattern for the pick env (envs/rl_pick_env.py:44-448 surface) on the drop-in class."""
    import random
    env = envs.RLPickEnv(is_render=False, is_good_view=False)               # resets once, as the reference does (:133)
    random.seed(0); np.random.seed(0)
    assert abs(float(env.action_space.high[0]) - 0.4) < 1e-7 and env.observation_space.shape == (3,)
    assert abs(float(env.observation_space.high[2]) - (0.55 + 0.257)) < 1e-6             # rl_pick_env.py:94-97
    state = env.reset()
    assert state.shape == (9,) and state.dtype == np.float64
    spawn = state[3:6].copy(); spawn[2] = 0.01                                            # :194
    d = np.linalg.norm(spawn - state[6:9])
    assert 0.22 <= d <= 0.25 and abs(state[5] - (0.01 - 10.0 / 240.0 ** 2)) < 1e-15 and 0.0 <= state[8] <= 0.55      # one step into its fall
    g = golden_json("py_random_pick_seed0.json")                                         # the reference's draw pattern
    # the fixture holds the SPAWN poses of the reference's loop (cube z = 0.01); the observed cube has come to rest on the table
    assert list(state[3:5]) == g["placements"][0]["cube"][:2] and list(state[6:9]) == g["placements"][0]["target"]
    for _ in range(5):
        action = np.zeros(3) + np.random.normal(0, 0.4 * 0.98, size=3)
        state, reward, done, info = env.step(action)
        assert state.shape == (9,) and isinstance(done, bool) and set(info) == {"is_success"}
        assert info["is_success"].dtype == np.float32 and (reward == -1.0 or -1.0 < reward < 1.0) and not done      # (the cube falls: its distance to the floating target moves by millimetres per step)
    assert env.gripper_state == 0
    state = env.reset()
    assert list(state[3:5]) == g["placements"][1]["cube"][:2] and list(state[6:9]) == g["placements"][1]["target"]
    env.close()


def test_diana_cam_reach_kinematics_preset(envs, O):
    """The Diana S1 set-up of envs/diana_cam_reach.py (yawed base, identity target orientation, dv 0.005, unclipped IK
    target) through BatchedReachEnv: a yawed base takes the generic FK path; steps match the oracle given the same
    constants; and the first reset puts the tool where SURVEY.md derives it from the URDF."""
    kw = envs.diana_cam_reach_kinematics()
    n = 512
    e = envs.BatchedReachEnv(n, device=DEV, seed=2, auto_reset=False, **kw)
    assert e.kernel_name == "reach_step<f64,generic>"
    ch = O.make_chain("diana", base_rpy=(0.0, 0.0, math.pi))
    cfg = O.default_config()
    cfg.target_quat[:] = kw["target_quat"]; cfg.dv = kw["dv"]
    cfg.goal_lo[:] = kw["goal_lo"]; cfg.goal_hi[:] = kw["goal_hi"]; cfg.box_lo[:] = kw["box_lo"]; cfg.box_hi[:] = kw["box_hi"]
    st = O.ReachState(n)
    obs_r = O.reach_reset(ch, cfg, st, seed=2)
    obs = _np(e.reset())
    assert np.abs(obs - obs_r).max() < 1e-6
    assert np.abs(obs[0, :3] - np.float32([0.591001, -0.1541, 0.421794])).max() < 2e-6
    rng = np.random.default_rng(96)
    for t in range(20):
        a = rng.normal(0.0, 0.5, (n, 3)).astype(np.float32)
        e.set_state(q=st.q, goal=st.goal, step=st.step, ep_return=st.ep_return)
        o, r, d, su = e.step(torch.from_numpy(a).to(DEV))
        o = _np(o).copy()
        o_r, r_r, d_r, s_r, iters = O.reach_step(ch, cfg, st, a)
        ok = (np.abs(_np(e.get_state()["q"]) - st.q).max(1) < 1e-6) & (iters < 20)
        assert ok.mean() > 0.97 and np.abs(o - o_r)[ok].max() < 1e-6, (t, ok.mean())
    e.close()


@pytest.mark.parametrize("task", ["push", "pick"])
@pytest.mark.parametrize("robot,fk_path", [("kuka", 1), ("diana", 0)])
def test_cube_tasks_on_other_chain_paths(envs, O, task, robot, fk_path):
    """Push / pick lanes instantiated for the generic-chain FK path and for the Diana table (a reachable Diana set-up as
    in test_step_teacher_forced_f64): teacher-forced steps against the oracle, and rollout == step launches bit for bit."""
    n, T = 192, 10
    rng = np.random.default_rng(95)
    ch = O.make_chain(robot)
    cfg = O.default_config(task)
    over = {}
    if robot == "diana":
        cfg.target_quat[:] = [1.0, 0.0, 0.0, 0.0]
        cfg.q_init[:] = [0.0, 0.5, 0.0, 1.6, 0.0, -1.0, 0.0]
        p_init, _ = O.fk(ch, cfg.q_init[:])
        lo = (p_init[0] - 0.25).tolist(); hi = (p_init[0] + 0.25).tolist()
        cfg.box_lo[:] = lo; cfg.box_hi[:] = hi; cfg.goal_lo[:] = lo; cfg.goal_hi[:] = hi
        over = dict(target_quat=list(cfg.target_quat), q_init=list(cfg.q_init), box_lo=lo, box_hi=hi, goal_lo=lo, goal_hi=hi)
    Env = envs.BatchedPushEnv if task == "push" else envs.BatchedPickEnv
    State, reset, step = (O.PushState, O.push_reset, O.push_step) if task == "push" else (O.PickState, O.pick_reset, O.pick_step)
    mk = lambda **kw: Env(n, device=DEV, seed=9, robot=robot, fk_path=fk_path, **over, **kw)
    e = mk(auto_reset=False)
    assert e.kernel_name == f"{task}_step<f64,{'generic' if fk_path == 1 else robot}>"
    st = State(n)
    reset(ch, cfg, st, seed=9)
    e.reset()
    for t in range(T):
        a = rng.normal(0.0, 0.3, (n, 3)).astype(np.float32)
        e.set_state(q=st.q, aux=st.aux, step=st.step, ep_return=st.ep_return)
        obs, rew, done, succ = e.step(torch.from_numpy(a).to(DEV))
        obs = _np(obs).copy()
        obs_r, rew_r, done_r, succ_r, iters = step(ch, cfg, st, a)
        ok = (np.abs(_np(e.get_state()["q"]) - st.q).max(1) < 1e-6) & (iters < 20)
        assert ok.mean() > 0.97 and np.abs(obs - obs_r)[ok].max() < 1e-6, (t, ok.mean())
    e.close()
    a_env, b_env = mk(), mk()
    a_env.reset(); b_env.reset()
    acts = torch.from_numpy(rng.normal(0.0, 0.3, (T, n, 3)).astype(np.float32)).to(DEV)
    out = a_env.rollout(T, acts)
    for t in range(T):
        o, r, d, su = b_env.step(acts[t])
        _same_rollout_step(out, t, o, r, d)
    a_env.close(); b_env.close()


# ------------------------------------------------------------------------------ fused TD3 actor (A1, config 3)

def _golden_actor():
    g = golden_npz("td3_actor_seed0.npz")
    sd = {k: g[k.replace(".", "_")] for k in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc3.weight", "fc3.bias")}
    return g, sd


def test_actor_forward_matches_reference_golden(envs, O):
    """G3 was produced by importing the reference's TD3_MLP (algo/TD3/TD3_mlp.py:33-97, net_mlp.py:29-40);
    tolerance 1e-5 on actions in f32."""
    g, sd = _golden_actor()
    e = _mk(envs, 64)
    e.set_policy("actor", action_bound=float(g["action_bound"]), actor_state_dict={k: torch.from_numpy(v) for k, v in sd.items()})
    for n in (1024, 1000, 1, 65):                                    # ragged tails of the last wavefront
        a = _np(e.actor_forward(torch.from_numpy(g["states"][:n])))
        assert a.shape == (n, 3)
        assert np.abs(a - g["actions"][:n]).max() < 1e-5
        assert np.abs(a - O.actor_forward(sd, g["states"][:n], float(g["action_bound"]))).max() < 1e-5
    e.close()


def test_actor_f16x3_within_tolerance_of_reference(envs, O):
    """Fast actor variant (f16 MFMA, hi/lo operand split, three passes): still within the 1e-5 actor tolerance of the
    reference's golden vectors; and usable as the fused rollout policy."""
    g, sd = _golden_actor()
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    e = _mk(envs, 256, seed=31, max_steps=20)
    e.set_policy("actor_f16x3", action_bound=float(g["action_bound"]), noise_sigma=0.1, noise_clip=0.7, actor_state_dict=tsd)
    for n in (1024, 777, 1):
        a = _np(e.actor_forward(torch.from_numpy(g["states"][:n])))
        assert np.abs(a - g["actions"][:n]).max() < 1e-5
    rng = np.random.default_rng(81)
    big = {"fc1.weight": rng.normal(0, 0.8, (256, 6)), "fc1.bias": rng.normal(0, 0.3, 256),
           "fc2.weight": rng.normal(0, 0.09, (256, 256)) * np.linspace(0.5, 1.5, 256)[None, :], "fc2.bias": rng.normal(0, 0.2, 256),
           "fc3.weight": rng.normal(0, 0.12, (3, 256)), "fc3.bias": rng.normal(0, 0.1, 3)}
    big = {k: v.astype(np.float32) for k, v in big.items()}
    st = rng.uniform(-1, 1, (640, 6)).astype(np.float32)
    e2 = _mk(envs, 64)
    e2.set_policy("actor_f16x3", action_bound=0.7, actor_state_dict={k: torch.from_numpy(v) for k, v in big.items()})
    assert np.abs(_np(e2.actor_forward(torch.from_numpy(st))) - O.actor_forward(big, st, 0.7)).max() < 1e-5
    e2.close()
    # as the fused policy: same trajectory as the exact-f32 actor to within the step tolerance
    ref = _mk(envs, 256, seed=31, max_steps=20)
    ref.set_policy("actor", action_bound=0.7, noise_sigma=0.1, noise_clip=0.7, actor_state_dict=tsd)
    e.reset(); ref.reset()
    oa = e.rollout(30, None, want_actions=True); ob = ref.rollout(30, None, want_actions=True)
    assert float((oa["actions"] - ob["actions"]).abs().max()) < 2e-5
    assert torch.equal(oa["done"], ob["done"]) and float((oa["obs"] - ob["obs"]).abs().max()) < 1e-5
    e.close(); ref.close()


@pytest.mark.parametrize("n", [64, 320, 448, 1024 + 192])
def test_actor_f16x3_ragged_workgroups(envs, n):
    """The f16x3 actor is a 4-wave workgroup phase (W2 through an LDS ring filled cooperatively): workgroups with 1, 2 or 3
    live waves (num_envs not a multiple of 256) take the slow fill path and must give the same trajectory as the
    per-wave exact-f32 actor."""
    g, sd = _golden_actor()
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    a, b = _mk(envs, n, seed=33, max_steps=15), _mk(envs, n, seed=33, max_steps=15)
    a.set_policy("actor_f16x3", action_bound=0.7, noise_sigma=0.2, noise_clip=0.7, actor_state_dict=tsd)
    b.set_policy("actor", action_bound=0.7, noise_sigma=0.2, noise_clip=0.7, actor_state_dict=tsd)
    a.reset(); b.reset()
    oa, ob = a.rollout(24, None, want_actions=True), b.rollout(24, None, want_actions=True)
    assert float((oa["actions"] - ob["actions"]).abs().max()) < 2e-5
    assert torch.equal(oa["done"], ob["done"]) and float((oa["obs"] - ob["obs"]).abs().max()) < 1e-5
    assert int(oa["done"].sum()) >= n                       # episodes ended and were reset inside the launch
    a.close(); b.close()


def _datd3_nets(g):
    keys = ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc3.weight", "fc3.bias")
    return [{k: g["%s_%s" % (n, k.replace(".", "_"))] for k in keys} for n in ("actor1", "actor2", "critic1", "critic2")]


@pytest.mark.parametrize("task", ["reach", "push"])
def test_fused_datd3_take_action_matches_reference_golden(envs, O, task):
    """DATD3_MLP.take_action (/root/reference/algo/DATD3/DATD3_mlp.py:88-109) on the device -- armenv_set_policy_datd3 +
    armenv_datd3_forward: four f16x3 MFMA passes (actor1, actor2, critic1 on cat(s, a1), critic2 on cat(s, a2)) through one set of LDS
    tables and one W2 ring -- against the vectors produced by calling the reference's own take_action one state at a time (G11: reach,
    6-float observations, critics 9 -> 256 -> 256 -> 1; G14: push / pick, 9-float observations, critics 12 -> ...): Q values within 1e-5,
    the same actor picked wherever |q1 - q2| > 1e-4 (both branches occur), actions within 1e-5; and against the oracle on 4 096 + 192
    random states (a ragged last workgroup), same bounds."""
    g = golden_npz("datd3_take_action_seed0.npz" if task == "reach" else "datd3_take_action9_seed0.npz")
    nets = _datd3_nets(g)
    tn = [{k: torch.from_numpy(v) for k, v in sd.items()} for sd in nets]
    bound = float(g["action_bound"])
    Env = envs.BatchedReachEnv if task == "reach" else envs.BatchedPushEnv
    e = Env(4096 + 192, device=DEV, seed=3)
    e.set_policy_datd3(*tn, action_bound=bound)
    a, q1, q2, pk = (_np(x) for x in e.datd3_forward(torch.from_numpy(g["states"]).to(DEV), want_q=True))
    assert np.abs(q1 - g["q1"]).max() < 1e-5 and np.abs(q2 - g["q2"]).max() < 1e-5
    clear = np.abs(g["q1"] - g["q2"]) > 1e-4
    assert clear.sum() >= 245 and np.array_equal(pk[clear], g["picked_actor"][clear].astype(np.uint8)) and 100 < pk[clear].sum() < 156
    assert np.abs(a - g["actions"])[clear].max() < 1e-5
    rng = np.random.default_rng(5)
    D = g["states"].shape[1]
    lo, hi = g["states"].min(0), g["states"].max(0)
    st = (lo + (hi - lo) * rng.random((4096 + 192, D), dtype=np.float32)).astype(np.float32)
    a, q1, q2, pk = (_np(x) for x in e.datd3_forward(torch.from_numpy(st).to(DEV), want_q=True))
    ar, q1r, q2r, pr = O.datd3_take_action(nets, st, bound)
    assert np.abs(q1 - q1r).max() < 1e-5 and np.abs(q2 - q2r).max() < 1e-5
    clear = np.abs(q1r - q2r) > 1e-4
    assert clear.mean() > 0.9 and np.array_equal(pk[clear], pr[clear]) and np.abs(a - ar)[clear].max() < 1e-5
    assert np.array_equal(_np(e.datd3_forward(torch.from_numpy(st).to(DEV))), a)
    e.close()


def test_fused_datd3_push_rollout(envs, O, kuka):
    """The DATD3 policy of the cube tasks (9-float observations, G14 nets) folded into the push rollout kernel: on a 2 048 + 64-env handle
    (ragged last workgroup) the actions of 40 fused steps are the oracle's take_action + noise on the observations the policy saw
    (wherever the critics are not within 1e-4 of each other) to 2e-5, and rollout(T) equals T step(None) launches bit for bit; the same
    on a pick handle runs and stays finite."""
    g = golden_npz("datd3_take_action9_seed0.npz")
    nets = _datd3_nets(g)
    tn = [{k: torch.from_numpy(v) for k, v in sd.items()} for sd in nets]
    n, T = 2048 + 64, 40
    mk = lambda: envs.BatchedPushEnv(n, device=DEV, seed=12, max_steps=25)
    e, e2 = mk(), mk()
    for x in (e, e2):
        x.set_policy_datd3(*tn, action_bound=0.4, noise_sigma=0.4 * 0.98, noise_clip=1e9)
    obs_prev = _np(e.reset()).copy(); e2.reset()
    out = e.rollout(T, None, want_actions=True)
    acts, obs = _np(out["actions"]), _np(out["obs"])
    ids = np.arange(n)
    episode, step = np.ones(n, dtype=np.uint32), np.zeros(n, dtype=np.int32)
    done = _np(out["done"])
    worst = 0.0
    for t in range(T):
        mu, q1, q2, _ = O.datd3_take_action(nets, obs_prev, 0.4)
        nz = O.policy_noise_ids(12, ids, episode, step)
        want = mu + np.float32(0.4 * 0.98) * nz
        clear = np.abs(q1 - q2) > 1e-4
        worst = max(worst, float(np.abs(acts[t] - want)[clear].max()))
        step += 1
        fin = done[t]
        episode[fin] += 1; step[fin] = 0
        obs_prev = obs[t].copy()
    assert worst < 2e-5 and done.sum() >= n, (worst, int(done.sum()))
    for t in range(T):
        o, r, d, s = e2.step(None)
        _same_rollout_step(out, t, o, r, d)
    e.close(); e2.close()
    p = envs.BatchedPickEnv(256, device=DEV, seed=2)
    p.set_policy_datd3(*tn, action_bound=0.4, noise_sigma=0.4 * 0.98, noise_clip=1e9)
    p.reset()
    o = p.rollout(20, None)
    assert bool(torch.isfinite(o["obs"]).all()) and p.counters()["nonfinite"] == 0
    p.close()


def test_fused_datd3_rollout_vs_oracle(envs, O, kuka):
    """The DATD3 policy folded into the rollout kernel (ARMENV_POLICY_DATD3) at 65 536 reach envs, 2 x armenv_rollout(50) with
    run()'s exploration noise, checked like BASELINE configs[2] on a strided sample of 2 048 envs: actions within 2e-5 of the
    oracle's take_action + noise on the observations the fused policy saw (wherever the two critics are not within 1e-4 of each
    other), observations within 1e-5 and identical flags with the oracle teacher-forced on the engine's actions; a 256 + 64-env
    handle (ragged workgroup) gives rollout == step launches bit for bit."""
    g = golden_npz("datd3_take_action_seed0.npz")
    nets = _datd3_nets(g)
    tn = [{k: torch.from_numpy(v) for k, v in sd.items()} for sd in nets]
    n, T, stride = 65536, 50, 32
    e = envs.BatchedReachEnv(n, device=DEV, seed=4)
    e.set_policy_datd3(*tn, action_bound=0.7, noise_sigma=0.7 * 0.98, noise_clip=0.7)
    obs0 = _np(e.reset()).copy()
    ids = np.arange(0, n, stride)
    cfg = O.default_config()
    st = O.ReachState(ids.size)
    st.q[:] = np.array(O.INIT_Q); st.goal[:] = obs0[ids, 3:]; st.episode[:] = 1
    obs_prev = obs0[ids].copy()
    worst_a = worst_o = 0.0
    flags = unclear = 0
    for launch in range(2):
        out = e.rollout(T, None, want_actions=True, want_terminal_obs=True)
        acts, obs, term, done, succ = (_np(out[k]) for k in ("actions", "obs", "terminal_obs", "done", "success"))
        for t in range(T):
            mu, q1, q2, _ = O.datd3_take_action(nets, obs_prev, 0.7)
            nz = O.policy_noise_ids(4, ids, st.episode, st.step)
            want = np.clip(mu + np.float32(0.7 * 0.98) * nz, -np.float32(0.7), np.float32(0.7))
            a = acts[t][ids]
            clear = np.abs(q1 - q2) > 1e-4
            unclear += int((~clear).sum())
            worst_a = max(worst_a, float(np.abs(a - want)[clear].max()))
            o_r, r_r, d_r, s_r, iters = O.reach_step(kuka, cfg, st, a)
            worst_o = max(worst_o, float(np.abs(term[t][ids] - o_r).max()))
            flags += int((done[t][ids] != d_r.astype(bool)).sum() + (succ[t][ids] != s_r.astype(bool)).sum())
            fin = d_r.astype(bool)
            if fin.any():
                st.q[fin] = np.array(O.INIT_Q); st.step[fin] = 0; st.episode[fin] += 1; st.ep_return[fin] = 0
                st.goal[fin] = obs[t][ids][fin, 3:]
            obs_prev = obs[t][ids].copy()
    e.close()
    assert worst_a < 2e-5 and worst_o < 1e-5 and flags == 0, (worst_a, worst_o, flags)
    assert unclear < 0.02 * 2 * T * ids.size
    a_env, b_env = (envs.BatchedReachEnv(256 + 64, device=DEV, seed=9, max_steps=12) for _ in range(2))
    for x in (a_env, b_env):
        x.set_policy_datd3(*tn, action_bound=0.7); x.reset()
    out = a_env.rollout(30, None)
    for t in range(30):
        o, r, d, s = b_env.step(None)
        _same_rollout_step(out, t, o, r, d)
    a_env.close(); b_env.close()


def _nets_of(g, names):
    keys = ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc3.weight", "fc3.bias")
    return [{k: g["%s_%s" % (n, k.replace(".", "_"))] for k in keys} for n in names]


@pytest.mark.parametrize("agent,task", [("daddpg", "reach"), ("daddpg", "push"), ("darc", "reach")])
def test_fused_daddpg_darc_take_action_match_reference_golden(envs, O, agent, task):
    """The reference's DEFAULT agent, opt.algo = 'DADDPG_MLP' (/root/reference/config.py:33): take_action with two actors and ONE critic
    valuing both proposals (/root/reference/algo/DADDPG/DADDPG_mlp.py:77-97) -- armenv_set_policy_daddpg: three packed nets, the
    critic's second pass on the LDS tables and W2 ring its first pass left -- and DARC_MLP.take_action
    (/root/reference/algo/DARC/DARC_mlp.py:92-113: DATD3's selection, through armenv_set_policy_datd3), against G15 (produced by calling
    the reference's own take_action one state at a time): Q values within 1e-5, the same actor picked wherever |q1 - q2| > 1e-4 (both
    branches occur; `>=`: a tie takes actor 1 in all three agents), actions within 1e-5; and against the oracle on 4 096 + 192 random
    states (ragged last workgroup).  The DADDPG install equals a DATD3 install given the same critic twice, bit for bit."""
    g = golden_npz({("daddpg", "reach"): "daddpg_take_action_seed0.npz", ("daddpg", "push"): "daddpg_take_action9_seed0.npz",
                    ("darc", "reach"): "darc_take_action_seed0.npz"}[(agent, task)])
    nets = _nets_of(g, ("actor1", "actor2", "critic", "critic") if agent == "daddpg" else ("actor1", "actor2", "critic1", "critic2"))
    tn = [{k: torch.from_numpy(v) for k, v in sd.items()} for sd in nets]
    bound = float(g["action_bound"])
    Env = envs.BatchedReachEnv if task == "reach" else envs.BatchedPushEnv
    e = Env(4096 + 192, device=DEV, seed=3)
    if agent == "daddpg":
        e.set_policy_daddpg(tn[0], tn[1], tn[2], action_bound=bound)
    else:
        e.set_policy_darc(*tn, action_bound=bound)
    a, q1, q2, pk = (_np(x) for x in e.datd3_forward(torch.from_numpy(g["states"]).to(DEV), want_q=True))
    assert np.abs(q1 - g["q1"]).max() < 1e-5 and np.abs(q2 - g["q2"]).max() < 1e-5
    clear = np.abs(g["q1"] - g["q2"]) > 1e-4
    assert clear.sum() >= 240 and np.array_equal(pk[clear], g["picked_actor"][clear].astype(np.uint8)) and 10 < pk[clear].sum() < 246
    assert np.abs(a - g["actions"])[clear].max() < 1e-5
    tie = g["q1"] == g["q2"]                      # `q1 >= q2` -> actor 1
    assert not tie.any() or (g["picked_actor"][tie] == 0).all()
    rng = np.random.default_rng(5)
    D = g["states"].shape[1]
    lo, hi = g["states"].min(0), g["states"].max(0)
    st = (lo + (hi - lo) * rng.random((4096 + 192, D), dtype=np.float32)).astype(np.float32)
    got = [_np(x) for x in e.datd3_forward(torch.from_numpy(st).to(DEV), want_q=True)]
    a, q1, q2, pk = got
    ar, q1r, q2r, pr = O.datd3_take_action(nets, st, bound)
    assert np.abs(q1 - q1r).max() < 1e-5 and np.abs(q2 - q2r).max() < 1e-5
    clear = np.abs(q1r - q2r) > 1e-4
    assert clear.mean() > 0.9 and np.array_equal(pk[clear], pr[clear]) and np.abs(a - ar)[clear].max() < 1e-5
    if agent == "daddpg":
        assert e._lib.armenv_kernel_name(e._h) and e._policy == "daddpg"
        e2 = Env(4096 + 192, device=DEV, seed=3)
        e2.set_policy_datd3(tn[0], tn[1], tn[2], {k: v.clone() for k, v in tn[2].items()}, action_bound=bound)   # four staged nets
        for x, y in zip(got, e2.datd3_forward(torch.from_numpy(st).to(DEV), want_q=True)):
            assert np.array_equal(x, _np(y))
        e2.close()
    e.close()


def test_fused_daddpg_rollout_vs_oracle(envs, O, kuka):
    """DADDPG_MLP.take_action folded into the rollout kernel (ARMENV_POLICY_DADDPG) at 65 536 reach envs, 2 x armenv_rollout(50) with
    run()'s exploration noise (main.py:116-117), on a strided sample of 2 048 envs: actions within 2e-5 of the oracle's take_action + noise
    on the observations the fused policy saw (wherever the critic does not value the two proposals within 1e-4 of each other), observations
    within 1e-5 and identical flags with the oracle teacher-forced on the engine's actions; rollout == step(None) launches bit for bit on a
    ragged 256 + 64-env handle; a push handle (G15's nine-input nets) stays finite."""
    g = golden_npz("daddpg_take_action_seed0.npz")
    nets = _nets_of(g, ("actor1", "actor2", "critic", "critic"))
    tn = [{k: torch.from_numpy(v) for k, v in sd.items()} for sd in nets]
    n, T, stride = 65536, 50, 32
    e = envs.BatchedReachEnv(n, device=DEV, seed=4)
    e.set_policy_daddpg(tn[0], tn[1], tn[2], action_bound=0.7, noise_sigma=0.7 * 0.98, noise_clip=0.7)
    obs0 = _np(e.reset()).copy()
    ids = np.arange(0, n, stride)
    cfg = O.default_config()
    st = O.ReachState(ids.size)
    st.q[:] = np.array(O.INIT_Q); st.goal[:] = obs0[ids, 3:]; st.episode[:] = 1
    obs_prev = obs0[ids].copy()
    worst_a = worst_o = 0.0
    flags = unclear = 0
    picks = 0
    for launch in range(2):
        out = e.rollout(T, None, want_actions=True, want_terminal_obs=True)
        acts, obs, term, done, succ = (_np(out[k]) for k in ("actions", "obs", "terminal_obs", "done", "success"))
        for t in range(T):
            mu, q1, q2, pk = O.datd3_take_action(nets, obs_prev, 0.7)
            picks += int(pk.sum())
            nz = O.policy_noise_ids(4, ids, st.episode, st.step)
            want = np.clip(mu + np.float32(0.7 * 0.98) * nz, -np.float32(0.7), np.float32(0.7))
            a = acts[t][ids]
            clear = np.abs(q1 - q2) > 1e-4
            unclear += int((~clear).sum())
            worst_a = max(worst_a, float(np.abs(a - want)[clear].max()))
            o_r, r_r, d_r, s_r, iters = O.reach_step(kuka, cfg, st, a)
            worst_o = max(worst_o, float(np.abs(term[t][ids] - o_r).max()))
            flags += int((done[t][ids] != d_r.astype(bool)).sum() + (succ[t][ids] != s_r.astype(bool)).sum())
            fin = d_r.astype(bool)
            if fin.any():
                st.q[fin] = np.array(O.INIT_Q); st.step[fin] = 0; st.episode[fin] += 1; st.ep_return[fin] = 0
                st.goal[fin] = obs[t][ids][fin, 3:]
            obs_prev = obs[t][ids].copy()
    e.close()
    assert worst_a < 2e-5 and worst_o < 1e-5 and flags == 0, (worst_a, worst_o, flags)
    assert unclear < 0.02 * 2 * T * ids.size and 0.02 < picks / (2 * T * ids.size) < 0.98        # both actors act
    a_env, b_env = (envs.BatchedReachEnv(256 + 64, device=DEV, seed=9, max_steps=12) for _ in range(2))
    for x in (a_env, b_env):
        x.set_policy_daddpg(tn[0], tn[1], tn[2], action_bound=0.7); x.reset()
    out = a_env.rollout(30, None)
    for t in range(30):
        o, r, d, s = b_env.step(None)
        _same_rollout_step(out, t, o, r, d)
    a_env.close(); b_env.close()
    g9 = golden_npz("daddpg_take_action9_seed0.npz")
    t9 = [{k: torch.from_numpy(v) for k, v in sd.items()} for sd in _nets_of(g9, ("actor1", "actor2", "critic"))]
    p = envs.BatchedPushEnv(2048 + 64, device=DEV, seed=2)
    p.set_policy_daddpg(*t9, action_bound=0.4, noise_sigma=0.4 * 0.98, noise_clip=1e9)
    p.reset()
    o = p.rollout(20, None)
    assert bool(torch.isfinite(o["obs"]).all()) and p.counters()["nonfinite"] == 0
    p.close()


def test_fused_actor_nine_inputs_push(envs, O):
    """obs_dim 9 (push / pick): both fused actor variants against the oracle's actor on the observations they saw."""
    rng = np.random.default_rng(82)
    sd = {"fc1.weight": rng.normal(0, 0.5, (256, 9)), "fc1.bias": rng.normal(0, 0.3, 256),
          "fc2.weight": rng.normal(0, 0.09, (256, 256)), "fc2.bias": rng.normal(0, 0.2, 256),
          "fc3.weight": rng.normal(0, 0.12, (3, 256)), "fc3.bias": rng.normal(0, 0.1, 3)}
    sd = {k: v.astype(np.float32) for k, v in sd.items()}
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    n = 320
    for kind in ("actor", "actor_f16x3"):
        e = envs.BatchedPushEnv(n, device=DEV, seed=6, max_steps=30)
        e.set_policy(kind, action_bound=0.4, noise_sigma=0.0, noise_clip=1e9, actor_state_dict=tsd)
        obs0 = e.reset().clone()
        out = e.rollout(12, None, want_actions=True)
        seen = torch.cat([obs0[None], out["obs"][:-1]])          # the observation each action was computed from
        ref = O.actor_forward(sd, _np(seen).reshape(-1, 9), 0.4).reshape(12, n, 3)
        assert np.abs(_np(out["actions"]) - ref).max() < 1e-5, kind
        st = rng.uniform(-1, 1, (777, 9)).astype(np.float32)
        assert np.abs(_np(e.actor_forward(torch.from_numpy(st))) - O.actor_forward(sd, st, 0.4)).max() < 1e-5, kind
        e.close()


def test_actor_forward_asymmetric_weights(envs, O):
    """Transpose-detecting check of the MFMA operand/accumulator maps: random, non-symmetric weights with a
    distinct scale per layer, and biases that make about half the units fire."""
    rng = np.random.default_rng(80)
    sd = {"fc1.weight": rng.normal(0, 0.8, (256, 6)), "fc1.bias": rng.normal(0, 0.3, 256),
          "fc2.weight": rng.normal(0, 0.09, (256, 256)) * np.linspace(0.5, 1.5, 256)[None, :], "fc2.bias": rng.normal(0, 0.2, 256),
          "fc3.weight": rng.normal(0, 0.12, (3, 256)), "fc3.bias": rng.normal(0, 0.1, 3)}
    sd = {k: v.astype(np.float32) for k, v in sd.items()}
    states = rng.uniform(-1, 1, (640, 6)).astype(np.float32)
    e = _mk(envs, 64)
    e.set_policy("actor", action_bound=0.7, actor_state_dict={k: torch.from_numpy(v) for k, v in sd.items()})
    a = _np(e.actor_forward(torch.from_numpy(states)))
    ref = O.actor_forward(sd, states, 0.7)
    t = torch.from_numpy
    h = torch.relu(t(states) @ t(sd["fc1.weight"]).T + t(sd["fc1.bias"]))
    h = torch.relu(h @ t(sd["fc2.weight"]).T + t(sd["fc2.bias"]))
    tref = (torch.tanh(h @ t(sd["fc3.weight"]).T + t(sd["fc3.bias"])) * 0.7).numpy()
    assert np.abs(ref - tref).max() < 2e-6 and np.abs(a - tref).max() < 5e-6 and ref.std() > 0.1
    e.close()


def test_rollout_fused_actor_matches_oracle(envs, O, kuka):
    """BASELINE config 3 path: TD3 actor folded into the rollout kernel + exploration noise (main.py:114-117)."""
    g, sd = _golden_actor()
    n, T = 256, 45
    cfg = O.default_config(); cfg.max_steps = 20
    e = _mk(envs, n, seed=31, max_steps=20)
    e.set_policy("actor", action_bound=0.7, noise_sigma=0.1, noise_clip=0.7, actor_state_dict={k: torch.from_numpy(v) for k, v in sd.items()})
    st = O.ReachState(n)
    obs0 = _np(e.reset()).copy(); obs0_r = O.reach_reset(kuka, cfg, st, seed=31)
    out = e.rollout(T, None, want_actions=True)
    ref = O.reach_rollout(kuka, cfg, st, T, None, seed=31, sigma=0.1, clip=0.7, actor=sd, bound=0.7, obs0=obs0_r)
    assert np.abs(_np(out["actions"]) - ref["actions"]).max() < 2e-5
    assert np.array_equal(_np(out["done"]), ref["done"].astype(bool))
    assert np.abs(_np(out["obs"]) - ref["obs"]).max() < 1e-5
    assert np.abs(_np(out["reward"]) - ref["reward"]).max() < 1e-4
    assert ref["done"].sum() >= 2 * n and np.abs(ref["actions"]).mean() > 0.02
    e.close()


def test_exploration_noise_is_fresh_after_every_kind_of_reset(envs):
    """The fused policy's noise is keyed by (env, episode, step); `episode` counts EVERY reset of an env -- Philox goals,
    caller goals (armenv_reset_with_goal: the N=1 classes' path) and in-place auto-resets alike -- so no episode replays
    the exploration sequence of the one before.  The same call sequence on a second handle is reproduced."""
    n = 256
    goal = torch.rand(n, 3) * torch.tensor([0.5, 0.6, 0.55]) + torch.tensor([0.2, -0.3, 0.0])
    runs = []
    for rep in range(2):
        e = _mk(envs, n, seed=4, auto_reset=False)
        e.set_policy("random")
        seqs = []
        for ep in range(3):
            e.reset(goal=goal)
            seqs.append(e.rollout(6, None, want_actions=True)["actions"].clone())
        assert _np(e.get_state()["episode"]).tolist() == [3] * n
        e.close()
        runs.append(seqs)
    a, b, c = runs[0]
    assert not torch.equal(a, b) and not torch.equal(b, c) and not torch.equal(a, c)
    assert (a - b).abs().mean().item() > 0.1                      # independent draws, not a shifted copy
    assert all(torch.equal(x, y) for x, y in zip(runs[0], runs[1]))


def test_set_policy_errors(envs):
    from armenv import ArmEnvError
    g, sd = _golden_actor()
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    e = _mk(envs, 100)                                               # not a multiple of 64
    with pytest.raises(ArmEnvError, match="multiple of 64"):
        e.set_policy("actor", actor_state_dict=tsd)
    e.close()
    e = _mk(envs, 64)
    with pytest.raises(ArmEnvError, match="no actor"):
        e.actor_forward(torch.zeros(4, 6))
    # every shape is checked before a pointer is taken (the packing kernel indexes with assumed strides)
    bad = dict(tsd); bad["fc1.weight"] = torch.zeros(128, 6); bad["fc1.bias"] = torch.zeros(128)
    with pytest.raises(ValueError, match="fc1.weight has shape"):
        e.set_policy("actor", actor_state_dict=bad)
    bad = dict(tsd); bad["fc1.weight"] = torch.zeros(256, 9)                 # an actor trained for the push / pick observation
    with pytest.raises(ValueError, match=r"fc1.weight has shape \(256, 9\)"):
        e.set_policy("actor_f16x3", actor_state_dict=bad)
    bad = dict(tsd); del bad["fc3.bias"]
    with pytest.raises(ValueError, match="no 'fc3.bias'"):
        e.set_policy("actor", actor_state_dict=bad)
    critic = dict(tsd); critic["fc1.weight"] = torch.zeros(256, 9); critic["fc3.weight"] = torch.zeros(1, 256); critic["fc3.bias"] = torch.zeros(1)
    with pytest.raises(ValueError, match="critic2: fc1.weight"):
        e.set_policy_datd3(tsd, tsd, critic, tsd)                            # an actor where critic 2 belongs
    with pytest.raises(ValueError, match="actor2: fc1.weight"):
        e.set_policy_datd3(tsd, critic, critic, critic)
    e.set_policy_datd3(tsd, tsd, critic, critic)                             # the right shapes pass
    with pytest.raises(ArmEnvError, match="noise"):
        e.set_policy("random", noise_sigma=-1.0)
    e.close()


# ------------------------------------------------------------------------------ trajectory store + HER sampler (8f.1)

def _store_from_golden(g):
    from armenv.replay import TrajectoryStore
    st = TrajectoryStore(device=DEV, seed=5)
    t = lambda k, dt=None: torch.from_numpy(g[k]).to(DEV)
    out = dict(obs=t("obs_after"), terminal_obs=t("next_obs"), reward=t("reward"), done_u8=t("done"), actions=t("action"))
    st.add_rollout(t("obs0"), out)
    return st


@pytest.mark.parametrize("task", ["reach", "push"])
def test_her_sampler_matches_reference_golden(envs, task):
    """G6 (outputs of the reference's ReplayBuffer_Trajectory_*.sample and the draws it made): the HIP sampler fed the
    same draws reproduces every value."""
    g = golden_npz(f"her_{task}_seed0.npz")
    st = _store_from_golden(g)
    assert st.size() == len(g["episodes"])
    assert np.array_equal(_np(st.chunk["episodes"])[: st.size()], g["episodes"])
    out = st.sample(len(g["picks"]), use_her=True, dis_threshold=float(g["dis_threshold"]), her_ratio=float(g["her_ratio"]),
                    picks=g["picks"], return_picks=True)
    assert np.array_equal(_np(out["picks"]), g["picks"])
    assert np.array_equal(_np(out["states"]).astype(np.float64), g["states"])
    assert np.array_equal(_np(out["next_states"]).astype(np.float64), g["next_states"])
    assert np.array_equal(_np(out["actions"]), g["actions"]) and np.array_equal(_np(out["dones"]), g["dones"])
    assert np.abs(_np(out["rewards"]).astype(np.float64) - g["rewards"]).max() < 1e-7


def test_her_sampler_own_draws_are_valid_and_uniform(envs):
    from oracle import her
    g = golden_npz("her_reach_seed0.npz")
    st = _store_from_golden(g)
    B = 200000
    out = st.sample(B, use_her=True, her_ratio=0.8, return_picks=True)
    pk = _np(out["picks"]); eps = g["episodes"]
    L = eps[pk[:, 0], 2]
    assert (pk[:, 0] >= 0).all() and (pk[:, 0] < len(eps)).all() and (pk[:, 1] >= 0).all() and (pk[:, 1] < L).all()
    h = pk[:, 2] == 1
    assert ((pk[h, 3] >= pk[h, 1] + 1) & (pk[h, 3] <= L[h])).all()
    assert abs(h.mean() - 0.8) < 0.01                                         # np.random.uniform() <= her_ratio
    cnt = np.bincount(pk[:, 0], minlength=len(eps)) / B
    assert np.abs(cnt - 1.0 / len(eps)).max() < 0.25 / len(eps)               # random.sample(buffer, 1): uniform over trajectories
    # every value equals the restatement run on the kernel's own draws
    ch = {k: g[k] for k in ("obs0", "obs_after", "next_obs", "action", "reward", "done")}
    ref = her.sample_with_picks(ch, eps, pk[:5000], 0.1)
    for k in ("states", "next_states", "actions", "dones"):
        assert np.array_equal(_np(out[k])[:5000], ref[k]), k
    assert np.abs(_np(out["rewards"])[:5000] - ref["rewards"]).max() < 1e-7
    # no HER: plain transitions
    out2 = st.sample(1000, use_her=False, return_picks=True)
    assert int(out2["picks"][:, 2].sum()) == 0
    # a second call draws a different batch (draw counter), the same store + seed + counter repeats
    st2 = _store_from_golden(g)
    a = st2.sample(64, return_picks=True)["picks"]; b = st2.sample(64, return_picks=True)["picks"]
    assert not torch.equal(a, b)


def test_rollout_into_store_end_to_end(envs):
    """main.py:108-138 on the device: reset -> rollout (fused random policy) -> trajectory store -> HER batches."""
    from armenv.replay import TrajectoryStore
    n, T = 4096, 60
    e = _mk(envs, n, seed=3, max_steps=12)                                   # 13-step episodes: ~4 complete per env
    e.set_policy("random")
    obs0 = e.reset().clone()
    out = e.rollout(T, None, want_actions=True, want_terminal_obs=True)
    st = TrajectoryStore(device=DEV, seed=1)
    st.add_rollout(obs0, out)
    n_done = int(out["done"].sum())
    assert st.size() == n_done and n_done >= 4 * n
    ep = st.chunk["episodes"][: st.size()]
    assert int(ep[:, 2].max()) <= 13 and int(ep[:, 2].min()) >= 1
    b = st.sample(65536, use_her=True, her_ratio=0.8, return_picks=True)
    pk = b["picks"]; her = pk[:, 2] == 1
    # relabelled goals are achieved eef positions, rewards follow the 0.1 threshold
    d = (b["next_states"][:, :3] - b["next_states"][:, 3:6]).norm(dim=1)
    assert bool(((b["rewards"][her] == 1.0) == (d[her] <= 0.1)).all())
    assert bool((b["states"][:, 3:6] == b["next_states"][:, 3:6]).all())
    # un-relabelled samples carry the env's own reward: -10 * distance or 0 on success (rl_reach_env.py:299-309)
    keep = ~her & (b["dones"] == 0)
    assert float((b["rewards"][keep] + 10.0 * d[keep]).abs().max()) < 1e-4
    # consecutive states of a transition differ by one bounded arm move (dv * 0.7 per axis + IK residual)
    assert float((b["next_states"][:, :3] - b["states"][:, :3]).abs().max()) < 0.02 * 0.7 + 2e-3
    e.close()


def test_trajectory_ring_matches_linear_window(envs):
    """Chunks accumulated in a ring (capacity smaller than the total) give the same episode index and the same samples
    as the restatement run on the equivalent linear window."""
    from armenv.replay import TrajectoryStore
    from oracle import her
    n, Tc, cap = 192, 25, 80
    e = _mk(envs, n, seed=6, max_steps=17)                       # 18-step episodes
    e.set_policy("random")
    obs = e.reset()
    st = TrajectoryStore(device=DEV, seed=2, capacity_steps=cap)
    keep = []
    for c in range(6):                                           # 150 steps through an 80-step ring (wraps twice)
        obs0 = obs.clone()
        out = e.rollout(Tc, None, want_actions=True, want_terminal_obs=True)
        obs = out["obs"][-1].clone()
        st.add_rollout(obs0, out, starts_at_reset=(c == 0))
        keep.append({k: _np(out[k]).copy() for k in ("obs", "terminal_obs", "actions", "reward", "done_u8")})
        cat = lambda k: np.concatenate([x[k] for x in keep])
        tot = len(keep) * Tc
        lo = max(0, tot - cap) if tot > cap else 0
        # the ring drops whole overflow: window = last min(tot, cap) steps
        T = min(tot, cap)
        lo = tot - T
        ch = dict(obs0=_np(obs0) * 0, obs_after=cat("obs")[lo:], next_obs=cat("terminal_obs")[lo:], action=cat("actions")[lo:],
                  reward=cat("reward")[lo:], done=cat("done_u8")[lo:])
        if lo == 0:
            ch["obs0"] = _np(st.chunk["obs0"])
        eps = her.index_episodes(ch["done"], starts_at_reset=(lo == 0))
        assert st.chunk["T"] == T and st.size() == len(eps)
        assert np.array_equal(_np(st.chunk["episodes"])[: len(eps)], eps)
        b = st.sample(4096, use_her=True, return_picks=True)
        ref = her.sample_with_picks(ch, eps, _np(b["picks"]), 0.1)
        for k in ("states", "next_states", "actions", "dones"):
            assert np.array_equal(_np(b[k]), ref[k]), (c, k)
        assert np.abs(_np(b["rewards"]) - ref["rewards"]).max() < 1e-7
    assert st.size() > n
    # an empty window yields an inert, defined batch
    empty = TrajectoryStore(device=DEV, seed=2, capacity_steps=40)
    e2 = _mk(envs, 64, seed=1)
    e2.set_policy("random"); o0 = e2.reset().clone()
    empty.add_rollout(o0, e2.rollout(10, None, want_actions=True, want_terminal_obs=True))
    assert empty.size() == 0
    z = empty.sample(32, return_picks=True)
    assert int(z["picks"][:, 0].max()) == -1 and float(z["states"].abs().max()) == 0.0 and bool((z["dones"] == 1).all())
    e.close(); e2.close()


# ------------------------------------------------------------------------------ stream / graph behaviour (boundary)

def test_step_is_capturable_in_a_hip_graph(envs):
    """armenv_step only enqueues a kernel on the caller's stream (no allocation, no sync), so a policy + step loop
    can be captured once in a HIP graph (torch.cuda.CUDAGraph) and replayed."""
    n = 4096
    torch.manual_seed(0)
    W = (torch.randn(6, 3, device=DEV) * 0.5)
    policy = lambda o: torch.tanh(o @ W) * 0.7
    eager = _mk(envs, n, seed=12); graphed = _mk(envs, n, seed=12)
    obs_e = eager.reset().clone()
    ref = []
    for _ in range(6):
        o, r, d, s = eager.step(policy(obs_e).contiguous())
        obs_e = o.clone(); ref.append((o.clone(), r.clone(), d.clone()))
    obs_g = graphed.reset()                                   # the env's persistent obs tensor: graph input and output
    side = torch.cuda.Stream(DEV)
    side.wait_stream(torch.cuda.current_stream(DEV))
    with torch.cuda.stream(side):                              # warm-up on the capture stream
        act = policy(obs_g).contiguous()
    torch.cuda.current_stream(DEV).wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        act = policy(graphed._obs).contiguous()
        o, r, d, s = graphed.step(act)
    # capture does not execute: state still at reset
    for t in range(6):
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(o, ref[t][0]) and torch.equal(r, ref[t][1]) and torch.equal(d, ref[t][2]), t
    assert graphed.counters()["env_steps"] == 6 * n
    eager.close(); graphed.close()


def test_rollout_is_capturable_in_a_hip_graph(envs):
    """armenv_rollout (both schedules) inside a HIP graph: a captured 16-step launch over static action / output buffers,
    replayed four times with fresh actions copied into the static buffer, equals the same 64 steps run eagerly."""
    n, T = 2048, 16
    for Env, kw in ((envs.BatchedReachEnv, {}), (envs.BatchedPickEnv, {})):          # lockstep, lane-asynchronous (pick default)
        gen = torch.Generator(device=DEV); gen.manual_seed(5)
        acts = (torch.randn((4, T, n, 3), device=DEV, generator=gen) * 0.4).contiguous()
        eager = Env(n, device=DEV, seed=12, max_steps=30, **kw); graphed = Env(n, device=DEV, seed=12, max_steps=30, **kw)
        eager.reset(); graphed.reset()
        ref = [{k: v.clone() for k, v in eager.rollout(T, acts[j]).items()} for j in range(4)]
        static = acts[0].clone()
        side = torch.cuda.Stream(DEV)
        side.wait_stream(torch.cuda.current_stream(DEV))
        bufs = {}
        with torch.cuda.stream(side):
            launch, out = graphed.bind_rollout(T, static, out=bufs)          # buffers allocated outside the capture
        torch.cuda.current_stream(DEV).wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            graphed.rollout(T, static, out=bufs)
        for j in range(4):
            static.copy_(acts[j])
            g.replay()
            torch.cuda.synchronize()
            for k in ("obs", "reward", "done", "success"):
                assert torch.equal(out[k], ref[j][k]), (Env.__name__, j, k)
        assert graphed.counters() == eager.counters()
        eager.close(); graphed.close()


def test_two_handles_on_two_streams_are_independent(envs):
    n = 8192
    a = _mk(envs, n, seed=1); b = _mk(envs, n, seed=2)
    s1, s2 = torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)
    act = torch.zeros(n, 3, device=DEV)
    torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        a.reset(); oa = [a.step(act)[0].clone() for _ in range(5)]
    with torch.cuda.stream(s2):
        b.reset(); ob = [b.step(act)[0].clone() for _ in range(5)]
    torch.cuda.synchronize()
    c = _mk(envs, n, seed=1); c.reset()
    oc = [c.step(act)[0].clone() for _ in range(5)]
    assert all(torch.equal(x, y) for x, y in zip(oa, oc)) and not torch.equal(oa[-1][:, 3:], ob[-1][:, 3:])
    a.close(); b.close(); c.close()


def test_training_loop_smoke(envs):
    """armenv.train (main.py:77-162 on the device) runs end to end: rollouts with the fused actor, ring store, HER
    batches, TD3 updates; a few iterations only (learning curves: profiles/r01_train_*.jsonl)."""
    from armenv.train import train_reach
    logs = []
    agent, hist = train_reach(num_envs=256, iterations=12, rollout_steps=16, updates=4, batch_size=256, window_steps=64,
                              max_steps=20, log_every=4, log=logs.append, use_graphs=True)      # (the opt-in hipGraph path still runs)
    assert len(hist) == 3 and hist[-1]["env_steps"] == 256 * 16 * 12 and hist[-1]["episodes"] >= 256 * 8
    assert agent.total_it > 0
    assert all(torch.isfinite(p).all() for p in agent.actor.parameters())


def test_plain_c_consumer(envs, tmp_path):
    """tests/c_abi/consumer.c -- plain C against include/armenv.h, its own process, no torch -- produces what the Python
    host side produces for the same seed and actions."""
    import subprocess
    from test_host_logic import build_c_consumer
    exe = build_c_consumer(str(tmp_path))
    n, steps = 4096, 10
    r = subprocess.run([exe, str(n), str(steps)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    f = r.stdout.split()
    e = envs.BatchedReachEnv(n, device=DEV, seed=7)
    e.reset()
    a = torch.tensor([[0.5, -0.25, -0.6]], device=DEV).repeat(n, 1).contiguous()
    for _ in range(steps):
        obs, rew, done, succ = e.step(a)
    assert np.array_equal(np.float32([float(x) for x in f[:6]]), _np(obs)[0])
    assert abs(float(f[6]) - float(rew.double().sum())) < 1e-3
    assert int(f[7]) == n * steps and int(f[8]) == 0 and f[9] == e.kernel_name
    e.close()


def test_td3_update_as_hipgraph_equals_eager(envs):
    """TD3.capture / train_graphed (the update replayed from hipGraphs over static buffers) follows the eager update:
    identical batches, target-policy noise off (its random stream differs inside a graph), 30 updates incl. ten delayed
    actor + soft updates.  Compared functionally (losses, Q values and actions on a held-out batch): Adam's
    normalisation turns last-bit differences of near-zero gradients into lr-sized parameter differences, so individual
    parameters are not comparable bit for bit between two executions.  capture() leaves parameters untouched."""
    from armenv.td3 import TD3
    torch.manual_seed(3)
    a = TD3(6, 3, 0.7, device=DEV, policy_noise=0.0)
    b = TD3(6, 3, 0.7, device=DEV, policy_noise=0.0)
    for nb, na in zip((b.actor, b.critic, b.target_actor, b.target_critic), (a.actor, a.critic, a.target_actor, a.target_critic)):
        nb.load_state_dict(na.state_dict())
    B = 512
    before = [p.detach().clone() for p in list(b.critic.parameters()) + list(b.actor.parameters())]
    static = b.capture(B)
    assert all(torch.equal(p, q) for p, q in zip(list(b.critic.parameters()) + list(b.actor.parameters()), before))
    assert all(float(v.abs().max()) == 0.0 for st in b.critic_opt.state.values() for v in st.values())   # fresh Adam state
    gen = torch.Generator(device=DEV); gen.manual_seed(11)
    mk = lambda: dict(states=torch.rand(B, 6, device=DEV, generator=gen), actions=torch.rand(B, 3, device=DEV, generator=gen) - 0.5,
                      next_states=torch.rand(B, 6, device=DEV, generator=gen), rewards=torch.rand(B, device=DEV, generator=gen),
                      dones=(torch.rand(B, device=DEV, generator=gen) < 0.1).to(torch.uint8))
    for it in range(30):
        batch = mk()
        la, lb = float(a.train(batch)), float(b.train_graphed(batch))
        assert abs(la - lb) < 2e-3 * max(1.0, abs(la)), (it, la, lb)
    assert a.total_it == b.total_it == 30
    held = mk()
    with torch.no_grad():
        qa, qb = a.critic(held["states"], held["actions"]), b.critic(held["states"], held["actions"])
        assert float((qa[0] - qb[0]).abs().max()) < 5e-3 and float((qa[1] - qb[1]).abs().max()) < 5e-3
        assert float((a.actor(held["states"]) - b.actor(held["states"])).abs().max()) < 5e-3
        assert float((a.target_actor(held["states"]) - b.target_actor(held["states"])).abs().max()) < 5e-3
        moved = float((b.actor(held["states"]) - TD3(6, 3, 0.7, device=DEV).actor(held["states"])).abs().max())
    assert moved > 1e-2                                    # the graphed learner really updated its actor
    assert static is b._graphs["buf"]


def test_training_loop_daddpg_default_agent(envs):
    """The reference's `run()` with its DEFAULT agent (main.py:77-162 with opt.algo = 'DADDPG_MLP', config.py:33) on the device:
    DADDPG_MLP.take_action fused into the rollouts (armenv_set_policy_daddpg, re-installed from the learner every iteration), HER
    batches from the device store, DADDPG's alternating updates replayed from hipGraphs -- and the graphed update follows the eager
    one (as test_td3_update_as_hipgraph_equals_eager)."""
    from armenv.daddpg import DADDPG
    from armenv.train import train_reach
    agent, hist = train_reach(num_envs=256, iterations=8, rollout_steps=16, updates=4, batch_size=256, window_steps=64,
                              max_steps=20, log_every=4, log=lambda s: None, algo="daddpg", use_graphs=True)
    assert isinstance(agent, DADDPG) and agent.total_it > 0
    assert len(hist) == 2 and hist[-1]["env_steps"] == 256 * 16 * 8 and hist[-1]["episodes"] >= 256 * 5
    assert all(torch.isfinite(p).all() for n_ in agent._nets() for p in n_.parameters())
    torch.manual_seed(3)
    a, b = DADDPG(6, 3, 0.7, device=DEV), DADDPG(6, 3, 0.7, device=DEV)
    for nb, na in zip(b._nets(), a._nets()):
        nb.load_state_dict(na.state_dict())
    B = 512
    b.capture(B)
    gen = torch.Generator(device=DEV); gen.manual_seed(11)
    mk = lambda: dict(states=torch.rand(B, 6, device=DEV, generator=gen), actions=torch.rand(B, 3, device=DEV, generator=gen) - 0.5,
                      next_states=torch.rand(B, 6, device=DEV, generator=gen), rewards=torch.rand(B, device=DEV, generator=gen),
                      dones=(torch.rand(B, device=DEV, generator=gen) < 0.1).to(torch.uint8))
    for it in range(20):
        batch = mk()
        la, lb = float(a.train(batch)), float(b.train_graphed(batch))
        assert abs(la - lb) < 5e-3 * max(1.0, abs(la)), (it, la, lb)      # (Adam: lr-sized differences from the fifth update on, see TD3's test)
    held = mk()
    with torch.no_grad():       # (functional comparison: Adam turns last-bit gradient differences into lr-sized parameter differences)
        for x, y in ((a.actor1, b.actor1), (a.actor2, b.actor2), (a.target_actor1, b.target_actor1)):
            assert float((x(held["states"]) - y(held["states"])).abs().max()) < 1e-2
        assert float((a.critic(held["states"], held["actions"]) - b.critic(held["states"], held["actions"])).abs().max()) < 1e-2


@pytest.mark.parametrize("algo,graphs", [("td3", True), ("daddpg", True), ("td3", False)])
def test_training_loop_learns_the_reach_task(envs, algo, graphs):
    """The on-device `run()` loop with its default settings LEARNS: >= 90 % of the episodes finished in the last 20 of 140 iterations end
    in success (reach_dis 0.01), for train_reach_with_TD3's agent and for the reference's default agent DADDPG, with the updates
    replayed from hipGraphs (the default) and issued eagerly.  A regression test of round 6: until then only smoke runs were tested, and
    the replayed updates had stopped learning (TD3: 20-60 %) -- every captured bias gradient was wrong from the second replay on, because
    a hipMemsetAsync captured into a hipGraph works once on this build (profiles/r06_td3_hipgraph_learning.txt)."""
    from armenv.train import train_reach
    hist = []
    import json
    train_reach(iterations=140, log_every=20, log=lambda s_: hist.append(json.loads(s_)), algo=algo, use_graphs=graphs)
    assert hist[-1]["success_rate"] >= 0.9 and hist[-1]["episodes"] > 5000, [round(h["success_rate"], 2) for h in hist]


@pytest.mark.parametrize("agent", ["td3", "daddpg"])
def test_captured_update_has_the_eager_updates_gradients(envs, agent):
    """The root cause test of round 6: from an identical state (parameters, Adam moments and step counters, batch) the update replayed
    from its hipGraph leaves the SAME gradients in the optimisers' .grad buffers as the eager update -- all of them, on every one of
    40 consecutive updates (before the fix: the bias gradients were off by 0.4-0.7 on two updates of three from the seventh on, every
    other gradient bit-identical) -- and reports the same loss."""
    from armenv.daddpg import DADDPG
    from armenv.td3 import TD3
    torch.manual_seed(0)
    G = TD3(6, 3, 0.7, policy_noise=0.0) if agent == "td3" else DADDPG(6, 3, 0.7)
    B = 2048
    buf = G.capture(B)
    gen = torch.Generator(device=DEV); gen.manual_seed(1)

    def snap():
        return ([{k: v.clone() for k, v in n_.state_dict().items()} for n_ in G._nets()],
                [[{k: v.clone() for k, v in st.items()} for st in o.state.values()] for o in G._opts()])

    def restore(s_):
        with torch.no_grad():
            for n_, sd in zip(G._nets(), s_[0]):
                for k, v in n_.state_dict().items():
                    v.copy_(sd[k])
            for o, sts in zip(G._opts(), s_[1]):
                for st, saved in zip(o.state.values(), sts):
                    for k, v in st.items():
                        v.copy_(saved[k])
    params = [p for o in G._opts() for p in o.param_groups[0]["params"]]
    worst = 0.0
    for it in range(40):
        for k, v in buf.items():
            v.copy_((torch.rand(v.shape, device=DEV, generator=gen) * (1.1 if v.dtype == torch.uint8 else 1)).to(v.dtype))
        G.total_it += 1
        flag = G._flag()
        s0 = snap()
        G._graphs["g"][flag].replay()
        gg, lg = [p.grad.clone() for p in params], float(G._graphs["loss"])
        restore(s0)
        le = float(G._update(buf["states"], buf["actions"], buf["rewards"].view(-1, 1), buf["next_states"],
                             buf["dones"].to(torch.float32).view(-1, 1), flag))
        worst = max(worst, max(float((a_ - p.grad).abs().max()) for a_, p in zip(gg, params)))
        assert abs(lg - le) < 1e-5 * max(1.0, abs(le)), (it, lg, le)
    assert worst < 1e-5, worst


def test_training_loop_smoke_pick(envs):
    """train_pick_with_TD3 (main.py:518-585) on the device: the 9-input fused actor drives PickLane rollouts."""
    from armenv.train import train_push
    agent, hist = train_push(num_envs=256, iterations=8, rollout_steps=16, updates=4, batch_size=256, window_steps=64,
                             max_steps=20, log_every=4, log=lambda s: None, task="pick")
    assert len(hist) == 2 and hist[-1]["env_steps"] == 256 * 16 * 8 and hist[-1]["episodes"] >= 256 * 5
    assert all(torch.isfinite(p).all() for p in agent.actor.parameters())
    # the default agent (two actors, one critic on 9-float observations) on the push task
    agent, hist = train_push(num_envs=256, iterations=8, rollout_steps=16, updates=4, batch_size=256, window_steps=64,
                             max_steps=20, log_every=4, log=lambda s: None, task="push", algo="daddpg")
    assert hist[-1]["env_steps"] == 256 * 16 * 8 and all(torch.isfinite(p).all() for n_ in agent._nets() for p in n_.parameters())


def test_device_summary_matches_host_reduction(envs, O, kuka):
    n = 4096 + 17                                              # ragged last wave
    e = _mk(envs, n, seed=14, max_steps=9)
    e.set_policy("random"); e.reset()
    e.rollout(25, None)
    s = e.summary()
    st = e.get_state()
    p, _ = O.fk(kuka, _np(st["q"]))
    d = np.linalg.norm(p - _np(st["goal"]).astype(np.float64), axis=1)
    ret, ln, su = e.episode_stats()
    assert abs(float(s["mean_distance"]) - d.mean()) < 1e-9 and abs(float(s["max_distance"]) - d.max()) < 1e-9
    assert abs(float(s["mean_last_return"]) - _np(ret).mean()) < 1e-9
    assert abs(float(s["mean_last_len"]) - _np(ln).mean()) < 1e-9 and float(s["raw"][5]) == n
    assert abs(float(s["last_success_rate"]) - _np(su).mean()) < 1e-12
    s2 = e.summary()                                           # fixed reduction order: repeat calls agree bit for bit
    assert torch.equal(s["raw"], s2["raw"])
    e.close()
    pe = envs.BatchedPushEnv(1024, device=DEV, seed=3)
    pe.reset()
    aux = _np(pe.get_state()["aux"])
    sp = pe.summary()
    assert abs(float(sp["mean_distance"]) - np.linalg.norm(aux[:, :3] - aux[:, 3:6], axis=1).mean()) < 1e-12
    pe.close()


# ------------------------------------------------------------------------------ the other consumers: DDPG, DATD3 (north_star)

def _sd_from(g, prefix):
    return {k: torch.from_numpy(g[f"{prefix}_{k.replace('.', '_')}"]) for k in
            ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc3.weight", "fc3.bias")}


def test_ddpg_actor_through_the_fused_policy(envs, O, kuka):
    """DDPG_MLP.take_action (algo/DDPG/DDPG_mlp.py:76-91) is the TD3 PolicyNet: the reference agent's own actions (golden
    G11, produced by importing algo.DDPG) come out of the fused MFMA actor -- exact f32 and f16x3 -- within 1e-5, and a fused
    rollout driven by those weights follows the oracle."""
    g = golden_npz("ddpg_take_action_seed0.npz")
    sd = _sd_from(g, "actor")
    states = torch.from_numpy(g["states"]).to(DEV)
    for kind in ("actor", "actor_f16x3"):
        e = _mk(envs, 256, seed=3)
        e.set_policy(kind, action_bound=float(g["action_bound"]), noise_sigma=0.0, noise_clip=0.7, actor_state_dict=sd)
        a = _np(e.actor_forward(states))
        assert np.abs(a - g["actions"]).max() < 1e-5, (kind, np.abs(a - g["actions"]).max())
        e.close()
    # fused rollout with DDPG's actor and the run() exploration noise against the oracle's actor + noise restatement
    n, T = 256, 30
    cfg = O.default_config()
    e = _mk(envs, n, seed=12)
    e.set_policy("actor", action_bound=0.7, noise_sigma=0.7 * 0.98, noise_clip=0.7, actor_state_dict=sd)
    st = O.ReachState(n)
    obs0 = _np(e.reset()).copy()
    O.reach_reset(kuka, cfg, st, seed=12)
    out = e.rollout(T, None, want_actions=True)
    ref = O.reach_rollout(kuka, cfg, st, T, None, seed=12, actor={k: v.numpy() for k, v in sd.items()}, bound=0.7, obs0=obs0)
    assert np.abs(_np(out["actions"]) - ref["actions"]).max() < 1e-4
    assert np.abs(_np(out["obs"]) - ref["obs"]).max() < 1e-4 and np.array_equal(_np(out["done"]), ref["done"].astype(bool))
    e.close()


def test_datd3_is_an_external_action_consumer(envs, O, kuka):
    """DATD3_MLP.take_action (algo/DATD3/DATD3_mlp.py:88-109: two actors, the critics' arg-max) has no fused form in the
    rollout kernel: it consumes the env through armenv_step with external actions.  The batched policy on the device
    reproduces the reference agent's own choices (golden G11), and a 40-step loop policy -> step -> obs follows the oracle
    driven with the same actions."""
    from armenv.policies import DATD3Policy
    g = golden_npz("datd3_take_action_seed0.npz")
    pol = DATD3Policy(6, 3, float(g["action_bound"]), device=DEV).load(*[_sd_from(g, k) for k in ("actor1", "actor2", "critic1", "critic2")])
    a, q1, q2 = pol.take_action(torch.from_numpy(g["states"]).to(DEV), return_q=True)
    clear = np.abs(g["q1"] - g["q2"]) > 1e-4
    assert np.abs(_np(a) - g["actions"])[clear].max() < 1e-5 and np.abs(_np(q1) - g["q1"]).max() < 1e-4
    n, T = 512, 40
    cfg = O.default_config()
    e = _mk(envs, n, seed=21)
    st = O.ReachState(n)
    obs = e.reset()
    O.reach_reset(kuka, cfg, st, seed=21)
    picked = np.zeros(2, dtype=np.int64)
    for t in range(T):
        a, q1, q2 = pol.take_action(obs, return_q=True)
        assert a.dtype == torch.float32 and a.is_contiguous() and tuple(a.shape) == (n, 3)
        picked += np.bincount(_np(q1 < q2).astype(np.int64), minlength=2)
        obs, rew, done, succ = e.step(a)
        obs_r, rew_r, done_r, succ_r, _ = O.reach_step_autoreset(kuka, cfg, st, _np(a), seed=21)
        assert np.abs(_np(obs) - obs_r).max() < 1e-5 and np.array_equal(_np(done), done_r.astype(bool)), t
    assert picked.min() > 0                     # both actors were chosen along the way
    with pytest.raises(Exception):              # and there is no fused form to install
        e.set_policy("datd3")
    e.close()


def test_td3_learner_golden_on_the_gpu(monkeypatch):
    """G8 on cuda:0: the reference's six TD3_MLP.train() updates reproduced by armenv.td3.TD3 with
    every network on the device.  The target-policy noise of the golden run came from torch's CPU generator; the test
    feeds the same stream (randn on the CPU, moved to the device) so that losses and parameters are comparable."""
    from armenv.td3 import TD3
    g = golden_npz("td3_train_seed0.npz")
    torch.manual_seed(0)
    agent = TD3(6, 3, 0.7, device="cpu")          # same initial parameters as the reference's constructor order ...
    init = [{k: v.clone() for k, v in n.state_dict().items()} for n in (agent.actor, agent.critic, agent.target_actor, agent.target_critic)]
    agent = TD3(6, 3, 0.7, device=DEV)            # ... moved into a learner that lives on the GPU
    for n_, sd in zip((agent.actor, agent.critic, agent.target_actor, agent.target_critic), init):
        n_.load_state_dict(sd)
    monkeypatch.setattr(torch, "randn_like", lambda t: torch.randn(t.shape, dtype=t.dtype).to(t.device))
    torch.manual_seed(123)
    for i, want in enumerate(g["losses"]):
        b = {k: torch.from_numpy(g[f"b{i}_{k}"]).to(DEV) for k in ("states", "actions", "next_states", "rewards", "dones")}
        loss = float(agent.train(b))
        assert abs(loss - want) < 2e-5 * max(1.0, abs(want)), (i, loss, want)
    # Parameters: Adam divides by sqrt(v): where a gradient is ~0 a last-bit difference between the CPU and GPU reductions
    # becomes an lr-sized (1e-3 per update) difference of that element, so element-wise equality holds for all but a handful;
    # the functions the networks compute are compared against the golden parameters on the first batch.
    ref = TD3(6, 3, 0.7, device=DEV)
    for name, net, rnet in (("actor", agent.actor, ref.actor), ("critic", agent.critic, ref.critic),
                            ("target_actor", agent.target_actor, ref.target_actor), ("target_critic", agent.target_critic, ref.target_critic)):
        rnet.load_state_dict({k: torch.from_numpy(g[f"{name}__{k.replace('.', '_')}"]) for k in net.state_dict()})
        for k, v in net.state_dict().items():
            assert v.device.type == "cuda"
            d = np.abs(_np(v) - g[f"{name}__{k.replace('.', '_')}"])
            assert (d < 2e-5).mean() > 0.999 and d.max() < 7e-3, (name, k, (d < 2e-5).mean(), d.max())
    s0, a0 = torch.from_numpy(g["b0_states"]).to(DEV), torch.from_numpy(g["b0_actions"]).to(DEV)
    with torch.no_grad():
        assert (agent.actor(s0) - ref.actor(s0)).abs().max().item() < 1e-4
        assert (agent.target_actor(s0) - ref.target_actor(s0)).abs().max().item() < 1e-4
        for x, y in zip(agent.critic(s0, a0) + agent.target_critic(s0, a0), ref.critic(s0, a0) + ref.target_critic(s0, a0)):
            assert (x - y).abs().max().item() < 1e-4


def _run_bench(argv, nproc=1, timeout=900, torchrun=False, env=None):
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable]
    if nproc > 1 or torchrun:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    r = subprocess.run(cmd + [os.path.join(root, "bench.py")] + argv, capture_output=True, text=True, timeout=timeout,
                       env=dict(os.environ, **env) if env else None)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{\"metric\"")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_contract():
    """The driver's own command (`--steps 20 --warmup 5`): ONE JSON line with the contract's keys (metric / value / unit /
    n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload +
    roofline + cpu_baseline), the binding VALU bound inside `roofline`, the one-thread and all-core CPU legs, and the
    parity fence of the workload."""
    d = _run_bench(["--steps", "20", "--warmup", "5"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "parity_fence"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["vs_baseline"] is None and d["dtype"] == "f64"
    assert d["unit"] == "env-steps/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and "workload" in d["config"]
    assert abs(d["value"] - 65536 * 20 / (d["ms_per_step"] * 20 / 1e3)) / d["value"] < 1e-6
    assert d["config"]["steps_per_launch"] == 20 and d["config"]["launches"] == 1
    ro, cb, pf = d["roofline"], d["cpu_baseline"], d["parity_fence"]
    assert ro["bound"] == "hbm" and ro["unit"] == "GB/s" and ro["peak"] == 8000.0 and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-12
    assert ro["achieved"] * 1e9 * ro["avg_launch_us"] * 1e-6 == pytest.approx(ro["algo_bytes_per_launch"], rel=1e-9)
    assert ro["traffic_key"] == "reach_rollout<f64,kuka>|policy=external|T=20|N=65536"
    assert ro["traffic"] is not None and 0.9 < ro["traffic"] / ro["algo_bytes_per_launch"] < 1.3      # PMC pass at this launch shape
    assert ro["binding_bound"] == "valu" and 0.2 < ro["valu"]["frac"] < 1.0 and ro["valu"]["unit"] == "TFLOP/s"
    one = ro["valu"]["one_wave_per_simd"]        # the single-wave f64 issue ceiling beside the nominal peak: measured in the run
    assert "measured in this run" in one["source"] and one["simds"] == 1024
    assert 4.0 <= one["cycles_per_f64_instruction"] < 9.0 and one["two_waves_per_simd_ns"] < one["ns_per_f64_instruction"]
    assert one["peak"] == pytest.approx(128 * 1024 / one["ns_per_f64_instruction"] * 1e-3) and one["peak"] < 78.6 and 0.5 < one["frac"] < 1.1
    # the headline as a distribution: the identical 20-step region 15 more times on fresh action rows
    assert d["regions"] == 16 and len(d["launch_us_samples"]) == 16 and d["value_min"] <= d["value_median"] <= d["value_max"]
    assert d["value_min"] <= d["value"] <= d["value_max"] and d["launch_us_samples"][0] == pytest.approx(ro["avg_launch_us"], abs=0.01)
    assert d["launch_us_max"] < 1.6 * d["launch_us_min"], d["launch_us_samples"]      # sanity, not a performance claim
    # clock probes around every region (ns per chained v_fma_f32, median over one probe wave per SIMD: 4-8 cycles at 1.4-2.5 GHz),
    # every region behind --busy-ahead-ms of scratch work, and round 4's two regimes (nothing in front; host / device restore) beside it
    assert d["config"]["busy_ahead_ms"] == 8.0 and d["clock_probe_xcds"] == 8
    assert len(d["clock_probe_ns_samples"]) == 16 and all(1.0 < x < 8.0 for x in d["clock_probe_ns_samples"] + d["clock_probe_ns_before"])
    assert len(d["launch_us_at_fastest_clock"]) == 16 and d["clock_probe_ns_fastest"] <= min(d["clock_probe_ns_samples"]) + 1e-3
    assert all(a_ <= b_ + 1e-9 for a_, b_ in zip(d["clock_probe_ns_samples"], d["clock_probe_ns_slowest_xcd"]))
    for mode in ("ab_host_restore", "ab_device_restore"):
        assert len(d[mode]["launch_us_samples"]) == 8 and len(d[mode]["clock_probe_ns_samples"]) == 8
        assert d[mode]["launch_us_max"] < 1.6 * d["launch_us_min"] and d[mode]["launch_us_min"] > 0.7 * d["launch_us_min"], d[mode]
    cf = d["config"]      # the flat copies
    assert cf["rccl_world_size"] == 1 and cf["value_median"] == d["value_median"] and cf["launch_us_median"] == d["launch_us_median"]
    assert cf["actor_f32_us_per_step"] == d["config3_actor_f32"]["us_per_step"] and cf["actor_f16x3_us_per_step"] == d["config3_actor_f16x3"]["us_per_step"]
    assert cf["push_us_per_step"] == d["config4_push"]["us_per_step"] and cf["step_api_us"] == d["step_api"]["avg_launch_us"]
    assert all(not isinstance(cf[k], (dict, list)) for k in ("rccl_backend", "ab_device_restore_launch_us_median", "ab_host_restore_launch_us_median",
                                                             "clock_probe_ns_min", "cpu_env_steps_per_s", "large_batch_env_steps_per_s"))
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert cb["threads_1"]["cores"] == 1 and cb["threads_1"]["value"] > 5e4 and cb["value"] >= 0.8 * cb["threads_1"]["value"]
    assert cb["cores"] <= cb["host"]["affinity_cpus"]
    assert 0.0 < pf["limit_step_rate"] < 0.5 and 0.0 <= pf["low_flange_step_rate"] < 0.1 and pf["env_steps_counted"] >= 65536 * 500
    assert 0.0 <= pf["cap_step_rate"] < 1e-3 and 0.0 < pf["illcond_step_rate"] < 0.01        # the four-term fence of config 2
    assert d["value"] > 1e9 and d["nonfinite_states"] == 0
    assert d["value_kernel"] >= d["value"] and d["value_kernel"] == pytest.approx(65536 * 20 / (ro["avg_launch_us"] * 1e-6), rel=1e-6)
    # the other single-GPU BASELINE configs, timed by the same command: config 3 (fused actor, both variants), config 4 (push)
    for key, pol, peak in (("config3_actor_f32", "actor", 157.3), ("config3_actor_f16x3", "actor_f16x3", 2500.0)):
        c3 = d[key]
        assert "error" not in c3, c3
        assert c3["envs"] == 65536 and c3["policy"] == pol and c3["steps"] == 400 and c3["value_kernel"] > 5e8
        assert c3["roofline"]["traffic"] is not None and 0.9 < c3["roofline"]["traffic"] / c3["roofline"]["algo_bytes_per_launch"] < 1.2
        assert c3["roofline_mfma"]["peak"] == peak and 0.2 < c3["roofline_mfma"]["frac"] < 1.0
        assert c3["roofline"]["traffic_key"] == "reach_rollout<f64,kuka>|policy=%s|T=100|N=65536" % pol
    c4 = d["config4_push"]
    assert "error" not in c4, c4
    # perf sanity bounds sit within ~25 % of the measured values: push 3.6-3.8e9, large_batch 11.9-12.3e9 / 0.63, DATD3 0.62e9
    assert c4["envs"] == 32768 and c4["task"] == "push" and c4["value_kernel"] > 2.7e9 and c4["kernel"] == "push_rollout<f64,kuka>"
    assert c4["roofline"]["traffic_key"] == "push_rollout<f64,kuka>|policy=external|T=100|N=32768" and 0.1 < c4["roofline"]["valu"]["frac"] < 1.0
    assert c4["roofline"]["traffic"] is not None and 0.9 < c4["roofline"]["traffic"] / c4["roofline"]["algo_bytes_per_launch"] < 1.2
    f4 = c4["parity_fence"]
    assert 0.05 < f4["limit_step_rate"] < 0.4 and 0.3 < f4["low_flange_step_rate"] < 0.7 and f4["cap_step_rate"] < 1e-3 and 1e-3 < f4["illcond_step_rate"] < 0.02
    assert d["step_api"]["value"] > 5e8
    lb = d["large_batch"]             # 1 048 576 envs on the one GPU: the two-waves-per-SIMD form of the rollout kernel
    assert lb["envs"] == 1048576 and lb["value"] > 9e9 and lb["valu"]["frac"] > 0.47      # (measured 11.9-12.3e9 / 0.63)
    dd = d["datd3_fused"]             # DATD3_MLP.take_action folded into the rollout kernel (beyond the configs)
    assert "error" not in dd, dd
    assert dd["policy"] == "datd3" and dd["envs"] == 65536 and dd["value_kernel"] > 0.5e9 and 0.2 < dd["roofline_mfma"]["frac"] < 1.0
    assert cf["datd3_us_per_step"] == dd["us_per_step"]
    da = d["daddpg_fused"]            # the reference's default agent (config.py:33): two actors, ONE critic valuing both proposals
    assert "error" not in da, da
    assert da["policy"] == "daddpg" and da["envs"] == 65536 and da["value_kernel"] > 0.5e9 and 0.2 < da["roofline_mfma"]["frac"] < 1.0
    assert cf["daddpg_us_per_step"] == da["us_per_step"] <= 1.02 * dd["us_per_step"]      # three staged nets: never slower than DATD3's four
    for leg_ in (dd, da):             # PMC traffic of the two-actor launch shapes
        assert leg_["roofline"]["traffic"] is not None and 0.9 < leg_["roofline"]["traffic"] / leg_["roofline"]["algo_bytes_per_launch"] < 1.2
    # the bracket's own numbers as flat keys (round 6)
    assert cf["procedure_version"] == 3 and cf["barrier_us"] < 50.0 and cf["collective_us"] == 0.0 and cf["collective_verified"] is None
    assert d["value_steps"] >= d["value"] >= 0.9 * d["value_steps"] and d["value_bracketed"] > 0


def test_bench_two_ranks_on_one_gpu_shard_the_trajectory(envs):
    """`bench.py --gpus 2` under torch.distributed.run on this one GPU (the ranks share it, so the logging collective falls
    back to gloo): the multi-rank control flow of BASELINE config 5 -- env-index sharding, barriers, max-over-ranks timing,
    at least one all-gather of episode returns inside the timed region -- and shard invariance: the two ranks' final joint
    states are the two halves of ONE 8 192-env handle driven with the concatenated action pools."""
    import hashlib
    n, K, W = 4096, 120, 5
    d = _run_bench(["--gpus", "2", "--steps", str(K), "--warmup", str(W), "--envs-per-gpu", str(n), "--state-digest",
                    "--prewarm-ms", "0"], nproc=2)
    assert d["n_gpus"] == 2 and d["config"]["total_envs"] == 2 * n and d["config"]["gathers_in_timed_region"] >= 1
    assert "gloo" in d["config"]["parallelism"] and d["scaling"] == "weak"
    assert abs(d["value"] - 2 * n * K / (d["ms_per_step"] * K / 1e3)) / d["value"] < 1e-6
    # the same trajectory on one handle: rank r draws its action pool from Generator(1000 + r), bench.py main()
    S = 1000
    pools = []
    for r in range(2):
        gen = torch.Generator(device=DEV); gen.manual_seed(1000 + r)
        pools.append((torch.randn((S, n, 3), device=DEV, generator=gen) * 0.686).clamp_(-0.7, 0.7))
    pool = torch.cat(pools, dim=1).contiguous()
    e = _mk(envs, 2 * n, seed=0)
    e.reset()
    e.rollout(W, pool[:W].contiguous())
    t = W
    while t < W + K:
        r = min(100, W + K - t)
        e.rollout(r, pool[t:t + r].contiguous())
        t += r
    q = _np(e.get_state()["q"])
    want = [hashlib.sha256(q[k * n:(k + 1) * n].tobytes()).hexdigest() for k in range(2)]
    assert d["config"]["state_digest"] == want
    e.close()


def test_bench_driver_shape_with_gathers_every_region():
    """`--gpus 2 --steps 20 --warmup 5` (the driver's arguments at N = 2): the 20-step region is shorter than --gather-every,
    and still carries one all-gather."""
    d = _run_bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--envs-per-gpu", "8192", "--prewarm-ms", "0"], nproc=2)
    assert d["n_gpus"] == 2 and d["config"]["total_envs"] == 16384 and d["config"]["gathers_in_timed_region"] == 1
    # a kernel-time figure beside the wall-clock one, and the host-side costs of the bracket (barrier, gather wait) stated
    assert d["value_kernel"] >= d["value"] > 0 and len(d["config"]["per_rank"]["kernel_ms"]) == 2
    assert d["value_kernel"] == pytest.approx(16384 * 20 / (max(d["config"]["per_rank"]["kernel_ms"]) * 1e-3), rel=1e-6)
    for k in ("barrier", "shm_barrier", "gather_issue", "gather_wait", "enqueue", "wait_for_gpu", "collective"):
        assert k in d["config"]["host_us"], k
    # round 6: `value` runs on synchronise + shared-memory barrier | K steps | launch-stream synchronise + shared-memory barrier (the
    # same code at N = 1); the rank-local clock and rounds 1-5's bracket (collective + RCCL barrier inside) are separate keys
    assert d["value_steps"] >= d["value"] >= d["value_bracketed"] and len(d["config"]["per_rank"]["wall_steps_ms"]) == 2
    assert d["config"]["collective_verified"] is True and "shared-memory barrier" in d["config"]["bracket"]
    assert max(d["config"]["per_rank"]["wall_ms"]) <= max(d["config"]["per_rank"]["wall_bracketed_ms"])
    assert max(d["config"]["per_rank"]["wall_steps_ms"]) <= max(d["config"]["per_rank"]["wall_ms"])
    assert d["config"]["rccl_ranks_seen"] == {"world_size": 2, "backend": "gloo"}


def test_bench_plain_python_launches_its_own_ranks():
    """`python3 bench.py --gpus 2 ...` with NO launcher around it and no RANK / WORLD_SIZE in the environment -- the way the round
    driver types the command: bench.py starts the two ranks itself (torch.distributed.run, a free port on
    127.0.0.1), relays rank 0's ONE line and the job's exit code.  The scalars of the multi-rank record sit in `config` as flat
    keys (a record that keeps only flat values loses nothing)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
                        "--envs-per-gpu", "8192"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{\"metric\"")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "weak"
    assert d["config"]["rccl_world_size"] == 2 and d["config"]["total_envs"] == 16384 and d["config"]["gathers_in_timed_region"] == 1
    assert d["config"]["rccl_backend"] == ("nccl" if torch.cuda.device_count() >= 2 else "gloo")
    assert abs(d["value"] - 16384 * 20 / (d["ms_per_step"] * 20 / 1e3)) / d["value"] < 1e-6
    # a failing job's exit code comes through as well (an unknown flag makes every rank exit with 2)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--no-such-flag"], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{\"metric\"")]
    # and WORLD_SIZE that contradicts --gpus is refused instead of silently running another job
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=600,
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_bench_eight_ranks_on_one_gpu_config5_control_flow(envs):
    """BASELINE configs[4] -- rl_reach_env sharded across 8 ranks, an all-gather of episode returns -- at world = 8 on this ONE
    GPU (the ranks share it, the logging collective falls back to gloo; on an 8-GPU node the same command runs one rank per GPU
    over RCCL): env_id_offset up to 7 x N, the 8-way gather inside the timed region, max-over-ranks timing.  The eight shard
    digests equal the eight slices of ONE 8 x N-env handle, and the gathered vector of episode returns equals that handle's
    armenv_episode_stats."""
    import hashlib
    n, K, W, world = 8192, 700, 5, 8          # 700 steps: every env finishes its first 501-step episode, so the returns are live
    d = _run_bench(["--gpus", str(world), "--steps", str(K), "--warmup", str(W), "--envs-per-gpu", str(n), "--state-digest",
                    "--prewarm-ms", "0", "--gather-every", "100"], nproc=world, timeout=1200)
    assert d["n_gpus"] == world and d["config"]["total_envs"] == world * n and d["config"]["gathers_in_timed_region"] >= 7
    assert d["config"]["rccl_ranks_seen"] == {"world_size": world, "backend": "gloo"} and "gloo" in d["config"]["parallelism"]
    assert len(d["config"]["per_rank"]["kernel_ms"]) == world and d["value_steps"] >= d["value"] > 0
    S = 1000
    pools = []
    for r in range(world):
        gen = torch.Generator(device=DEV); gen.manual_seed(1000 + r)
        pools.append((torch.randn((S, n, 3), device=DEV, generator=gen) * 0.686).clamp_(-0.7, 0.7))
    e = _mk(envs, world * n, seed=0)
    e.reset()
    t = 0
    for r in [W] + [100] * (K // 100):
        a = torch.cat([p_[t:t + r] for p_ in pools], dim=1).contiguous()
        e.rollout(r, a)
        t += r
    q = _np(e.get_state()["q"])
    want = [hashlib.sha256(q[k * n:(k + 1) * n].tobytes()).hexdigest() for k in range(world)]
    assert d["config"]["state_digest"] == want
    ret = _np(e.episode_stats()[0]).astype(np.float32)
    assert d["config"]["gathered_returns_sha256"] == hashlib.sha256(ret.tobytes()).hexdigest()
    assert abs(d["config"]["gathered_returns_mean"] - float(ret.astype(np.float64).mean())) < 1e-6 and ret.min() < 0.0
    e.close()
