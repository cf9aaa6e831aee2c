"""GPU tests of the round-4 switches and diagnostics of the C ABI (ABI 4):
  ArmEnvConfig.ik_tip_offset    the point of link 7 at which calculateInverseKinematics takes its position error and linear
                                Jacobian (/root/reference/envs/rl_reach_env.py:244-250): the last un-switched unknown of the IK
                                restatement -- URDF link frame (default) or Bullet's inertial frame, 2 cm along the tool axis;
  armenv_step / armenv_rollout diag_dev   the step's f64 end-effector position and reward (the reference returns Python floats).
All against the CPU oracle, teacher-forced (every step starts from the oracle's state)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
INERTIAL = (0.0, 0.0, 0.02)          # KUKA link-7 inertial origin
SKEW = (0.01, -0.02, 0.03)           # off joint 7's axis: the last joint's Jacobian column is no longer zero


@pytest.fixture(scope="module")
def envs():
    from armenv import envs
    return envs


def _np(t):
    return t.detach().cpu().numpy()


def _noise(rng, n, task):
    if task == "reach":
        return np.clip(rng.normal(0.0, 0.686, (n, 3)), -0.7, 0.7).astype(np.float32)
    return rng.normal(0.0, 0.392, (n, 3)).astype(np.float32)


@pytest.mark.parametrize("off", [INERTIAL, SKEW])
@pytest.mark.parametrize("task,fk_path,precision", [("reach", 0, 64), ("reach", 1, 64), ("push", 0, 64), ("pick", 0, 64), ("reach", 0, 32)])
def test_ik_tip_offset_teacher_forced(envs, O, kuka, task, fk_path, precision, off):
    """The switch in kernel (MODE 2 build of the lanes, armenv_kin.h) and oracle (orc_ik_ex), both chain paths, the three
    tasks: q within 1e-6 rad, the same IK update counts, the same observations; and the switch does something (the joints
    differ from the offset-0 engine's by far more than the tolerance)."""
    n = 1024 + 5
    rng = np.random.default_rng(77)
    Env = dict(reach=envs.BatchedReachEnv, push=envs.BatchedPushEnv, pick=envs.BatchedPickEnv)[task]
    State, reset, stepf = dict(reach=(O.ReachState, O.reach_reset, O.reach_step), push=(O.PushState, O.push_reset, O.push_step),
                               pick=(O.PickState, O.pick_reset, O.pick_step))[task]
    cfg = O.default_config(task); cfg.ik_tip_offset[:] = list(off)
    e = Env(n, device=DEV, seed=3, auto_reset=False, fk_path=fk_path, precision=precision, ik_tip_offset=list(off))
    base = Env(n, device=DEV, seed=3, auto_reset=False, fk_path=fk_path, precision=precision)
    assert e.cfg.ik_tip_offset[2] == off[2]
    st = State(n)
    reset(kuka, cfg, st, seed=3)
    e.reset(); base.reset()
    tol = 1e-6 if precision == 64 else 1e-4
    flips = 0
    moved = 0.0
    for t in range(12):
        a = _noise(rng, n, task)
        kw = dict(q=st.q, step=st.step, ep_return=st.ep_return)
        kw.update(dict(goal=st.goal) if task == "reach" else dict(aux=st.aux))
        e.set_state(**kw); base.set_state(**kw)
        obs, rew, done, succ = e.step(torch.from_numpy(a).to(DEV), want_ik_updates=True)
        obs = _np(obs).copy(); upd = _np(e.ik_updates).astype(np.int32)
        base.step(torch.from_numpy(a).to(DEV))
        mp = np.zeros(n)
        obs_r, rew_r, done_r, succ_r, iters = stepf(kuka, cfg, st, a, minpiv=mp)
        dq = np.abs(_np(e.get_state()["q"]) - st.q).max(1)
        ok = dq < tol
        # a call that ran to the iteration cap or through an ill-conditioned system is outside every parity statement (the fence)
        flips += int((~ok & (iters < cfg.ik_max_iters) & (mp >= cfg.fence_pivot)).sum())
        if precision == 64:
            assert np.array_equal(upd[ok], iters[ok]), t
            assert np.abs(obs - obs_r)[ok].max() <= 2e-7, t
        else:
            assert np.abs(obs - obs_r)[ok].max() < 1e-4, t
        moved = max(moved, float(np.abs(_np(base.get_state()["q"]) - st.q).max()))
    assert flips <= (2 if precision == 64 else 0.01 * 12 * n), flips
    assert moved > 1e-3, moved
    # the stand-alone entry point (armenv_ik) follows the same switch
    if precision == 64:
        q0 = st.q.copy()
        p0, _ = O.fk(kuka, q0)
        tgt = p0 + rng.normal(0, 0.01, (n, 3))
        q_g, it_g = e.ik(torch.from_numpy(q0), torch.from_numpy(tgt))
        q_r, it_r = O.ik(kuka, cfg, q0, tgt)
        diag, _ = O.ik_diag(kuka, cfg, q0, tgt)                  # pick's arms visit near-singular poses: outside the fence only
        same = (_np(it_g) == it_r) & (it_r < cfg.ik_max_iters) & (diag[:, 0] >= cfg.fence_pivot)
        assert same.mean() > 0.9 and np.abs(_np(q_g) - q_r)[same].max() < 1e-6
    e.close(); base.close()


@pytest.mark.parametrize("task", ["reach", "push", "pick"])
def test_vanishing_tip_offset_reproduces_the_default_kernels_bitwise(envs, task):
    """The default kernels leave the last joint's linear Jacobian column out (its lever arm is x - x = +0, armenv_kin.h
    dls_update); the MODE 2 build keeps all seven columns with the lever arms taken from p + W * offset.  With an offset too
    small to change p (1e-300) the two builds must produce the same bits -- outputs, state, counters: the claim "the same
    bits for every finite state" behind the shortcut, checked on a free-running rollout with in-place resets."""
    n, T = 2048 + 9, 60
    Env = dict(reach=envs.BatchedReachEnv, push=envs.BatchedPushEnv, pick=envs.BatchedPickEnv)[task]
    gen = torch.Generator(device=DEV); gen.manual_seed(8)
    sig = 0.686 if task == "reach" else 0.392
    acts = (torch.randn((T, n, 3), device=DEV, generator=gen) * sig).clamp_(-0.7, 0.7).contiguous()
    a = Env(n, device=DEV, seed=4, max_steps=25, fence_counters=1)
    b = Env(n, device=DEV, seed=4, max_steps=25, ik_tip_offset=[0.0, 0.0, 1e-300])
    a.reset(); b.reset()
    oa = a.rollout(T, acts, want_ik_updates=True)
    ob = b.rollout(T, acts, want_ik_updates=True)
    for k in ("obs", "reward", "done", "success", "ik_updates"):
        assert torch.equal(oa[k], ob[k]), (task, k)
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), (task, k)
    ca, cb = a.counters(), b.counters()
    assert ca == cb and ca["episodes"] >= 2 * n
    a.close(); b.close()


@pytest.mark.parametrize("task", ["reach", "push"])
def test_step_diag_is_the_f64_view_of_the_step(envs, O, kuka, task):
    """armenv_step / armenv_rollout diag_dev: [eef xyz, reward] in f64 before the f32 stores -- rounds to the obs / reward
    buffers bit for bit; for reach the reward is -10 |eef - goal| evaluated in f64 (/root/reference/envs/rl_reach_env.py:281-309),
    to 1e-15 of numpy's; success <=> that distance < reach_dis, never otherwise."""
    n, T = 4096, 40
    Env = dict(reach=envs.BatchedReachEnv, push=envs.BatchedPushEnv)[task]
    rng = np.random.default_rng(3)
    e = Env(n, device=DEV, seed=5, auto_reset=False, fence_counters=2, reach_dis=0.05)
    e.reset()
    goal = _np(e.get_state()["goal"]).astype(np.float64) if task == "reach" else None
    hits = 0
    for t in range(T):
        a = torch.from_numpy(_noise(rng, n, task)).to(DEV)
        obs, rew, done, succ = e.step(a, want_diag=True)
        d = _np(e.diag)
        assert np.array_equal(d[:, :3].astype(np.float32), _np(obs)[:, :3])
        assert np.array_equal(d[:, 3].astype(np.float32), _np(rew))
        if task == "reach":
            dist = np.sqrt(((d[:, :3] - goal) ** 2).sum(1))
            s = _np(succ)
            assert np.array_equal(s, dist < 0.05) or np.abs(dist - 0.05)[s != (dist < 0.05)].max() < 1e-15
            assert np.abs(np.where(s, 0.0, -10.0 * dist) - d[:, 3]).max() < 1e-14
            hits += int(s.sum())
    assert task != "reach" or hits > 0
    gen = torch.Generator(device=DEV); gen.manual_seed(1)
    acts = (torch.randn((7, n, 3), device=DEV, generator=gen) * 0.3).contiguous()
    out = e.rollout(7, acts, want_diag=True)
    assert out["diag"].shape == (7, n, 4)
    assert torch.equal(out["diag"][..., :3].float(), out["obs"][..., :3]) and torch.equal(out["diag"][..., 3].float(), out["reward"])
    from armenv import ArmEnvError
    for fc in (0, 1):                   # the diagnostics belong to the fence_counters = 2 build of the kernels
        plain = Env(64, device=DEV, fence_counters=fc)
        plain.reset()
        with pytest.raises(ArmEnvError):
            plain.step(torch.zeros((64, 3), device=DEV), want_diag=True)
        plain.close()
    with pytest.raises(ArmEnvError):
        Env(64, device=DEV, fence_counters=3)
    e.close()


def test_n1_reach_env_reward_and_flags_come_from_the_same_numbers(envs):
    """RLReachEnv.step took done / success from the kernel's carried-trig frame and distance / reward from a second
    FK with re-derived trig.  Now one launch, one set of numbers: `distance < reach_dis` <=> is_success, reward = -10 distance
    (0 on success) to the last bit, observation = float32(robot_state)."""
    import random
    from armenv.config import opt
    env = envs.RLReachEnv()
    random.seed(3)
    old = opt.reach_dis
    try:
        opt.reach_dis = 0.08                       # successes within a short walk
        obs = env.reset()
        succ = 0
        for t in range(120):
            goal = obs[3:].astype(np.float64)
            a = np.clip((goal - obs[:3].astype(np.float64)) / 0.02, -0.7, 0.7) if t % 2 else np.random.default_rng(t).normal(0, 0.5, 3)
            obs, r, done, s = env.step(a)
            assert (env.distance < 0.08) == bool(s)
            assert r == (0.0 if s else -env.distance * 10) or abs(r + env.distance * 10) < 1e-15
            assert np.array_equal(obs[:3], np.asarray(env.robot_state, dtype=np.float32))
            if done:
                succ += int(s)
                obs = env.reset()
        assert succ > 0
    finally:
        opt.reach_dis = old
        env.close()


def test_create_rejects_what_the_bookkeeping_cannot_represent(envs):
    """the per-step IK update count is a u8 and the wave's trip maximum is folded over 8 bits -- a bookkeeping handle
    with ik_max_iters beyond 254 would saturate silently.  Rejected at create; the default build takes it."""
    from armenv import ArmEnvError
    with pytest.raises(ArmEnvError):
        envs.BatchedReachEnv(64, device=DEV, fence_counters=1, ik_max_iters=300)
    with pytest.raises(ArmEnvError):
        envs.BatchedReachEnv(64, device=DEV, ik_tip_offset=[0.0, 0.0, 0.02], ik_max_iters=300)
    with pytest.raises(ArmEnvError):
        envs.BatchedReachEnv(64, device=DEV, ik_tip_offset=[0.0, float("nan"), 0.0])
    e = envs.BatchedReachEnv(64, device=DEV, ik_max_iters=300)
    e.reset(); e.step(torch.zeros((64, 3), device=DEV)); e.close()
    # the tip offset lives in the bookkeeping build: no fused actor beside it
    t = envs.BatchedReachEnv(64, device=DEV, ik_tip_offset=[0.0, 0.0, 0.02])
    sd = {k: torch.zeros(s) for k, s in (("fc1.weight", (256, 6)), ("fc1.bias", (256,)), ("fc2.weight", (256, 256)), ("fc2.bias", (256,)),
                                        ("fc3.weight", (3, 256)), ("fc3.bias", (3,)))}
    with pytest.raises(ArmEnvError):
        t.set_policy("actor", actor_state_dict=sd)
    t.set_policy("random")
    t.reset(); t.rollout(5, None); t.close()


def test_add_trajectory_keeps_every_episodes_own_first_state():
    """TrajectoryStore.add_trajectory (rl_utils.py:112-113 for the single-env loops of main.py:108-129) stored an
    episode's first state only for the very first trajectory; every later one started, for the sampler, at the previous
    episode's terminal observation.  Three trajectories with distinct first states, a ring small enough to wrap, then samples
    of (episode e, step 0) without HER: states must be traj.states[0], next_states traj.states[1]; and the last step of the
    PREVIOUS episode still returns its own terminal next_state."""
    from armenv.replay import Trajectory, TrajectoryStore
    rng = np.random.default_rng(0)
    store = TrajectoryStore(device=DEV, seed=1, capacity_steps=16)
    trajs = []
    for e in range(4):
        L_ = 5 + e
        tr = Trajectory(rng.uniform(0.2, 0.5, 6).astype(np.float32) + 10.0 * (e + 1))     # unmistakable first states
        for t in range(L_):
            tr.store_step(rng.normal(0, 0.3, 3).astype(np.float32), rng.uniform(0.2, 0.5, 6).astype(np.float32) + 10.0 * (e + 1) + t + 1,
                          float(-t), t == L_ - 1)
        trajs.append(tr)
        store.add_trajectory(tr)
    assert store.size() >= 2                       # 5 + 6 + 7 + 8 = 26 steps through a 16-step ring: the oldest fell out
    eps = _np(store.chunk["episodes"][: store.size()])
    kept = trajs[-store.size():]                   # complete episodes still in the window, oldest first
    assert [int(x) for x in eps[:, 2]] == [t.length for t in kept]
    picks = []
    for e, tr in enumerate(kept):
        picks += [[e, 0, 0, 0], [e, tr.length - 1, 0, 0]]
    out = store.sample(len(picks), use_her=False, picks=np.asarray(picks, dtype=np.int32))
    st, nx, ac = _np(out["states"]), _np(out["next_states"]), _np(out["actions"])
    for e, tr in enumerate(kept):
        assert np.array_equal(st[2 * e], np.asarray(tr.states[0], dtype=np.float32)), e
        assert np.array_equal(nx[2 * e], np.asarray(tr.states[1], dtype=np.float32)), e
        assert np.array_equal(ac[2 * e], np.asarray(tr.actions[0], dtype=np.float32)), e
        assert np.array_equal(st[2 * e + 1], np.asarray(tr.states[-2], dtype=np.float32)), e
        assert np.array_equal(nx[2 * e + 1], np.asarray(tr.states[-1], dtype=np.float32)), e      # the episode's own terminal state


@pytest.mark.parametrize("task,parts", [("reach", 2), ("reach", 4), ("push", 2)])
def test_pipelined_env_equals_the_single_handle_bitwise(envs, task, parts):
    """armenv.envs.PipelinedEnv: the batch cut into `parts` handles on `parts` HIP streams (a part's launch
    t+1 runs under the other parts' launch-t tails; a closed-loop policy's kernels for part A under part B's step).  Same
    trajectory as ONE handle, bit for bit: outputs of every step, final state, counters -- with the open-loop step(), with the
    bare bound launches, and closed loop through a torch policy; across in-place resets (30-step episodes).
    Matches /root/reference/main.py:111-128."""
    n, T = 4096, 70
    Env = dict(reach=envs.BatchedReachEnv, push=envs.BatchedPushEnv)[task]
    gen = torch.Generator(device=DEV); gen.manual_seed(3)
    sig = 0.686 if task == "reach" else 0.392
    acts = (torch.randn((T, n, 3), device=DEV, generator=gen) * sig).clamp_(-0.7, 0.7).contiguous()
    one = Env(n, device=DEV, seed=11, max_steps=30)
    ref_obs = one.reset().clone()
    ref = []
    for t in range(T):
        o, r, d, s = one.step(acts[t])
        ref.append((o.clone(), r.clone(), d.clone(), s.clone()))
    st_one, c_one = one.get_state(), one.counters()
    # open loop through step()
    pe = envs.PipelinedEnv(Env, n, parts=parts, device=DEV, seed=11, max_steps=30)
    assert torch.equal(pe.reset(), ref_obs)
    for t in range(T):
        o, r, d, s = pe.step(acts[t])
        for x, y in zip((o, r, d, s), ref[t]):
            assert torch.equal(x, y), (task, parts, t)
    st = pe.get_state()
    for k in st_one:
        assert torch.equal(st[k], st_one[k]), k
    c = pe.counters()
    assert {k: c[k] for k in ("episodes", "successes", "env_steps", "ik_updates")} == {k: c_one[k] for k in ("episodes", "successes", "env_steps", "ik_updates")}
    pe.close()
    # the bare bound launches, no per-step stream traffic
    pe = envs.PipelinedEnv(Env, n, parts=parts, device=DEV, seed=11, max_steps=30)
    pe.reset()
    torch.cuda.synchronize()
    for f in pe.bind_steps(acts):
        f()
    pe.join()
    for x, y in zip((pe._obs, pe._reward), ref[-1][:2]):
        assert torch.equal(x, y)
    assert torch.equal(pe.get_state()["q"], st_one["q"])
    pe.close()
    # closed loop: a deterministic torch policy of the observation
    W = torch.linspace(-1.0, 1.0, 3 * one.obs_dim, device=DEV).reshape(one.obs_dim, 3)
    policy = lambda o: (torch.tanh(o @ W) * 0.5).contiguous()
    one.close()
    one = Env(n, device=DEV, seed=11, max_steps=30)          # a fresh handle: the goal stream is keyed by the env's reset count
    o = one.reset()
    for t in range(40):
        o, r, d, s = one.step(policy(o))
    want = (o.clone(), r.clone(), one.get_state()["q"].clone())
    pe = envs.PipelinedEnv(Env, n, parts=parts, device=DEV, seed=11, max_steps=30)
    pe.reset()
    o2, r2, _, _ = pe.run_closed_loop(policy, 40)
    assert torch.equal(o2, want[0]) and torch.equal(r2, want[1]) and torch.equal(pe.get_state()["q"], want[2])
    pe.close(); one.close()


def test_pipelined_get_state_with_steps_in_flight(envs):
    """PipelinedEnv.get_state / episode_stats rely on the per-part ordering of BatchedArmEnv._ordered alone (no join()).
    With a queue of un-joined launches on the parts' streams, the unjoined read equals the joined one -- and when the wrapped call
    raises, the caller's stream is still ordered behind the fixed stream (try / finally), so the next read is sound."""
    n, K = 8192, 40
    gen = torch.Generator(device=DEV); gen.manual_seed(5)
    acts = (torch.randn((K, n, 3), device=DEV, generator=gen) * 0.686).clamp_(-0.7, 0.7).contiguous()
    one = envs.BatchedReachEnv(n, device=DEV, seed=4, max_steps=25)
    one.reset()
    for t in range(K):
        one.step(acts[t])
    want, want_stats = one.get_state(), one.episode_stats()
    one.close()
    for attempt in range(3):
        pe = envs.PipelinedEnv(envs.BatchedReachEnv, n, parts=4, device=DEV, seed=4, max_steps=25)
        pe.reset()
        torch.cuda.synchronize()
        for f in pe.bind_steps(acts):
            f()                                   # 160 launches queued on four streams, nothing joined
        got = pe.get_state()                      # read while they are in flight
        stats = pe.episode_stats()
        junk = [torch.full_like(v, -3.0) for v in got.values()]     # allocator traffic on the caller's stream right behind the read
        del junk
        for k in want:
            assert torch.equal(got[k], want[k]), (attempt, k)
        for x, y in zip(stats, want_stats):
            assert torch.equal(x, y), attempt
        # a failing call inside _ordered() leaves the streams ordered
        e0 = pe.envs[0]
        with pytest.raises(Exception):
            with e0._ordered():
                raise RuntimeError("boom")
        assert torch.equal(pe.get_state()["q"], want["q"])
        pe.close()
