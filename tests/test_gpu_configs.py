"""BASELINE.json configs inside the suite, at their own sizes:
  configs[0]  rl_reach_env.py single env, TD3 from algo/TD3, 1k steps (plumbing)   -> test_config0_*
  configs[2]  rl_reach_env 65 536 envs + TD3 actor forward fused into the step kernel -> test_config2_*
(configs[1] and [3] at full size: tests/test_gpu_parity.py::test_benchmarked_launch_shape_free_running_vs_oracle_f64[65536],
tests/test_gpu_fence.py::test_push_config4_free_running_vs_oracle.)"""
import random

import numpy as np
import pytest
import torch

from conftest import golden_npz

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def envs():
    from armenv import envs
    return envs


def _np(t):
    return t.detach().cpu().numpy()


def test_config0_single_env_td3_1000_steps(envs, O, kuka):
    """BASELINE configs[0]: the loop of /root/reference/main.py:100-138 -- reset, (take_action -> exploration noise -> step ->
    store_step) until done, add_trajectory, and opt.n_train TD3 updates on HER batches once the store holds
    opt.minimal_episodes episodes -- for 1 000 steps of the N=1 drop-in ``envs.RLReachEnv`` with the build's TD3 / Trajectory /
    TrajectoryStore counterparts.  Episodes are 101 steps long here (opt.max_steps_one_episode = 100) so that the training
    branch runs inside the 1 000 steps.  The N=1 trajectory is checked against the oracle fed the same actions: f32
    observations to 1e-6, identical done / success; the reward -- a Python float computed in f64, as the reference returns it
    (/root/reference/envs/rl_reach_env.py:300-319) -- equals the oracle's reward of the env's own joint state to 1e-12 and the
    free-running oracle's to 2e-8."""
    from armenv import opt
    from armenv.replay import Trajectory, TrajectoryStore
    from armenv.td3 import TD3
    saved = opt.max_steps_one_episode
    opt.max_steps_one_episode = 100
    try:
        env = envs.RLReachEnv(is_render=False, is_good_view=False)                   # main.py:83 (the ctor resets once, :125)
        state_dim, action_dim = env.observation_space.shape[0], env.action_space.shape[0]
        action_bound = float(env.action_space.high[0]) + 0.3                         # main.py:85-87
        assert (state_dim, action_dim) == (6, 3) and abs(action_bound - 0.7) < 1e-7
        random.seed(opt.random_seed); np.random.seed(opt.random_seed); torch.manual_seed(opt.random_seed)   # main.py:89-91
        store = TrajectoryStore(device=DEV, seed=0, capacity_steps=4096)             # main.py:93
        agent = TD3(state_dim, action_dim, action_bound, device=DEV)                 # main.py:95
        cfg = O.default_config(); cfg.max_steps = 100
        st = O.ReachState(1)
        steps = episodes = updates = successes = 0
        worst_obs = worst_rew = worst_rew_state = 0.0
        w0 = agent.actor.fc1.weight.detach().clone()
        while steps < 1000:
            state = env.reset()                                                      # main.py:108
            assert state.dtype == np.float32 and state.shape == (6,)
            obs_r = O.reach_reset_with_goal(kuka, cfg, st, state[3:].reshape(1, 3))
            assert np.abs(state - obs_r[0]).max() < 1e-6
            traj = Trajectory(state)                                                 # main.py:109
            done, ep_ret = False, 0.0
            while not done:
                action = agent.take_action(state)                                    # main.py:114
                action = (action + np.random.normal(0, action_bound * opt.gamma, size=action_dim)).clip(-action_bound, action_bound)
                state, reward, done, is_success = env.step(action)                   # main.py:124
                o_r, r_r, d_r, s_r, _ = O.reach_step(kuka, cfg, st, action.astype(np.float32).reshape(1, 3))
                assert isinstance(reward, float) and isinstance(done, bool) and isinstance(is_success, bool)
                assert done == bool(d_r[0]) and is_success == bool(s_r[0]), steps
                worst_obs = max(worst_obs, float(np.abs(state - o_r[0]).max()))
                worst_rew = max(worst_rew, abs(reward - float(r_r[0])))
                # the reward arithmetic itself (:271-309) on the env's own f64 joint state: the oracle's FK and outcome, 1e-12
                q_env = _np(env._eng.get_state()["q"])
                p_or, _ = O.fk(kuka, q_env)
                d_or = float(np.sqrt(np.sum((p_or[0] - state[3:].astype(np.float64)) ** 2)))
                r_or, d_flag, s_flag = O.reach_outcome(cfg, d_or, env.step_counter)
                assert d_flag == done and s_flag == is_success
                worst_rew_state = max(worst_rew_state, abs(reward - r_or))
                if is_success:
                    assert reward == 0 and done                                      # main.py:125 compares reward == 0
                    successes += 1
                ep_ret += reward
                traj.store_step(action, state, reward, done)                         # main.py:128
                steps += 1
            store.add_trajectory(traj)                                               # main.py:129
            episodes += 1
            assert env.step_counter == traj.length <= 101
            if store.size() >= opt.minimal_episodes:                                 # main.py:135-138
                for _ in range(opt.n_train):
                    batch = store.sample(opt.batch_size, use_her=True, her_ratio=opt.her_ratio)
                    loss = agent.train(batch)
                    updates += 1
                assert bool(torch.isfinite(loss))
        assert store.size() == episodes >= 9 and updates >= 5 * opt.n_train
        assert worst_obs < 1e-6, worst_obs
        assert worst_rew_state < 1e-12, worst_rew_state      # f64 all the way (an f32 reward buffer would show 1e-7 here); the N=1
                                                             # classes use the generic FK path: the URDF's own rpy = 1.57079632679
        assert worst_rew < 2e-8, worst_rew                    # free-running against the oracle's own trajectory (101-step episodes)
        assert not torch.equal(w0, agent.actor.fc1.weight.detach())                  # the delayed actor update ran
        env.close()
    finally:
        opt.max_steps_one_episode = saved


@pytest.mark.parametrize("kind", ["push", "pick"])
def test_n1_cube_envs_return_the_f64_reward(envs, O, kuka, kind):
    """RLPushEnv / RLPickEnv N=1 drop-ins: the shaped reward -100 * (d_now - d_last) as a Python float in f64
    (/root/reference/envs/rl_push_env.py:388-397,427) IS the kernel's -- the step's f64 diagnostics (armenv_step diag_dev), bit for
    bit, d_last carried in the engine's state -- against the f64 oracle fed the same actions and against the reference's own numpy
    expression on the returned observation (same value up to the rounding of two f64 distances; the kernel's sum of squares is
    FMA-contracted).  The host classes hold no reward arithmetic."""
    import inspect
    from armenv.envs import rl_pick_env, rl_push_env
    for mod in (rl_push_env, rl_pick_env):
        src = inspect.getsource(mod.RLPushEnv.step if mod is rl_push_env else mod.RLPickEnv.step)
        code = "\n".join(l.split("#")[0] for l in src.split('"""')[2].splitlines())
        for token in ("linalg", "_d_last", "* 100", "*100", "0.01", "1e-5", "- test", "-test"):
            assert token not in code, (mod.__name__, token)
    random.seed(3); np.random.seed(3)
    Env = envs.RLPushEnv if kind == "push" else envs.RLPickEnv
    env = Env(is_render=False, is_good_view=False)
    state = env.reset()
    cfg = O.default_config(kind)
    st = (O.PushState if kind == "push" else O.PickState)(1)
    reset_g, stepf = (O.push_reset_with_goal, O.push_step) if kind == "push" else (O.pick_reset_with_goal, O.pick_step)
    reset_g(kuka, cfg, st, state[3:9].reshape(1, 6))
    # the cube's height is the engine's on both sides (one step into its fall); the f64 placement in the plane (reset_with_goal takes f32)
    assert abs(st.aux[0, 2] - state[5]) < 1e-15 and abs(state[5] - (0.01 - 10.0 / 240.0 ** 2)) < 1e-15
    st.aux[0, 0:2] = state[3:5]; st.aux[0, 3:6] = state[6:9]
    st.aux[0, 6] = np.linalg.norm(st.aux[0, 0:3] - st.aux[0, 3:6])
    worst = 0.0
    moved = 0
    d_last = float(np.linalg.norm(state[3:6] - state[6:9], axis=-1))     # np.linalg.norm(..., axis=-1) as the reference calls it (:388)
    for t in range(60):
        # steer the tool through the cube at table height so that the shaped reward is not just the idle -1
        eef, cube = state[:3].astype(np.float64), state[3:6]
        tip = eef - np.array([0.0, 0.0, 0.257 if kind == "pick" else 0.0])
        want = np.array([cube[0], cube[1], 0.0]) + np.array([0.0, 0.0, 0.02]) + 0.05 * np.sign(cube - tip) * np.array([1, 1, 0])
        action = np.clip((want - tip) / 0.08, -0.5, 0.5)
        state, reward, done, info = env.step(action)
        o_r, r_r, d_r, s_r, _ = stepf(kuka, cfg, st, action.astype(np.float32).reshape(1, 3))
        assert done == bool(d_r[0])
        worst = max(worst, abs(float(reward) - float(r_r[0])))
        # the kernel's own f64 reward, bit for bit, and the engine's carried d_last is the distance it was computed from
        assert isinstance(reward, float) and reward == float(env._eng.diag[0, 3].item())
        aux = env._eng.get_state()["aux"][0].cpu().numpy()
        assert abs(aux[6] - np.linalg.norm(aux[0:3] - aux[3:6])) < 1e-15
        # the reference computes the reward from the observation it returns (:388-397)
        d_cur = float(np.linalg.norm(state[3:6] - state[6:9], axis=-1))
        test = d_cur - d_last
        d_last = d_cur
        if not done:
            assert abs(reward - -(0.01 if abs(test) < 1e-5 else test) * 100) < 1e-12, (t, reward)
        moved += int(float(reward) != -1.0)
        if done:
            break
    assert worst < 1e-6, worst               # free-running against the oracle for 60 steps (the arm differs by ~1e-10 by then)
    assert moved >= 3, moved
    env.close()


def _golden_actor_sd():
    g = golden_npz("td3_actor_seed0.npz")
    return {k: torch.from_numpy(g[k.replace(".", "_")]) for k in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc3.weight", "fc3.bias")}


@pytest.mark.parametrize("kind", ["actor", "actor_f16x3"])
def test_config2_fused_actor_65536_envs_vs_oracle(envs, O, kuka, kind):
    """BASELINE configs[2] at its own size: 65 536 reach envs, the reference agent's actor (golden weights produced by importing
    algo.TD3) folded into the rollout kernel with run()'s exploration noise, 2 x armenv_rollout(100).  Checked on a strided
    sample of 2 048 envs (global ids 0, 32, 64, ...: the noise and the goals are keyed by global env id, so the oracle
    reproduces exactly those envs): actions within 2e-5 of oracle actor + noise, observations within 1e-5 while the oracle is
    teacher-forced with the engine's own actions (so that the MFMA actor's 1e-6 does not compound through the policy
    loop), identical done / success flags.  Matches /root/reference/algo/TD3/TD3_mlp.py:82-97, main.py:114-117."""
    n, T, stride = 65536, 100, 32
    sd = _golden_actor_sd()
    sd_np = {k: v.numpy() for k, v in sd.items()}
    e = envs.BatchedReachEnv(n, device=DEV, seed=4)
    e.set_policy(kind, action_bound=0.7, noise_sigma=0.7 * 0.98, noise_clip=0.7, actor_state_dict=sd)
    obs0 = _np(e.reset()).copy()
    ids = np.arange(0, n, stride)
    m = ids.size
    cfg = O.default_config()
    for gid in ids[:64]:          # goals of the first sampled envs reproduce bit for bit (Philox keyed by global id)
        s1 = O.ReachState(1)
        o1 = O.reach_reset(kuka, cfg, s1, seed=4, env_id0=int(gid))
        assert np.array_equal(o1[0, 3:], obs0[gid, 3:])
    # batch oracle state for the whole sample, initialised from the engine's own reset (goals verified above)
    st = O.ReachState(m)
    st.q[:] = np.array(O.INIT_Q); st.goal[:] = obs0[ids, 3:]; st.episode[:] = 1
    obs_prev = obs0[ids].copy()
    worst_a = worst_o = 0.0
    flags = 0
    resets = 0
    for launch in range(2):
        out = e.rollout(T, None, want_actions=True, want_terminal_obs=True)
        acts, obs, term, done, succ = (_np(out[k]) for k in ("actions", "obs", "terminal_obs", "done", "success"))
        for t in range(T):
            # what the fused policy must have produced from the observation it saw: actor(obs) + sigma * noise, clipped
            mu = O.actor_forward(sd_np, obs_prev, 0.7)
            nz = O.policy_noise_ids(4, ids, st.episode, st.step)
            want = np.clip(mu + np.float32(0.7 * 0.98) * nz, -np.float32(0.7), np.float32(0.7))
            a = acts[t][ids]
            worst_a = max(worst_a, float(np.abs(a - want).max()))
            # teacher-forced on the actions: the oracle env takes the engine's action
            o_r, r_r, d_r, s_r, iters = O.reach_step(kuka, cfg, st, a)
            worst_o = max(worst_o, float(np.abs(term[t][ids] - o_r).max()))
            flags += int((done[t][ids] != d_r.astype(bool)).sum() + (succ[t][ids] != s_r.astype(bool)).sum())
            # auto-reset on the oracle side for finished envs: next goal from the engine's obs (verified against Philox above)
            fin = d_r.astype(bool)
            if fin.any():
                resets += int(fin.sum())
                st.q[fin] = np.array(O.INIT_Q); st.step[fin] = 0; st.episode[fin] += 1; st.ep_return[fin] = 0
                st.goal[fin] = obs[t][ids][fin, 3:]
            obs_prev = obs[t][ids].copy()            # what the fused policy sees next (a finished env: its new episode's first obs)
    assert worst_a < 2e-5, worst_a
    assert worst_o < 1e-5, worst_o
    assert flags == 0
    c = e.counters()
    assert c["env_steps"] == n * 2 * T and c["nonfinite"] == 0
    e.close()


def test_n1_env_reproduces_the_reference_runs_first_episodes(envs):
    """BASELINE configs[0] against the reference's own numbers: the first five episodes of the recorded train_reach_with_TD3 run
    (tests/golden/visdata_reach_td3.json; real PyBullet, opt.random_seed = 0; see tests/reference_run.py) through the N=1 drop-in
    `envs.RLReachEnv` -- every env step an armenv_step launch of the HIP engine -- with the protocol of main.py:176-204: the
    untrained TD3 actor of torch.manual_seed(0) (golden G3), a + N(0, 0.98) unclipped from np.random.seed(0), goals from
    random.seed(0).  Episode lengths 64 (success) / 501 x 4 and the five returns to 1e-3 (measured 6e-4 = 4e-7 relative)."""
    import reference_run as R
    from armenv.td3 import TD3
    fx = R.fixture_returns()
    env = envs.RLReachEnv(is_render=False, is_good_view=False)                     # main.py:171 (its reset precedes the seeding)
    random.seed(0); np.random.seed(0); torch.manual_seed(0)                        # main.py:176-178
    agent = TD3(6, 3, 0.7, device=DEV)
    agent.actor.load_state_dict({k: torch.from_numpy(v) for k, v in R.actor_weights().items()})
    got = []
    for ep in range(5):
        state = env.reset(); done, ret, n = False, 0.0, 0
        while not done:
            action = agent.take_action(state) + np.random.normal(0, 1 * 0.98, size=3)          # main.py:199-200
            state, reward, done, is_success = env.step(action)
            ret += reward; n += 1
        got.append((ret, n, is_success))
    c = env._eng.counters()
    env.close()
    assert [n for _, n, _ in got] == [64, 501, 501, 501, 501] and [s for _, _, s in got] == [True, False, False, False, False]
    diffs = [abs(r - x) for (r, _, _), x in zip(got, fx)]
    assert max(diffs) < 1e-3, diffs
    assert c["limit_steps"] > 200 and c["low_flange_steps"] > 200 and c["cap_steps"] == 0


def test_n1_push_env_on_the_recorded_push_runs_first_episodes(envs):
    """The first five episodes of the reference's two recorded train_push_with_TD3 runs (tests/golden/visdata_push_td3.json, real
    PyBullet, seed 0, the same trajectories under two rewards: tests/reference_run.py) through the N=1 drop-in `envs.RLPushEnv` on
    the HIP engine.  All five run to the time limit; the arm touches the cube in episodes 1, 2, 3, 5 and not in 4 (the recorded runs'
    own pattern); episode 4 -- which depends on nothing but the cube's free fall, the placement stream, the draw counts and the
    reward arithmetic of rl_push_env.py:368-440 -- returns the SHIPPED reward's recorded -504.1221 to 2e-3 (eight falling steps above
    the 1e-5 threshold) and, re-scored with the earlier reward, the other run's -512.0719 to 2e-3; the touched episodes move the cube on
    149 / 192 / 90 / 43 steps against Bullet's 149 / 192 / 84 / 32 (the oracle's counts; the engine's may differ by a step or two: the
    trajectory is sensitive to every contact), return within 15 of the shipped reward's recorded values and end with the cube within
    4.7 cm of where Bullet left it."""
    import reference_run as R
    from armenv.td3 import TD3
    org, upd = R.push_fixture_returns("origin"), R.push_fixture_returns("updata")
    rec = R.push_recorded_observables(5)
    env = envs.RLPushEnv(is_render=False, is_good_view=False)                      # main.py:454
    random.seed(0); np.random.seed(0); torch.manual_seed(0)                        # main.py:459-461
    agent = TD3(9, 3, 0.4, device=DEV)
    agent.actor.load_state_dict({k: torch.from_numpy(v) for k, v in R.actor9_weights().items()})
    got = []
    for ep in range(5):
        state = env.reset(); done, ret, ret_o, n, moved, M = False, 0.0, 0.0, 0, 0, 0
        cube0 = state[3:6].copy()
        d_last = float(np.linalg.norm(state[3:6] - state[6:9]))
        while not done:
            action = agent.take_action(state) + np.random.normal(0, 0.4 * 0.98, size=3)        # main.py:481-484
            state, reward, done, info = env.step(action)
            moved += int(np.abs(state[3:5] - cube0[0:2]).max() > 0); cube0 = state[3:6].copy()
            d_cur = float(np.linalg.norm(state[3:6] - state[6:9]))
            M += int(abs(d_cur - d_last) >= 1e-5); d_last = d_cur
            ret += reward; n += 1
            ret_o += reward if (done or reward == 100) else -1.0
        d_f = float(np.linalg.norm(state[3:6].astype(np.float32) - state[6:9].astype(np.float32)))
        got.append(dict(ret=ret, ret_origin=ret_o, n=n, moved=moved, M=M, d_f=d_f))
    env.close()
    assert [g["n"] for g in got] == [501] * 5
    assert [g["moved"] > 20 for g in got] == [True, True, True, False, True] and [g["M"] > 8 for g in got] == [True, True, True, False, True]
    assert abs(got[3]["ret"] - upd[3]) < 2e-3 and abs(got[3]["ret_origin"] - org[3]) < 2e-3 and got[3]["M"] == 8, got[3]
    for k in (0, 1, 2, 4):
        assert abs(got[k]["M"] - rec[k][1]) <= 14 and abs(got[k]["ret"] - upd[k]) < 15.0, (k, got[k], rec[k], upd[k])
        assert abs(got[k]["d_f"] - rec[k][0]) < 0.047 and abs(got[k]["ret_origin"] - org[k]) < 2.3, (k, got[k], rec[k])


@pytest.mark.parametrize("precision", [64, 32])
def test_benchmarked_rollout_kernel_reproduces_the_reference_run(envs, O, precision):
    """The kernel bench.py times -- env_rollout_kernel<ReachLane<KukaChain, double, 0>, double, 0, 1>: the compile-time KUKA fast
    path, default build, external actions, one launch per episode -- against real PyBullet's numbers: the first five episodes of
    the reference's recorded run (tests/reference_run.py) in open-loop form (goals and the 2 068 actions as the oracle's replay of
    the run produces them), 64 lanes fed the same env.  Sum of the launch's f32 reward rows against the recorded returns: 3e-3
    (f64 engine; measured 6e-4 + the f32 rounding of 501 rewards), success exactly at step 64 of episode 1, every lane the same
    bits.  precision = 32: the f32 engine on the four 501-step episodes, within 1.0 of ~1 500 (its stated 1e-4-per-step class;
    episode 1's success test at 1 cm is a threshold an f32 trajectory may cross a step early or late)."""
    import reference_run as R
    fx = R.fixture_returns()
    rec = []
    out, _ = R.replay_on_oracle(O, 5, record=rec)
    n = 64
    e = envs.BatchedReachEnv(n, device=DEV, auto_reset=False, precision=precision)
    assert e.kernel_name == "reach_step<f%d,kuka>" % precision
    for ep, ((goal, acts), (_, length, succ)) in enumerate(zip(rec, out)):
        T = len(acts)
        assert T == length
        if precision == 32 and succ:
            continue
        e.reset(goal=torch.from_numpy(np.tile(goal, (n, 1))))
        a = torch.from_numpy(np.tile(acts.astype(np.float32)[:, None, :], (1, n, 1))).contiguous().to(DEV)
        o = e.rollout(T, a)
        ret = o["reward"].double().sum(0)
        assert bool((ret == ret[0]).all()) and bool((o["obs"] == o["obs"][:, :1]).all())          # 64 lanes, one trajectory
        tol = 3e-3 if precision == 64 else 1.0
        assert abs(float(ret[0]) - fx[ep]) < tol, (ep, float(ret[0]), fx[ep])
        done = o["done"][:, 0].cpu().numpy()
        assert done[-1] and not done[:-1].any() and bool(o["success"][-1, 0]) == succ
    e.close()
