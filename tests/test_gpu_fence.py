"""GPU tests of the parity fence and of the cube tasks at their benchmarked workload.

The fence names the env steps on which a parity statement cannot be made:
  limit   the IK result lies outside the URDF joint limits       (Bullet's limit constraint pushes back in stepSimulation)
  flange  the step ends with the flange below z = 0.05            (arm-table contact)
  cap     the IK call ran to its 20-iteration cap                 (it does not converge: the update oscillates by 0.2-0.8 rad
                                                                   per iteration, twenty of them amplify last-bit differences
                                                                   to 1e-5 rad and more)
  cond    one of the call's damped systems was ill-conditioned    (an LDL^T pivot of J J^T + lambda I below fence_pivot = 1e-2:
                                                                   a near-singular pose -- stretched elbow at the edge of the
                                                                   arm's reach, aligned wrist -- where the solve amplifies
                                                                   rounding differences by ~1 / pivot)
The first two are about Bullet; the last two are about ANY pair of implementations of the reference's algorithm, the CPU
oracle and this engine included (the oracle's own primal and dual solve forms part ways there: tests/tools/fence_study.py).
So the free-running tests below follow every env of BASELINE config 4 (push, 32 768 envs) and of the pick task at the same
size under the reference's own exploration noise (main.py:484, unclipped N(0, 0.392)) for six 100-step launches and 501-step
episodes, and assert in two tiers:
  strict      100 % agreement with the oracle (observation 1e-4, flags, IK update counts) on every env-step whose env has had no
              capped or ill-conditioned IK call since its last reset (FenceBook);
  task space  on EVERY env-step whose env is still in the same episode as its oracle twin -- the excluded ones included -- the
              observation stays within a loose bound, all but a stated share within 1e-4, and wherever the observations agree the
              done / success flags do: past an ill-conditioned call the two sides part in the arm's null space (radians in q),
              not in what the task observes (tests/tools/fence_study.py).  Pick's capped calls do not converge at all (target
              out of reach): there the end-effector itself parts, and the tier states how much of the run that touches.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def envs():
    from armenv import envs
    return envs


def _np(t):
    return t.detach().cpu().numpy()


class FenceBook:
    """Which envs are comparable at which step.  `sync`: GPU env and oracle env have finished all their episodes at the same
    steps so far (an env whose `done` flags ever differ is in a different episode from then on, for good).  `clean`: no IK
    call of the env has run to the iteration cap or through an ill-conditioned system since the last reset both sides did
    together (the oracle's own update count and pivots decide)."""

    def __init__(self, n, cap, pivot):
        self.n, self.cap, self.pivot = n, cap, pivot
        self.sync = np.ones(n, dtype=bool)
        self.clean = np.ones(n, dtype=bool)
        self.checked = self.tainted = self.desynced = self.total = self.resynced = 0
        self.cap_calls = self.cond_calls = 0
        # task-space tier: every env-step of an env that is in step with its twin (clean or not)
        self.t2_steps = self.t2_within_1e4 = self.t2_within_1e3 = self.t2_flags = 0
        self.t2_worst = self.t2_worst_eef = 0.0

    def comparable(self, iters_o, minpiv_o):
        """mask of the envs to compare at this step (call before `advance`)"""
        capped, illc = iters_o >= self.cap, minpiv_o < self.pivot
        self.cap_calls += int(capped.sum()); self.cond_calls += int(illc.sum())
        self.clean &= ~(capped | illc)
        chk = self.sync & self.clean
        self.total += self.n
        self.checked += int(chk.sum())
        self.tainted += int((self.sync & ~self.clean).sum())
        self.desynced += int((~self.sync).sum())
        return chk

    def task_space(self, d_obs, flags_differ, d_eef=None):
        """second tier (call before `advance`): d_obs = max |obs_gpu - obs_oracle| per env (d_eef: over the end-effector part
        only), flags_differ = done or success differ"""
        m = self.sync
        if d_eef is not None:
            self.t2_worst_eef = max(self.t2_worst_eef, float(d_eef[m].max(initial=0.0)))
        self.t2_steps += int(m.sum())
        self.t2_within_1e4 += int((d_obs[m] < 1e-4).sum())
        self.t2_within_1e3 += int((d_obs[m] < 1e-3).sum())
        self.t2_worst = max(self.t2_worst, float(d_obs[m].max(initial=0.0)))
        self.t2_flags += int((flags_differ & m & (d_obs < 1e-4)).sum())

    def advance(self, done_g, done_o):
        self.sync &= done_g == done_o
        self.clean |= self.sync & done_g & done_o       # a reset both sides did together starts a clean episode

    def resync(self, st, gpu_state):
        """Teacher-forced resynchronisation at a launch boundary: every env the strict tier has dropped --
        tainted by a capped / ill-conditioned IK call, or out of step with its twin -- has its oracle twin re-seeded from the GPU's
        own state (q, aux, step, episode, ep_return) and is re-admitted: from here on the two sides start from the same numbers
        again, and the only env-steps the strict tier never sees are the capped / ill-conditioned calls themselves (plus the rest
        of the launch they happened in)."""
        m = ~(self.sync & self.clean)
        if m.any():
            idx = np.nonzero(m)[0]
            st.q[idx] = gpu_state["q"][idx]
            st.aux[idx] = gpu_state["aux"][idx]
            st.step[idx] = gpu_state["step"][idx]
            st.episode[idx] = gpu_state["episode"][idx].view(np.uint32) if gpu_state["episode"].dtype != np.uint32 else gpu_state["episode"][idx]
            st.ep_return[idx] = gpu_state["ep_return"][idx]
            self.resynced += int(m.sum())
            self.sync[idx] = True
            self.clean[idx] = True


def _free_run(envs, O, kuka, task, n, launches, R, sigma, seed, max_steps=500, scripted=None, resync=False):
    """`launches` x armenv_rollout(R) with i.i.d. N(0, sigma) actions (fence counters on, per-step IK update counts out)
    against the oracle's *_step_autoreset on the same actions.  resync: after every launch the oracle twins of the envs the
    strict tier has dropped are re-seeded from the GPU's state (FenceBook.resync)."""
    Env = dict(push=envs.BatchedPushEnv, pick=envs.BatchedPickEnv)[task]
    State, reset, stepf = dict(push=(O.PushState, O.push_reset, O.push_step_autoreset),
                               pick=(O.PickState, O.pick_reset, O.pick_step_autoreset))[task]
    cfg = O.default_config(task); cfg.max_steps = max_steps
    e = Env(n, device=DEV, seed=seed, fence_counters=1, max_steps=max_steps)
    st = State(n)
    reset(kuka, cfg, st, seed=seed)
    e.reset()
    gen = torch.Generator(device=DEV); gen.manual_seed(100 + seed)
    book = FenceBook(n, int(cfg.ik_max_iters), float(cfg.fence_pivot))
    iters, minpiv = np.zeros(n, dtype=np.int32), np.zeros(n)
    worst_obs = worst_rew = 0.0
    upd_mismatch = flag_mismatch = rew_flips = 0
    gpu_cap = 0
    ep_g = ep_o = 0
    bufs = {}
    for b in range(launches):
        acts = (torch.randn((R, n, 3), device=DEV, generator=gen) * sigma).contiguous()
        out = e.rollout(R, acts, out=bufs, want_ik_updates=True)
        a_np = _np(acts)
        obs_g, rew_g, done_g, succ_g, upd_g = (_np(out[k]) for k in ("obs", "reward", "done", "success", "ik_updates"))
        gpu_cap += int((upd_g >= cfg.ik_max_iters).sum())
        for t in range(R):
            obs_o, rew_o, done_o, succ_o, _ = stepf(kuka, cfg, st, a_np[t], seed=seed, iters=iters, minpiv=minpiv)
            chk = book.comparable(iters, minpiv)
            done_o = done_o.astype(bool)
            d = np.abs(obs_g[t] - obs_o).max(1)
            worst_obs = max(worst_obs, float(d[chk].max(initial=0.0)))
            fl = (done_g[t] != done_o) | (succ_g[t] != succ_o.astype(bool))
            flag_mismatch += int(fl[chk].sum())
            book.task_space(d, fl, np.abs(obs_g[t][:, :3] - obs_o[:, :3]).max(1))
            same = chk & (done_g[t] == done_o)
            # the shaped reward has a threshold (rl_push_env.py:393-394: -1 when the distance moved by less than 1e-5, -100 x the
            # change otherwise): a change within rounding of 1e-5 -- the falling / settling cube passes through it -- lands on either side
            rg = rew_g[t].astype(np.float64)
            flip = ((rg == -1.0) ^ (rew_o == -1.0)) & (np.minimum(np.abs(rg), np.abs(rew_o)) < 2e-3)
            rew_flips += int((flip & same).sum())
            worst_rew = max(worst_rew, float(np.abs(rg - rew_o)[same & ~flip].max(initial=0.0)))
            upd_mismatch += int((upd_g[t].astype(np.int32) != iters)[chk].sum())
            ep_g += int(done_g[t].sum()); ep_o += int(done_o.sum())
            book.advance(done_g[t], done_o)
        if resync:
            book.resync(st, {k: _np(v) for k, v in e.get_state().items()})
    cnt = e.counters()
    e.close()
    assert rew_flips <= 1e-5 * book.total + 2, rew_flips
    return dict(book=book, worst_obs=worst_obs, worst_rew=worst_rew, upd_mismatch=upd_mismatch, flag_mismatch=flag_mismatch, rew_flips=rew_flips,
                gpu_cap=gpu_cap, counters=cnt, ep_g=ep_g, ep_o=ep_o)


def _report(task, r):
    b, c = r["book"], r["counters"]
    return (f"{task}: {b.total} env-steps, compared {b.checked} ({100.0 * b.checked / b.total:.2f} %), twins re-seeded from the GPU state {b.resynced}, excluded after a capped / "
            f"ill-conditioned IK call {b.tainted} ({100.0 * b.tainted / b.total:.2f} %), out of step {b.desynced} "
            f"({100.0 * b.desynced / b.total:.3f} %); capped calls oracle {b.cap_calls} gpu {r['gpu_cap']} (counter {c['cap_steps']}), "
            f"ill-conditioned calls oracle {b.cond_calls} gpu counter {c['illcond_steps']}; worst |obs| {r['worst_obs']:.2e} "
            f"worst |reward| {r['worst_rew']:.2e} ({r['rew_flips']} steps on either side of the reward's 1e-5 threshold); IK update counts differing {r['upd_mismatch']}; flags differing {r['flag_mismatch']}; "
            f"episodes gpu {r['ep_g']} oracle {r['ep_o']} || task-space tier: {b.t2_steps} env-steps ({100.0 * b.t2_steps / b.total:.3f} %), "
            f"within 1e-4 {100.0 * b.t2_within_1e4 / max(1, b.t2_steps):.4f} %, within 1e-3 {100.0 * b.t2_within_1e3 / max(1, b.t2_steps):.4f} %, "
            f"worst |obs| {b.t2_worst:.2e} (end-effector part {b.t2_worst_eef:.2e}), flags differing where obs agree {b.t2_flags}")


@pytest.mark.parametrize("seed", [6, 16, 26])
def test_push_config4_free_running_vs_oracle(envs, O, kuka, record_property, seed):
    """BASELINE config 4 at its own size: rl_push_env, 32 768 envs, train_push_with_TD3's exploration noise (main.py:484),
    501-step episodes (rl_push_env.py:418), 6 x armenv_rollout(100) against push_step_autoreset, three seeds.
    Strict tier -- every env that has had no capped or ill-conditioned IK call since its last reset: observation (eef, cube,
    target) within 1e-4 at every step, identical done / success flags, reward within 2e-2 (= -100 x the change of a distance
    between two positions that are within 1e-4), the same IK update counts.
    Task-space tier -- EVERY env-step (the excluded ones too; nothing is out of step on this task): end-effector within 1e-3
    (measured 3e-4 .. 4e-4 on the three seeds), the whole observation within 1e-4 on 99.9 % of them, no flag differing where
    the observations agree.  The cube part of the observation has no per-step bound: the build-defined contact model is
    discontinuous (the tool sphere touches the cube or misses it), so an end-effector difference of 1e-4 at a grazing contact
    moves the cube by centimetres on one side only -- 3 to 4 env-steps per million.
    Matches /root/reference/envs/rl_push_env.py:310-356."""
    n = 32768
    r = _free_run(envs, O, kuka, "push", n, 6, 100, 0.4 * 0.98, seed=seed)
    msg = _report("push seed %d" % seed, r)
    print(msg); record_property("fence", msg)
    b = r["book"]
    assert r["worst_obs"] < 1e-4 and r["flag_mismatch"] == 0, msg
    assert r["worst_rew"] < 2e-2, msg
    assert r["upd_mismatch"] <= 1e-4 * b.checked, msg            # a residual within rounding of 1e-4 may flip one trip
    assert b.checked >= 0.72 * b.total, msg                     # 0.57 % of push's IK calls are ill-conditioned (the table-height corners of the box)
    assert r["counters"]["cap_steps"] == r["gpu_cap"], msg      # the counter is the sum of the per-step view
    assert abs(r["gpu_cap"] - b.cap_calls) <= max(8, 0.05 * b.cap_calls), msg
    assert abs(r["counters"]["illcond_steps"] - b.cond_calls) <= max(8, 0.05 * b.cond_calls), msg
    assert r["ep_g"] >= n and abs(r["ep_g"] - r["ep_o"]) <= 1e-3 * r["ep_o"], msg
    assert r["counters"]["nonfinite"] == 0
    # task-space tier over 100 % of the env-steps
    assert b.t2_steps >= 0.9999 * b.total, msg
    assert b.t2_worst_eef < 1e-3 and b.t2_within_1e4 >= 0.999 * b.t2_steps and b.t2_within_1e3 >= 0.9999 * b.t2_steps and b.t2_flags == 0, msg


def test_pick_32768_free_running_vs_oracle(envs, O, kuka, record_property):
    """The pick task at config 4's size under the same exploration noise (main.py:552), lane-asynchronous rollouts (the
    pick default), 501-step episodes.  0.8 % of its IK calls run to Bullet's 20-iteration cap (the arm wanders to the top of
    the 0.807 m box, where the tool-down pose is out of reach): those env-steps and the rest of their episodes are the cap
    term of the fence; everything else agrees with the oracle at every step.  Matches
    /root/reference/envs/rl_pick_env.py:310-355."""
    n = 32768
    r = _free_run(envs, O, kuka, "pick", n, 6, 100, 0.4 * 0.98, seed=7)
    msg = _report("pick", r)
    print(msg); record_property("fence", msg)
    b = r["book"]
    assert r["worst_obs"] < 1e-4 and r["flag_mismatch"] == 0, msg
    assert r["worst_rew"] < 2e-2, msg
    assert r["upd_mismatch"] <= 1e-4 * b.checked, msg
    assert b.checked >= 0.58 * b.total, msg                     # the comparison covers the head of every episode
    # task-space tier: every env-step of an env still in the same episode as its twin.  A capped call does not converge (the
    # target is out of the arm's reach with the tool orientation asked for): twenty oscillating updates end centimetres apart
    # on the two sides, and the env carries that offset to its next reset -- so the tier states what share of the run agrees
    # (thresholds set from the measured line, recorded in profiles/r04_fence_free_running.txt) instead of a per-step bound
    assert b.t2_steps >= 0.995 * b.total, msg
    assert b.t2_within_1e4 >= 0.80 * b.t2_steps and b.t2_flags <= 1e-5 * b.t2_steps, msg
    assert r["counters"]["cap_steps"] == r["gpu_cap"], msg
    assert abs(r["gpu_cap"] - b.cap_calls) <= 0.05 * b.cap_calls, msg       # the cap RATE is a property of the workload
    assert abs(r["counters"]["illcond_steps"] - b.cond_calls) <= 0.05 * b.cond_calls, msg
    assert 0.002 < b.cap_calls / b.total < 0.03, msg
    assert r["counters"]["nonfinite"] == 0


# (two seeds per task since round 6 -- the third, 26 / 27, gave the same rates to the second decimal in round 5 and cost a minute of
# the suite: profiles/r05_fence_free_running.txt)
@pytest.mark.parametrize("task,seed", [("push", 6), ("push", 16), ("pick", 7), ("pick", 17)])
def test_cube_tasks_strict_tier_with_resync(envs, O, kuka, record_property, task, seed):
    """The strict tier over (nearly) everything: the same workloads as the two free-running tests above (32 768 envs, exploration
    noise of main.py:484 / :552, 501-step episodes, 600 steps), one env step per launch, and after every launch the oracle twin of
    an env that has just had a capped or ill-conditioned IK call (or whose done flag differed) is re-seeded from the GPU's own state
    (q, aux, step, episode, ep_return through armenv_get_state).  The env is back in the strict comparison at the next step, so
    the only env-steps never compared strictly are the capped / ill-conditioned calls themselves: push >= 99 %, pick >= 97 % of
    all env-steps (free-running: 74 % / 61 %), each within 1e-4 on the whole observation with identical flags and IK update counts.
    Matches /root/reference/envs/rl_push_env.py:310-356, rl_pick_env.py:310-355."""
    n = 32768
    r = _free_run(envs, O, kuka, task, n, 600, 1, 0.4 * 0.98, seed=seed, resync=True)
    msg = _report("%s seed %d, twins re-seeded after every step" % (task, seed), r)
    print(msg); record_property("fence", msg)
    b = r["book"]
    assert r["worst_obs"] < 1e-4 and r["flag_mismatch"] == 0 and r["worst_rew"] < 2e-2, msg
    assert r["upd_mismatch"] <= 1e-4 * b.checked, msg
    assert b.checked >= (0.99 if task == "push" else 0.97) * b.total, msg
    assert b.resynced >= 0.5 * (b.cap_calls + b.cond_calls) and b.desynced <= 1e-4 * b.total, msg
    assert r["ep_g"] >= n and r["counters"]["nonfinite"] == 0, msg


@pytest.mark.parametrize("task", ["push", "pick"])
def test_short_episodes_free_running_vs_oracle(envs, O, kuka, task):
    """Many resets: 25-step episodes, 8 192 envs, 3 x rollout(50); the time-limit resets happen at the same steps on both
    sides, so an env excluded after a capped call comes back into the comparison with its next episode."""
    r = _free_run(envs, O, kuka, task, 8192, 3, 50, 0.4 * 0.98, seed=11, max_steps=24)
    msg = _report(task, r)
    b = r["book"]
    assert r["worst_obs"] < 1e-4 and r["flag_mismatch"] == 0 and r["worst_rew"] < 2e-2, msg
    assert b.checked >= 0.9 * b.total and r["ep_g"] >= 5 * 8192, msg
    assert r["counters"]["cap_steps"] == r["gpu_cap"], msg


def test_cap_counter_and_update_counts_teacher_forced(envs, O, kuka):
    """The third fence term, teacher-forced.  States from 150 oracle steps of the pick workload (the arms that reach the top
    of the box are in there), then every step starts from the oracle's state: armenv_step's ik_updates output equals the
    oracle's update count, armenv_counters out[7] counts exactly the calls that reached ik_max_iters, and both equal bit 2
    of the oracle's fence flags.  Same for the reach task with an out-of-reach target box (every call capped)."""
    n = 4096
    rng = np.random.default_rng(17)
    cfg = O.default_config("pick")
    st = O.PickState(n)
    O.pick_reset(kuka, cfg, st, seed=5)
    for _ in range(150):
        O.pick_step_autoreset(kuka, cfg, st, (rng.standard_normal((n, 3)) * 0.392).astype(np.float32), seed=5)
    e = envs.BatchedPickEnv(n, device=DEV, seed=5, auto_reset=False, fence_counters=1)
    e.reset()
    caps = conds = 0
    for t in range(12):
        a = (rng.standard_normal((n, 3)) * 0.392).astype(np.float32)
        e.set_state(q=st.q, aux=st.aux, step=st.step, ep_return=st.ep_return)
        q0 = st.q.copy()
        p0, _ = O.fk(kuka, q0)
        tgt = np.clip(p0.astype(np.float32).astype(np.float64) + 0.08 * a.astype(np.float64), cfg.box_lo[:], cfg.box_hi[:])
        flags = O.fence_flags(kuka, cfg, q0, tgt)
        c0 = e.counters()
        e.step(torch.from_numpy(a).to(DEV), want_ik_updates=True)
        upd = _np(e.ik_updates).astype(np.int32)
        _, _, _, _, iters = O.pick_step(kuka, cfg, st, a)
        c1 = e.counters()
        assert (upd != iters).sum() <= 2, (t, int((upd != iters).sum()))
        assert c1["cap_steps"] - c0["cap_steps"] == int((upd >= 20).sum())
        assert c1["ik_updates"] - c0["ik_updates"] == int(upd.sum())
        assert abs(int(((flags & 4) != 0).sum()) - int((upd >= 20).sum())) <= 2
        assert np.array_equal((flags & 4) != 0, iters >= 20)
        # conditioning term: the kernel's count of calls with a pivot below fence_pivot against bit 3 of the oracle's flags
        # (a pivot within rounding of the threshold may fall on either side)
        assert abs((c1["illcond_steps"] - c0["illcond_steps"]) - int(((flags & 8) != 0).sum())) <= 3, t
        caps += int((upd >= 20).sum()); conds += int(((flags & 8) != 0).sum())
    assert caps >= 50 and conds >= 50, (caps, conds)                 # the test has teeth
    e.close()
    # reach with targets far outside the arm's reach: every IK call runs to the cap
    m = 512
    r = envs.BatchedReachEnv(m, device=DEV, seed=1, auto_reset=False, fence_counters=1, dv=1.0, box_hi=[2.0, 2.0, 2.0])
    r.reset()
    r.step(torch.ones((m, 3), device=DEV), want_ik_updates=True)
    assert bool((r.ik_updates == 20).all()) and r.counters()["cap_steps"] == m
    off = envs.BatchedReachEnv(m, device=DEV, seed=1, auto_reset=False, dv=1.0, box_hi=[2.0, 2.0, 2.0])    # bookkeeping off (default)
    off.reset(); off.step(torch.ones((m, 3), device=DEV))
    assert off.counters()["cap_steps"] == 0 and off.counters()["ik_updates"] == 20 * m
    from armenv import ArmEnvError
    with pytest.raises(ArmEnvError):               # the per-step view belongs to the bookkeeping build of the kernels
        off.step(torch.ones((m, 3), device=DEV), want_ik_updates=True)
    r.close(); off.close()


@pytest.mark.parametrize("precision", [64, 32])
def test_limit_pushback_model_teacher_forced(envs, O, kuka, precision):
    """R7 as a named model, clamp_joint_limits = 2: a joint the IK left beyond its URDF limit
    (/root/reference/envs/bmirobot_joints_info_pybullet.txt:1-7) is moved back by limit_erp (default 0.2, Bullet's constraint
    ERP) of its violation per step (rl_reach_env.py:252-258); joints inside keep their bits.  Oracle and kernel, teacher-forced."""
    from test_gpu_parity import _actions, _limit_fence_states
    n = 4096
    rng = np.random.default_rng(321)
    cfg0, cfg2 = O.default_config(), O.default_config()
    cfg2.clamp_joint_limits = 2
    lim = np.array(O.KUKA["limit"])
    free = envs.BatchedReachEnv(n, device=DEV, auto_reset=False, precision=precision, fence_counters=1)
    erp = envs.BatchedReachEnv(n, device=DEV, auto_reset=False, precision=precision, clamp_joint_limits=2, fence_counters=1)
    assert erp.cfg.limit_erp == 0.2
    tol = 1e-6 if precision == 64 else 1e-4
    hits = 0
    for rep in range(3):
        q = _limit_fence_states(O, kuka, cfg0, n, rng)
        a = _actions(rng, n)
        st0, st2 = O.ReachState(n), O.ReachState(n)
        for st in (st0, st2):
            st.q[:] = q; st.goal[:] = np.float32([0.45, 0.1, 0.3])
        for env in (free, erp):
            env.reset(); env.set_state(q=q, goal=st0.goal, step=st0.step)
        at = torch.from_numpy(a).to(DEV)
        obs_f = _np(free.step(at)[0]).copy(); obs_e = _np(erp.step(at)[0]).copy()
        O.reach_step(kuka, cfg0, st0, a)
        obs_r2, *_ = O.reach_step(kuka, cfg2, st2, a)
        qf, qe = _np(free.get_state()["q"]), _np(erp.get_state()["q"])
        ok = (np.abs(qf - st0.q).max(1) < tol) & (np.abs(qe - st2.q).max(1) < tol)
        assert ok.mean() > (0.999 if precision == 64 else 0.99)
        assert np.abs(obs_e - obs_r2)[ok].max() < max(tol, 2e-7)
        out = (np.abs(st0.q) > lim).any(axis=1)                               # the raw IK result leaves the limits
        assert np.array_equal(qf[~out & ok], qe[~out & ok])                    # envs inside the limits keep their bits
        viol0 = np.clip(np.abs(st0.q) - lim, 0.0, None)                        # violation of the raw result ...
        viol2 = np.clip(np.abs(st2.q) - lim, 0.0, None)                        # ... and after one push-back: 80 % of it
        assert np.abs(viol2 - 0.8 * viol0).max() < 1e-12
        hits += int(out.sum())
    assert hits >= 100
    from armenv import ArmEnvError
    with pytest.raises(ArmEnvError):
        envs.BatchedReachEnv(64, device=DEV, clamp_joint_limits=2, limit_erp=0.0)
    with pytest.raises(ArmEnvError):
        envs.BatchedReachEnv(64, device=DEV, clamp_joint_limits=3)
    free.close(); erp.close()


@pytest.mark.parametrize("task", ["reach", "push", "pick"])
def test_trig_rederivation_at_step_512_in_every_launch_grouping(envs, task):
    """The carried (cos q, sin q) pair is re-derived from q when an env's own step counter reaches a multiple of 512 (never inside
    the reference's 501-step episodes; round 4 moved the re-derivation from the top of the next step to the tail of this one, into
    the rare out-of-line region it shares with `done`).  With 1 300-step episodes: (i) right after step 512 the carried pair IS the
    pair set_state derives from the same q, bit for bit -- and one step earlier it is not, for most envs (the incremental rotations
    have drifted by last bits); (ii) the trajectory across steps 512 and 1 024 is the same bits as ONE rollout launch, as 100-step
    launches (the re-derivation falls inside a launch, and -- at 1 024 -- on a launch's last step: the lane-asynchronous and the
    lockstep kernels, full and half-filled waves) and as one-step launches around the two boundaries."""
    n = 256 + 5
    Env = dict(reach=envs.BatchedReachEnv, push=envs.BatchedPushEnv, pick=envs.BatchedPickEnv)[task]
    gen = torch.Generator(device=DEV); gen.manual_seed(11)
    sig = 0.686 if task == "reach" else 0.392
    acts = (torch.randn((1100, n, 3), device=DEV, generator=gen) * sig).clamp_(-0.7, 0.7).contiguous()
    kw = dict(device=DEV, seed=4, max_steps=1300, reach_dis=1e-9) if task == "reach" else dict(device=DEV, seed=4, max_steps=1300, push_success_dis=1e-9)

    def derived(e):      # the pair as set_state derives it from the handle's own q
        st = {k: v.clone() for k, v in e.get_state().items()}
        f = Env(n, **kw); f.reset()
        f.set_state(**{k: v for k, v in st.items() if k != "trig"})
        t = f.get_state()["trig"].clone(); f.close()
        return st["trig"], t

    a = Env(n, **kw); a.reset()
    a.rollout(511, acts[:511].contiguous())
    carried, fresh = derived(a)
    assert int((carried != fresh).any(dim=1).sum()) > n // 2          # the drift the re-derivation removes is there
    a.rollout(1, acts[511:512].contiguous())
    assert int(a.get_state()["step"].min()) == 512 and int(a.get_state()["step"].max()) == 512
    carried, fresh = derived(a)
    assert torch.equal(carried, fresh)
    a.rollout(588, acts[512:].contiguous())
    ref = {k: v.clone() for k, v in a.get_state().items()}
    a.close()

    def same(e, what):
        st = e.get_state()
        for k in ref:
            assert torch.equal(ref[k], st[k]), (task, what, k)
        e.close()

    b = Env(n, **kw); b.reset()
    b.rollout(1100, acts)                                               # one launch
    same(b, "one launch")
    for ready, lanes in ((0, 64), (62, 32), (0, 32)):                    # 100-step launches: step 512 inside one, 1 024 = a launch's last but... 1 000 + 24
        c = Env(n, rollout_ready_lanes=ready, rollout_lanes_per_wave=lanes, **kw); c.reset()
        for k in range(11):
            c.rollout(100, acts[100 * k:100 * (k + 1)].contiguous())
        same(c, ("100-step launches", ready, lanes))
    d = Env(n, **kw); d.reset()                                          # launches that END on the re-derivation steps, step launches around them
    d.rollout(510, acts[:510].contiguous())
    for t in range(510, 515):
        d.step(acts[t])
    d.rollout(1024 - 515, acts[515:1024].contiguous())
    for t in range(1024, 1030):
        d.step(acts[t])
    d.rollout(1100 - 1030, acts[1030:].contiguous())
    same(d, "boundaries as launch ends and step launches")


@pytest.mark.parametrize("task", ["reach", "push", "pick"])
def test_checkpoint_restores_the_trajectory_bitwise(envs, task):
    """get_state / set_state as a checkpoint: the carried (cos q, sin q) pair travels with q, so a handle
    restored mid-episode continues the uninterrupted run bit for bit -- outputs and final state; restoring q alone
    (resetJointState semantics) re-derives the pair and is equal to the IK's noise floor only."""
    n, T = 1024 + 7, 60
    Env = dict(reach=envs.BatchedReachEnv, push=envs.BatchedPushEnv, pick=envs.BatchedPickEnv)[task]
    gen = torch.Generator(device=DEV); gen.manual_seed(5)
    sig = 0.686 if task == "reach" else 0.392
    acts = (torch.randn((2 * T, n, 3), device=DEV, generator=gen) * sig).clamp_(-0.7, 0.7).contiguous()
    a, b, c = (Env(n, device=DEV, seed=9, max_steps=45) for _ in range(3))
    a.reset(); b.reset(); c.reset()
    a.rollout(T, acts[:T].contiguous())
    snap = {k: v.clone() for k, v in a.get_state().items()}
    assert snap["trig"].shape == (n, 14)
    q = snap["q"]
    assert float((snap["trig"][:, :7] - torch.cos(q)).abs().max()) < 1e-12 and float((snap["trig"][:, 7:] - torch.sin(q)).abs().max()) < 1e-12
    ref = {k: v.clone() for k, v in a.rollout(T, acts[T:].contiguous()).items()}
    b.set_state(**snap)
    got = b.rollout(T, acts[T:].contiguous())
    for k in ("obs", "reward", "done", "success"):
        assert torch.equal(ref[k], got[k]), (task, k)
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), (task, k)
    # without the pair: the same trajectory to the IK's noise floor (not required to be bitwise)
    c.set_state(**{k: v for k, v in snap.items() if k != "trig"})
    got_c = c.rollout(T, acts[T:].contiguous())
    within = (got_c["done"] == ref["done"]).all(0) & ((got_c["obs"] - ref["obs"]).abs().amax(dim=(0, 2)) < 1e-4)
    assert float(within.float().mean()) > (0.97 if task == "pick" else 0.995)     # pick: envs that pass through a capped IK call
    for x in (a, b, c):
        x.close()


def test_step_before_first_reset_is_well_defined(envs):
    """a fresh handle sits at q = 0 with (cos q, sin q) = (1, 0), not at all-zero rotation frames."""
    e = envs.BatchedReachEnv(256, device=DEV, auto_reset=False)
    st = e.get_state()
    assert bool((st["trig"][:, :7] == 1).all()) and bool((st["trig"][:, 7:] == 0).all()) and bool((st["q"] == 0).all())
    obs, rew, done, succ = e.step(torch.zeros((256, 3), device=DEV))
    assert bool(torch.isfinite(obs).all()) and bool(torch.isfinite(rew).all()) and e.counters()["nonfinite"] == 0
    assert bool((obs == obs[0]).all())                                   # every env did the same well-defined thing
    e.close()


# ------------------------------------------------------------------------------ multi-GPU plumbing on the one GPU

def test_return_gatherer_gpu_branch_survives_freed_producers():
    """armenv.dist.ReturnGatherer's GPU branch (side stream, two alternating (stage, out) slots, record_stream on the
    producer's tensor) with world = 1 semantics, against what a synchronous gather would return: 60 launches, the producer
    tensor freed right after each launch and its block immediately re-used for garbage (the round-1 hazard: the side stream
    read a tensor the caching allocator had already handed out again), results consumed on the producer stream while the
    next launches are queued (the write-after-read hazard of ADVICE r02)."""
    from armenv.dist import ReturnGatherer
    n = 1 << 18
    g = ReturnGatherer(n, DEV, world=1)
    ramp = torch.arange(n, device=DEV, dtype=torch.float32) * 1e-3
    consumed = []
    for k in range(60):
        x = ramp + float(k)                          # a fresh producer tensor every launch
        for _ in range(4):                           # queued work in front of it: the tensor is not complete at launch time
            x = x * 1.0
        g.launch(x)
        del x
        junk = [torch.full((n,), -7.0, device=DEV) for _ in range(3)]      # re-uses the freed blocks at once
        del junk
        if k % 2 == 1:
            out = g.result()                         # orders the producer stream behind the collective(s)
            consumed.append((k, (out - ramp).sum() / n))          # a consumer kernel reading `out` on the producer stream
    torch.cuda.synchronize()
    for k, v in consumed:
        assert abs(float(v) - k) < 1e-3, (k, float(v))
    assert g.launches == 60 and torch.equal(g.result(), ramp + 59.0)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL cannot run between ranks that share one device")
def test_bench_two_gpus_over_rccl():
    """`bench.py --gpus 2` under torch.distributed.run on two real GPUs: the logging all-gather goes through RCCL (backend
    "nccl"), every rank owns its device, and the line carries the kernel-time figure beside the wall-clock one."""
    from test_gpu_parity import _run_bench
    d = _run_bench(["--gpus", "2", "--steps", "200", "--warmup", "20", "--no-cpu-baseline"], nproc=2)
    assert d["n_gpus"] == 2 and d["config"]["total_envs"] == 131072 and "RCCL" in d["config"]["parallelism"]
    assert d["config"]["gathers_in_timed_region"] >= 1 and d["value_kernel"] >= d["value"] > 5e9
    assert len(d["config"]["per_rank"]["kernel_ms"]) == 2


def test_bench_one_rank_over_rccl(envs):
    """What a 1-GPU box can execute of BASELINE configs[4]'s RCCL path: `bench.py` as ONE rank under torch.distributed.run with
    ARMENV_BENCH_COLLECTIVE=1 -- init_process_group("nccl", device_id), the barriers of the bracket, the logging
    all_gather_into_tensor on the side stream inside the timed region, the max-over-ranks all_gather / all_reduce on device
    tensors, all_gather_object, destroy_process_group: every collective call of the multi-GPU run goes through RCCL (a
    communicator of one rank).  The gathered vector is the handle's own episode returns and the trajectory is the single-handle
    one (same digest as a plain rollout of the same pool)."""
    import hashlib
    n, K, W = 8192, 700, 5
    d = _run_bench_env(["--gpus", "1", "--steps", str(K), "--warmup", str(W), "--envs-per-gpu", str(n), "--state-digest",
                        "--prewarm-ms", "0", "--gather-every", "100", "--no-cpu-baseline"])
    assert d["n_gpus"] == 1 and d["config"]["rccl_ranks_seen"] == {"world_size": 1, "backend": "nccl"}
    assert "RCCL" in d["config"]["parallelism"] and d["config"]["gathers_in_timed_region"] >= 7
    assert d["value_steps"] >= d["value"] >= d["value_bracketed"] > 0 and len(d["config"]["per_rank"]["kernel_ms"]) == 1
    for k in ("barrier", "shm_barrier", "gather_wait", "gather_issue", "collective"):
        assert k in d["config"]["host_us"], k
    # the clock of `value` closes on the launch stream + the shared-memory barrier; the collective is verified after it
    assert d["config"]["collective_verified"] is True and d["config"]["collective_us"] > 0 and d["config"]["barrier_us"] < 100.0
    gen = torch.Generator(device=DEV); gen.manual_seed(1000)
    pool = (torch.randn((1000, n, 3), device=DEV, generator=gen) * 0.686).clamp_(-0.7, 0.7)
    e = envs.BatchedReachEnv(n, device=DEV, seed=0)
    e.reset()
    t = 0
    for r in [W] + [100] * (K // 100):
        e.rollout(r, pool[t:t + r].contiguous())
        t += r
    q = e.get_state()["q"].cpu().numpy()
    assert d["config"]["state_digest"] == [hashlib.sha256(q.tobytes()).hexdigest()]
    ret = e.episode_stats()[0].detach().float().cpu().numpy()
    assert d["config"]["gathered_returns_sha256"] == hashlib.sha256(ret.tobytes()).hexdigest()
    assert d["config"]["gathered_returns_mean"] == pytest.approx(float(ret.astype(np.float64).mean()), rel=1e-12) and ret.min() < -1.0
    e.close()


def test_episode_returns_f32_is_the_logging_vector(envs):
    """armenv_episode_returns_f32 (ABI 6): the per-env return of the last finished episode as ONE f32 vector -- the send buffer of the
    logging all-gather -- equals armenv_episode_stats' f64 vector rounded once, into a fresh tensor, into a caller's buffer, and on a
    raw side stream ordered by the caller (ReturnGatherer.launch_into(..., takes_stream=True)); a wrong buffer is refused."""
    from armenv.dist import ReturnGatherer
    n = 4096 + 64
    e = envs.BatchedReachEnv(n, device=DEV, seed=5, max_steps=15)
    e.reset()
    gen = torch.Generator(device=DEV); gen.manual_seed(2)
    e.rollout(40, (torch.randn((40, n, 3), device=DEV, generator=gen) * 0.686).clamp_(-0.7, 0.7))
    want = e.episode_stats()[0].to(torch.float32)
    assert float(want.min()) < 0.0 and torch.equal(e.episode_returns_f32(), want)
    out = torch.empty(n, dtype=torch.float32, device=DEV)
    assert e.episode_returns_f32(out=out) is out and torch.equal(out, want)
    with pytest.raises(ValueError):
        e.episode_returns_f32(out=torch.empty(n, dtype=torch.float64, device=DEV))
    g = ReturnGatherer(n, DEV, 1)                      # one rank, no process group: the side-stream plumbing without a collective
    g.launch_into(lambda stage, stream=None: e.episode_returns_f32(out=stage, stream=stream), takes_stream=True)
    g.order_after_read()
    e.rollout(3, torch.zeros((3, n, 3), device=DEV))  # the next launch waits until the returns have been read
    assert torch.equal(g.result(), want)
    e.close()


def test_bench_one_rank_over_rccl_keeps_the_single_gpu_value():
    """the N-rank `value` measures the engine.  The driver's shape (`--steps 20 --warmup 5`, 65 536 envs) as ONE
    rank down the whole RCCL path (process group, dist.barrier() around the bracket, the logging all-gather issued inside the region)
    against the plain single-GPU run of the same session: the same bracket code, so the two `value`s are commensurable -- round 5's
    bracket lost 41 % here.  Both runs are single regions of ~150 us on a shared host, so the bound is on the best of three tries
    (measured pairs: 0.955-0.965, profiles/r06_bench_rccl_world1.json beside r06_bench_plain_beside_rccl.json)."""
    from test_gpu_parity import _run_bench
    argv = ["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--secondary-legs", "0", "--fence-steps", "0",
            "--large-batch", "0", "--ab-regions", "0", "--repeat-regions", "2"]
    best = 0.0
    for _ in range(3):
        plain = _run_bench(argv)
        rccl = _run_bench_env(argv)
        assert rccl["config"]["rccl_backend"] == "nccl" and rccl["config"]["gathers_in_timed_region"] == 1
        assert rccl["config"]["collective_verified"] is True
        assert rccl["config"]["collective_transport"] == "rccl-direct", rccl["config"]["collective_direct_error"]
        assert rccl["value_steps"] >= rccl["value"] >= rccl["value_bracketed"]
        best = max(best, rccl["value"] / plain["value"])
        if best >= 0.95:
            break
    assert best >= 0.95, best


def _run_bench_env(argv):
    from test_gpu_parity import _run_bench
    return _run_bench(argv, torchrun=True, env={"ARMENV_BENCH_COLLECTIVE": "1"})


@pytest.mark.parametrize("task", ["reach", "push", "pick"])
def test_half_filled_waves_equal_full_waves(envs, task):
    """ArmEnvConfig.rollout_lanes_per_wave: 32 envs per wavefront (lanes 32..63 idle; the default for push / pick rollouts of at
    most 32 x #SIMDs envs) against full wavefronts: the same bits -- outputs, final state, every counter incl. the four fence
    terms -- on a ragged batch (not a multiple of 32), with external actions and the in-kernel random policy, lockstep and
    lane-asynchronous, across in-place resets (20-step episodes) and a second launch."""
    n, T = 2048 + 32 + 7, 45
    Env = dict(reach=envs.BatchedReachEnv, push=envs.BatchedPushEnv, pick=envs.BatchedPickEnv)[task]
    rng = np.random.default_rng(5)
    sig = 0.686 if task == "reach" else 0.392
    acts = torch.from_numpy((rng.standard_normal((T, n, 3)) * sig).clip(-0.7, 0.7).astype(np.float32)).to(DEV)
    for policy in ("external", "random"):
        for ready in (0, 62):
            ref = None
            for lanes in (64, 32, 0):
                e = Env(n, device=DEV, seed=17, max_steps=20, rollout_lanes_per_wave=lanes, rollout_ready_lanes=ready, fence_counters=1)
                if policy == "random":
                    e.set_policy("random", noise_sigma=sig, noise_clip=0.7)
                e.reset()
                out = e.rollout(T, acts if policy == "external" else None, want_actions=True, want_terminal_obs=True, want_ik_updates=True)
                got = {k: out[k].clone() for k in ("obs", "reward", "done", "success", "actions", "terminal_obs", "ik_updates")}
                out2 = e.rollout(9, acts[:9].contiguous() if policy == "external" else None)
                got.update(obs2=out2["obs"].clone(), done2=out2["done"].clone())
                got.update({"st_" + k: v.clone() for k, v in e.get_state().items()})
                cnt = e.counters()
                for k in ("wave_trips", "wave_rounds"):      # what the schedule cost the waves: depends on the wave filling
                    assert cnt.pop(k) > 0
                e.close()
                if ref is None:
                    ref, ref_cnt = got, cnt
                    assert cnt["episodes"] >= 2 * n and cnt["env_steps"] == n * (T + 9)
                else:
                    for k in ref:
                        assert torch.equal(ref[k], got[k]), (task, policy, ready, lanes, k)
                    assert cnt == ref_cnt, (task, policy, ready, lanes)
    from armenv import ArmEnvError
    with pytest.raises(ArmEnvError):
        Env(64, device=DEV, rollout_lanes_per_wave=16)


def test_pick_schedule_counters_at_32768(envs, record_property):
    """What the three rollout schedules cost pick's wavefronts at its benchmark size (32 768 envs, exploration noise of
    main.py:484, steady state after 600 steps), by the bookkeeping build's schedule counters (armenv_counters out[9], out[10]):
    lockstep pays its slowest lane at every step (~8.8 trips per wave-step for 4.4 per env-step), the lane-asynchronous count
    rule less (~7.2), the straggler rule -- the default -- less again (~6.8), each for a few more step tails; the three
    trajectories are the same bits.  (timings: profiles/r03_async_schedule.txt.)"""
    n, T = 32768, 100
    gen = torch.Generator(device=DEV); gen.manual_seed(1000)
    pool = torch.randn((1000, n, 3), device=DEV, generator=gen) * 0.392
    waves = n // 32                      # half-filled waves at this size
    res, ref = {}, None
    for name, over in (("lockstep", dict(rollout_ready_lanes=0)), ("count", dict(rollout_straggler_trips=0)), ("straggler", {})):
        e = envs.BatchedPickEnv(n, device=DEV, seed=0, fence_counters=1, **over)
        assert name != "straggler" or (e.cfg.rollout_ready_lanes, e.cfg.rollout_straggler_trips) == (62, 6)
        e.reset()
        for k in range(6):
            e.rollout(T, pool[k * T:(k + 1) * T])
        c0 = e.counters()
        for k in range(6, 10):
            e.rollout(T, pool[k * T:(k + 1) * T])
        c1 = e.counters()
        q = e.get_state()["q"].clone()
        e.close()
        ws = waves * 4 * T
        res[name] = ((c1["wave_trips"] - c0["wave_trips"]) / ws, (c1["wave_rounds"] - c0["wave_rounds"]) / ws,
                     (c1["ik_updates"] - c0["ik_updates"]) / (n * 4 * T) + 1.0)
        if ref is None:
            ref = q
        else:
            assert torch.equal(ref, q), name
    record_property("pick_schedule", {k: [round(x, 3) for x in v] for k, v in res.items()})
    print("pick 32768 schedules (wave trips, tails per wave-step; trips per env-step):", res)
    assert res["lockstep"][1] == 1.0 and res["lockstep"][2] == res["count"][2] == res["straggler"][2]
    assert 4.2 < res["lockstep"][2] < 4.7
    assert res["straggler"][0] < res["count"][0] < res["lockstep"][0]
    assert res["straggler"][0] < 0.82 * res["lockstep"][0] and res["straggler"][1] < 1.4
