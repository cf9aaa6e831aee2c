"""Replays the head of the reference's recorded training run on the CPU oracle (test infrastructure).

tests/golden/visdata_reach_td3.json holds the per-episode returns of ONE `train_reach_with_TD3` run of the reference
(/root/reference/main.py:165-231; visdata/reach/TD3_0.01/Reach_TD3.json) -- real PyBullet, opt.random_seed = 0.  Until five
episodes are stored no network update happens (main.py:209, opt.minimal_episodes = 5), so the first five episodes are a pure
function of things that are reproducible here:
    random.seed(0)          -> the goal of every reset (envs/rl_reach_env.py:180-183: 7 draws per reset, 3 per step :316-318)
    torch.manual_seed(0)    -> the untrained TD3 actor's weights (= golden G3, tests/golden/td3_actor_seed0.npz)
    np.random.seed(0)       -> the exploration noise N(0, 1 * opt.gamma) added to every action (main.py:200)
and of the env's arithmetic: FK, the clipped target, calculateInverseKinematics, resetJointState, stepSimulation, _reward.
Five episodes = 2 068 free-running env steps (one success after 64 steps, four 501-step time-outs), 276 of them with a joint
beyond its URDF limit and 274 with the flange below z = 0.05 -- the first two terms of the parity fence.  Their returns are the
only numbers under /root/reference that real PyBullet computed THROUGH the IK and the physics step: a known answer for the whole
path, and the data the named switches of the restatement are fitted on (tests/tools/fit_bullet.py).
"""
import json
import os
import random

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_LO, _HI = (0.2, -0.3, 0), (0.7, 0.3, 0.55)         # envs/rl_reach_env.py:65-70


def fixture_returns():
    return json.load(open(os.path.join(GOLDEN, "visdata_reach_td3.json")))["return_per_episode"]


def actor_weights():
    g = np.load(os.path.join(GOLDEN, "td3_actor_seed0.npz"))
    return {k: g[k.replace(".", "_")] for k in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc3.weight", "fc3.bias")}


def replay_on_oracle(O, episodes=5, configure=None, action_f32=False, record=None):
    """The first `episodes` episodes of the run on the oracle.  configure(cfg): flips switches of the OrcConfig.
    Returns [(return, length, success)], and the number of steps on which the limit / flange fence terms fired.
    record (optional list): receives per episode (goal f32[3], actions f64[T][3]) -- the open-loop form of the run."""
    chain = O.make_chain("kuka")
    cfg = O.default_config()
    if configure:
        configure(cfg)
    raw = O.default_config()
    for f in ("ik_exit_mode", "ik_angle_f32", "ik_form", "ik_max_iters", "ik_residual", "ik_lambda", "ik_max_dtheta"):
        setattr(raw, f, getattr(cfg, f))
    raw.ik_tip_offset[:] = list(cfg.ik_tip_offset)
    sd = actor_weights()
    random.seed(0); np.random.seed(0)                  # main.py:176-177 (the env's constructor reset came before)
    st = O.ReachState(1)
    out, fence = [], [0, 0]
    lim = np.array(O.KUKA["limit"])
    for ep in range(episodes):
        goal = [random.uniform(_LO[k], _HI[k]) for k in range(3)]          # :180-182
        random.random()                                                     # :183
        for k in range(3):
            random.uniform(_LO[k], _HI[k])                                  # :210-212
        obs = O.reach_reset_with_goal(chain, cfg, st, np.float32([goal]))[0]
        g64 = obs[3:].astype(np.float64)
        done, ret, n, succ = False, 0.0, 0, False
        acts = []
        while not done:
            a = O.actor_forward(sd, obs[None].astype(np.float32), 0.7)[0].astype(np.float64)      # TD3_MLP.take_action
            a = a + np.random.normal(0, 1 * 0.98, size=3)                   # main.py:200
            acts.append(a.copy())
            q0 = st.q.copy()
            if action_f32:
                o, r, d, s, _ = O.reach_step(chain, cfg, st, a.astype(np.float32)[None])
            else:       # the reference hands the f64 action to the env: target = p + 0.02 * a in f64 (rl_reach_env.py:231-242)
                p0, _ = O.fk(chain, q0)
                tgt = np.clip(p0[0] + a * cfg.dv, cfg.box_lo[:], cfg.box_hi[:])
                qn, _ = O.ik(chain, cfg, q0, tgt[None])
                st.q[:] = qn
                st.step += 1
                p1, _ = O.fk(chain, st.q)
                dist = float(np.sqrt(np.sum((p1[0] - g64) ** 2)))
                r_, d_, s_ = O.reach_outcome(cfg, dist, int(st.step[0]))
                o = [np.concatenate([p1[0].astype(np.float32), obs[3:]])]
                r, d, s = [r_], [d_], [s_]
            for k in range(3):
                random.uniform(_LO[k], _HI[k])                              # :316-318
            qr, _ = O.ik(chain, raw, q0, (np.clip(O.fk(chain, q0)[0][0] + a * cfg.dv, cfg.box_lo[:], cfg.box_hi[:]))[None])
            fence[0] += int((np.abs(qr[0]) > lim).any()); fence[1] += int(o[0][2] < 0.05)
            obs = np.asarray(o[0], dtype=np.float32)
            ret += float(r[0]); n += 1; done = bool(d[0]); succ = bool(s[0])
        out.append((ret, n, succ))
        if record is not None:
            record.append((obs[3:].astype(np.float32).copy(), np.asarray(acts)))
    return out, fence


# ------------------------------------------------------------------------------ the recorded push runs
# tests/golden/visdata_push_td3.json: the first 40 per-episode returns of BOTH recorded train_push_with_TD3 runs (main.py:449-515, seed 0):
#   "origin"  visdata/push/origin_TD3/TD3.json -- recorded with an EARLIER reward: every step that is neither the last nor a success
#             costs -1 (the line `# reward = -1` the shipped _reward still carries, rl_push_env.py:426), so an episode returns
#             -500 - 50 |cube - target|_final (:418-420);
#   "updata"  visdata/push/updata_TD3/TD3.json -- recorded with the reward the reference SHIPS (:388-397,427): -1 on a step whose
#             cube-target distance changes by less than 1e-5, -100 x the change otherwise.
# Same seeds, and no network update before five episodes are stored (main.py:497): episodes 1-5 are the SAME trajectories in both runs,
# seen through two rewards.  Per episode the pair of returns therefore yields two observables of Bullet's cube:
#   d_f = (-500 - R_origin) / 50                      the final cube-target distance,
#   M   = R_updata - R_origin + 100 (d_f - d_0)       the number of steps on which the distance changed by >= 1e-5 (d_0 = distance after
#                                                     reset(); the sub-threshold changes of the other steps sum to < 1e-3) --
# M comes out integer to 1e-2 for every one of the first six episodes, which is the check that the two runs do share them.
#   * untouched episodes (4 and 6): M = 8 -- the cube is in free fall from its spawn height for reset()'s stepSimulation and the first
#     twelve env steps, eight of which change the distance by >= 1e-5 -- and d_f = sqrt(planar^2 + dz^2) with the placement distance of
#     that reset (a fixed function of random.seed(0): 6 draws per placement try :197-209, 3 per step :435-437) and dz = 14.74 mm
#     (ArmEnvConfig.push_rest_z).  The engine's free-fall model has no other fitted number and reproduces both returns to 3e-4.
#   * touched episodes (1, 2, 3, 5): M = 149, 192, 84, 32 and d_f - d_0 = +32.9, +1.7, +53.4, +7.4 mm: what the contact model is fitted
#     on (tests/tools/fit_bullet.py part C).

def push_fixture_returns(run="origin"):
    return json.load(open(os.path.join(GOLDEN, "visdata_push_td3.json")))["return_per_episode" if run == "origin" else "return_per_episode_updata"]


def push_recorded_observables(episodes=5):
    """[(d_f, M, planar)] of the first `episodes` episodes from the two recorded runs' returns (see above)"""
    org, upd = push_fixture_returns("origin"), push_fixture_returns("updata")
    random.seed(0)
    out = []
    for e in range(episodes):
        _, _, planar = draw_push_placement()
        for _ in range(501 * 3):
            random.uniform(0.0, 1.0)
        d_f = (-500.0 - org[e]) / 50.0
        out.append((d_f, upd[e] - org[e] + 100.0 * (d_f - planar), planar))
    return out


def actor9_weights():
    g = np.load(os.path.join(GOLDEN, "td3_actor9_seed0.npz"))
    return {k: g[k.replace(".", "_")] for k in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc3.weight", "fc3.bias")}


def draw_push_placement():
    """rl_push_env.py:195-214 on Python's global `random`: (cube xy, target xy, planar distance)"""
    import math
    x = y = xt = yt = d = 0.0
    for _ in range(1000):
        x = random.uniform(_LO[0], _HI[0]); y = random.uniform(_LO[1], _HI[1]); random.random()
        xt = random.uniform(_LO[0], _HI[0]); yt = random.uniform(_LO[1], _HI[1]); random.random()
        d = math.sqrt((x - xt) ** 2 + (y - yt) ** 2 + (0.01 - 0.01) ** 2)
        if 0.22 <= d <= 0.25:
            break
    return [x, y], [xt, yt], d


def push_untouched_returns(episodes, dz):
    """what each of the first `episodes` episodes returns if its cube is never touched (valid while all earlier ones had 501 steps)"""
    import math
    random.seed(0)
    out = []
    for _ in range(episodes):
        _, _, d = draw_push_placement()
        for _ in range(501 * 3):
            random.uniform(0.0, 1.0)
        out.append(-500.0 - 50.0 * float(np.float32(math.sqrt(d * d + dz * dz))))
    return out


def replay_push_on_oracle(O, episodes=5, configure=None):
    """The first `episodes` episodes of the recorded push runs on the oracle's push env (its own contact model).
    Returns per episode a dict: ret (the shipped reward = the "updata" run's), ret_origin (the earlier reward: -1 per step, same
    trajectory), n (steps), M (steps whose cube-target distance changed by >= 1e-5), d_f (final distance, float32 states as :400),
    planar (placement distance), moved (steps on which the cube's xy moved at all)."""
    import math
    chain = O.make_chain("kuka")
    cfg = O.default_config("push")
    if configure:
        configure(cfg)
    sd = actor9_weights()
    random.seed(0); np.random.seed(0)
    st = O.PushState(1)
    out = []
    for ep in range(episodes):
        c, t, planar = draw_push_placement()
        cube, tgt = c + [float(cfg.push_place_z)], t + [float(cfg.push_place_z)]          # both spawned at z = 0.01 (:199,206)
        obs = O.push_reset_with_goal(chain, cfg, st, np.float32([cube + tgt]))[0]
        st.aux[0, 0:2] = c; st.aux[0, 3:6] = tgt      # the f64 placement (reset_with_goal takes f32); the cube's z is the engine's: it has
        st.aux[0, 6] = math.sqrt(sum((a - b) ** 2 for a, b in zip(st.aux[0, 0:3], tgt)))      # begun to fall in reset()'s stepSimulation
        done, ret, ret_o, n, moved, M = False, 0.0, 0.0, 0, 0, 0
        while not done:
            state = np.hstack((obs[:3].astype(np.float32), st.aux[0, 0:3], st.aux[0, 3:6])).astype(np.float32)   # :308, then torch.float
            a = O.actor_forward(sd, state[None], 0.4)[0].astype(np.float64) + np.random.normal(0, 0.4 * 0.98, size=3)   # main.py:481-484
            c0, dl = st.aux[0, 0:3].copy(), float(st.aux[0, 6])
            o, r, d, s, _ = O.push_step(chain, cfg, st, a.astype(np.float32)[None])
            for k in range(3):
                random.uniform(_LO[k], _HI[k])                                   # :435-437
            moved += int(np.abs(st.aux[0, 0:2] - c0[0:2]).max() > 0)
            M += int(abs(float(st.aux[0, 6]) - dl) >= 1e-5)
            obs = o[0]; ret += float(r[0]); n += 1; done = bool(d[0])
            ret_o += float(r[0]) if (done or float(r[0]) == 100.0) else -1.0
        d32 = float(np.linalg.norm(st.aux[0, 0:3].astype(np.float32) - st.aux[0, 3:6].astype(np.float32)))
        out.append(dict(ret=ret, ret_origin=ret_o, n=n, M=M, d_f=d32, planar=planar, moved=moved))
    return out
