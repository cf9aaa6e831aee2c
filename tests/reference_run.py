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


# ------------------------------------------------------------------------------ the recorded push run
# tests/golden/visdata_push_td3.json: first 40 per-episode returns of visdata/push/origin_TD3/TD3.json (train_push_with_TD3,
# main.py:449-515, seed 0).  Bullet's cube dynamics are not restated (the build's contact model is its own), so only what does
# not depend on them is compared:
#   * an episode in which the arm never touches the cube returns 500 x (-1) (rl_push_env.py:393-394,427) and a final
#     -50 * |cube - target| (:418-420), i.e. -500 - 50 sqrt(planar^2 + dz^2): planar = the placement distance of that reset --
#     a fixed function of random.seed(0) while every episode lasts 501 steps (6 draws per placement try :197-209, 3 per step
#     :435-437) -- and dz = what the dynamic cube sinks below the fixed target while it settles on the table.  Episodes 4, 6, 13,
#     24 of the recorded run fit ONE dz = 14.74 mm to 2e-3: the placement stream, the draw counts and ArmEnvConfig.push_rest_z;
#   * WHICH of the first five episodes (before any network update) touch the cube at all: T T T - T.

def push_fixture_returns():
    return json.load(open(os.path.join(GOLDEN, "visdata_push_td3.json")))["return_per_episode"]


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
    """The first `episodes` episodes of the recorded push run on the oracle's push env (its own contact model).
    Returns [(return, length, steps on which the cube moved)]."""
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
        c, t, _ = draw_push_placement()
        cube, tgt = c + [float(cfg.push_rest_z)], t + [float(cfg.push_place_z)]
        obs = O.push_reset_with_goal(chain, cfg, st, np.float32([cube + tgt]))[0]
        st.aux[0, 0:3] = cube; st.aux[0, 3:6] = tgt
        st.aux[0, 6] = math.sqrt(sum((a - b) ** 2 for a, b in zip(cube, tgt)))
        done, ret, n, moved = False, 0.0, 0, 0
        while not done:
            state = np.hstack((obs[:3].astype(np.float32), st.aux[0, 0:3], st.aux[0, 3:6])).astype(np.float32)   # :308, then torch.float
            a = O.actor_forward(sd, state[None], 0.4)[0].astype(np.float64) + np.random.normal(0, 0.4 * 0.98, size=3)   # main.py:481-484
            c0 = st.aux[0, 0:3].copy()
            o, r, d, s, _ = O.push_step(chain, cfg, st, a.astype(np.float32)[None])
            for k in range(3):
                random.uniform(_LO[k], _HI[k])                                   # :435-437
            moved += int(np.abs(st.aux[0, 0:3] - c0).max() > 0)
            obs = o[0]; ret += float(r[0]); n += 1; done = bool(d[0])
        out.append((ret, n, moved))
    return out
