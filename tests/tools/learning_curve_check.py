#!/usr/bin/env python3
"""The reference's own learning curve against this build.

/root/reference/visdata/reach/TD3_0.01/Reach_TD3.json is the only record the reference holds of how rl_reach_env BEHAVES: one
`train_reach_with_TD3` run (main.py:165-231), 351 episodes, success rate 0.96 from episode 125, 1.0 from 175, mean return of
the last 25 episodes -67.9 (tests/golden/visdata_reach_td3.json: the numbers, extracted by gen_fixtures.py).  It is one sample
of a stochastic process, so the comparison is statistical -- but a wrong dv, reward scale, success threshold, episode length
or an IK that does not track its target would move these numbers by factors, which no restatement-vs-restatement test can see.

This tool runs that function's protocol on the N=1 drop-in `armenv.envs.RLReachEnv` (every env step one armenv_step launch on
the GPU) with the build's TD3 / Trajectory / TrajectoryStore counterparts:
    per episode:  s = env.reset();  until done:  a = actor(s) + N(0, 1 * opt.gamma) (UNCLIPPED, main.py:200);  env.step(a);
                  then, once 5 episodes are stored, n_train = 40 updates on HER batches of 256 (her_ratio 0.8, x 0.75 whenever a
                  25-episode success rate is a new maximum, main.py:221-225)
for K independent learners (seeds 0..K-1; `--procs` of them at a time as processes sharing the GPU) and prints, beside the
fixture's: episodes to the first 25-episode block with success >= 0.9, the success-rate series, the mean return of episodes
327..351, and the mean return of the first five (untrained) episodes -- which pins reward scale x episode length x workspace.

Usage: python tests/tools/learning_curve_check.py [--learners 16] [--procs 8] [--episodes 351]"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "drl-on-robot-arm_amd"))


def learner(seed, episodes):
    import numpy as np
    import torch
    from armenv import envs, opt
    from armenv.replay import Trajectory, TrajectoryStore
    from armenv.td3 import TD3
    dev = "cuda:0"
    env = envs.RLReachEnv(is_render=False, is_good_view=False)                    # main.py:171
    action_bound = float(env.action_space.high[0]) + 0.3                           # :174
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)              # :176-178
    store = TrajectoryStore(device=dev, seed=seed, capacity_steps=episodes * 501 + 8)
    agent = TD3(6, 3, action_bound, device=dev)               # defaults = config.py:54-72, as TD3_MLP is built at main.py:182-183
    her_ratio = float(opt.her_ratio)
    returns, lengths, rates = [], [], []
    rate, max_rate = 0.0, 0.0
    t0 = time.time()
    for ep in range(episodes):
        state = env.reset()
        traj = Trajectory(state)
        done, ret = False, 0.0
        while not done:
            action = agent.take_action(state)
            action = action + np.random.normal(0, 1 * opt.gamma, size=3)          # :200 (unclipped)
            state, reward, done, is_success = env.step(action)
            if is_success:
                rate += 1
            ret += reward
            traj.store_step(action, state, reward, done)
        store.add_trajectory(traj)
        returns.append(ret); lengths.append(traj.length)
        if store.size() >= opt.minimal_episodes:                                   # :209-213
            for _ in range(opt.n_train):
                agent.train(store.sample(opt.batch_size, use_her=True, her_ratio=her_ratio))
        if (ep + 1) % 25 == 0:                                                     # :221-228
            rate /= 25.0
            rates.append(rate)
            if rate >= max_rate:
                max_rate = rate
                her_ratio *= 0.75
            rate = 0.0
    env.close()
    return dict(seed=seed, returns=returns, lengths=lengths, success_rate=rates, seconds=time.time() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--learners", type=int, default=16)
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--episodes", type=int, default=351)
    ap.add_argument("--one", type=int, default=None, help="(internal) run one learner with this seed and print its JSON")
    a = ap.parse_args()
    if a.one is not None:
        print("LEARNER " + json.dumps(learner(a.one, a.episodes)), flush=True)
        return
    import subprocess
    import numpy as np
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "visdata_reach_td3.json")))
    pend, runs, live = list(range(a.learners)), [], []
    while pend or live:
        while pend and len(live) < a.procs:
            s = pend.pop(0)
            live.append((s, subprocess.Popen([sys.executable, os.path.abspath(__file__), "--one", str(s), "--episodes", str(a.episodes)],
                                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
        s, p = live.pop(0)
        out, err = p.communicate()
        line = [l for l in out.splitlines() if l.startswith("LEARNER ")]
        if p.returncode != 0 or not line:
            print("learner %d failed: %s" % (s, err[-500:]), flush=True)
            continue
        runs.append(json.loads(line[0][8:]))

    def summary(ret, rates):
        first90 = next((25 * (i + 1) for i, r in enumerate(rates) if r >= 0.9), None)
        return first90, float(np.mean(ret[-25:])), float(np.mean(ret[:5]))
    f90, l25, f5 = summary(fx["return_per_episode"], fx["success_rate_every_25_episodes"])
    print("reference (visdata/reach/TD3_0.01/Reach_TD3.json, one run): episodes to the first 25-episode block with success >= 0.9: %s | "
          "mean return of the last 25 episodes %.1f | of the first 5 episodes %.1f | success rate per 25 episodes %s"
          % (f90, l25, f5, " ".join("%.2f" % r for r in fx["success_rate_every_25_episodes"])))
    rows = [summary(r["returns"], r["success_rate"]) for r in runs]
    for r, (a90, b25, c5) in zip(runs, rows):
        print("  this build, seed %2d: to 0.9: %s | last 25: %8.1f | first 5: %8.1f | mean episode length %5.1f | %s | %.0f s"
              % (r["seed"], a90, b25, c5, float(np.mean(r["lengths"])), " ".join("%.2f" % x for x in r["success_rate"]), r["seconds"]))
    got90 = [x[0] for x in rows if x[0] is not None]
    if rows:
        print("this build, %d learners: episodes to 0.9 success: median %s (min %s, max %s; %d of %d got there) | last-25 mean return: "
              "median %.1f (min %.1f, max %.1f) | first-5 mean return: median %.1f | final success rate: mean %.3f"
              % (len(rows), np.median(got90) if got90 else None, min(got90) if got90 else None, max(got90) if got90 else None, len(got90),
                 len(rows), np.median([x[1] for x in rows]), min(x[1] for x in rows), max(x[1] for x in rows),
                 np.median([x[2] for x in rows]), float(np.mean([r["success_rate"][-1] for r in runs]))))
        ref_rates = np.array(fx["success_rate_every_25_episodes"])
        mine = np.array([r["success_rate"] for r in runs if len(r["success_rate"]) == len(ref_rates)])
        if len(mine):
            print("success rate per 25-episode block, mean over learners: " + " ".join("%.2f" % x for x in mine.mean(0)))
            print("                                    reference's run:   " + " ".join("%.2f" % x for x in ref_rates))


if __name__ == "__main__":
    main()
