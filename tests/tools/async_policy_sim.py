#!/usr/bin/env python3
"""Which transition rule should the lane-asynchronous rollout use?  (CPU oracle trip record, no GPU.)

The lane-asynchronous kernel (csrc/armenv_env.h: env_rollout_async_kernel) alternates IK trips of the lanes that are still
iterating with "transition rounds" in which the lanes whose IK has stopped finish their step and begin the next one.  A
round costs the whole wave a step tail + a step head; a trip costs it a trip, however few lanes take part.  This script replays
the oracle's per-step trip counts of one task through wave models of
  lockstep            one round per step, all lanes
  count >= k          round when k lanes wait (what ArmEnvConfig.rollout_ready_lanes selects)
  stragglers >= K     round when every lane that has NOT yet spent K trips on its step waits: lanes on their way to Bullet's
                      20-iteration cap carry on, the others stay in phase with each other
and prices them with instruction counts per wave (trip with an update ~700, exit trip ~230, round ~R).
Usage: async_policy_sim.py [task] [envs] [steps] [pre_steps] [lanes]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O

task = sys.argv[1] if len(sys.argv) > 1 else "pick"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
T = int(sys.argv[3]) if len(sys.argv) > 3 else 300
pre = int(sys.argv[4]) if len(sys.argv) > 4 else 600
LW = int(sys.argv[5]) if len(sys.argv) > 5 else 32
cache = f"/tmp/armenv_trips_{task}_{n}_{T}_{pre}.npy"
if os.path.exists(cache):
    trips = np.load(cache)
else:
    ch, cfg = O.make_chain("kuka"), O.default_config(task)
    State, reset, stepf = dict(reach=(O.ReachState, O.reach_reset, O.reach_step_autoreset), push=(O.PushState, O.push_reset, O.push_step_autoreset),
                               pick=(O.PickState, O.pick_reset, O.pick_step_autoreset))[task]
    st = State(n); reset(ch, cfg, st, seed=0)
    rng = np.random.default_rng(1)
    sig, clip = (0.686, 0.7) if task == "reach" else (0.392, 1e9)
    it = np.zeros(n, dtype=np.int32)
    trips = np.zeros((T, n), dtype=np.int32)
    for t in range(pre + T):
        a = np.clip(rng.standard_normal((n, 3)) * sig, -clip, clip).astype(np.float32)
        if task == "reach":
            stepf(ch, cfg, st, a, seed=0, want_terminal=False, iters=it)
        else:
            stepf(ch, cfg, st, a, seed=0, iters=it)
        if t >= pre:
            trips[t - pre] = it + 1
    np.save(cache, trips)

C_UPD, C_EXIT = 700.0, 230.0
W = n // LW
tr = trips.reshape(T, W, LW)


def simulate(rule, R):
    """rule(ready [W, LW] bool, spent [W, LW] trips spent on the current step, active [W, LW] lanes with steps left) -> [W] bool."""
    t = np.zeros((W, LW), dtype=np.int64)
    rem = tr[0].copy()
    spent = np.zeros((W, LW), dtype=np.int64)
    ready = np.zeros((W, LW), dtype=bool)
    cost = np.zeros(W)
    rounds = np.zeros(W)
    iters = np.zeros(W)
    wi = np.arange(W)[:, None]
    li = np.arange(LW)[None, :]
    while True:
        active = t < T
        if not active.any():
            break
        run = active & ~ready
        upd = run & (rem > 1)
        anyrun = run.any(1)
        cost += np.where(upd.any(1), C_UPD, np.where(anyrun, C_EXIT, 0.0))
        iters += anyrun
        rem = np.where(run, rem - 1, rem)
        spent = np.where(run, spent + 1, spent)
        ready = ready | (run & (rem == 0))
        go = (rule(ready, spent, active) | ~(active & ~ready).any(1)) & (ready.any(1))
        adv = ready & go[:, None]
        cost += np.where(go, R, 0.0)
        rounds += go
        t = np.where(adv, t + 1, t)
        nxt = tr[np.minimum(t, T - 1), wi, li]
        rem = np.where(adv, nxt, rem)
        spent = np.where(adv, 0, spent)
        ready = ready & ~adv
    return cost.mean() / T, rounds.mean() / T, iters.mean() / T


print(f"{task} {n} envs, {T} steps after {pre}, {LW} lanes per wave: mean trips per env-step {trips.mean():.3f}, "
      f"capped {100 * (trips >= 21).mean():.2f} % of env-steps")
for R in (400.0, 600.0, 800.0):
    print(f" round = {R:.0f} instructions")
    c, r, i = simulate(lambda ready, spent, active: (ready | ~active).all(1), R)
    print(f"  lockstep                    : {c:7.0f} per wave-step ({i:.2f} trips, {r:.2f} rounds)")
    for k in (LW - 1, LW - 2, LW - 4, LW // 2):
        c, r, i = simulate(lambda ready, spent, active, k=k: ready.sum(1) >= k, R)
        print(f"  count >= {k:2d}                 : {c:7.0f} per wave-step ({i:.2f} trips, {r:.2f} rounds)")
    for K in (5, 6, 7, 8, 10):
        c, r, i = simulate(lambda ready, spent, active, K=K: ~(active & ~ready & (spent < K)).any(1), R)
        print(f"  stragglers after {K:2d} trips   : {c:7.0f} per wave-step ({i:.2f} trips, {r:.2f} rounds)")
    for K in (6, 8):
        for k in (LW // 2, LW * 3 // 4):
            c, r, i = simulate(lambda ready, spent, active, K=K, k=k: ~(active & ~ready & (spent < K)).any(1) & (ready.sum(1) >= k), R)
            print(f"  stragglers {K:2d} & count >= {k:2d}: {c:7.0f} per wave-step ({i:.2f} trips, {r:.2f} rounds)")
