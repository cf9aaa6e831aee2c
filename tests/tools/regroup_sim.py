#!/usr/bin/env python3
"""Would regrouping envs across wavefronts between launches pay?  (CPU oracle, no GPU.)

A wave pays its slowest lane: lockstep sum_t max_lane trips(e, t), lane-asynchronous max_lane sum_t trips(e, t).  If an env's
IK cost is persistent from one launch to the next (pick: an arm that wandered to the top of the box runs Bullet's loop to
its 20-iteration cap for hundreds of steps in a row), sorting the envs by the trips they needed in the PREVIOUS launch puts
the slow ones into the same waves.  This script replays the oracle's per-step trip counts through both schedules with
  identity   env e in slot e (what the engine did up to round 2)
  previous   slots sorted by the previous T-step window's trip total (what a launch-end regroup can know)
  oracle     slots sorted by THIS window's trip total (the bound for any predictor)
Usage: regroup_sim.py [task] [envs] [T] [windows] [pre_steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O

task = sys.argv[1] if len(sys.argv) > 1 else "pick"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
T = int(sys.argv[3]) if len(sys.argv) > 3 else 100
K = int(sys.argv[4]) if len(sys.argv) > 4 else 6
pre = int(sys.argv[5]) if len(sys.argv) > 5 else 600
ch, cfg = O.make_chain("kuka"), O.default_config(task)
State, reset, stepf = dict(reach=(O.ReachState, O.reach_reset, O.reach_step_autoreset), push=(O.PushState, O.push_reset, O.push_step_autoreset),
                           pick=(O.PickState, O.pick_reset, O.pick_step_autoreset))[task]
st = State(n); reset(ch, cfg, st, seed=0)
rng = np.random.default_rng(1)
sig, clip = (0.686, 0.7) if task == "reach" else (0.392, 1e9)
it = np.zeros(n, dtype=np.int32)
trips = np.zeros((K * T, n), dtype=np.int32)
for t in range(pre + K * T):
    a = np.clip(rng.standard_normal((n, 3)) * sig, -clip, clip).astype(np.float32)
    if task == "reach":
        stepf(ch, cfg, st, a, seed=0, want_terminal=False, iters=it)
    else:
        stepf(ch, cfg, st, a, seed=0, iters=it)
    if t >= pre:
        trips[t - pre] = it + 1


def cost(W, perm):
    """W [T, n] trips; perm slot -> env.  Returns (lockstep, async) trips per step, mean over waves."""
    G = W[:, perm].reshape(W.shape[0], n // 64, 64)
    return G.max(2).sum(0).mean() / W.shape[0], G.sum(0).max(1).mean() / W.shape[0]


ident = np.arange(n)
print(f"{task} {n} envs, {K} windows of {T} steps after {pre}: mean trips per env-step {trips.mean():.3f}")
prev_tot = None
rows = []
for k in range(K):
    W = trips[k * T:(k + 1) * T]
    tot = W.sum(0)
    r = dict(k=k, mean=W.mean(), ident=cost(W, ident), oracle=cost(W, np.argsort(tot, kind="stable")))
    if prev_tot is not None:
        r["prev"] = cost(W, np.argsort(prev_tot, kind="stable"))
        r["corr"] = float(np.corrcoef(prev_tot, tot)[0, 1])
        # two-bucket partition: envs above 1.5 x the mean of the previous window go last
        slow = prev_tot > 1.5 * prev_tot.mean()
        r["bucket"] = cost(W, np.concatenate([ident[~slow], ident[slow]]))
        r["slow_frac"] = float(slow.mean())
    prev_tot = tot
    rows.append(r)
for r in rows:
    s = f"  window {r['k']}: mean {r['mean']:.3f} | identity lock {r['ident'][0]:.2f} async {r['ident'][1]:.2f}"
    if "prev" in r:
        s += (f" | sorted by previous window lock {r['prev'][0]:.2f} async {r['prev'][1]:.2f} (corr {r['corr']:.2f})"
              f" | two buckets lock {r['bucket'][0]:.2f} async {r['bucket'][1]:.2f} (slow {100 * r['slow_frac']:.1f} %)")
    s += f" | clairvoyant lock {r['oracle'][0]:.2f} async {r['oracle'][1]:.2f}"
    print(s)
