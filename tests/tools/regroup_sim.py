#!/usr/bin/env python3
"""Would regrouping envs across wavefronts between launches pay?  (CPU oracle, no GPU.)

A wave pays its slowest lane: lockstep sum_t max_lane trips(e, t), lane-asynchronous max_lane sum_t trips(e, t).  If an env's
IK cost is persistent from one launch to the next (pick: an arm that wandered to the top of the box runs Bullet's loop to
its 20-iteration cap for hundreds of steps in a row), sorting the envs by the trips they needed in the PREVIOUS launch puts
the slow ones into the same waves.  This script replays the oracle's per-step trip counts through both schedules with
  identity   env e in slot e (what the engine did up to round 2)
  previous   slots sorted by the previous T-step window's trip total (what a launch-end regroup can know)
  oracle     slots sorted by THIS window's trip total (the bound for any predictor)
Then two sweeps over the same trip record: the launch length T (regrouping more often needs shorter launches, which cost the
lane-asynchronous schedule its averaging), and the number of envs per wave (half-filled waves).
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


# launch-length sweep: predictors a launch-end regroup could use -- the previous window's total, its last step, an EWMA
cap = trips >= cfg.ik_max_iters + 1
print(f"  per-step persistence: P(capped at t+1 | capped at t) = {(cap[1:] & cap[:-1]).sum() / max(1, cap[:-1].sum()):.2f}; "
      + "lag correlation of trips: " + " ".join(f"{lag}:{np.corrcoef(trips[:-lag].ravel(), trips[lag:].ravel())[0, 1]:.2f}" for lag in (1, 5, 20, 100)))
S = trips.shape[0]
for Tw in (5, 10, 20, 50, 100):
    acc = {k: [] for k in ("identity", "previous", "last step", "ewma", "clairvoyant")}
    prev = None; ew = np.zeros(n)
    for k in range(S // Tw):
        W = trips[k * Tw:(k + 1) * Tw]; tot = W.sum(0)
        acc["identity"].append(cost(W, ident)); acc["clairvoyant"].append(cost(W, np.argsort(tot, kind="stable")))
        if prev is not None:
            acc["previous"].append(cost(W, np.argsort(prev, kind="stable")))
            acc["last step"].append(cost(W, np.argsort(last, kind="stable")))
            acc["ewma"].append(cost(W, np.argsort(ew, kind="stable")))
        prev, last, ew = tot, W[-1], 0.5 * ew + tot / Tw
    print(f"  launches of {Tw:3d} steps: " + " | ".join(f"{k} lock {np.mean([c[0] for c in v]):.2f} async {np.mean([c[1] for c in v]):.2f}" for k, v in acc.items()))
# envs per wave
for LW in (64, 32, 16):
    lock, asy = [], []
    for k in range(S // T):
        W = trips[k * T:(k + 1) * T].reshape(T, n // LW, LW)
        lock.append(W.max(2).sum(0).mean() / T); asy.append(W.sum(0).max(1).mean() / T)
    print(f"  {LW:2d} envs per wave ({T}-step launches): lockstep {np.mean(lock):.2f}  lane-asynchronous {np.mean(asy):.2f}  (mean over envs {trips.mean():.2f})")
