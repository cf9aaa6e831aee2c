#!/usr/bin/env python3
"""IK trip statistics of the benchmark workloads from the CPU oracle (no GPU): what a T-step launch of one-env-per-lane
waves must cost at least, and what the lockstep wave actually pays.

A step of env e costs trips(e, t) = updates + 1 FKs.  For a wave of 64 envs over a T-step launch:
  lockstep   sum_t max_lane trips      every step ends with the wave's slowest lane (what env_rollout_kernel does)
  async      max_lane sum_t trips      lanes never wait for each other inside the launch (critical path of the slowest env)
  mean       mean_lane sum_t trips     perfect load balance
and the LAUNCH ends with its slowest wave.  Usage: trip_stats.py [task] [envs] [T] [pre_steps] [envs per wave]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O
task = sys.argv[1] if len(sys.argv) > 1 else "reach"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
T = int(sys.argv[3]) if len(sys.argv) > 3 else 100
pre = int(sys.argv[4]) if len(sys.argv) > 4 else 600
LW = int(sys.argv[5]) if len(sys.argv) > 5 else 64        # envs per wave (64 = full wavefronts)
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
W = trips.reshape(T, n // LW, LW)
lock = W.max(2).sum(0) / T; asyn = W.sum(0).max(1) / T; mean = W.mean()
print(f"{task} {n} envs, {LW} per wave, {T}-step launch after {pre} steps: updates per env-step {trips.mean() - 1:.3f}; trips (FKs) per step:")
print(f"  mean over envs            {mean:.3f}")
print(f"  lockstep wave  (sum_t max_lane): mean over waves {lock.mean():.3f}  slowest wave {lock.max():.3f}")
print(f"  async wave     (max_lane sum_t): mean over waves {asyn.mean():.3f}  slowest wave {asyn.max():.3f}")
print(f"  slowest env of the launch {trips.sum(0).max() / T:.3f}   histogram of per-step trips: " +
      " ".join(f"{k}:{(trips == k).mean():.4f}" for k in range(1, 22) if (trips == k).any()))
