#!/usr/bin/env python3
"""Measurement of the rows downstream of the env step on one MI355X: episode indexing
and HER-"future" sampling of the device-resident trajectory store, and one TD3 update of the torch learner.
Prints one JSON line per measurement (HIP events on the launch stream); results are kept in profiles/.

Algorithmic bytes per HER sample (reach, D = 6 floats): read episode entry 12 + state row 24 + next-state row 24 +
action 12 + reward 4 + done 1 + the future step's row 24 (relabel) = 101; write state 24 + next 24 + action 12 + reward 4
+ done 1 = 65; total 166 B (push / pick, D = 9: 226 B)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "drl-on-robot-arm_amd"))
import torch  # noqa: E402
from armenv import envs  # noqa: E402
from armenv.replay import TrajectoryStore  # noqa: E402
from armenv.td3 import TD3  # noqa: E402

DEV = "cuda:0"


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters      # us


def main():
    for task, Env, D in (("reach", envs.BatchedReachEnv, 6), ("push", envs.BatchedPushEnv, 9)):
        n, T = 65536, 64
        env = Env(n, device=DEV, seed=0, max_steps=20)         # short episodes: ~3 complete episodes per env in the window
        env.set_policy("random")
        obs0 = env.reset().clone()
        out = env.rollout(T, None, want_actions=True, want_terminal_obs=True)
        store = TrajectoryStore(device=DEV, seed=1)
        store.add_rollout(obs0, out)
        E = store.size()
        us = timed(store._index, 20)
        print(json.dumps({"what": "index_episodes (count + prefix sum + write)", "task": task, "envs": n, "steps": T,
                          "episodes": E, "us": round(us, 1), "done_bytes_read_twice_GBps": round(2 * n * T / us / 1e3, 1)}))
        per = 166 if D == 6 else 226
        for B in (2048, 65536, 1 << 20):
            us = timed(lambda: store.sample(B, use_her=True, her_ratio=0.8), 50 if B < (1 << 20) else 20)
            print(json.dumps({"what": "her_sample", "task": task, "batch": B, "us": round(us, 1),
                              "samples_per_s": round(B / us * 1e6), "algorithmic_GBps": round(B * per / us / 1e3, 1),
                              "hbm_peak_GBps": 8000}))
        env.close()
    agent = TD3(6, 3, 0.7, device=DEV)
    for B in (256, 2048, 16384):
        batch = dict(states=torch.rand(B, 6, device=DEV), actions=torch.rand(B, 3, device=DEV) - 0.5,
                     next_states=torch.rand(B, 6, device=DEV), rewards=torch.rand(B, device=DEV),
                     dones=torch.zeros(B, dtype=torch.uint8, device=DEV))
        us = timed(lambda: agent.train(batch), 50)
        print(json.dumps({"what": "TD3 update, eager (torch learner, armenv/td3.py)", "batch": B, "us": round(us, 1),
                          "transitions_per_s": round(B / us * 1e6)}))
        g = TD3(6, 3, 0.7, device=DEV)
        static = g.capture(B)
        for k, v in static.items():
            v.copy_(batch[k])
        us = timed(lambda: g.train_graphed(static), 50)
        print(json.dumps({"what": "TD3 update, hipGraph replay (TD3.capture / train_graphed)", "batch": B, "us": round(us, 1),
                          "transitions_per_s": round(B / us * 1e6)}))


if __name__ == "__main__":
    main()
