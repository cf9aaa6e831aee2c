#!/usr/bin/env python3
"""Long free-running comparison of the HIP rollout with the CPU oracle (one-off confidence run, not part of the suite):
N envs x T steps with auto-reset and the same pre-generated actions; reports episode / success totals and how far the two
trajectories are apart per 500-step block (teacher forcing is what the parity tests use; this is the drift picture)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "drl-on-robot-arm_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from armenv import envs  # noqa: E402
from oracle import oracle as O  # noqa: E402

n, T, B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, int(sys.argv[2]) if len(sys.argv) > 2 else 3000, 500
task = sys.argv[3] if len(sys.argv) > 3 else "reach"
dev = "cuda:0"
ch, cfg = O.make_chain("kuka"), O.default_config(task)
Env = dict(reach=envs.BatchedReachEnv, push=envs.BatchedPushEnv, pick=envs.BatchedPickEnv)[task]
State, reset, stepf = dict(reach=(O.ReachState, O.reach_reset, O.reach_step_autoreset), push=(O.PushState, O.push_reset, O.push_step_autoreset),
                           pick=(O.PickState, O.pick_reset, O.pick_step_autoreset))[task]
e = Env(n, device=dev, seed=5)
st = State(n)
reset(ch, cfg, st, seed=5)
e.reset()
gen = torch.Generator(device=dev); gen.manual_seed(9)
sig, clip = (0.686, 0.7) if task == "reach" else (0.392, 1e9)
ep_g = ep_o = su_g = su_o = 0
t0 = time.time()
for b in range(T // B):
    acts = (torch.randn((B, n, 3), device=dev, generator=gen) * sig).clamp_(-clip, clip).contiguous()
    out = e.rollout(B, acts)
    a_np = acts.cpu().numpy()
    worst = 0.0; agree = []
    obs_g = out["obs"].cpu().numpy(); done_g = out["done"].cpu().numpy(); succ_g = out["success"].cpu().numpy()
    for t in range(B):
        r = stepf(ch, cfg, st, a_np[t], seed=5)
        obs_o, done_o, succ_o = r[0], r[2], r[3]
        d = np.abs(obs_g[t] - obs_o).max(1)
        agree.append((d < 1e-4).mean())
        ep_o += int(done_o.sum()); su_o += int((done_o.astype(bool) & succ_o.astype(bool)).sum())
    ep_g += int(done_g.sum()); su_g += int((done_g & succ_g).sum())
    print(f"steps {(b + 1) * B:5d}: envs within 1e-4 of the oracle: min over block {min(agree):.4f}, last {agree[-1]:.4f} | episodes gpu {ep_g} oracle {ep_o} | "
          f"successes gpu {su_g} oracle {su_o} | nonfinite {e.counters()['nonfinite']} | {time.time() - t0:.0f}s", flush=True)
