#!/usr/bin/env python3
"""GPU sweep of the lane-asynchronous schedule's transition threshold (ArmEnvConfig.rollout_ready_lanes) and of the wave
filling (rollout_lanes_per_wave) for one task at one batch size: us per step of 100-step rollouts in steady state
(600 warm-up steps first: the arms have to reach the top of the box before pick shows its capped IK calls).
Usage: ready_lanes_sweep.py [task] [envs] [lanes,...] [ready,...] [straggler_trips,...]    (ready 0 = lockstep; straggler
trips K > 0: the straggler rule, for which ready only has to be non-zero)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "drl-on-robot-arm_amd"))
import torch
from armenv import envs

task = sys.argv[1] if len(sys.argv) > 1 else "pick"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
lanes = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "32,64").split(",")]
readies = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "0,16,32,48,56,60,62,63,64").split(",")]
stragglers = [int(x) for x in (sys.argv[5] if len(sys.argv) > 5 else "0").split(",")]
extra = json.loads(os.environ.get("ARMENV_SWEEP_OVERRIDES", "{}"))     # further ArmEnvConfig fields, e.g. {"max_steps": 100000}
Env = {"reach": envs.BatchedReachEnv, "push": envs.BatchedPushEnv, "pick": envs.BatchedPickEnv}[task]
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1000)
T, S = int(os.environ.get("ARMENV_SWEEP_T", "100")), 1000      # steps per launch; rows of the action pool
NL = 1200 // T if T <= 400 else 2                             # timed launches
pool = torch.randn((S, n, 3), device=dev, generator=gen) * (0.392 if task != "reach" else 0.686)
if task == "reach":
    pool.clamp_(-0.7, 0.7)
ref = None
for lw in lanes:
    for r, K in [(r, K) for r in readies for K in (stragglers if r > 0 else [0])]:
        env = Env(n, device=dev, seed=0, rollout_ready_lanes=r, rollout_lanes_per_wave=lw, rollout_straggler_trips=K, **extra)
        env.reset()
        bufs = {}
        for k in range(6):
            env.rollout(100, pool[k * 100:(k + 1) * 100])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        e0.record()
        for k in range(NL):        # the pool's rows are i.i.d.: launches reuse them cyclically
            o = (k * T) % (S - T + 1)
            env.rollout(T, pool[o:o + T], out=bufs)
        e1.record()
        torch.cuda.synchronize(dev)
        us = e0.elapsed_time(e1) * 1e3 / (NL * T)
        q = env.get_state()["q"]
        same = "" if ref is None else (" bits equal" if torch.equal(q, ref) else " BITS DIFFER")
        if ref is None:
            ref = q.clone()
        env.close()
        # the same 1 000 steps in the bookkeeping build: what the schedule cost the waves (armenv_counters out[9], out[10])
        env = Env(n, device=dev, seed=0, rollout_ready_lanes=r, rollout_lanes_per_wave=lw, rollout_straggler_trips=K, fence_counters=1, **extra)
        env.reset()
        for k in range(6):
            env.rollout(100, pool[k * 100:(k + 1) * 100])
        c0 = env.counters()
        for k in range(NL):
            o = (k * T) % (S - T + 1)
            env.rollout(T, pool[o:o + T], out=bufs)
        c1 = env.counters()
        env.close()
        ws = ((n + lw - 1) // lw) * NL * T
        print(f"{task} {n} envs, {T}-step launches, {lw} lanes per wave, ready_lanes {r:2d}, straggler_trips {K:2d}: {us:7.3f} us per step{same}; per wave-step "
              f"{(c1['wave_trips'] - c0['wave_trips']) / ws:.2f} trips, {(c1['wave_rounds'] - c0['wave_rounds']) / ws:.2f} tails "
              f"(per env-step {(c1['ik_updates'] - c0['ik_updates']) / (n * NL * T) + 1:.2f} trips, {100 * (c1['episodes'] - c0['episodes']) / (n * NL * T):.2f} % of env-steps end an episode)", flush=True)
