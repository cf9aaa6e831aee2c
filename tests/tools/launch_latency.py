#!/usr/bin/env python3
"""Where the wall-clock of ONE 20-step rollout launch goes (the driver's `bench.py --steps 20`): host time of each call
between the clock's start and stop, against the kernel's own duration (HIP events)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "drl-on-robot-arm_amd"))
import torch
from armenv import envs
dev = torch.device("cuda:0")
n, T = 65536, 20
e = envs.BatchedReachEnv(n, device=dev, seed=0)
gen = torch.Generator(device=dev); gen.manual_seed(1)
pool = (torch.randn((200, n, 3), device=dev, generator=gen) * 0.686).clamp_(-0.7, 0.7)
e.reset()
bufs = {}
for rep in range(6):
    launch, _ = e.bind_rollout(T, pool[rep * T:(rep + 1) * T], out=bufs)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    if rep >= 3:
        time.sleep(0.05)          # an idle gap like the one before the driver's timed region
    p = time.perf_counter
    t0 = p(); ev0.record(); t1 = p(); launch(); t2 = p(); ev1.record(); t3 = p(); ev1.synchronize(); t4 = p()
    torch.cuda.synchronize(dev); t5 = p()
    print("rep %d idle=%d: ev0.record %.1f  launch %.1f  ev1.record %.1f  ev1.synchronize %.1f  cuda.synchronize %.1f | wall %.1f us, kernel (events) %.1f us"
          % (rep, rep >= 3, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t4 - t3) * 1e6, (t5 - t4) * 1e6, (t5 - t0) * 1e6, ev0.elapsed_time(ev1) * 1e3))
# without events
for rep in range(3):
    launch, _ = e.bind_rollout(T, pool[rep * T:(rep + 1) * T], out=bufs)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter(); launch(); torch.cuda.synchronize(dev); t1 = time.perf_counter()
    print("no events: wall %.1f us" % ((t1 - t0) * 1e6))
s = torch.cuda.current_stream(dev)
for rep in range(3):
    launch, _ = e.bind_rollout(T, pool[rep * T:(rep + 1) * T], out=bufs)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter(); launch(); s.synchronize(); t1 = time.perf_counter()
    print("no events, stream.synchronize: wall %.1f us" % ((t1 - t0) * 1e6))
