#!/usr/bin/env python3
"""Standalone TD3 actor forward (armenv_actor_forward) for 65 536 states, both kinds, by HIP events; ARMENV_LIB selects another
build of the library (A/B of actor kernels without the env step around them)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.getcwd(), "drl-on-robot-arm_amd")); sys.path.insert(0, os.getcwd())
import torch, numpy as np
from armenv import envs
import bench
sd = bench.golden_actor()
n = 65536
e = envs.BatchedReachEnv(256, device="cuda:0")
for kind in ("actor_f16x3", "actor"):
    e.set_policy(kind, action_bound=0.7, actor_state_dict=sd)
    st = torch.rand((n, 6), device="cuda:0")
    for _ in range(5): e.actor_forward(st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): e.actor_forward(st)
    e1.record(); torch.cuda.synchronize()
    print(os.environ.get("ARMENV_LIB", "current"), kind, "standalone actor_forward 65536 states: %.2f us" % (e0.elapsed_time(e1) * 1e3 / 50))
