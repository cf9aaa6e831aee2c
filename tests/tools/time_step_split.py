#!/usr/bin/env python3
"""The gym-style step path (one armenv_step launch per env step) at 65 536 envs: what the options of VERDICT r03 #5 buy.

  plain        one handle, one stream (bench.py's step_api)
  split P      PipelinedEnv: P handles over consecutive env ranges, P streams; a part's launch t+1 runs under the other parts'
               launch-t tails (a launch ends with its slowest env)
  w2 half      one handle built for two waves per SIMD, launched as half-filled waves (rollout_waves_per_simd=2 cannot be combined
               with half-filled waves in this engine: reported as the plain two-waves build at this size for reference)
  each eager (K bare C calls per step) and replayed from a hipGraph of G steps
  closed loop  DATD3Policy.take_action -> step, plain vs run_closed_loop on 2 parts: policy(A) under step(B)

Open loop: pre-generated actions (the bench's pool), steady state after `pre` steps.  Prints us per full-batch env step (HIP
events around the whole run on the caller's stream, parts joined) and env-steps/s.
Usage: python tests/tools/time_step_split.py [envs] [steps] [pre]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "drl-on-robot-arm_amd"))
import torch
from armenv import envs
from armenv.policies import DATD3Policy

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 600
pre = int(sys.argv[3]) if len(sys.argv) > 3 else 600
dev = "cuda:0"
gen = torch.Generator(device=dev); gen.manual_seed(1000)
S = 1000
pool = (torch.randn((S, n, 3), device=dev, generator=gen) * 0.686).clamp_(-0.7, 0.7)

w = envs.BatchedReachEnv(n, device=dev, seed=99); w.set_policy("random"); w.reset()
t0 = time.time(); b = {}
while time.time() - t0 < 0.3:
    w.rollout(100, None, out=b); torch.cuda.synchronize()
w.close(); del b


def timed(fn, reps=3):
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        us = max(e0.elapsed_time(e1) * 1e3, wall * 1e6) / K
        best = us if best is None else min(best, us)
    return best


def report(name, us):
    print("%-44s %7.2f us per step   %.2f G env-steps/s" % (name, us, n / us / 1e3), flush=True)


def open_loop(parts, graph, **kw):
    if parts == 1:
        e = envs.BatchedReachEnv(n, device=dev, seed=0, **kw)
    else:
        e = envs.PipelinedEnv(envs.BatchedReachEnv, n, parts=parts, device=dev, seed=0, **kw)
    e.reset()
    acts = pool[:K] if K <= S else pool
    # steady state: past the first time-limit resets
    if parts == 1:
        for t in range(pre):
            e.step(pool[t % S])
        steps = [e.bind_step(acts[t % S]) for t in range(K)]
        join = lambda: None
    else:
        for t in range(pre):
            e.step(pool[t % S], fork=False, join=False)
        e.join()
        steps = e.bind_steps(acts[:K])
        join = e.join
    torch.cuda.synchronize()
    if not graph:
        def run():
            if parts > 1:
                e._fork_from_current()
            for f in steps:
                f()
            join()
        us = timed(run)
    else:
        G = 50
        g = torch.cuda.CUDAGraph()
        # capture G steps: the parts' streams fork from / join into the capturing stream
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            if parts > 1:
                e._fork_from_current()
            for f in steps[:4]:
                f()
            join()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        if parts == 1:
            with torch.cuda.graph(g):
                cs = torch.cuda.current_stream(dev)
                import ctypes as C
                e._fixed_stream = C.c_void_p(cs.cuda_stream)
                caps = [e.bind_step(acts[t]) for t in range(G)]
                for f in caps:
                    f()
            e._fixed_stream = None
        else:
            with torch.cuda.graph(g):
                e._fork_from_current()
                for f in steps[:G]:
                    f()
                e.join()
        g.replay(); torch.cuda.synchronize()

        def run():
            for _ in range(K // G):
                g.replay()
        us = timed(run)
    e.close()
    return us


for parts in (1, 2, 4):
    for graph in (False, True):
        try:
            report("open loop, %s, %s" % ("plain" if parts == 1 else "split %d" % parts, "hipGraph of 50" if graph else "eager"), open_loop(parts, graph))
        except Exception as ex:       # noqa: BLE001
            print("open loop parts=%d graph=%s FAILED: %s" % (parts, graph, str(ex)[:300]), flush=True)
try:
    report("open loop, plain, two-waves-per-SIMD build, eager", open_loop(1, False, rollout_waves_per_simd=2))
except Exception as ex:           # noqa: BLE001
    print("w2 FAILED:", str(ex)[:300])

# closed loop: DATD3's two-actor / two-critic arg-max in front of every step
pol = DATD3Policy(6, 3, 0.7, device=dev)
Kc = K


def closed(parts):
    if parts == 1:
        e = envs.BatchedReachEnv(n, device=dev, seed=0)
        obs = e.reset()

        def run():
            o = obs
            for _ in range(Kc):
                o, _, _, _ = e.step(pol.take_action(o))
        run(); us = timed(run, reps=2)
    else:
        e = envs.PipelinedEnv(envs.BatchedReachEnv, n, parts=parts, device=dev, seed=0)
        e.reset()
        run = lambda: e.run_closed_loop(pol.take_action, Kc)
        run(); us = timed(run, reps=2)
    e.close()
    return us


for parts in (1, 2, 4):
    try:
        report("closed loop DATD3Policy -> step, %s" % ("plain" if parts == 1 else "split %d" % parts), closed(parts))
    except Exception as ex:       # noqa: BLE001
        print("closed loop parts=%d FAILED: %s" % (parts, str(ex)[:300]), flush=True)
