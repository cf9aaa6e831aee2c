"""Bitwise comparison of two builds of libarmenv.so (argv[1], default build/ab/libarmenv_base.so, against the current one): sha256 over every
output row and the final state of reach / push / pick trajectories (rollout and step launches).  GPU."""
import os, sys, hashlib, subprocess, json
ROOT = os.getcwd()
CODE = r'''
import sys, os, hashlib
sys.path.insert(0, "drl-on-robot-arm_amd")
import torch
from armenv import envs
out = {}
for task, Env, sig in (("reach", envs.BatchedReachEnv, 0.686), ("push", envs.BatchedPushEnv, 0.392), ("pick", envs.BatchedPickEnv, 0.392)):
    n, T = 8192, 100
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(5)
    pool = torch.randn((700, n, 3), device="cuda:0", generator=gen) * sig
    e = Env(n, device="cuda:0", seed=3)
    e.reset()
    h = hashlib.sha256()
    for k in range(7):
        o = e.rollout(T, pool[k * T:(k + 1) * T].contiguous())
        for key in ("obs", "reward", "done", "success"):
            h.update(o[key].cpu().numpy().tobytes())
    for _ in range(30):
        ob, r, d, s = e.step(pool[0])
        h.update(ob.cpu().numpy().tobytes()); h.update(r.cpu().numpy().tobytes())
    st = e.get_state()
    for key in sorted(st):
        h.update(st[key].cpu().numpy().tobytes())
    out[task] = h.hexdigest()[:16]
print("DIGEST", out)
'''
for lib in (sys.argv[1] if len(sys.argv) > 1 else "drl-on-robot-arm_amd/build/ab/libarmenv_base.so", None):
    env = dict(os.environ)
    if lib: env["ARMENV_LIB"] = os.path.join(ROOT, lib)
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print(lib or "current", [l for l in r.stdout.splitlines() if l.startswith("DIGEST")], r.stderr[-300:] if r.returncode else "")
