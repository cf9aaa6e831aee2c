"""Per-launch time of the headline rollout (65 536 reach envs, f64, 100 steps per launch, i.i.d. action pool) from a cold
start: how long does the GPU take to reach its steady clock, i.e. how much warm-up does a steady-state figure need?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "drl-on-robot-arm_amd"))
from armenv.envs.batched import BatchedReachEnv
n, T, L = 65536, 100, 60
dev = torch.device("cuda:0")
env = BatchedReachEnv(n, device=dev, precision=64, seed=0)
g = torch.Generator(device=dev); g.manual_seed(1000)
pool = (torch.randn((1000, n, 3), device=dev, generator=g) * 0.686).clamp_(-0.7, 0.7)
env.reset(); bufs = {}
torch.cuda.synchronize()
time.sleep(2.0)                       # let the GPU fall back to its idle state
ev = [torch.cuda.Event(enable_timing=True) for _ in range(L + 1)]
ev[0].record()
for k in range(L):
    a = pool[(k * T) % 1000:(k * T) % 1000 + T]
    env.rollout(T, a, out=bufs); ev[k + 1].record()
torch.cuda.synchronize()
dt = [ev[k].elapsed_time(ev[k + 1]) * 1e3 / T for k in range(L)]
print("cold start (2 s idle before the first launch):")
for r in range(0, L, 10):
    print(f"launches {r:2d}-{r+9:2d} (steps {r*T:5d}+): " + " ".join(f"{x:5.2f}" for x in dt[r:r + 10]) + "  us per step")
# the same workload again from a fresh reset, GPU warm: separates the clock ramp from the evolution of the env population
env.reset()
ev[0].record()
for k in range(L):
    a = pool[(k * T) % 1000:(k * T) % 1000 + T]
    env.rollout(T, a, out=bufs); ev[k + 1].record()
torch.cuda.synchronize()
dt = [ev[k].elapsed_time(ev[k + 1]) * 1e3 / T for k in range(L)]
print("same workload from a fresh reset, GPU warm:")
for r in range(0, L, 10):
    print(f"launches {r:2d}-{r+9:2d} (steps {r*T:5d}+): " + " ".join(f"{x:5.2f}" for x in dt[r:r + 10]) + "  us per step")
