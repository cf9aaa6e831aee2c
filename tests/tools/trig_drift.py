"""How far does a long rollout launch (carried (cos q, sin q)) drift from one-step launches (re-derived every step)?
4 096 reach envs, 500-step episodes, i.i.d. actions; max |obs difference| per 100-step block, f64 and f32 engines."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "drl-on-robot-arm_amd"))
from armenv.envs.batched import BatchedReachEnv
n, T = 4096, 500
dev = "cuda:0"
g = torch.Generator(device=dev); g.manual_seed(3)
acts = (torch.randn((T, n, 3), device=dev, generator=g) * 0.686).clamp_(-0.7, 0.7)
for prec in (64, 32):
    a = BatchedReachEnv(n, device=dev, seed=9, precision=prec); b = BatchedReachEnv(n, device=dev, seed=9, precision=prec)
    a.reset(); b.reset()
    out = a.rollout(T, acts)
    worst, flags = [], 0
    for t in range(T):
        o, r, d, s = b.step(acts[t])
        worst.append((out["obs"][t] - o).abs().max().item()); flags += int((out["done"][t] != d).sum().item())
    w = np.array(worst)
    print(f"f{prec}: max |obs diff| per 100-step block: " + " ".join(f"{w[k:k+100].max():.2e}" for k in range(0, T, 100)) + f" | done flags differing: {flags}")
    a.close(); b.close()
