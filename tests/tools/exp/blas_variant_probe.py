"""Does the BLAS backend decide whether the hipGraph-replayed TD3 update learns?  (round 6; profiles/r06_td3_hipgraph_learning.txt)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "drl-on-robot-arm_amd"))
import torch
which = sys.argv[1]
if which != "default":
    torch.backends.cuda.preferred_blas_library(which)
print("preferred blas:", torch.backends.cuda.preferred_blas_library(), flush=True)
from armenv import train
for seed in (0, 1):
    hist = []
    t0 = time.perf_counter()
    train.train_reach(iterations=160, log_every=20, log=lambda s: hist.append(json.loads(s)), use_graphs=True, seed=seed)
    print("td3 hipGraphs blas=%s seed %d: %s  %.1f s" % (which, seed, [round(h["success_rate"], 2) for h in hist], time.perf_counter() - t0), flush=True)
