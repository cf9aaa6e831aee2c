"""Which Adam implementation survives hipGraph replay?  (round 6; see profiles/r06_td3_hipgraph_learning.txt)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "drl-on-robot-arm_amd"))
import torch
from armenv import train, td3
def run(tag, **kw):
    hist = []
    t0 = time.perf_counter()
    train.train_reach(iterations=160, log_every=20, log=lambda s: hist.append(json.loads(s)), **kw)
    print("%-44s success rate per 20 iterations %s   %.1f s" % (tag, [round(h["success_rate"], 2) for h in hist], time.perf_counter() - t0), flush=True)
def variant(**adam_kw):
    class V(td3.TD3):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.actor_opt = torch.optim.Adam(self.actor.parameters(), lr=1e-3, **adam_kw)
            self.critic_opt = torch.optim.Adam(self.critic.parameters(), lr=1e-3, **adam_kw)
    return V
for tag, kw in (("foreach=False capturable", dict(foreach=False, capturable=True)), ("fused=True", dict(fused=True, capturable=True))):
    train.TD3 = variant(**kw)
    for seed in (0, 1):
        try:
            run("td3 hipGraphs Adam(%s) seed %d" % (tag, seed), use_graphs=True, seed=seed)
        except Exception as e:
            print(tag, "failed:", type(e).__name__, str(e)[:200], flush=True)
