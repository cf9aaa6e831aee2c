"""Round 6: the TD3 update replayed from hipGraphs does not learn the reach task; the eager one does (profiles/r06_td3_hipgraph_learning.txt).
python tests/tools/exp/train_bisect.py  (one MI355X, ~1 min)"""
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
for seed in (0, 1, 2):
    run("td3 eager seed %d" % seed, use_graphs=False, seed=seed)
    run("td3 hipGraphs seed %d" % seed, use_graphs=True, seed=seed)
for seed in (0, 1, 2):
    run("daddpg hipGraphs seed %d" % seed, use_graphs=True, algo="daddpg", seed=seed)
run("daddpg eager seed 0", use_graphs=False, algo="daddpg")
for pf in (1, 2):
    class PF(td3.TD3):
        def __init__(self, *a, **k):
            super().__init__(*a, policy_freq=pf, **k)
    train.TD3 = PF
    run("td3 hipGraphs policy_freq %d" % pf, use_graphs=True)
    run("td3 eager     policy_freq %d" % pf, use_graphs=False)
train.TD3 = td3.TD3
# replays of ONE captured update from an identical state: the same numbers?
dev = "cuda:0"; B = 2048
torch.manual_seed(0)
G = td3.TD3(6, 3, 0.7, policy_noise=0.0)
buf = G.capture(B)
gen = torch.Generator(device=dev); gen.manual_seed(1)
def snap():
    return ([{k: v.clone() for k, v in n.state_dict().items()} for n in G._nets()], [[{k: v.clone() for k, v in st.items()} for st in o.state.values()] for o in G._opts()])
def restore(s):
    with torch.no_grad():
        for n, sd in zip(G._nets(), s[0]):
            for k, v in n.state_dict().items():
                v.copy_(sd[k])
        for o, sts in zip(G._opts(), s[1]):
            for st, saved in zip(o.state.values(), sts):
                for k, v in st.items():
                    v.copy_(saved[k])
flat = lambda: torch.cat([p.detach().flatten() for n in G._nets() for p in n.parameters()]).clone()
args = lambda flag: (buf["states"], buf["actions"], buf["rewards"].view(-1, 1), buf["next_states"], buf["dones"].to(torch.float32).view(-1, 1), flag)
gg = ee = ge = 0
N = 60
for it in range(1, N + 1):
    for k, v in buf.items():
        v.copy_((torch.rand(v.shape, device=dev, generator=gen) * (1.1 if v.dtype == torch.uint8 else 1)).to(v.dtype))
    G.total_it += 1
    flag = G._flag()
    s0 = snap()
    G._graphs["g"][flag].replay(); A = flat()
    restore(s0); G._graphs["g"][flag].replay(); Bv = flat()
    restore(s0); G._update(*args(flag)); C = flat()
    restore(s0); G._update(*args(flag)); D = flat()
    gg += int(not torch.equal(A, Bv)); ee += int(not torch.equal(C, D)); ge += int(not torch.equal(A, C))
print("one update from an identical state, %d states: two replays of the graph differ on %d, two eager updates on %d, graph vs eager on %d" % (N, gg, ee, ge))
