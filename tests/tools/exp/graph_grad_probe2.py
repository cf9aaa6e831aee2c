"""After the fix (no multi-block reduction inside the captured update): do the captured update's gradients equal the eager ones, and do
the replayed learners learn?  (round 6; profiles/r06_td3_hipgraph_learning.txt)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "drl-on-robot-arm_amd"))
import torch
from armenv import td3, train
from armenv.daddpg import DADDPG
dev = "cuda:0"; B = 2048
def count(mk, n=150):
    torch.manual_seed(0)
    G = mk()
    buf = G.capture(B)
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    def snap():
        return ([{k: v.clone() for k, v in n_.state_dict().items()} for n_ in G._nets()], [[{k: v.clone() for k, v in st.items()} for st in o.state.values()] for o in G._opts()])
    def restore(s):
        with torch.no_grad():
            for n_, sd in zip(G._nets(), s[0]):
                for k, v in n_.state_dict().items():
                    v.copy_(sd[k])
            for o, sts in zip(G._opts(), s[1]):
                for st, saved in zip(o.state.values(), sts):
                    for k, v in st.items():
                        v.copy_(saved[k])
    names = [(str(i) + "." + k, p) for i, n_ in enumerate(G._nets()[:3]) for k, p in n_.named_parameters() if p.grad is not None or True]
    args = lambda flag: (buf["states"], buf["actions"], buf["rewards"].view(-1, 1), buf["next_states"], buf["dones"].to(torch.float32).view(-1, 1), flag)
    bad, worst, lossbad = 0, 0.0, 0
    for it in range(1, n + 1):
        for k, v in buf.items():
            v.copy_((torch.rand(v.shape, device=dev, generator=gen) * (1.1 if v.dtype == torch.uint8 else 1)).to(v.dtype))
        G.total_it += 1
        flag = G._flag()
        s0 = snap()
        G._graphs["g"][flag].replay(); gg = {k: p.grad.clone() for k, p in names if p.grad is not None}; lg = float(G._graphs["loss"])
        restore(s0)
        le = float(G._update(*args(flag))); ge = {k: p.grad.clone() for k, p in names if p.grad is not None}
        w = max(float((gg[k] - ge[k]).abs().max()) for k in gg)
        worst = max(worst, w); bad += int(w > 1e-4); lossbad += int(abs(lg - le) > 1e-5 * max(1.0, abs(le)))
    print("%-8s updates (of %d) with a gradient off by > 1e-4: %d (worst difference %.1e); with a different loss value: %d" % (type(G).__name__, n, bad, worst, lossbad), flush=True)
count(lambda: td3.TD3(6, 3, 0.7, policy_noise=0.0))
count(lambda: DADDPG(6, 3, 0.7))
def run(tag, **kw):
    hist = []
    t0 = time.perf_counter()
    train.train_reach(iterations=160, log_every=20, log=lambda s: hist.append(json.loads(s)), **kw)
    print("%-32s success rate per 20 iterations %s   %.1f s" % (tag, [round(h["success_rate"], 2) for h in hist], time.perf_counter() - t0), flush=True)
for seed in (0, 1, 2):
    run("td3 hipGraphs seed %d" % seed, use_graphs=True, seed=seed)
for seed in (0, 1, 2):
    run("daddpg hipGraphs seed %d" % seed, use_graphs=True, seed=seed, algo="daddpg")
run("td3 eager seed 0", use_graphs=False)
run("daddpg eager seed 1", use_graphs=False, seed=1, algo="daddpg")
