"""Which gradient differs between the captured update and the eager one, from an identical state?  (round 6; profiles/r06_td3_hipgraph_learning.txt)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "drl-on-robot-arm_amd"))
import torch
from armenv import td3
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
names = [("critic." + k, p) for k, p in G.critic.named_parameters()] + [("actor." + k, p) for k, p in G.actor.named_parameters()]
grads = lambda: {k: p.grad.clone() for k, p in names if p.grad is not None}
args = lambda flag: (buf["states"], buf["actions"], buf["rewards"].view(-1, 1), buf["next_states"], buf["dones"].to(torch.float32).view(-1, 1), flag)
shown = 0
for it in range(1, 40):
    for k, v in buf.items():
        v.copy_((torch.rand(v.shape, device=dev, generator=gen) * (1.1 if v.dtype == torch.uint8 else 1)).to(v.dtype))
    G.total_it += 1
    flag = G._flag()
    s0 = snap()
    G._graphs["g"][flag].replay(); gg = grads(); lg = float(G._graphs["loss"])
    restore(s0)
    le = float(G._update(*args(flag))); ge = grads()
    worst = max(float((gg[k] - ge[k]).abs().max()) for k in gg)
    if worst > 1e-6 and shown < 4:
        shown += 1
        print("update %d (%s): loss graph %.7f eager %.7f" % (it, "critic + actor" if flag else "critic only", lg, le))
        for k in gg:
            d = (gg[k] - ge[k]).abs()
            print("   %-22s |grad| max %.3e  max diff %.3e  elements differing by > 1e-6: %d of %d" % (k, float(ge[k].abs().max()), float(d.max()), int((d > 1e-6).sum()), d.numel()))
print("done; updates with any gradient difference > 1e-6 shown:", shown)

# what IS the wrong bias gradient?  a few elements over consecutive updates, beside the eager value and the bias itself
print("\nupdate | graph grad[:4] | eager grad[:4] | bias value[:4] | graph - eager [:4]")
p = G.critic.fc1.bias
prev = None
for it in range(40, 52):
    for k, v in buf.items():
        v.copy_((torch.rand(v.shape, device=dev, generator=gen) * (1.1 if v.dtype == torch.uint8 else 1)).to(v.dtype))
    G.total_it += 1
    flag = G._flag()
    s0 = snap()
    G._graphs["g"][flag].replay(); gg = p.grad.clone()
    restore(s0)
    G._update(*args(flag)); ge = p.grad.clone()
    f = lambda t: "[" + ", ".join("%+.4f" % float(x) for x in t[:4]) + "]"
    print(it, "T" if flag else "F", f(gg), f(ge), f(s0[0][1]["fc1.bias"]), f(gg - ge), " ratio (graph-eager)/bias:", f((gg - ge) / s0[0][1]["fc1.bias"]))
