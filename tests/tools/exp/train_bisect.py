import sys, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "drl-on-robot-arm_amd"))
import torch
from armenv import td3
dev = "cuda:0"; B = 2048
torch.manual_seed(0)
G = td3.TD3(6, 3, 0.7, policy_noise=0.0)
buf = G.capture(B)
print("reserved after capture MB", torch.cuda.memory_reserved() / 1e6, "allocated", torch.cuda.memory_allocated() / 1e6, flush=True)
shapes = [(256, 256), (2048, 256), (256,), (2048, 6), (2048, 9), (2048, 1), (256, 9), (3, 256), (2048, 3), (1,), (2048,), (512, 512), (4096, 256)]
sent = []
for rep in range(40):
    for sh in shapes:
        sent.append(torch.full(sh, 1.0, device=dev))
# some freed again, to mix the free lists
del sent[::3]
torch.cuda.synchronize()
gen = torch.Generator(device=dev); gen.manual_seed(1)
for it in range(1, 61):
    for k, v in buf.items():
        v.copy_((torch.rand(v.shape, device=dev, generator=gen) * (2 if v.dtype == torch.uint8 else 1)).to(v.dtype))
    G.total_it += 1
    G._graphs["g"][G._flag()].replay()
    tmp = [torch.full(sh, 1.0, device=dev) for sh in shapes]      # eager traffic between replays
    sent.extend(tmp[:2])
torch.cuda.synchronize()
bad = [(i, tuple(t.shape), float((t - 1.0).abs().max())) for i, t in enumerate(sent) if not bool((t == 1.0).all())]
print("sentinel tensors overwritten by graph replays:", len(bad), bad[:8], flush=True)
# where do the graph pools live relative to the sentinels?
print(torch.cuda.memory_summary(abbreviated=True)[:1500])
