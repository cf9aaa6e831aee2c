import torch
dev = "cuda:0"
x = torch.zeros(8, device=dev)
out = torch.zeros(8, device=dev)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        out.copy_(torch.randn_like(x))
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out.copy_(torch.randn_like(x))
vals = []
for _ in range(4):
    g.replay(); torch.cuda.synchronize(); vals.append(out.clone().cpu())
print("replays differ:", [bool((vals[0] != v).any()) for v in vals[1:]])
print(vals[0][:4], vals[1][:4])
# and with a clamp chain like TD3's
a = torch.zeros(4, 3, device=dev)
g2 = torch.cuda.CUDAGraph()
o2 = torch.zeros(4, 3, device=dev)
with torch.cuda.graph(g2):
    o2.copy_((torch.randn_like(a) * 0.2).clamp(-0.5, 0.5))
r = []
for _ in range(3):
    g2.replay(); torch.cuda.synchronize(); r.append(o2.clone().cpu())
print("TD3-style noise differs:", bool((r[0] != r[1]).any()), bool((r[1] != r[2]).any()))
