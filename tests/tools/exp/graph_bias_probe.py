"""Minimal reproduction attempt: the first layer's bias gradient under hipGraph capture (round 6; profiles/r06_td3_hipgraph_learning.txt)"""
import os, sys
import torch
import torch.nn as nn
import torch.nn.functional as F
dev = "cuda:0"; B = 2048
torch.manual_seed(0)
def trial(tag, input_requires_grad=False, use_cat=False, layers=3, explicit_bias=False):
    net = nn.Sequential(nn.Linear(9 if use_cat else 6, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 1)).to(dev)
    params = list(net.parameters())
    x6 = torch.zeros(B, 6, device=dev); x3 = torch.zeros(B, 3, device=dev); y = torch.zeros(B, 1, device=dev)
    gbuf = [torch.zeros_like(p) for p in params]
    def step():
        x = torch.cat([x6, x3], dim=1) if use_cat else x6
        if input_requires_grad:
            x = x.detach().requires_grad_(True)
        loss = F.mse_loss(net(x), y)
        gs = torch.autograd.grad(loss, params)
        for b_, g_ in zip(gbuf, gs):
            b_.copy_(g_)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    bad = {}
    for it in range(30):
        x6.copy_(torch.rand(B, 6, device=dev, generator=gen)); x3.copy_(torch.rand(B, 3, device=dev, generator=gen)); y.copy_(torch.rand(B, 1, device=dev, generator=gen))
        # other eager work between replays, like a training loop has (allocations, GEMMs)
        junk = torch.rand(512, 512, device=dev, generator=gen) @ torch.rand(512, 512, device=dev, generator=gen)
        g.replay(); gg = [b_.clone() for b_ in gbuf]
        step(); ge = [b_.clone() for b_ in gbuf]
        for (n_, _), a_, b_ in zip(net.named_parameters(), gg, ge):
            d = float((a_ - b_).abs().max())
            if d > 1e-5:
                bad[n_] = max(bad.get(n_, 0.0), d)
    print("%-46s gradients that differ graph vs eager (max |diff| over 30 replays): %s" % (tag, bad or "none"), flush=True)
trial("6 inputs, input does not require grad")
trial("cat(6, 3) inputs, input does not require grad", use_cat=True)
trial("6 inputs, input requires grad", input_requires_grad=True)
trial("cat(6, 3) inputs, input requires grad", use_cat=True, input_requires_grad=True)
