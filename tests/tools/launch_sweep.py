"""Step-API launch cost against batch size: one armenv_step launch per env step, 50 launches per hipGraph, timed with
events over 20 replays.  Separates what a launch costs (dispatch + drain of the 512-register waves) from what the
kernel computes.  Usage: python tests/tools/launch_sweep.py [precision]"""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "drl-on-robot-arm_amd"))
from armenv.envs.batched import BatchedReachEnv

prec = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
print(f"{'envs':>8} {'waves':>6} {'policy':>8} {'us/launch(graph)':>17} {'us/launch(eager)':>17}")
for n in (64, 1024, 4096, 16384, 32768, 65536, 131072, 262144):
    for pol in ("zero", "random"):
        env = BatchedReachEnv(n, device=dev, precision=prec, seed=1)
        env.reset()
        a = torch.zeros(n, 3, device=dev) if pol == "zero" else (torch.rand(n, 3, device=dev) * 2 - 1)
        def run(k):
            for _ in range(k):
                env.step(a)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            run(4)
        torch.cuda.current_stream(dev).wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            run(50)
        g.replay(); torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record(); torch.cuda.synchronize(dev)
        tg = e0.elapsed_time(e1) * 1e3 / 1000
        e0.record(); run(500); e1.record(); torch.cuda.synchronize(dev)
        te = e0.elapsed_time(e1) * 1e3 / 500
        print(f"{n:8d} {(n + 63) // 64:6d} {pol:>8} {tg:17.2f} {te:17.2f}")
        env.close()
