#!/usr/bin/env python3
"""Host-side cost of the bench's timed bracket around ONE 20-step rollout launch (the driver's --steps 20): wall clock from
before the launch to after the closing synchronise, for several ways of waiting, against the kernel's own duration.
Usage: sync_latency.py [envs] [steps] [repeats]"""
import ctypes as C, os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "drl-on-robot-arm_amd"))
sys.path.insert(0, ROOT)
import torch
from armenv import envs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 20
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
dev = torch.device("cuda", 0)
env = envs.BatchedReachEnv(n, device=dev, seed=0)
gen = torch.Generator(device=dev); gen.manual_seed(1000)
pool = (torch.randn((1000, n, 3), device=dev, generator=gen) * 0.686).clamp_(-0.7, 0.7)
env.reset()
bufs = {}
launches = [env.bind_rollout(T, pool[k * T:(k + 1) * T], out=bufs)[0] for k in range(1000 // T)]
path = next(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l)
h = C.CDLL(path)
for f in (h.hipEventCreate, h.hipEventRecord, h.hipEventSynchronize, h.hipEventQuery, h.hipStreamSynchronize, h.hipStreamQuery, h.hipEventElapsedTime):
    f.restype = C.c_int
h.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
h.hipEventSynchronize.argtypes = [C.c_void_p]
h.hipEventQuery.argtypes = [C.c_void_p]
h.hipStreamSynchronize.argtypes = [C.c_void_p]
h.hipStreamQuery.argtypes = [C.c_void_p]
h.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
ev = [C.c_void_p(), C.c_void_p()]
for e in ev:
    h.hipEventCreate(C.byref(e))
for l in launches[:20]:
    l()
torch.cuda.synchronize()
p = time.perf_counter


def run(mode, events):
    walls, kern = [], []
    for r in range(reps):
        l = launches[r % len(launches)]
        h.hipStreamSynchronize(st)
        t0 = p()
        if events:
            h.hipEventRecord(ev[0], st)
        l()
        if events:
            h.hipEventRecord(ev[1], st)
        if mode == "event_sync":
            h.hipEventSynchronize(ev[1]); h.hipStreamSynchronize(st)
        elif mode == "event_spin":
            while h.hipEventQuery(ev[1]) != 0:
                pass
            h.hipStreamSynchronize(st)
        elif mode == "stream_sync":
            h.hipStreamSynchronize(st)
        elif mode == "stream_spin":
            while h.hipStreamQuery(st) != 0:
                pass
        walls.append((p() - t0) * 1e6)
        if events:
            ms = C.c_float(); h.hipEventElapsedTime(C.byref(ms), ev[0], ev[1]); kern.append(ms.value * 1e3)
    return statistics.median(walls), min(walls), (statistics.median(kern) if kern else float("nan"))


for mode, events in (("event_sync", True), ("event_spin", True), ("stream_sync", True), ("stream_spin", True), ("stream_sync", False), ("stream_spin", False)):
    med, mn, k = run(mode, events)
    print(f"{mode:12s} events={int(events)}: wall median {med:7.1f} us  min {mn:7.1f} us  kernel (events) median {k:7.1f} us", flush=True)
