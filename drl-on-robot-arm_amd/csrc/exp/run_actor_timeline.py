"""Cycle attribution of the fused f16x3 actor inside the rollout kernel (instrumented build: `make actor_timeline`).
s_memtime deltas per section, summed over the 1024 waves; printed per wave-step in cycles of the 100 MHz counter -> us."""
import sys, os, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, '..', '..'))
from armenv import _lib as L
L.LIB_PATH = os.path.join(ROOT, 'libarmenv_atl.so')
from armenv import envs
n, T = 65536, 20
g = np.load(os.path.join(ROOT, '..', '..', '..', 'tests', 'golden', 'td3_actor_seed0.npz'))
sd = {k: torch.from_numpy(g[k.replace('.', '_')]) for k in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc3.weight", "fc3.bias")}
e = envs.BatchedReachEnv(n, device='cuda:0')
e.set_policy('actor_f16x3', actor_state_dict=sd)
lib = L.load()
lib.armenv_dbg_actor_sections.argtypes = [C.POINTER(C.c_uint64 * 16), C.c_int]
buf = (C.c_uint64 * 16)()
e.reset()
e.rollout(T, None); torch.cuda.synchronize()
lib.armenv_dbg_actor_sections(C.byref(buf), 1)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record(); e.rollout(T, None); ev1.record(); torch.cuda.synchronize()
lib.armenv_dbg_actor_sections(C.byref(buf), 1)
v = np.array(list(buf), dtype=np.float64) / (n // 64) / T      # per wave per step, in s_memtime ticks (100 MHz)
names = ['entry vmcnt(0)', 'fill issue', 'counted vmcnt wait', 'barrier', 'LDS read issue', 'MFMA blocks (+split)', 'pass prologue',
         'pass epilogue', 'tanh']
print(f'instrumented step: {ev0.elapsed_time(ev1) * 1e3 / T:.1f} us')
for nme, x in zip(names, v):
    print(f'   {nme:22s} {x / 100.0:7.2f} us')
print(f'   {"sum":22s} {v[:9].sum() / 100.0:7.2f} us')
