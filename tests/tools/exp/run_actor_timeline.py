"""Where a fused f16x3 actor pass spends its time: shader-clock stamps (s_memtime) of workgroup 0's four waves at the phases of the
pass, accumulated in scalar registers by class -- prologue, resident k-steps, streamed even / odd k-steps, layer 3 -- from the instrumented build
(`make -C drl-on-robot-arm_amd timeline`), for both env tiles of the LAST env step of a 100-step launch.  GPU.
  python tests/tools/exp/run_actor_timeline.py"""
import sys, os, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "..", "..", "..", "drl-on-robot-arm_amd"))
from armenv import _lib as L
L.LIB_PATH = os.environ.get("ARMENV_TL_LIB", os.path.join(ROOT, "libarmenv_tl.so"))
from armenv import envs
n = 65536
e = envs.BatchedReachEnv(n, device="cuda:0")
lib = L.load()
g = np.load(os.path.join(ROOT, "..", "..", "golden", "td3_actor_seed0.npz"))
sd = {k: torch.from_numpy(g[k.replace(".", "_")]) for k in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc3.weight", "fc3.bias")}
e.set_policy("actor_f16x3", action_bound=0.7, noise_sigma=0.686, noise_clip=0.7, actor_state_dict=sd)
tl = torch.zeros((4, 64), dtype=torch.int64, device="cuda:0")
lib.armenv_dbg_set_actor_timeline.argtypes = [C.c_void_p]
assert lib.armenv_dbg_set_actor_timeline(C.c_void_p(tl.data_ptr())) == 0
e.reset()
b = {}
for _ in range(8):
    e.rollout(100, None, out=b)
torch.cuda.synchronize()
t = tl.cpu().numpy().astype(np.int64)
names = ("prologue (obs operands, layer 1 + split of row tile 0)", "4 resident k-step bodies", "6 streamed even k-step bodies", "6 streamed odd k-step bodies",
         "layer 3 + pass epilogue", "16 heads: s_waitcnt vmcnt", "16 heads: s_barrier")
for w in range(4):
    for p in range(2):
        a = t[w, 8 * p:8 * p + 7]
        print("wave %d env tile %d: %s | total %d cycles" % (w, p, "  ".join("%s %d" % (nm.split(" (")[0], x) for nm, x in zip(names, a)), a.sum()))
a = np.array([t[w, 8 * p:8 * p + 7] for w in range(4) for p in range(2)], dtype=np.float64).mean(0)
tot = a.sum()
print("mean per pass (cycles, share of the pass): prologue %.0f (%.1f %%) | resident k-step body %.0f each | streamed even body %.0f | odd %.0f | layer 3 + epilogue %.0f (%.1f %%) | "
      "head vmcnt wait %.0f per k-step (%.1f %% of the pass) | head barrier %.0f per k-step (%.1f %%) | pass %.0f cycles; the 24 MFMAs of a k-step hold the pipe 768 cycles "
      "(bodies: %.1f %% of the pass, of which matrix pipe 16 x 768 = %.1f %% of the pass)"
      % (a[0], 100 * a[0] / tot, a[1] / 4, a[2] / 6, a[3] / 6, a[4], 100 * a[4] / tot, a[5] / 16, 100 * a[5] / tot, a[6] / 16, 100 * a[6] / tot, tot,
         100 * (a[1] + a[2] + a[3]) / tot, 100 * 16 * 768 / tot))
