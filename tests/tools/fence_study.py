#!/usr/bin/env python3
"""Where do two implementations of the reference's IK part ways?  (CPU oracle only, no GPU.)

The oracle carries two algebraically identical forms of Bullet's damped-least-squares update: the primal 7x7
(J^T J + lambda I) dtheta = J^T e solved by Gaussian elimination (BussIK, `ik_form = 0`) and the dual 6x6
dtheta = J^T (J J^T + lambda I)^-1 e (`ik_form = 1`, the form the HIP kernels use).  Run free (auto-reset, the tasks' own
exploration noise) on the same actions they differ by rounding only -- and still end up radians apart in joint space on the
push task within a few hundred steps.  This script measures where that starts, which is what the parity fence's `cap` and
`cond` terms name: an env is compared from a reset until its first IK call that ran to the iteration
cap or whose damped system had an LDL^T pivot below `fence_pivot`.

Usage: fence_study.py [task] [envs] [steps] [pivot thresholds ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O

task = sys.argv[1] if len(sys.argv) > 1 else "push"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
S = int(sys.argv[3]) if len(sys.argv) > 3 else 600
kappas = [float(x) for x in sys.argv[4:]] or [0.0, 1e-3, 3e-3, 5e-3, 1e-2, 2e-2]
kuka = O.make_chain("kuka")
St, reset, stepf = dict(reach=(O.ReachState, O.reach_reset, O.reach_step_autoreset), push=(O.PushState, O.push_reset, O.push_step_autoreset),
                        pick=(O.PickState, O.pick_reset, O.pick_step_autoreset))[task]
cfgA, cfgB = O.default_config(task), O.default_config(task); cfgB.ik_form = 1
A, B = St(n), St(n); reset(kuka, cfgA, A, seed=7); reset(kuka, cfgB, B, seed=7)
rng = np.random.default_rng(3)
sig, clip = (0.686, 0.7) if task == "reach" else (0.392, 1e9)
itA, itB, mp = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n)
rec = dict(its=[], mp=[], dobs=[], dq=[], drew=[], dflag=[], done=[], dupd=[])
kw = dict(want_terminal=False) if task == "reach" else {}
for t in range(S):
    a = np.clip(rng.standard_normal((n, 3)) * sig, -clip, clip).astype(np.float32)
    oA, rA, dA, sA, _ = stepf(kuka, cfgA, A, a, seed=7, iters=itA, minpiv=mp, **kw)
    oB, rB, dB, sB, _ = stepf(kuka, cfgB, B, a, seed=7, iters=itB, **kw)
    for k, v in zip(rec, (itA.copy(), mp.copy(), np.abs(oA - oB).max(1), np.abs(A.q - B.q).max(1), np.abs(rA - rB), (dA != dB) | (sA != sB),
                          dA.astype(bool) & dB.astype(bool), itA != itB)):
        rec[k].append(v)
rec = {k: np.stack(v) for k, v in rec.items()}
print(f"{task}: {n} envs x {S} steps, primal (BussIK 7x7 Gauss) vs dual (6x6) form of the same update, free-running with auto-reset")
print(f"  IK calls at the iteration cap: {(rec['its'] >= cfgA.ik_max_iters).mean():.5f}; smallest pivot quantiles (min, 1e-4, 1e-3, 1e-2, median): "
      + " ".join("%.2e" % x for x in np.quantile(rec["mp"], [0, 1e-4, 1e-3, 1e-2, 0.5])))
print(f"  final joint-space difference quantiles (50, 90, 99, 99.9 %, max): " + " ".join("%.1e" % x for x in np.quantile(rec["dq"][-1], [0.5, 0.9, 0.99, 0.999, 1])))
for kappa in kappas:
    clean, sync = np.ones(n, bool), np.ones(n, bool)
    st = dict(chk=0, bad=0, wo=0.0, wq=0.0, wr=0.0, fl=0, up=0)
    for t in range(S):
        bad = (rec["its"][t] >= cfgA.ik_max_iters) | (rec["mp"][t] < kappa)
        st["bad"] += int(bad.sum())
        clean &= ~bad
        chk = clean & sync
        st["chk"] += int(chk.sum())
        if chk.any():
            st["wo"] = max(st["wo"], rec["dobs"][t][chk].max()); st["wq"] = max(st["wq"], rec["dq"][t][chk].max())
            st["fl"] += int(rec["dflag"][t][chk].sum()); st["up"] += int(rec["dupd"][t][chk].sum())
            ok = chk & ~rec["dflag"][t]
            st["wr"] = max(st["wr"], rec["drew"][t][ok].max(initial=0.0))
        sync &= ~rec["dflag"][t]
        clean |= sync & rec["done"][t]
    print(f"  fence_pivot {kappa:<7g} flagged calls {st['bad'] / (S * n):.5f} | compared {st['chk'] / (S * n):.4f} of the env-steps: worst |obs| {st['wo']:.2e} "
          f"|q| {st['wq']:.2e} |reward| {st['wr']:.2e}, flags differing {st['fl']}, update counts differing {st['up']}")
