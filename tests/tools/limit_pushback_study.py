#!/usr/bin/env python3
"""How the limit term of the parity fence changes under the joint-limit models (CPU oracle, reach, 8 192 envs, run() exploration
noise, steps 600..1200): clamp_joint_limits = 0 (reference behaviour: nothing), 2 (push back by limit_erp = 0.2 of the violation per
step), 1 (hard projection).  Recorded in profiles/r03_limit_pushback.txt."""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as O
kuka = O.make_chain("kuka"); n = 8192
lim = np.array(O.KUKA["limit"])
for mode in (0, 2, 1):
    cfg = O.default_config(); cfg.clamp_joint_limits = mode
    raw = O.default_config()
    st = O.ReachState(n); O.reach_reset(kuka, cfg, st, seed=0)
    rng = np.random.default_rng(1)
    hits = tot = 0; exc = []; post = 0
    for t in range(1200):
        a = np.clip(rng.standard_normal((n, 3)) * 0.686, -0.7, 0.7).astype(np.float32)
        if t >= 600:
            p0, _ = O.fk(kuka, st.q)
            tgt = np.clip(p0 + 0.02 * a.astype(np.float64), [0.2, -0.3, 0], [0.7, 0.3, 0.55])
            qraw, _ = O.ik(kuka, raw, st.q, tgt)
            v = np.clip(np.abs(qraw) - lim, 0, None).max(1)
            hits += int((v > 0).sum()); tot += n; exc.append(v[v > 0])
        O.reach_step_autoreset(kuka, cfg, st, a, seed=0, want_terminal=False)
        if t >= 600: post += int((np.clip(np.abs(st.q) - lim, 0, None).max(1) > 0).sum())
    e = np.concatenate(exc)
    print("clamp mode", mode, "IK result outside limits: %.4f of steps; median excess %.3f rad, p90 %.3f, max %.3f; state outside limits after the step: %.4f" % (hits / tot, np.median(e), np.quantile(e, 0.9), e.max(), post / tot))
