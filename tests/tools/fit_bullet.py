#!/usr/bin/env python3
"""Fits the named switches of the IK / stepSimulation restatement to what real PyBullet computed.

Two sources of truth, used in this order:

 (A) ALWAYS: the reference's recorded run.  tests/golden/visdata_reach_td3.json holds the per-episode returns of one
     `train_reach_with_TD3` run of the reference (real PyBullet, seed 0); its first five episodes -- 2 068 free-running env steps,
     276 of them with a joint beyond its URDF limit and 274 with the flange below z = 0.05 -- are reproducible from seeded streams
     (tests/reference_run.py).  Every combination of
         ik_exit_mode {0, 1} x ik_angle_f32 {0, 1} x ik_form {primal, dual} x ik_tip_offset {link frame, inertial (0,0,0.02)}
         x clamp_joint_limits {0, 1, 2 at limit_erp 0.05 / 0.2 / 0.5}
     is replayed on the CPU oracle and ranked by the worst |return - recorded return| over the five episodes.

 (B) WHERE `import pybullet` WORKS (not in the build image, not on the GPU box): the same sweep against `_BulletReach`
     (tests/test_pybullet_oracle.py: the reference's own p.* call sequence in DIRECT mode), teacher-forced from Bullet's joint
     state over --steps steps that include limit / flange steps; prints the best setting and its worst |dq|.
     Elsewhere this part prints "pybullet unavailable".

 (C) ALWAYS: the cube of the push task against the reference's TWO recorded push runs (tests/reference_run.py: the same first five
     episodes under two rewards give, per episode, the final cube-target distance d_f and the number M of steps on which that distance
     changed by >= 1e-5).  The planar contact model (ArmEnvConfig.push_contact_model = 1; oracle push_contact_dyn) is swept over
         push_tool_radius x push_friction x push_contact_erp x push_tool_below
     and every setting is ranked by the worst |d_f - recorded| over the four touched episodes (x 50 = the error of the first run's
     return) and by the sum of |M - recorded| (= the error of the second run's returns).  Bullet's own values -- contact ERP 0.2,
     friction 5 x 0.5 -- are rows of the table.  --push-only skips (A) and (B).

Usage: python tests/tools/fit_bullet.py [--steps 10000] [--push-only]        (CPU only)"""
import argparse
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np

from oracle import oracle as O
import reference_run as R

INERTIAL = (0.0, 0.0, 0.02)


def settings():
    clamps = [(0, 0.2), (1, 0.2), (2, 0.05), (2, 0.2), (2, 0.5)]
    for em, af, form, tip, (cl, erp) in itertools.product((0, 1), (1, 0), (0, 1), ((0.0, 0.0, 0.0), INERTIAL), clamps):
        yield dict(ik_exit_mode=em, ik_angle_f32=af, ik_form=form, ik_tip_offset=tip, clamp_joint_limits=cl, limit_erp=erp)


def apply(cfg, s):
    for k, v in s.items():
        if k == "ik_tip_offset":
            cfg.ik_tip_offset[:] = list(v)
        else:
            setattr(cfg, k, v)


def label(s):
    return ("exit_mode %d  angle_f32 %d  %s  tip %-8s  clamp %d%s" % (
        s["ik_exit_mode"], s["ik_angle_f32"], "dual  " if s["ik_form"] else "primal", "inertial" if s["ik_tip_offset"][2] else "link",
        s["clamp_joint_limits"], (" erp %.2f" % s["limit_erp"]) if s["clamp_joint_limits"] == 2 else "         "))


def fit_recorded_run():
    fx = R.fixture_returns()
    rows = []
    for s in settings():
        out, fence = R.replay_on_oracle(O, 5, lambda c, s=s: apply(c, s))
        diffs = [abs(r - x) for (r, _, _), x in zip(out, fx)]
        rows.append((max(diffs), [n for _, n, _ in out], s))
    rows.sort(key=lambda r: r[0])
    print("(A) the reference's recorded run (visdata/reach/TD3_0.01/Reach_TD3.json, first five episodes = 2 068 env steps of real PyBullet):")
    print("    worst |return - recorded| over the five episodes, all %d switch settings, best first" % len(rows))
    for w, lens, s in rows[:12]:
        print("    %10.3e   %s   episode lengths %s" % (w, label(s), lens))
    print("    ...")
    for w, lens, s in rows[-3:]:
        print("    %10.3e   %s   episode lengths %s" % (w, label(s), lens))
    good = [r for r in rows if r[0] < 10 * rows[0][0]]
    keys = ("ik_exit_mode", "ik_tip_offset", "clamp_joint_limits", "ik_angle_f32", "ik_form")
    print("    settings within 10x of the best (%d): " % len(good) + "; ".join(
        "%s in %s" % (k, sorted({str(r[2][k]) for r in good})) for k in keys))
    print("    => decided by the data: " + ", ".join(k for k in keys if len({str(r[2][k]) for r in good}) == 1)
          + "; not distinguishable at this level: " + ", ".join(k for k in keys if len({str(r[2][k]) for r in good}) > 1))
    return rows


def fit_pybullet(steps):
    try:
        import pybullet  # noqa: F401
        import pybullet_data  # noqa: F401
    except Exception as e:      # noqa: BLE001
        print("(B) pybullet unavailable (%s): the teacher-forced sweep against the real engine is skipped here" % e)
        return
    from test_pybullet_oracle import _BulletReach
    kuka = O.make_chain("kuka")
    b = _BulletReach()
    b.reset()
    rng = np.random.default_rng(0)
    rec = []
    for t in range(steps):
        if t % 501 == 0:
            b.reset()
        a = np.clip(rng.normal(0, 0.686, 3), -0.7, 0.7)
        q0 = b.q()
        pos, jp = b.step(a)
        rec.append((q0, a, b.q(), pos))
    b.close()
    q0 = np.array([r[0] for r in rec]); a = np.array([r[1] for r in rec]); q1 = np.array([r[2] for r in rec])
    lim = np.array(O.KUKA["limit"])
    rows = []
    for s in settings():
        cfg = O.default_config(); apply(cfg, s)
        p0, _ = O.fk(kuka, q0)
        tgt = np.clip(p0 + 0.02 * a, cfg.box_lo[:], cfg.box_hi[:])
        qn, _ = O.ik(kuka, cfg, q0, tgt)
        d = np.abs(qn - q1).max(1)
        fenced = (np.abs(qn) > lim).any(1) | (O.fk(kuka, qn)[0][:, 2] < 0.05)
        rows.append((float(d.max()), float(d[~fenced].max(initial=0.0)), float(d[fenced].max(initial=0.0)), int(fenced.sum()), s))
    rows.sort(key=lambda r: r[0])
    print("(B) teacher-forced against pybullet over %d steps: worst |dq| (all / outside the limit+flange steps / inside them, count)" % steps)
    for w, wo, wi, nf, s in rows[:10]:
        print("    %10.3e  %10.3e  %10.3e  %6d   %s" % (w, wo, wi, nf, label(s)))
    print("    best setting: " + label(rows[0][4]))


def _push_row(par):
    r, mu, erp, below = par

    def conf(c):
        c.push_contact_model = 1
        c.push_tool_radius, c.push_friction, c.push_contact_erp, c.push_tool_below = r, mu, erp, below
    out = R.replay_push_on_oracle(O, 5, conf)
    rec = R.push_recorded_observables(5)
    org, upd = R.push_fixture_returns("origin"), R.push_fixture_returns("updata")
    touched = (0, 1, 2, 4)
    e_u = [abs(out[k]["ret"] - upd[k]) for k in touched]            # the shipped reward's returns (the "updata" run)
    e_o = [abs(out[k]["ret_origin"] - org[k]) for k in touched]     # the earlier reward's returns = 50 x the final-distance error
    e_m = [abs(out[k]["M"] - rec[k][1]) for k in touched]
    ok4 = out[3]["M"] == 8 and abs(out[3]["ret"] - upd[3]) < 2e-3
    return par, max(e_u), max(e_o), sum(e_m), [o["M"] for o in out], [o["d_f"] - o["planar"] for o in out], ok4


def fit_push():
    from multiprocessing import Pool
    rec = R.push_recorded_observables(5)
    org, upd = R.push_fixture_returns("origin"), R.push_fixture_returns("updata")
    print("(C) the cube of the push task against the two recorded push runs (first five episodes; the arm touches the cube in 1, 2, 3, 5):")
    print("    recorded: M = %s   d_f - placement distance = %s" % ([round(m) for _, m, _ in rec], ["%+.4f" % (d - p) for d, _, p in rec]))
    legacy = R.replay_push_on_oracle(O, 5, lambda c: setattr(c, "push_contact_model", 0))
    print("    rounds 1-4 model (tool sphere, full push-out, cube at rest from reset on): M = %s   d_f - placement = %s   return errors, shipped reward %s, earlier reward %s"
          % ([o["M"] for o in legacy], ["%+.4f" % (o["d_f"] - o["planar"]) for o in legacy],
             ["%+.1f" % (o["ret"] - u) for o, u in zip(legacy, upd)], ["%+.2f" % (o["ret_origin"] - u) for o, u in zip(legacy, org)]))
    fmt = lambda row: ("    radius %.3f  friction %.3f  erp %.4f  below %.4f : worst return error, shipped reward %5.1f / earlier reward %4.2f   sum |dM| %3d   M %s   d_f - placement %s%s"
                       % (*row[0], row[1], row[2], row[3], row[4], ["%+.4f" % x for x in row[5]], "" if row[6] else "   [episode 4 disturbed]"))
    with Pool(min(8, os.cpu_count() or 1)) as pool:
        # (1) the fit that is shipped: the tool is the KUKA flange as drawn (radius 0.045, face 0.045 below the link-7 frame); TWO free
        #     numbers, the contact ERP and the cube / table friction
        grid2 = [(0.045, mu, erp, 0.045) for mu in (0.015, 0.02, 0.025, 0.03, 0.035, 0.04, 0.05, 0.06, 0.1, 0.5, 2.5)
                 for erp in (0.006, 0.008, 0.009, 0.01, 0.011, 0.012, 0.014, 0.016, 0.02, 0.05, 0.2)]
        rows2 = pool.map(_push_row, grid2, chunksize=4)
        # (2) context: all four numbers free
        grid4 = list(itertools.product((0.03, 0.035, 0.04, 0.045, 0.05, 0.055), (0.03, 0.05, 0.08, 0.1, 0.15, 0.2, 0.3, 0.5, 1.0, 2.5),
                                       (0.01, 0.015, 0.02, 0.03, 0.04, 0.05, 0.07, 0.1, 0.2), (0.03, 0.045, 0.06)))
        rows4 = pool.map(_push_row, grid4, chunksize=8)
    print("    (1) nominal flange geometry, %d settings of (friction, erp); best by the worst return error under the SHIPPED reward:" % len(rows2))
    for row in sorted(rows2, key=lambda x: x[1])[:10]:
        print(fmt(row))
    print("        shipped default:")
    print(fmt(next(x for x in rows2 if x[0] == (0.045, 0.03, 0.01, 0.045))))
    print("        Bullet's own constants (contact ERP 0.2, friction 5 x 0.5 = 2.5):")
    print(fmt(next(x for x in rows2 if x[0] == (0.045, 2.5, 0.2, 0.045))))
    print("    (2) all four numbers free, %d settings; best by the final distances (the earlier reward's returns):" % len(rows4))
    for row in sorted(rows4, key=lambda x: x[2])[:6]:
        print(fmt(row))
    print("        best by the shipped reward's returns:")
    for row in sorted(rows4, key=lambda x: x[1])[:6]:
        print(fmt(row))
    print("    risk: two fitted numbers (four in (2)) against eight numbers (d_f and M of four episodes) of a trajectory that is sensitive to every")
    print("    contact: neighbouring settings differ by 10-20 in return.  The moving-step counts ask for far longer slides than Bullet's friction")
    print("    allows a rigid push-out (M rises as friction and erp fall), so the fitted values are effective values of this planar stand-in, not")
    print("    Bullet's parameters; no setting reproduces both observables of all four episodes.")
    return rows2, rows4


if __name__ == "__main__":
    os.environ.setdefault("OMP_NUM_THREADS", "1")       # the sweeps run one process per core: no OpenMP team inside each
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10000)
    ap.add_argument("--push-only", action="store_true")
    a = ap.parse_args()
    if not a.push_only:
        fit_recorded_run()
        fit_pybullet(a.steps)
    fit_push()
