#!/usr/bin/env python3
"""does a cube that can TIP explain the reference's recorded push runs with Bullet's own constants?

The five first episodes of the two recorded push runs (tests/reference_run.py) on the oracle's arm pipeline, with the cube under a
planar rigid-body step written here (test tooling, numpy scalars): the 4 cm / 1 kg box as a SQUARE in the vertical plane spanned by the
contact normal n and z -- three degrees of freedom (slide along n, height, tilt about the horizontal axis perpendicular to n) -- its four
corners against the table, the tool (link 7: a vertical cylinder) against its near face, solved the way Bullet solves a step: gravity
into the velocities, contacts found at the positions the step starts from, sequential impulses (normal rows with an ERP share of the
penetration as velocity target, friction rows bounded by mu x the normal impulse), integration.  Constants are Bullet's / the URDFs':
ERP 0.2, friction 5 (cube) x 0.5 (table, link) = 2.5, g = 10, dt = 1/240, 50 iterations.  Swept: the nominal height of the tool contact.

Prints per setting M (steps whose cube-target distance moved by >= 1e-5), d_f - d_0 and the return errors under both recorded rewards."""
import math
import os
import random
import sys

os.environ.setdefault("OMP_NUM_THREADS", "1")       # one oracle thread per worker of the sweep's process pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O          # noqa: E402
import reference_run as R               # noqa: E402

H = 0.02; MASS = 1.0; INERTIA = MASS * (2 * H) ** 2 / 6.0; G = 10.0; DT = 1.0 / 240.0


class Cube:
    """the square in the (n, z) plane; pos = COM xyz, n = unit xy vector of the plane, th = tilt (top towards +n), vel (vs, vz, w)"""

    def __init__(self, P):
        self.P = P
        self.n = np.array([1.0, 0.0]); self.th = 0.0
        self.vs = self.vz = self.w = 0.0
        self.awake = False

    def corners(self):
        c, s = math.cos(self.th), math.sin(self.th)
        return [(bs * c + bz * s, -bs * s + bz * c) for bs, bz in ((-H, -H), (H, -H), (H, H), (-H, H))]

    def step(self, pos, tool_p, k, fall_z):
        """one stepSimulation; pos: COM xyz (in / out); tool_p: link-7 frame position; fall_z(k, z_prev): the pinned free-fall height"""
        P = self.P
        r, lo = P["radius"], tool_p[2] - P["below"]
        # --- tool contact at the start positions (footprint test on the COM-centred square, as push_contact_dyn)
        tool = None
        if lo < pos[2] + H:
            qx = min(max(tool_p[0], pos[0] - H), pos[0] + H); qy = min(max(tool_p[1], pos[1] - H), pos[1] + H)
            gx, gy = qx - tool_p[0], qy - tool_p[1]
            gap = math.hypot(gx, gy)
            if gap < r:
                if gap > 1e-9:
                    pen, nx, ny = r - gap, gx / gap, gy / gap
                else:
                    ex = [pos[0] + H - tool_p[0], tool_p[0] - (pos[0] - H), pos[1] + H - tool_p[1], tool_p[1] - (pos[1] - H)]
                    b = int(np.argmin(ex)); pen = ex[b] + r
                    nx, ny = ((-1.0, 0.0), (1.0, 0.0), (0.0, -1.0), (0.0, 1.0))[b]
                pen_v = (pos[2] + H) - lo
                tool = dict(pen=pen, n=np.array([nx, ny]), pen_v=pen_v, vertical=pen_v < pen)
                if tool["vertical"] and P["press"] == "freeze":      # push_contact_dyn's rule: the tool presses the cube onto the table, nothing moves
                    tool = None
        if tool is None and not self.awake:
            pos[2] = fall_z(k, pos[2])
            return
        if not self.awake:                      # wakes flat, at the pinned height, moving as the fall model says
            self.awake = True
            self.th = self.vs = self.w = 0.0
            self.vz = 0.0
            self.zground = (pos[2] - H) + (G * DT * DT / P["erp"] if k > P["land"] else 0.0)   # resting penetration g dt^2 / erp
            if k <= P["land"]:
                self.zground = P["table"]
                self.vz = -G * DT * (k - 1)
        flat = abs(self.th) < 1e-4 and abs(self.w) < 1e-3
        if tool is not None and not tool["vertical"] and flat and math.hypot(self.vs, 0.0) < 1e-3:
            self.n = tool["n"].copy(); self.th = 0.0
        # --- gravity
        self.vz -= G * DT
        rows = []      # (kind, r_s, r_z, dir_s, dir_z, target, mu_parent)
        cs = self.corners()
        for (rs, rz) in cs:
            d = pos[2] + rz - self.zground
            if d < P["margin"]:
                tgt = -d / DT if d > 0 else P["erp"] * (-d) / DT
                rows.append(["n", rs, rz, 0.0, 1.0, tgt, None, 0.0])
                rows.append(["f", rs, rz, 1.0, 0.0, 0.0, len(rows) - 1, 0.0])
        if tool is not None:
            if tool["vertical"]:
                tgt = P["erp"] * tool["pen_v"] / DT
                rows.append(["n", 0.0, H, 0.0, -1.0, tgt, None, 0.0])
                if P["tool_mu"] > 0:
                    rows.append(["f", 0.0, H, 1.0, 0.0, 0.0, len(rows) - 1, 0.0])
            else:
                c = float(tool["n"] @ self.n)
                sgn = 1.0 if c >= 0 else -1.0
                top = pos[2] + H; bot = max(lo, pos[2] - H)
                zc = bot + P["hfrac"] * (top - bot)
                tgt = P["erp"] * tool["pen"] / DT * abs(c)
                rows.append(["n", -sgn * H, zc - pos[2], sgn, 0.0, tgt, None, 0.0])
                if P["tool_mu"] > 0:
                    rows.append(["f", -sgn * H, zc - pos[2], 0.0, 1.0, 0.0, len(rows) - 1, 0.0])
        mu_g, mu_t = P["mu"], P["tool_mu"]
        for it in range(P["iters"]):
            for kind in ("n", "f"):
                for i, row in enumerate(rows):
                    if row[0] != kind:
                        continue
                    _, rs, rz, ds, dz, tgt, parent, lam = row
                    # velocity of the point: (vs + w rz, vz - w rs)
                    v = (self.vs + self.w * rz) * ds + (self.vz - self.w * rs) * dz
                    ang = rz * ds - rs * dz
                    K = 1.0 / MASS + ang * ang / INERTIA
                    dl = (tgt - v) / K
                    if kind == "n":
                        new = max(0.0, lam + dl)
                    else:
                        is_tool = rows[parent][4] != 1.0
                        lim = (mu_t if is_tool else mu_g) * rows[parent][7]
                        new = min(max(lam + dl, -lim), lim)
                    dl = new - lam
                    row[7] = new
                    self.vs += dl * ds / MASS; self.vz += dl * dz / MASS; self.w += dl * ang / INERTIA
        # --- integrate
        pos[0] += self.n[0] * self.vs * DT; pos[1] += self.n[1] * self.vs * DT
        pos[2] += self.vz * DT
        self.th += self.w * DT
        # a quarter turn later the square is the same square
        if self.th > math.pi / 4 + 1e-9 and False:
            pass
        # --- back to rest: flat (a multiple of 90 degrees), slow, no tool
        thm = (self.th + math.pi / 4) % (math.pi / 2) - math.pi / 4
        if tool is None and abs(thm) < P["sleep_th"] and abs(self.w) < P["sleep_w"] and abs(self.vs) < P["sleep_v"] and abs(self.vz) < 0.02 and k > P["land"] + 3:
            self.awake = False
            self.th = self.vs = self.vz = self.w = 0.0


def replay(P, episodes=5, verbose=False):
    chain = O.make_chain("kuka")
    cfg = O.default_config("push")
    sd = R.actor9_weights()
    random.seed(0); np.random.seed(0)
    st = O.PushState(1)
    land = 1
    cfall = 0.5 * G * DT * DT
    while cfall * land * (land + 1) < cfg.push_drop_contact:
        land += 1
    P = dict(P, land=land, table=cfg.push_place_z - H - cfg.push_drop_contact)

    def fall_z(k, zprev):
        if k <= land:
            return cfg.push_place_z - cfall * k * (k + 1)
        return cfg.push_rest_z + (zprev - cfg.push_rest_z) * (1.0 - cfg.push_drop_relax)
    out = []
    for ep in range(episodes):
        c, t, planar = R.draw_push_placement()
        cube, tgt = c + [float(cfg.push_place_z)], t + [float(cfg.push_place_z)]
        obs = O.push_reset_with_goal(chain, cfg, st, np.float32([cube + tgt]))[0]
        st.aux[0, 0:2] = c; st.aux[0, 3:6] = tgt
        pos = np.array([c[0], c[1], float(st.aux[0, 2])])
        target = np.array(tgt)
        d_last = float(np.linalg.norm(pos - target))
        body = Cube(P)
        done, ret, ret_o, n, M, touch = False, 0.0, 0.0, 0, 0, 0
        while not done:
            st.aux[0, 0:3] = pos
            state = np.hstack((obs[:3].astype(np.float32), pos, target)).astype(np.float32)
            a = O.actor_forward(sd, state[None], 0.4)[0].astype(np.float64) + np.random.normal(0, 0.4 * 0.98, size=3)
            step0 = int(st.step[0])
            o, r, d, s, _ = O.push_step(chain, cfg, st, a.astype(np.float32)[None])      # the arm (the oracle's own cube is overwritten)
            for k_ in range(3):
                random.uniform(R._LO[k_], R._HI[k_])
            p1 = O.fk(chain, st.q)[0][0]
            was = body.awake
            body.step(pos, p1, step0 + 2, fall_z)
            touch += int(body.awake and not was)
            d_cur = float(np.linalg.norm(pos - target))
            test = d_cur - d_last
            M += int(abs(test) >= 1e-5)
            if abs(test) < 1e-5:
                test = 0.01
            d_last = d_cur
            n += 1
            d32 = float(np.linalg.norm(pos.astype(np.float32) - target.astype(np.float32)))
            if n > cfg.max_steps:
                rew, done = -d32 * 50.0, True
            elif d32 < cfg.push_success_dis:
                rew, done = 100.0, True
            else:
                rew = -test * 100.0
            ret += rew
            ret_o += rew if (done or rew == 100.0) else -1.0
            obs = np.asarray(o[0], dtype=np.float32)
            st.aux[0, 6] = d_last
        out.append(dict(ret=ret, ret_origin=ret_o, n=n, M=M, d_f=d32, planar=planar, touch=touch, th=body.th))
    return out


def main():
    rec = R.push_recorded_observables(5)
    org, upd = R.push_fixture_returns("origin"), R.push_fixture_returns("updata")
    print("recorded: M", [round(r[1]) for r in rec], " d_f - d_0", ["%+.4f" % (r[0] - r[2]) for r in rec])
    base = dict(radius=0.045, below=0.045, erp=0.2, mu=2.5, tool_mu=2.5, iters=50, margin=0.02, hfrac=0.5,
                sleep_th=2e-3, sleep_w=0.05, sleep_v=2e-3, press="erp")
    sweeps = [("nominal (press: ERP share)", {}), ("nominal, press: freeze", dict(press="freeze"))]
    for hf in (0.0, 0.1, 0.2, 0.3, 0.4, 0.45, 0.55, 0.6, 0.7, 0.8, 0.9, 1.0):
        sweeps.append(("freeze, contact height %.2f" % hf, dict(hfrac=hf, press="freeze")))
    sweeps += [("freeze, tool friction 0", dict(tool_mu=0.0, press="freeze")), ("freeze, tool friction 0.5", dict(tool_mu=0.5, press="freeze")),
               ("freeze, 10 iterations", dict(iters=10, press="freeze")), ("freeze, 200 iterations", dict(iters=200, press="freeze")),
               ("freeze, contact margin 0", dict(margin=0.0, press="freeze")),
               ("freeze, sleeps 4x sooner", dict(sleep_th=8e-3, sleep_w=0.2, sleep_v=8e-3, press="freeze")),
               ("freeze, table friction 1.0", dict(mu=1.0, press="freeze")), ("freeze, ERP 0.1", dict(erp=0.1, press="freeze"))]
    if len(sys.argv) > 1 and sys.argv[1] == "quick":
        sweeps = sweeps[:2]
    import concurrent.futures as cf
    with cf.ProcessPoolExecutor(max_workers=6) as ex:
        results = list(ex.map(replay, [dict(base, **ch) for _, ch in sweeps]))
    worst_s, worst_m = [], []
    for (name, ch), res in zip(sweeps, results):
        worst_s.append(max(abs(r["ret"] - upd[i]) for i, r in enumerate(res)))
        worst_m.append(sum(abs(r["M"] - round(rec[i][1])) for i, r in enumerate(res)))
        print("%-26s M %s  d_f - d_0 %s  return errors: shipped %s  earlier %s  wakes %s" % (
            name, [r["M"] for r in res], ["%+.4f" % (r["d_f"] - r["planar"]) for r in res],
            ["%+.1f" % (r["ret"] - upd[i]) for i, r in enumerate(res)], ["%+.2f" % (r["ret_origin"] - org[i]) for i, r in enumerate(res)],
            [r["touch"] for r in res]))
    print("worst |return error| under the shipped reward per setting:", ["%.0f" % x for x in worst_s])
    print("sum |dM| per setting:", worst_m)


if __name__ == "__main__":
    main()
