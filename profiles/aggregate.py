#!/usr/bin/env python3
"""Turns the raw rocprofv3 outputs of profiles/collect.sh (gpurun_out/prof/) into the summaries kept in profiles/
(ROUND = r05 unless given as argv[1]):
  <round>_kernel_stats_<config>.csv   --stats tables, armenv kernels only (config "driver" = `bench.py --steps 20 --warmup 5`)
  <round>_kernel_stats_<config>_by_T.csv  the same launches from the per-dispatch kernel trace, one row per (kernel, steps per
                                      launch): rocprofv3's --stats merges the T = 5 warm-up, the T = 20 timed launches and the
                                      T = 100 launches of the fence leg of one symbol into one average
  <round>_bench_<config>.json         the bench.py line of the same run
  <round>_pmc_default_bench_f64.json  per-launch means of every counter for the rollout / step kernels of the default bench
  <round>_pmc_actor_f16x3.json        same for the fused f16x3 actor rollout kernel
  <round>_pmc_push_pick.json          VALU instructions / wave cycles of the push and pick rollout kernels (32 768 envs)
  traffic.json                    HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB counters; gfx950 reports
                                  half of the fetched bytes, MI355X_MICROARCH.md HBM section) -- read by bench.py; one entry
                                  per launch shape, keyed "<kernel>|policy=<p>|T=<steps per launch>|N=<envs>"
Only launches with the full step count are averaged (the warm-up launch of a different length is dropped by taking the
most frequent duration class: launches whose duration is within 30 % of the median)."""
import collections
import csv
import glob
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r06"


def short(name):
    if "env_rollout_async_kernel" in name or "env_rollout_kernel" in name:
        # template arguments <Lane, T, POLICY, WAVES>.  POLICY: 0 external actions (the headline), 1 in-kernel Philox policy
        # (the bench line's "in_kernel_policy" leg and the device pre-warm), 2 / 3 fused actors.  WAVES = 2: the
        # two-waves-per-SIMD build (bench.py's large_batch leg), kept apart from the headline kernel.
        import re
        m = re.search(r", (\d+), (\d+)>\(", name)
        pol, waves = (m.group(1), m.group(2)) if m else ("0", "1")
        return ("rollout_philox" if pol == "1" else "rollout") + ("_w2" if waves == "2" else "")
    if "env_step_kernel" in name:
        return "step"
    return None


def pmc(files, want):
    """{kernel: {counter: {launches, mean_per_launch}}} for kernels `want` maps to a label"""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for f in files:
        rows = list(csv.DictReader(open(f)))
        dur = collections.defaultdict(list)
        for r in rows:
            k = want(r["Kernel_Name"])
            if k:
                dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        med = {k: statistics.median(v) for k, v in dur.items()}
        for r in rows:
            k = want(r["Kernel_Name"])
            if not k:
                continue
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            if abs(d - med[k]) > 0.3 * med[k]:
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = {"_kernel": r["Kernel_Name"][:140],
                       "_meta": {"VGPR": r["VGPR_Count"], "AGPR": r["Accum_VGPR_Count"], "SGPR": r["SGPR_Count"],
                                 "workgroup": r["Workgroup_Size"], "grid": r["Grid_Size"], "lds": r["LDS_Block_Size"],
                                 "scratch": r["Scratch_Size"]}}
    out = {}
    for k, cs in acc.items():
        out[k] = dict(meta[k])
        for c, v in sorted(cs.items()):
            out[k][c] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
    return out


def by_steps(trace_csv, bench_json, dst):
    """Per-dispatch kernel trace -> one row per (kernel symbol, steps per launch).  A rollout launch's length is not in the
    trace; it is recovered from the duration: bench.py's line gives the time per env step of its own launches
    (roofline.avg_launch_us / steps_per_launch), and a dispatch is assigned the candidate length (the run's warm-up remainder,
    its steps per launch, the 100 of the fence / secondary legs) whose predicted duration is nearest on a log scale."""
    import math
    rows = list(csv.DictReader(open(trace_csv)))
    try:
        b = json.load(open(bench_json))
    except Exception:
        return
    spl = float(b["config"]["steps_per_launch"])
    # (the median of the run's repeated regions where the line has it: the contract region alone can carry a host hiccup between its
    # opening event and its launch -- under the profiler it did, 331 us for a 120 us kernel, and every launch was mislabelled)
    us_step = (b.get("launch_us_median") or b["roofline"]["avg_launch_us"]) / spl
    R = int(round(spl))
    W = int(b["warmup"])
    cands = sorted({R, 100, (W % R) or R})          # the warm-up's last (short) launch, the timed launches, the 100-step legs
    # The headline kernel's time per step comes from the bench line; every other rollout variant of the run (the scratch handle's
    # in-kernel policy, the fused-actor / push / DATD3 legs, the fence handles) is launched 100 steps at a time by bench.py, so its
    # own median launch is a 100-step one (round 6: these rows carried "?" before).
    pol = {"external": 0, "random": 1, "actor": 2, "actor_f16x3": 3, "datd3": 4, "daddpg": 4}.get(b["config"].get("policy", "external"), 0)
    lane = {"reach": "ReachLane", "push": "false, 0>", "pick": "true, 0>"}[("push" if "rl_push" in b["metric"] else ("pick" if "rl_pick" in b["metric"] else "reach"))]
    prec = "double" if b.get("dtype", "f64") == "f64" else "float"

    def is_headline(name):
        return lane in name and (", %d, 1>" % pol) in name and ("%s, 0>" % prec in name or lane != "ReachLane") and "async" not in name
    durs = collections.defaultdict(list)
    for r in rows:
        if "env_rollout" in r["Kernel_Name"]:
            durs[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    per_step = {n_: (us_step if is_headline(n_) else statistics.median(v_) / 100.0) for n_, v_ in durs.items()}
    groups = collections.defaultdict(list)
    for r in rows:
        name = r["Kernel_Name"]
        if not any(s_ in name for s_ in ("env_", "actor_", "her_", "index_episodes", "datd3")):
            continue
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        T = ""
        if "env_rollout" in name:
            T = min(cands, key=lambda c: abs(math.log(max(d, 1e-3) / (c * per_step[name]))))
        groups[(name, T)].append(d)
    with open(dst, "w") as fh:
        fh.write("Kernel_Name,steps_per_launch,calls,avg_us,min_us,max_us,median_us\n")
        for (name, T), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
            fh.write('"%s",%s,%d,%.2f,%.2f,%.2f,%.2f\n' % (name, T, len(v), sum(v) / len(v), min(v), max(v), statistics.median(v)))


def main():
    for f in glob.glob(os.path.join(SRC, "*_kernel_trace.csv")):
        cfg = os.path.basename(f)[: -len("_kernel_trace.csv")]
        by_steps(f, os.path.join(SRC, f"{cfg}_bench.json"), os.path.join(DST, f"{ROUND}_kernel_stats_{cfg}_by_T.csv"))
    for f in glob.glob(os.path.join(SRC, "*_kernel_stats.csv")):
        cfg = os.path.basename(f)[: -len("_kernel_stats.csv")]
        rows = [l for i, l in enumerate(open(f)) if i == 0 or any(s in l for s in ("env_", "actor_", "her_", "index_episodes"))]
        open(os.path.join(DST, f"{ROUND}_kernel_stats_{cfg}.csv"), "w").writelines(rows)
        b = os.path.join(SRC, f"{cfg}_bench.json")
        if os.path.exists(b) and open(b).read().strip().startswith("{"):
            shutil.copy(b, os.path.join(DST, f"{ROUND}_bench_{cfg}.json"))
    default = pmc(sorted(glob.glob(os.path.join(SRC, "pmc[0-9]_counters.csv"))), short)
    if default:
        json.dump(default, open(os.path.join(DST, f"{ROUND}_pmc_default_bench_f64.json"), "w"), indent=1)
    actor = pmc(sorted(glob.glob(os.path.join(SRC, "pmc_actor*_counters.csv"))), short)
    if actor:
        json.dump(actor, open(os.path.join(DST, f"{ROUND}_pmc_actor_f16x3.json"), "w"), indent=1)
    pp = {t: pmc([f], short) for t in ("push", "pick") for f in glob.glob(os.path.join(SRC, f"pmc_{t}_counters.csv"))}
    if pp:
        json.dump(pp, open(os.path.join(DST, f"{ROUND}_pmc_push_pick.json"), "w"), indent=1)
    # HBM traffic per launch shape
    tpath = os.path.join(DST, "traffic.json")
    traffic = {}
    if os.path.exists(tpath):        # keep the entries of shapes this run did not collect
        try:
            traffic = {k: v for k, v in json.load(open(tpath)).items() if not k.startswith("_")}
        except Exception:
            traffic = {}

    def add(d, k, key, n):
        if k in d and "FETCH_SIZE" in d[k] and "WRITE_SIZE" in d[k]:
            fe, wr = d[k]["FETCH_SIZE"]["mean_per_launch"], d[k]["WRITE_SIZE"]["mean_per_launch"]
            traffic[key] = {"hbm_bytes_per_launch": (2 * fe + wr) * 1024, "fetch_size_kb_raw": fe, "write_size_kb_raw": wr,
                            "launches_averaged": d[k]["FETCH_SIZE"]["launches"], "round": ROUND}

    shapes = {"": 100, "_T20": 20}    # suffix of the pmc3 / pmc4 passes -> steps per rollout launch (profiles/collect.sh B100 / B20)
    for suf, T in shapes.items():
        d = pmc(sorted(glob.glob(os.path.join(SRC, f"pmc[34]{suf}_counters.csv"))), short)
        add(d, "rollout", "reach_rollout<f64,kuka>|policy=external|T=%d|N=65536" % T, 65536)
        add(d, "step", "reach_step<f64,kuka>|policy=external|T=1|N=65536", 65536)
    # the launch shapes of bench.py's config3 / config4 legs (one bench run per shape: its headline kernel is "rollout")
    for suf, key in (("_actor", "reach_rollout<f64,kuka>|policy=actor|T=100|N=65536"),
                     ("_actor_f16x3", "reach_rollout<f64,kuka>|policy=actor_f16x3|T=100|N=65536"),
                     ("_push", "push_rollout<f64,kuka>|policy=external|T=100|N=32768"),
                     ("_datd3", "reach_rollout<f64,kuka>|policy=datd3|T=100|N=65536"),
                     ("_daddpg", "reach_rollout<f64,kuka>|policy=daddpg|T=100|N=65536")):
        d = pmc(sorted(glob.glob(os.path.join(SRC, f"pmc[34]{suf}_counters.csv"))), short)
        add(d, "rollout", key, 0)
    traffic["_note"] = ("FETCH_SIZE / WRITE_SIZE from separate rocprofv3 --pmc passes over bench.py (KiB per launch, mean over "
                        "launches, N=65536; profiles/collect.sh + aggregate.py). hbm_bytes_per_launch = 2*FETCH + WRITE: FETCH "
                        "doubled per the gfx950 note in MI355X_MICROARCH.md (HBM section).  Keys: "
                        "<kernel>|policy=<p>|T=<steps per launch>|N=<envs>; bench.py reports `roofline.traffic` only for an "
                        "exact match of its own launch shape (T=20 is the driver's `--steps 20`).")
    json.dump(traffic, open(tpath, "w"), indent=1)
    print(json.dumps(traffic, indent=1))
    for k in ("rollout", "step"):
        if k in default:
            d = default[k]
            g = lambda c: d.get(c, {}).get("mean_per_launch")
            print(k, {c: g(c) for c in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE")})
    if "rollout" in actor:
        print("actor rollout", {c: v["mean_per_launch"] for c, v in actor["rollout"].items() if not c.startswith("_")}, actor["rollout"]["_meta"])
    for t, d in pp.items():
        if "rollout" in d:
            print(t, {c: v["mean_per_launch"] for c, v in d["rollout"].items() if not c.startswith("_")})


if __name__ == "__main__":
    main()
