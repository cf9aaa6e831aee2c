#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of rl_reach_env at 65 536 parallel envs per GPU (BASELINE.json
configs[1]; N>1 = configs[4], envs sharded over GPUs, RCCL all-gather of episode returns for logging).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of actions: every env of the rank advances once with a
pre-generated random-policy action batch already resident in HBM (an i.i.d. [<=1000, N, 3] pool consumed in order).
Before the W warm-up steps the device is brought to its steady clocks on a scratch handle (--prewarm-ms; an idle
MI355X ramps for ~30 ms, tests/tools/clock_ramp.py), and every timed region has --busy-ahead-ms of the same scratch work
enqueued right in front of its opening synchronise (the region starts from the clocks of a chip under sustained load).  Two launch shapes run the same per-step code:
  --mode rollout (default, `value`): armenv_rollout, R = 100 steps per kernel launch, env state kept in registers,
                 per-step outputs written to [R][N][...] buffers (the trajectory of R armenv_step calls);
  --mode step:   armenv_step, one launch per step (the gym-style call).  In rollout mode this path is also timed
                 beside the headline and reported under "step_api".
Timing -- ONE bracket for 1 and N ranks (round 6): W untimed steps, then device synchronise + barrier, the clock, EXACTLY K steps, the
launch stream's synchronise + barrier, the clock; the barrier of the bracket is the single-node shared-memory one
(armenv.dist.ShmBarrier, ~3 us; RCCL's dist.barrier() runs before and after, outside the clock).  A multi-rank run issues its logging
all-gather INSIDE the region -- RCCL's own ncclAllGather on a side stream behind the steps (armenv.dist.RcclComm; ~40 us of host time
under the running kernel) -- and waits for it and verifies it right after the clock (config.collective_us, collective_verified).
`value` = all ranks' env-steps / MAX over ranks of that wall time; `value_steps` = the same with each rank's clock stopped when its
own launch stream is idle (no barrier); `value_bracketed` = rounds 1-5's multi-rank clock (the collective's completion, a device
synchronise and dist.barrier() inside it); `value_kernel` = the same steps / MAX over ranks of the kernels' own time (HIP events on
the launch stream).  Single-GPU runs repeat the identical region 15 more
times on fresh action rows and report the spread (value_median / value_min / value_max, launch_us_samples) beside the
contract's first region.  Rank 0 prints ONE JSON line; besides the headline it carries
short legs for the other single-GPU BASELINE configs (config3_actor_f32, config3_actor_f16x3, config4_push) and the fused two-actor
policies (datd3_fused, daddpg_fused).
The CPU oracle is timed beside it (rank 0, N=1 only) on a bounded sample -- one thread, then every core the process may
use -- as a baseline, never as the thing measured.  A short extra leg on a second handle with the parity-fence counters on
reports how often the workload crosses the URDF joint limits / drives the flange below z = 0.05 / runs an IK call to its
iteration cap or through an ill-conditioned system (diagnostics: DESIGN.md section 2).

The driver runs `--steps 20 --warmup 5`: ONE 20-step launch in the timed region.  Everything but the C call is prepared
before the clock starts (armenv's bind_rollout), and profiles/traffic.json holds the PMC traffic of that launch shape too.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "drl-on-robot-arm_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np

ENVS_PER_GPU = 65536


def _cpu_leg(n, threads, budget):
    """One timed leg of the CPU oracle: `n` envs, `threads` OpenMP threads, about `budget` seconds.  Runs in a child process
    of its own (no torch, OMP_PROC_BIND=close set before libgomp starts) so that thread pinning never touches the process
    that drives the GPU."""
    from oracle import oracle as O
    O.build()
    chain, cfg = O.make_chain("kuka"), O.default_config()
    O.set_num_threads(threads)
    rng = np.random.default_rng(0)
    st = O.ReachState(n)
    O.reach_reset(chain, cfg, st, seed=0)
    # i.i.d. actions per step like the device pool (a short cycle of action arrays would pin the envs in a corner of the box)
    A = np.clip(rng.standard_normal((64, n, 3), dtype=np.float32) * np.float32(0.686), -0.7, 0.7)
    O.reach_step_autoreset(chain, cfg, st, A[0], seed=0, want_terminal=False)   # warm-up
    t0 = time.perf_counter(); k = 0
    while True:
        O.reach_step_autoreset(chain, cfg, st, A[k % 64], seed=0, want_terminal=False)
        k += 1
        dt = time.perf_counter() - t0
        if (dt >= budget and k >= 3) or k >= 5000:
            break
    return {"value": n * k / dt, "steps": k, "seconds": dt, "threads": O.num_threads()}


if len(sys.argv) > 1 and sys.argv[1] == "--cpu-leg":      # child of cpu_baseline(): python bench.py --cpu-leg N THREADS SECONDS
    print(json.dumps(_cpu_leg(int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]))))
    sys.exit(0)

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
F64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X f64 vector (FMA = 2 flop), 1/2 of the 157.3 TF f32 vector peak
# algorithmic bytes: caller I/O per env-step, and state read+written once per launch
IO_BYTES = 12 + 24 + 4 + 1 + 1                 # action in; obs, reward, done, success out
# q, (cos q, sin q), ep_return read + written; goal read; step read + written
STATE_BYTES = {64: 2 * (56 + 112 + 8) + 12 + 2 * 4, 32: 2 * (28 + 56 + 4) + 12 + 2 * 4}
F64_ISSUE_CYCLES_ONE_WAVE = 6.6   # round 3's stand-alone probe (profiles/r03_valu_f64_rate_probe.txt); since round 4 the bench line measures it in the run (issue_probe)
NOMINAL_GHZ = 2.4
# algorithmic flops of the f64/f32 reach step: per IK update and per FK-only exit trip
FLOPS_PER_UPDATE, FLOPS_PER_EXIT_FK = 1250, 510   # update trip; exit FK + residual + per-step sincos/reward


class HipEvents:
    """Two HIP timing events driven straight through the HIP runtime torch has loaded (ctypes): hipEventRecord costs ~1 us this
    way and 5-9 us through torch.cuda.Event.record(), and the first record of the timed region sits between the clock's start
    and the launch.  Recorded on the stream the kernels are launched on; falls back to torch events if the runtime cannot be
    found."""

    def __init__(self, dev):
        import ctypes as C
        self.C, self.h, self.ev = C, None, None
        self.stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        try:
            path = next(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l)
            h = C.CDLL(path)
            for f in (h.hipEventCreate, h.hipEventRecord, h.hipEventSynchronize, h.hipEventElapsedTime, h.hipEventDestroy):
                f.restype = C.c_int
            h.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
            h.hipEventSynchronize.argtypes = [C.c_void_p]
            h.hipStreamSynchronize.restype, h.hipStreamSynchronize.argtypes = C.c_int, [C.c_void_p]
            h.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
            ev = [C.c_void_p(), C.c_void_p()]
            if any(h.hipEventCreate(C.byref(e)) != 0 for e in ev):
                raise OSError("hipEventCreate failed")
            self.h, self.ev = h, ev
        except Exception:       # noqa: BLE001
            self.t = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
            for e in self.t:
                e.record()

    def record(self, k):
        if self.h is not None:
            self.h.hipEventRecord(self.ev[k], self.stream)
        else:
            self.t[k].record()

    def synchronize(self, k):
        if self.h is not None:
            self.h.hipEventSynchronize(self.ev[k])
        else:
            self.t[k].synchronize()

    def stream_synchronize(self):
        if self.h is not None:
            self.h.hipStreamSynchronize(self.stream)
        else:
            torch.cuda.current_stream().synchronize()

    def elapsed_ms(self):
        if self.h is None:
            return self.t[0].elapsed_time(self.t[1])
        ms = self.C.c_float()
        rc = self.h.hipEventElapsedTime(self.C.byref(ms), self.ev[0], self.ev[1])
        if rc != 0:
            raise RuntimeError("hipEventElapsedTime failed: %d" % rc)
        return float(ms.value)

    def close(self):
        if self.h is not None:
            for e in self.ev:
                self.h.hipEventDestroy(e)
            self.h = None


class ClockProbes:
    """armenv_probe_clock samples around the timed regions: one wavefront on every SIMD runs a fixed dependent chain of v_fma_f32
    between two readings of the device's 100 MHz counter, so a sample's duration is inversely proportional to the shader clock at
    that moment (whole chip loaded, ~10 us).  Enqueued OUTSIDE every clock and every pair of HIP events (before the opening
    synchronise, after the closing one); read back once at the end of the run."""

    def __init__(self, dev, cap=128):
        import ctypes as C
        from armenv import _lib
        self.C, self.lib, self.dev = C, _lib.load(), dev
        rows = C.c_int32(0)
        self.ok = self.lib.armenv_probe_clock(dev.index or 0, None, C.byref(rows), None) == 0 and rows.value > 0
        self.rows = rows.value
        self.buf = torch.zeros((cap, max(1, self.rows), 4), dtype=torch.int64, device=dev) if self.ok else None
        self.tags = []

    def mark(self, tag):
        if tag is None or not self.ok or len(self.tags) >= self.buf.shape[0]:
            return
        C = self.C
        ptr = C.c_void_p(self.buf[len(self.tags)].data_ptr())
        if self.lib.armenv_probe_clock(self.dev.index or 0, ptr, None, C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)) == 0:
            self.tags.append(tag)

    def read(self):
        """per tag: ns per chained instruction (median over the waves), the same in s_memtime ticks, the slowest XCD's median ns,
        and the device time of the sample (us since the first one)"""
        if not self.tags:
            return {}
        b = self.buf[:len(self.tags)].cpu().numpy().astype(np.float64)
        chain = float(int(b[0, 0, 3]) >> 8 & 0xFFFFFF)
        xcd = self.buf[0, :, 3].cpu().numpy() & 15
        t0 = b[0, :, 2].min()
        out = {}
        for i, t in enumerate(self.tags):
            ns = b[i, :, 0] * 10.0 / chain
            per_xcd = [float(np.median(ns[xcd == x])) for x in sorted(set(xcd.tolist()))]
            out[t] = {"ns": float(np.median(ns)), "memtime_ticks": float(np.median(b[i, :, 1]) / chain), "ns_slowest_xcd": max(per_xcd),
                      "ns_fastest_xcd": min(per_xcd), "at_us": float((b[i, :, 2].min() - t0) * 0.01)}
        self.xcds = len(set(xcd.tolist()))
        return out


def algo_bytes_per_launch(task, policy, precision, n, steps_per_launch):
    """Algorithmic bytes one launch moves: caller I/O per env-step + the state read and written once."""
    io_b, st_b = IO_BYTES, STATE_BYTES[precision]
    if policy != "external":
        io_b -= 12                      # no action read
    if task != "reach":  # obs 36 B instead of 24; state: cube / target / d_last / cube velocity (9 reals; pick 11) r+w instead of goal
        io_b += 12
        st_b += 2 * (9 if task == "push" else 11) * (precision // 8) - 12
    return (io_b * steps_per_launch + st_b) * n


def traffic_lookup(kernel, policy, steps_per_launch, n):
    """HBM bytes per launch from separate rocprofv3 --pmc passes (profiles/README.md), one entry per launch shape:
    "<kernel>|policy=<p>|T=<steps per launch>|N=<envs>"; only an exact match is reported."""
    key = "%s|policy=%s|T=%d|N=%d" % (kernel, policy, int(steps_per_launch), n)
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    traffic = None
    if os.path.exists(tpath) and steps_per_launch == int(steps_per_launch):
        try:
            traffic = json.load(open(tpath)).get(key, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    return traffic, key


_ISSUE_PROBE = {}


def issue_probe():
    """The single-wave f64 issue interval of THIS device in THIS run (library diagnostic armenv_probe_issue_rate: sixteen
    independent v_fma_f64 chains per wave, one / two waves on every SIMD; v_fma_f32 beside it).  Measured once per process, on a
    device the caller has already warmed.  None if the probe fails (the bench line then carries no one_wave_per_simd)."""
    if not _ISSUE_PROBE:
        import ctypes as C
        from armenv import _lib
        lib, d = _lib.load(), torch.cuda.current_device()
        out = {}
        for key, prec, w in (("ns_1", 64, 1), ("ns_2", 64, 2), ("ns_f32", 32, 1)):
            v = C.c_double(0.0)
            if lib.armenv_probe_issue_rate(d, prec, w, C.byref(v)) != 0 or not v.value > 0:
                _ISSUE_PROBE["failed"] = True
                return None
            out[key] = v.value
        out["simds"] = 4 * torch.cuda.get_device_properties(d).multi_processor_count
        _ISSUE_PROBE.update(out)
    return None if _ISSUE_PROBE.get("failed") else _ISSUE_PROBE


def rooflines(task, policy, precision, n, steps_per_launch, launch_us, updates, kernel):
    """(roofline, valu, mfma-or-None) of one launch shape: algorithmic HBM bytes against 8 TB/s (with the PMC traffic of the
    same shape when profiles/traffic.json has it), algorithmic flops against the vector peak (the bound that binds: 29 flop/B,
    SURVEY.md section 8d), and the fused actor's dense flops against the MFMA peak."""
    algo = algo_bytes_per_launch(task, policy, precision, n, steps_per_launch)
    achieved = algo / (launch_us * 1e-6) / 1e9
    traffic, key = traffic_lookup(kernel, policy, steps_per_launch, n)
    flops = (updates * FLOPS_PER_UPDATE + FLOPS_PER_EXIT_FK) * n * steps_per_launch
    vpeak = F64_VECTOR_PEAK_TFLOPS if precision == 64 else 157.3
    tf = flops / (launch_us * 1e-6) / 1e12
    valu = {"bound": "valu", "achieved": tf, "peak": vpeak, "unit": "TFLOP/s", "frac": tf / vpeak,
            "ik_updates_per_env_step": updates, "algo_flops_per_launch": flops}
    if precision == 64:
        # measured IN THIS RUN (armenv_probe_issue_rate, ~30 ms once per process): one wave per SIMD issues independent v_fma_f64
        # every ~6.6 cycles, the pipe's nominal 4-cycle rate needs several waves per SIMD; the env kernels run one wave per SIMD
        # (two in large_batch).  The nominal peak above stays the creditable one.
        pr = issue_probe()
        if pr:
            peak1 = 64 * 2 * pr["simds"] / pr["ns_1"] * 1e-3      # TFLOP/s of pure FMAs at the measured single-wave issue interval
            valu["one_wave_per_simd"] = {"ns_per_f64_instruction": pr["ns_1"], "cycles_per_f64_instruction": pr["ns_1"] * NOMINAL_GHZ,
                                         "peak": peak1, "frac": tf / peak1, "two_waves_per_simd_ns": pr["ns_2"],
                                         "f32_ns_per_instruction": pr["ns_f32"], "simds": pr["simds"],
                                         "source": "measured in this run: armenv_probe_issue_rate (back-to-back independent v_fma_f64, vector "
                                                   "operands, one wave on every SIMD); cycles at the nominal %.1f GHz; round 3's stand-alone probe "
                                                   "(profiles/r03_valu_f64_rate_probe.txt) gave %.1f" % (NOMINAL_GHZ, F64_ISSUE_CYCLES_ONE_WAVE)}
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_key": key, "kernel": kernel, "avg_launch_us": launch_us, "algo_bytes_per_launch": algo,
            "binding_bound": "valu", "valu": valu}
    mfma = None
    if policy.startswith("actor") or policy in ("datd3", "daddpg"):
        # 2 * (6*256 + 256*256 + 256*3) flop per env-step; both layers on the MFMA
        af = 2 * (6 * 256 + 256 * 256 + 256 * 3) * n * steps_per_launch
        if policy in ("datd3", "daddpg"):     # two actors + two critic passes (9 -> 256 -> 256 -> 1): DATD3_mlp.py:88-109, DADDPG_mlp.py:90-94
            af = 2 * (2 * (6 * 256 + 256 * 256 + 256 * 3) + 2 * (9 * 256 + 256 * 256 + 256)) * n * steps_per_launch
        if policy == "actor":
            peak, dt, mult = 157.3, "f32 (v_mfma_f32_32x32x2_f32)", 1
        else:   # three f16 MFMA passes per useful multiply-add; priced against the dense f16 MFMA peak
            peak, dt, mult = 2500.0, "f32 emulated by 3 x f16 (v_mfma_f32_32x32x16_f16, hi/lo split)", 3
        mfma = {"bound": "mfma", "achieved": mult * af / (launch_us * 1e-6) / 1e12, "peak": peak, "unit": "TFLOP/s",
                "frac": mult * af / (launch_us * 1e-6) / 1e12 / peak, "dtype": dt, "useful_tflops": af / (launch_us * 1e-6) / 1e12}
    return roof, valu, mfma


def golden_actor():
    """weights of TD3_MLP(6,3,0.7) under torch.manual_seed(0): golden G3 (produced by importing the reference's algo.TD3)"""
    g = np.load(os.path.join(ROOT, "tests", "golden", "td3_actor_seed0.npz"))
    return {k: torch.from_numpy(g[k.replace(".", "_")]) for k in
            ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc3.weight", "fc3.bias")}


def golden_datd3():
    """the four nets of DATD3_MLP(6, 3, 0.7) under torch.manual_seed(0): golden G11 (produced by importing the reference's algo.DATD3)"""
    g = np.load(os.path.join(ROOT, "tests", "golden", "datd3_take_action_seed0.npz"))
    keys = ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc3.weight", "fc3.bias")
    return [{k: torch.from_numpy(g["%s_%s" % (n_, k.replace(".", "_"))]) for k in keys} for n_ in ("actor1", "actor2", "critic1", "critic2")]


def golden_daddpg():
    """the three nets of DADDPG_MLP(6, 3, 0.7) -- the reference's default agent, config.py:33: golden G15 (produced by importing algo.DADDPG)"""
    g = np.load(os.path.join(ROOT, "tests", "golden", "daddpg_take_action_seed0.npz"))
    keys = ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc3.weight", "fc3.bias")
    return [{k: torch.from_numpy(g["%s_%s" % (n_, k.replace(".", "_"))]) for k in keys} for n_ in ("actor1", "actor2", "critic")]


def secondary_leg(envs, dev, task, n, policy, precision, launches, pre_launches, fence_steps=0):
    """One more BASELINE config timed by the same command: `launches` x armenv_rollout(100) on a fresh
    handle after `pre_launches` untimed ones, HIP events on the launch stream, with the same roofline bookkeeping as the
    headline.  config3 = reach 65 536 + fused TD3 actor (policy actor / actor_f16x3), config4 = push 32 768, external actions."""
    T = 100
    Env = {"reach": envs.BatchedReachEnv, "push": envs.BatchedPushEnv, "pick": envs.BatchedPickEnv}[task]
    e = Env(n, device=dev, seed=0, precision=precision)
    bound, sig = (0.7, 0.7 * 0.98) if task == "reach" else (0.4, 0.4 * 0.98)
    pool = None
    if policy == "external":          # i.i.d. rows, consumed in order (the headline's pool shape)
        S = T * (launches + pre_launches)
        gen = torch.Generator(device=dev); gen.manual_seed(2000)
        pool = torch.randn((S, n, 3), device=dev, generator=gen) * sig
        if task == "reach":
            pool.clamp_(-bound, bound)
    elif policy == "datd3":          # DATD3_MLP.take_action fused: two actors, two critics, the better-valued action
        e.set_policy_datd3(*golden_datd3(), action_bound=bound, noise_sigma=sig, noise_clip=bound)
    elif policy == "daddpg":         # DADDPG_MLP.take_action fused: two actors, ONE critic on both proposals (three staged nets, four passes)
        e.set_policy_daddpg(*golden_daddpg(), action_bound=bound, noise_sigma=sig, noise_clip=bound)
    else:
        e.set_policy(policy, action_bound=bound, noise_sigma=sig, noise_clip=bound if task == "reach" else 1e9,
                     actor_state_dict=golden_actor() if policy.startswith("actor") else None)
    e.reset()
    bufs = {}
    rows = lambda j: None if pool is None else pool[j * T:(j + 1) * T]
    for j in range(pre_launches):
        e.rollout(T, rows(j), out=bufs)
    torch.cuda.synchronize(dev)
    c0 = e.counters()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(); ev1.record(); torch.cuda.synchronize(dev)      # event creation outside the region
    t0 = time.perf_counter()
    ev0.record()
    for j in range(launches):
        e.rollout(T, rows(pre_launches + j), out=bufs)
    ev1.record(); torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    c1 = e.counters()
    kernel = e.kernel_name.replace("_step", "_rollout")
    e.close()
    launch_us = ev0.elapsed_time(ev1) * 1e3 / launches
    updates = (c1["ik_updates"] - c0["ik_updates"]) / max(1, c1["env_steps"] - c0["env_steps"])
    roof, valu, mfma = rooflines(task, policy, precision, n, T, launch_us, updates, kernel)
    out = {"value": n * T * launches / wall, "value_kernel": n * T / (launch_us * 1e-6), "unit": "env-steps/s", "envs": n, "task": task,
           "policy": policy, "steps": T * launches, "steps_per_launch": T, "untimed_steps_before": T * pre_launches,
           "us_per_step": launch_us / T, "dtype": "f64" if precision == 64 else "f32", "kernel": kernel, "roofline": roof}
    if mfma:
        out["roofline_mfma"] = mfma
    if fence_steps > 0 and pool is not None:
        out["parity_fence"] = parity_fence(Env, n, dev, precision, pool, min(fence_steps, pool.shape[0]))
    return out


def cpu_baseline(precision, seconds=8.0):
    """Oracle (C, fp64, gcc -O3 -march=native, OpenMP over envs) on the same workload -- same action distribution, auto-reset --
    timed twice on a bounded sample: ONE thread (8 192 envs) and every core this process may use (65 536 envs;
    `cores` = affinity mask capped by the cgroup CPU quota, oracle.usable_cores).  Each leg is a child process."""
    import subprocess
    from oracle import oracle as O
    O.build()
    cores = O.usable_cores()

    def leg(n, threads, budget):
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="close", OMP_PLACES="cores")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-leg", str(n), str(threads), str(budget)],
                           env=env, capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            raise RuntimeError("cpu_baseline child failed: " + r.stderr[-1000:])
        return json.loads(r.stdout.strip().splitlines()[-1])

    one = leg(8192, 1, seconds * 0.6)
    full = leg(ENVS_PER_GPU, cores["usable"], seconds)
    try:                      # the metric's own "PyBullet CPU path", only if it exists here
        import pybullet  # noqa: F401
        pyb = "installed (not timed by this script)"
    except Exception as e:    # expected: the wheel is not in the image and there is no network
        pyb = "unavailable: %s" % e
    return {"value": full["value"], "unit": "env-steps/s", "cores": full["threads"], "kind": "port",
            "sample": f"{full['steps']} steps x {ENVS_PER_GPU} envs of the same reach workload in {full['seconds']:.1f}s on "
                      f"{full['threads']} threads (C oracle fp64, gcc -O3 -march=native, OpenMP, OMP_PROC_BIND=close)",
            "threads_1": {"value": one["value"], "unit": "env-steps/s", "cores": 1,
                          "sample": f"{one['steps']} steps x 8192 envs in {one['seconds']:.1f}s on one thread"},
            "parallel_efficiency": full["value"] / (one["value"] * full["threads"]),
            "host": {"affinity_cpus": cores["affinity"], "cgroup_cpu_quota": cores["cgroup_quota"], "os_cpu_count": os.cpu_count()},
            "pybullet": pyb}



def step_api_graph(env, pool, next_actions, n, k2, dev):
    """50 armenv_step launches captured in one hipGraph over a static [50, N, 3] action buffer; every replay is fed the
    next 50 rows of the action pool by one device-to-device copy (39 MB, part of the timed region)."""
    G = 50
    gbuf = torch.empty((G,) + tuple(pool.shape[1:]), dtype=pool.dtype, device=dev)
    gbuf.copy_(next_actions(G))

    def run_g():
        for j in range(G):
            env.step(gbuf[j])
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for j in range(4):
            env.step(gbuf[j])
    torch.cuda.current_stream(dev).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        run_g()
    for _ in range(2):
        gbuf.copy_(next_actions(G))
        graph.replay()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(k2 // G):
        gbuf.copy_(next_actions(G))
        graph.replay()
    e1.record()
    torch.cuda.synchronize(dev)
    w3 = time.perf_counter() - t0
    k3 = (k2 // G) * G
    return {"value": n * k3 / w3, "unit": "env-steps/s", "steps": k3, "steps_per_graph": G,
            "avg_launch_us": e0.elapsed_time(e1) * 1e3 / k3}

def large_batch(Env, dev, args):
    """One GPU holds far more than 65 536 envs (288 GB of HBM; this state is 209 B per env): the same 100-step rollout at
    --large-batch envs.  With more waves than SIMDs the engine launches env_rollout_kernel<..., WAVES = 2> (<= 256 registers
    per lane, two waves per SIMD: the second wave issues into the first one's dependent-f64 waits), same bits."""
    n, T = int(args.large_batch), 100
    gen = torch.Generator(device=dev); gen.manual_seed(4242)
    fresh = lambda: (torch.randn((T, n, 3), device=dev, generator=gen) * 0.686).clamp_(-0.7, 0.7)   # i.i.d. rows, never replayed
    e = Env(n, device=dev, seed=0, precision=args.precision)
    e.reset()
    out = {}
    acts = fresh()
    for _ in range(7):                 # past the first time-limit resets: the steady state of the workload
        e.rollout(T, acts, out=out)
        acts = fresh()
    k = 3
    timed_acts = [acts] + [fresh() for _ in range(k - 1)]
    torch.cuda.synchronize(dev)
    c0 = e.counters()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for a in timed_acts:
        e.rollout(T, a, out=out)
    ev1.record(); torch.cuda.synchronize(dev)
    c1 = e.counters()
    e.close()
    us = ev0.elapsed_time(ev1) * 1e3 / (k * T)
    upd = (c1["ik_updates"] - c0["ik_updates"]) / (c1["env_steps"] - c0["env_steps"])
    tf = (upd * FLOPS_PER_UPDATE + FLOPS_PER_EXIT_FK) * n / (us * 1e-6) / 1e12
    peak = F64_VECTOR_PEAK_TFLOPS if args.precision == 64 else 157.3
    return {"envs": n, "value": n / (us * 1e-6), "unit": "env-steps/s", "us_per_step": us, "steps": k * T,
            "kernel": "reach_rollout<f%d,kuka> built for two waves per SIMD" % args.precision,
            "valu": {"achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak, "ik_updates_per_env_step": upd}}


def parity_fence(Env, n, dev, precision, pool, fence_steps):
    """The same workload on a second handle with the fence counters on (ArmEnvConfig.fence_counters): the share of env
    steps whose IK result lies outside the URDF joint limits, that end with the flange below z = 0.05, whose IK call ran to its
    iteration cap, or whose IK call passed through an ill-conditioned damped system.  The first two were believed to be where
    Bullet's stepSimulation (/root/reference/envs/rl_reach_env.py:258) acts and this kinematic engine does not; the reference's
    recorded run says Bullet does nothing observable there and they are diagnostics now.  On the last two no
    two implementations of the algorithm agree.  Counted over the second half of the leg (steady state: past the first
    time-limit resets), with the cost of the bookkeeping."""
    T = 100
    k = max(2, fence_steps // T)
    S = pool.shape[0]
    out = {}
    # the two handles' launches alternate, each timed by its own pair of events: both see the same clocks and the same phase of
    # the episodes (timed one after the other the first leg ran on a colder chip and the "cost" moved between 8 % and 18 %)
    hs = {fence: Env(n, device=dev, seed=0, precision=precision, fence_counters=fence) for fence in (1, 0)}
    for e in hs.values():
        e.reset()
    evs = {fence: [] for fence in hs}
    c0 = None
    for j in range(k):
        if j == k // 2:
            c0 = hs[1].counters()
        lo = (j * T) % (S - T + 1)
        for fence, e in hs.items():
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
            e.rollout(T, pool[lo:lo + T], out=out)
            ev[1].record()
            if j >= k // 2:
                evs[fence].append(ev)
    torch.cuda.synchronize(dev)
    c1 = hs[1].counters()
    ms = {fence: sum(a_.elapsed_time(b_) for a_, b_ in evs[fence]) for fence in hs}
    for e in hs.values():
        e.close()
    res = {1: (c0, c1, ms[1]), 0: (None, None, ms[0])}
    c0, c1, ms_on = res[1]
    steps = c1["env_steps"] - c0["env_steps"]
    rate = lambda key: (c1[key] - c0[key]) / steps
    return {"limit_step_rate": rate("limit_steps"), "low_flange_step_rate": rate("low_flange_steps"),
            "cap_step_rate": rate("cap_steps"), "illcond_step_rate": rate("illcond_steps"),
            "fence_z": 0.05, "fence_pivot": 1e-2, "ik_max_iters": 20,
            "env_steps_counted": steps, "steps_before_counting": (k // 2) * T,
            "bookkeeping_cost_frac": ms_on / res[0][2] - 1.0,
            "meaning": "share of env steps (limit) whose IK result lies outside the URDF joint limits / (low_flange) that end with the flange "
                       "below fence_z -- diagnostics only since round 4: the reference's own recorded run (real PyBullet, tests/reference_run.py) "
                       "contains 13 % of each and is reproduced to 4e-7 with this engine doing nothing there -- and (cap, illcond) whose IK call did "
                       "not converge / passed through a near-singular pose, where no two implementations of the algorithm agree to 1e-4 in joint "
                       "space (the strict tier of the parity tests excludes an env from such a call to its next reset; the task-space tier does not)"}


class Scratch:
    """A second env handle that only makes load (in-kernel random policy, outputs discarded): `prewarm(ms)` brings the GPU to its
    steady clocks before the W warm-up steps (from idle the first ~3 000 steps, 30 ms, of the headline workload run 10-25 % slower
    than the rest, profiles/r01_launch_costs.txt section 4; the W warm-up steps of the contract are too short to cover that and they
    belong to the benchmarked handle's trajectory), `ahead(ms)` ENQUEUES about `ms` of the same work without waiting for it --
    bench.py puts it in front of every timed region so that each region starts from the clocks of a chip under sustained load, the
    state a rollout engine runs in (round 4's regions followed host round trips of ~0.3 ms each, and what the clocks did in those
    gaps differed from box to box: VERDICT r04 weak #5).  The benchmarked handle is never touched."""

    def __init__(self, Env, n, dev, precision):
        self.dev = dev
        self.env = Env(n, device=dev, seed=987654321, precision=precision)
        self.env.set_policy("random", action_bound=0.7, noise_sigma=0.686, noise_clip=0.7)
        self.env.reset()
        self.bufs = {}
        self.launch, _ = self.env.bind_rollout(100, None, out=self.bufs)
        self.ms_per_launch = 0.7

    def prewarm(self, ms):
        if ms <= 0:
            return
        t0 = time.perf_counter()
        k, busy = 0, 0.0
        duty = float(os.environ.get("ARMENV_BENCH_PREWARM_DUTY", "1.0"))
        while (time.perf_counter() - t0) * 1e3 < ms:
            tb = time.perf_counter()
            for _ in range(4):
                self.launch()
            k += 4
            torch.cuda.synchronize(self.dev)
            busy += time.perf_counter() - tb
            if duty < 1.0:      # (diagnostic, ARMENV_BENCH_PREWARM_DUTY: idle a share of the time -- a chip held at full f64 load may sit at its power cap)
                time.sleep((time.perf_counter() - tb) * (1.0 / duty - 1.0))
        self.ms_per_launch = busy * 1e3 / k

    def ahead(self, ms):
        for _ in range(int(round(ms / self.ms_per_launch))):
            self.launch()

    def close(self):
        self.env.close()
        self.bufs = None


def self_launch(n_ranks, argv):
    """`python bench.py --gpus N` without a launcher around it (how the round driver types it): start the N ranks here, one
    per GPU, as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>
    bench.py <same argv>` in a process group of their own, relay their stdout (rank 0's ONE JSON line) and stderr, and return
    their exit code.  The group is killed if it outlives ARMENV_BENCH_LAUNCH_TIMEOUT seconds (default 1800) or if this
    process is interrupted.  The torch.distributed.run form keeps working: it sets WORLD_SIZE and never gets here."""
    import signal
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver supports dmabuf IPC only (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    limit = float(os.environ.get("ARMENV_BENCH_LAUNCH_TIMEOUT", "1800"))
    proc = subprocess.Popen(cmd, env=env, start_new_session=True)          # stdout / stderr inherited: the line passes through

    def kill_group(*_):
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
    old = {s: signal.signal(s, lambda *_: (kill_group(), sys.exit(130))) for s in (signal.SIGINT, signal.SIGTERM)}
    try:
        return proc.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        kill_group()
        proc.wait()
        print("bench.py: the %d-rank job did not finish within %.0f s and was killed" % (n_ranks, limit), file=sys.stderr)
        return 124
    finally:
        if proc.poll() is None:      # no rank outlives the launcher
            kill_group()
        for s, h in old.items():
            signal.signal(s, h)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--prewarm-ms", type=float, default=150.0,
                    help="busy time on a SCRATCH env handle before the warm-up steps: an idle MI355X needs ~30 ms of load to "
                         "reach its steady clocks (tests/tools/clock_ramp.py); the benchmarked handle is not touched")
    ap.add_argument("--precision", type=int, default=64, choices=[32, 64])
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--gather-every", type=int, default=100, help="steps between episode-return all-gathers")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--repeat-regions", type=int, default=15,
                    help="single-GPU runs: after the contract's timed region, the identical region this many more times on fresh "
                         "action rows -> value_median / value_min / value_max / launch_us_samples (0 = skip)")
    ap.add_argument("--busy-ahead-ms", type=float, default=8.0,
                    help="scratch-handle work enqueued in front of every timed region's opening synchronise, so that the region starts "
                         "from the clocks of a chip under sustained load (0 = nothing in front, round 4's procedure)")
    ap.add_argument("--repeat-same-rows", action="store_true",
                    help="diagnostic: the repeated regions re-use the contract region's own rows of the action pool (the same work, bit "
                         "for bit) instead of fresh ones: separates what a region costs because of WHEN it runs from what its action window costs")
    ap.add_argument("--rehearsals", type=int, default=0,
                    help="untimed passes through the contract's own procedure (W warm-up steps + the bracketed K steps) before the real one; the "
                         "env state and the action cursor are restored afterwards, so the timed trajectory is unchanged (0 = none)")
    ap.add_argument("--ab-regions", type=int, default=8,
                    help="single-GPU runs with --repeat-regions > 0: the region this many more times under each of round 4's two "
                         "regimes (nothing in front of the region; state restored through the host / on the device) with clock "
                         "probes around every region -> line.ab_host_restore / ab_device_restore (0 = skip)")
    ap.add_argument("--state-digest", action="store_true",
                    help="add config.state_digest: per rank, the sha256 of the joint angles of its envs right after the timed "
                         "region (tests: rank shards reproduce the single-handle trajectory)")
    ap.add_argument("--large-batch", type=int, default=1048576,
                    help="extra leg (reach, external actions, one GPU): the same rollout at this many envs on ONE GPU -- more "
                         "waves than SIMDs, so the engine runs the two-waves-per-SIMD form of the kernel (0 = skip)")
    ap.add_argument("--fence-steps", type=int, default=1200,
                    help="length of the extra parity-fence leg (second handle, fence_counters=1; 0 = skip): how often this "
                         "workload leaves the URDF joint limits / drives the flange below z = 0.05")
    ap.add_argument("--secondary-legs", type=int, default=1,
                    help="1: after the headline (reach, external actions) also time BASELINE configs[2] (reach + fused TD3 actor, exact "
                         "f32 and f16x3) and configs[3] (push, 32 768 envs) with short legs on fresh handles; 0 = skip")
    ap.add_argument("--task", default="reach", choices=["reach", "push", "pick"],
                    help="reach = BASELINE configs[1] (headline); push = configs[3] (use --envs-per-gpu 32768); "
                         "pick = the next-row env (SURVEY.md section 8f.4)")
    ap.add_argument("--policy", default="external", choices=["external", "random", "actor", "actor_f16x3", "datd3", "daddpg"],
                    help="external = pre-generated actions in HBM (configs[1], headline); random / actor = fused in-kernel "
                         "policy (actor = configs[2]: TD3 actor forward folded into the rollout kernel)")
    ap.add_argument("--mode", default="rollout", choices=["rollout", "step"])
    ap.add_argument("--rollout-steps", type=int, default=100,
                    help="env steps fused per armenv_rollout launch (default: the reference's logging period, main.py:130)")
    args = ap.parse_args()

    from armenv import envs
    from armenv.dist import ReturnGatherer, ShmBarrier, env_rank_world, init_process_group

    rank, local_rank, world = env_rank_world()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process becomes the launcher of N ranks (one per GPU) and relays rank 0's line
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d, or with plain "
                         "`python bench.py --gpus %d` (no WORLD_SIZE in the environment)" % (args.gpus, world, args.gpus, args.gpus))
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", local_rank % max(1, ndev))
    torch.cuda.set_device(dev)
    # one rank per GPU over RCCL; if the node has fewer GPUs than ranks (debugging on a 1-GPU box) the ranks share
    # GPUs and the logging collective falls back to gloo -- reported in config.parallelism
    backend = "nccl" if world <= ndev else "gloo"
    # ARMENV_BENCH_COLLECTIVE=1: a ONE-rank job takes the multi-rank path too -- process group, barriers, the logging all-gather on
    # the side stream, the max-over-ranks reductions -- so that a 1-GPU box executes every RCCL call an 8-GPU node would
    # (tests/test_gpu_fence.py::test_bench_one_rank_over_rccl).  Launch under torch.distributed.run --nproc-per-node 1.
    multi = world > 1 or os.environ.get("ARMENV_BENCH_COLLECTIVE") == "1"
    init_process_group(backend, dev if backend == "nccl" else None, force=multi)

    n = args.envs_per_gpu
    Env = {"reach": envs.BatchedReachEnv, "push": envs.BatchedPushEnv, "pick": envs.BatchedPickEnv}[args.task]
    env = Env(n, device=dev, seed=0, env_id_offset=rank * n, precision=args.precision)
    gen = torch.Generator(device=dev); gen.manual_seed(1000 + rank)
    # the random policy "pre-generated on device for 1 000 steps as [1000, N, 3] f32
    # (786 MB)": i.i.d. across steps, so every env does a genuine random walk.  (A short ring of action tensors replayed
    # in a cycle makes every env drift ballistically into a corner of the workspace box and sit there -- a different,
    # slower workload: more lanes with 5+ IK trips per wave.)  Capped at 2 GiB for very large batches.
    # ARMENV_BENCH_POOL: rows of the pool (experiments with shorter, i.e. periodic, action sequences)
    S = int(max(64, min(int(os.environ.get("ARMENV_BENCH_POOL", "1000")), (2 << 30) // (12 * n))))
    if args.task == "reach":      # run() exploration with a zero actor, main.py:116-117
        pool = (torch.randn((S, n, 3), device=dev, generator=gen) * 0.686).clamp_(-0.7, 0.7)
    else:                         # train_push_with_TD3 exploration, unclipped, main.py:457,484
        pool = torch.randn((S, n, 3), device=dev, generator=gen) * 0.392
    if args.policy != "external":
        args.mode = "rollout"
        bound, sig = (0.7, 0.7 * 0.98) if args.task == "reach" else (0.4, 0.4 * 0.98)
        sd = None
        if args.policy.startswith("actor"):      # weights of TD3_MLP(6,3,0.7) under torch.manual_seed(0): golden G3
            sd = golden_actor()
            if args.task != "reach":
                raise SystemExit("--policy actor: the golden actor has 6 inputs (reach)")
        if args.policy in ("datd3", "daddpg"):       # the fused two-actor policies (golden nets G11 / G15: 6-float observations)
            if args.task != "reach":
                raise SystemExit("--policy %s: the golden nets have 6 inputs (reach)" % args.policy)
            if args.policy == "datd3":
                env.set_policy_datd3(*golden_datd3(), action_bound=bound, noise_sigma=sig, noise_clip=bound)
            else:
                env.set_policy_daddpg(*golden_daddpg(), action_bound=bound, noise_sigma=sig, noise_clip=bound)
        else:
            env.set_policy(args.policy, action_bound=bound, noise_sigma=sig, noise_clip=bound if args.task == "reach" else 1e9,
                           actor_state_dict=sd)
    gather = ReturnGatherer(n, dev, world, collective=multi)
    if multi:
        gather.warm_up()       # the communicator's first collectives set up channels: not inside a rollout loop, not inside the clock
    env.reset()
    R = max(1, min(args.rollout_steps, args.steps))
    R = min(R, S)
    bufs = {}
    cursor = [0]     # next row of the action pool

    def next_actions(r):
        """r consecutive rows of the pool as a contiguous view (wraps to row 0 when fewer than r are left)"""
        if cursor[0] + r > S:
            cursor[0] = 0
        a = pool[cursor[0]:cursor[0] + r]
        cursor[0] += r
        return a

    def do_gather():
        """A logging all-gather in the middle of a region: armenv_episode_stats on the launch stream (the next launch overwrites
        what it reads, so it has to sit in front of it anyway), the collective on the side stream."""
        gather.launch(env.episode_returns_f32())

    def do_gather_trailing():
        """The all-gather that ends a region: armenv_episode_stats too goes to the side stream (behind the region's steps), so the
        launch stream carries the K steps and nothing else and its synchronise closes the clock on them; the collective is waited
        for -- and checked -- after the clock (logging only, never on the step critical path)."""
        gather.launch_into(lambda stage, stream=None: env.episode_returns_f32(out=stage, stream=stream), takes_stream=True)

    def plan(k):
        """The launches of exactly k env steps of every env of this rank, prepared up front (buffers, pointers, the
        positions of the logging all-gathers): what remains for the timed region is one C call per launch.
        Multi-rank runs fire the episode-return all-gather every --gather-every steps AND at least once per region."""
        ops, gathers = [], 0
        if args.mode == "step":
            for i in range(k):
                a = next_actions(1)[0]
                ops.append(lambda a=a: env.step(a))
                if multi and (i + 1) % args.gather_every == 0:
                    ops.append(do_gather); gathers += 1
        else:
            done_steps = 0
            while done_steps < k:
                r = min(R, k - done_steps)
                a_in = None if args.policy != "external" else next_actions(r)
                launch, _ = env.bind_rollout(r, a_in, out=bufs if r == R else None)
                ops.append(launch)
                done_steps += r
                if multi and (done_steps // args.gather_every) != ((done_steps - r) // args.gather_every):
                    ops.append(do_gather); gathers += 1
        if multi and ops and ops[-1] is do_gather:
            ops[-1] = do_gather_trailing
        if multi and gathers == 0:
            # a region shorter than --gather-every (the driver's 20 steps) still carries one all-gather, of the returns as they
            # stand AFTER its steps: issued inside the region, on the side stream behind the launches
            ops.append(do_gather_trailing); gathers = 1
        launches = len(ops) - gathers
        return ops, launches, gathers

    def run(k):
        ops, launches, gathers = plan(k)
        for op in ops:
            op()
        if gathers:
            gather.order_after_read()      # the next launch overwrites what a side-stream armenv_episode_stats may still be reading
        return launches

    host_us = {}
    extra = {}
    shm = ShmBarrier(rank, world)

    # every region of the run marks the probes twice: the contract's, its repeats, both A/B regimes, the rehearsals
    probes = ClockProbes(dev, cap=2 * (2 + args.rehearsals + args.repeat_regions + 2 * args.ab_regions) + 8)

    def timed(k, tag=None, count=True, ahead_ms=0.0):
        """tag: name of the region for the clock probes (one sample enqueued ahead of the opening synchronise, one after the
        closing one; both outside the clock and the events).  count=False: no armenv_counters round trips around the region.
        ahead_ms: this much scratch-handle work is enqueued right before the opening synchronise (Scratch.ahead)."""
        ops, launches, gathers = plan(k)
        # the launches' output buffers may be fresh allocations (the first region of a run: plan() has just made them): written
        # once here so that the region does not pay their first touch (cold TLB entries: the contract's region measured 3-6 us
        # slower than its 15 repeats into the same buffers, gpurun_out/r05/bench_driver_c.json)
        for v_ in bufs.values():
            v_.zero_()
        evs = HipEvents(dev)
        evs.record(0); evs.record(1)  # first use outside the region
        if count:
            torch.cuda.synchronize(dev)
        c0 = env.counters() if count else None
        if multi:
            dist.barrier()           # RCCL's own barrier, OUTSIDE the clock: the ranks arrive together, then load their chips
        if ahead_ms > 0:
            scratch.ahead(ahead_ms)
        probes.mark(tag and tag + ":before")
        # The bracket, the same code for one rank and for N: device synchronise + barrier, the clock, the K steps, the launch
        # stream's synchronise + barrier, the clock.  The barrier is the single-node shared-memory one (armenv.dist.ShmBarrier,
        # ~1 us): through round 5 the multi-rank bracket closed on gather.result() + a device synchronise + dist.barrier(), and a
        # ONE-rank job through RCCL lost 41 % of a 20-step region to that -- N > 1 measured the bracket,
        # not the engine, and was not commensurable with N = 1.  The logging all-gather is still issued inside the region (side
        # stream); it is waited for and verified right after the clock, and the old figure is kept as value_bracketed.
        torch.cuda.synchronize(dev)
        shm.wait()
        p = time.perf_counter
        trailing = multi and len(ops) > 1 and ops[-1] is do_gather_trailing     # the region's last op is a logging gather
        step_ops = ops[:-1] if trailing else ops
        t0 = p()
        evs.record(0)
        ta = p()
        for op in step_ops:
            op()
        tb = p()
        evs.record(1)                # closes the K steps on the launch stream
        tc = p()
        if trailing:
            do_gather_trailing()     # host work under the running kernels; its device work is on the side stream, behind the steps
        tt = p()
        # The K steps are done on this rank when its launch stream is idle (wall_steps): ONE stream synchronise.  (Rounds 1-3 waited
        # for ev1 first and synchronised the stream after it: the second call returned at once as far as the GPU was concerned and
        # still cost ~5 us of host time inside the clock.)
        evs.stream_synchronize()
        td = p()
        wall_steps = td - t0
        shm.wait()                   # ... and on every rank when the slowest one is through
        te = p()
        wall = te - t0               # `value`: barrier + synchronise to synchronise + barrier
        if multi:
            gather.result()          # orders the launch stream behind the collective ...
            torch.cuda.synchronize(dev)   # ... and the device synchronise waits for it
        tf = p()
        if multi:
            dist.barrier()
        tg = p()
        wall_bracketed = tg - t0     # rounds 1-5's clock: + the logging collective's completion + RCCL's barrier (value_bracketed)
        host_us.update(event0_record=(ta - t0) * 1e6, enqueue=(tb - ta) * 1e6, event1_record=(tc - tb) * 1e6,
                       gather_issue=(tt - tc) * 1e6, wait_for_gpu=(td - tt) * 1e6, shm_barrier=(te - td) * 1e6,
                       gather_wait=(tf - te) * 1e6, barrier=(tg - tf) * 1e6,
                       # issue -> complete as the host sees it (the collective starts behind the region's steps on the device)
                       collective=((tf - tc) * 1e6 if trailing or gathers else 0.0))
        extra.update(wall_bracketed=wall_bracketed)
        if multi and count and gathers:
            # after the clock: what the collective delivered is this rank's own vector in this rank's place
            g_ = gather.result()
            mine_ = env.episode_stats()[0].to(torch.float32)
            extra["collective_verified"] = bool(torch.equal(g_[rank * n:(rank + 1) * n], mine_)) and g_.numel() == world * n
        probes.mark(tag and tag + ":after")
        c1 = env.counters() if count else None
        gpu_ms = evs.elapsed_ms()
        evs.close()
        return wall, gpu_ms, launches, gathers, ({k_: c1[k_] - c0[k_] for k_ in c1} if count else None), wall_steps

    scratch = Scratch(Env, n, dev, args.precision)
    # the timed launches' output buffers are allocated (and touched) BEFORE the device is warmed: a fresh 60 MB block is a hipMalloc
    # of milliseconds with the GPU idle, and an idle gap between the pre-warm and the contract's region costs that region its clocks
    # (gpurun_out/r05/prewarm_*.json: the first region 125-131 us against 121 for its repeats, whatever the pre-warm's length)
    env.bind_rollout(R, None if args.policy != "external" else pool[:R], out=bufs)
    for v_ in bufs.values():
        v_.zero_()
    scratch.prewarm(args.prewarm_ms)
    # Dress rehearsal (--rehearsals, default 0): the contract's own procedure -- W warm-up steps, then the bracketed K steps -- untimed,
    # after which the env state and the action-pool cursor are put back, so that the contract's region below runs exactly the launches
    # it would have run.  Built to test whether the contract's region is 5-8 us slower than the median of its repeats because it is the
    # FIRST pass through this code path: it is not (profiles/r05_region_clock_probe.txt, appendix) -- kept as an option.
    if args.rehearsals > 0:
        snap0 = {k: v.clone() for k, v in env.get_state().items()}
        for _ in range(args.rehearsals):
            c_keep = cursor[0]
            run(args.warmup)
            timed(args.steps, ahead_ms=args.busy_ahead_ms)
            env.set_state(**snap0, sync=False)
            cursor[0] = c_keep
        del snap0
    run(args.warmup)
    cursor0 = cursor[0]
    snap = {k: v.clone() for k, v in env.get_state().items()} if (not multi and args.repeat_regions > 0) else None
    if snap is not None and os.environ.get("ARMENV_BENCH_RESTORE_FIRST") == "1":
        env.set_state(**snap, sync=False)       # diagnostic: the contract region behind the same (idempotent) restore as its repeats
    wall, gpu_ms, launches, gathers, dc, wall_steps = timed(args.steps, tag="r0", ahead_ms=args.busy_ahead_ms)
    host_us_main = dict(host_us)
    extra_main = dict(extra)

    # The headline is ONE sample of a short region (the driver's 20 steps are one ~120 us launch): the same region again, 15
    # times -- the env state restored ON THE DEVICE (one kernel on the launch stream, no host round trip) to what it was when the
    # contract's region started (same phase of the episodes: the cost of a step drifts with the time since the common reset, 6.6 us
    # at step 25, 7.9 at step 330), fresh rows of the action pool, same handle, same launch shape, same bracket, the same
    # --busy-ahead-ms of scratch work in front -- for the spread.  `value` stays the first region, the contract's; the repeats are
    # reported beside it.  The trajectory then continues from the last repeat's end.
    repeats = []
    ab = {}
    if snap is not None:
        for j in range(args.repeat_regions):
            env.set_state(**snap, sync=False)
            if args.repeat_same_rows:
                cursor[0] = cursor0
            w_, g_, l_, _, _, _ = timed(args.steps, tag="r%d" % (j + 1), count=os.environ.get("ARMENV_BENCH_COUNT_REPEATS") == "1",
                                        ahead_ms=args.busy_ahead_ms)
            repeats.append((w_, g_ * 1e3 / l_))
        # Round 4's procedure beside it, eight regions each, same clock probes: NOTHING in front of the region, the state restored
        #   host_restore:   through the host (set_state + synchronise, armenv_counters D2H before and after: ~0.3 ms between regions);
        #   device_restore: on the device.
        # Where these differ from the samples above the difference is what the clocks do when the chip is left idle between short
        # kernels -- chip- and firmware-dependent, which is why the headline regions no longer depend on it.
        for mode in ("host_restore", "device_restore")[:2 if args.ab_regions > 0 else 0]:
            res = []
            for j in range(args.ab_regions):
                env.set_state(**snap, sync=(mode == "host_restore"))
                w_, g_, l_, _, _, _ = timed(args.steps, tag="%s%d" % (mode, j), count=(mode == "host_restore"))
                res.append((w_, g_ * 1e3 / l_))
            ab[mode] = res
    scratch.close()

    t = torch.tensor([wall, gpu_ms * 1e-3, wall_steps, extra_main["wall_bracketed"], 0.0 if extra_main.get("collective_verified", True) else 1.0],
                     dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    per_rank = None
    if multi:       # every rank's wall clock and kernel time of the region (stragglers show here); value uses the MAX wall
        allt = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank = {"wall_ms": [float(x[0]) * 1e3 for x in allt], "kernel_ms": [float(x[1]) * 1e3 for x in allt],
                    "wall_steps_ms": [float(x[2]) * 1e3 for x in allt], "wall_bracketed_ms": [float(x[3]) * 1e3 for x in allt]}
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall_max = float(t[0].item())
    wall_steps_max = float(t[2].item())
    wall_bracketed_max = float(t[3].item())
    collective_ok = float(t[4].item()) == 0.0
    counters = env.counters()
    digests = None
    gathered = None
    if args.state_digest:
        import hashlib
        mine = hashlib.sha256(env.get_state()["q"].cpu().numpy().tobytes()).hexdigest()
        digests = [mine]
        if multi:
            digests = [None] * world
            dist.all_gather_object(digests, mine)
            # one more logging all-gather, of the returns as they stand after the run: what every rank now holds
            do_gather()
            g_all = gather.result().detach().float().cpu().numpy()
            gathered = {"sha256": hashlib.sha256(g_all.tobytes()).hexdigest(), "mean": float(g_all.astype(np.float64).mean())}

    step_api = None
    if args.mode == "rollout" and not multi and args.policy == "external":
        args.mode = "step"                      # the gym-style one-launch-per-step path, timed beside the headline
        k2 = min(args.steps, 500)
        run(10)
        w2, g2, _, _, _, _ = timed(k2)
        step_api = {"value": n * k2 / w2, "unit": "env-steps/s", "steps": k2, "avg_launch_us": g2 * 1e3 / k2,
                    "kernel": env.kernel_name}
        # the same launches replayed from a hipGraph (50 armenv_step calls per graph): host launch cost removed
        if k2 >= 50:
            step_api["hipgraph"] = step_api_graph(env, pool, next_actions, n, k2, dev)
        args.mode = "rollout"

    # "pre-generated on device ... or generated in-kernel for the persistent variant --
    # report both".  Same rollout kernel family with the random policy drawn in-kernel (Philox), timed beside the headline.
    in_kernel = None
    if step_api is not None and args.steps >= R:
        bound, sig = (0.7, 0.7 * 0.98) if args.task == "reach" else (0.4, 0.4 * 0.98)
        env.set_policy("random", action_bound=bound, noise_sigma=sig, noise_clip=bound if args.task == "reach" else 1e9)
        lr = max(2, min(args.steps // R, 10))
        for _ in range(2):
            env.rollout(R, None, out=bufs)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(lr):
            env.rollout(R, None, out=bufs)
        e1.record()
        torch.cuda.synchronize(dev)
        w4 = time.perf_counter() - t0
        in_kernel = {"value": n * lr * R / w4, "unit": "env-steps/s", "steps": lr * R,
                     "us_per_step": e0.elapsed_time(e1) * 1e3 / (lr * R), "kernel": env.kernel_name.replace("_step", "_rollout") + "+philox_policy"}

    if rank == 0:
        total_envs = n * world
        value = total_envs * args.steps / wall_max
        # the same K steps against the slowest rank's KERNEL time (HIP events on its launch stream): what the GPUs did, without
        # the host-side bracket (event records, enqueue, the barrier's wake-up latency -- fixed costs of the order of one
        # 20-step launch, see config.host_us); equals the events' figure of the one rank when n_gpus = 1
        kernel_ms_max = max(per_rank["kernel_ms"]) if per_rank else gpu_ms
        value_kernel = total_envs * args.steps / (kernel_ms_max * 1e-3)
        launch_us = gpu_ms * 1e3 / launches              # HIP events on the launch stream, per kernel launch
        steps_per_launch = args.steps / launches
        kernel = env.kernel_name if args.mode == "step" else env.kernel_name.replace("_step", "_rollout")
        updates = dc["ik_updates"] / max(1, dc["env_steps"])
        roof, valu, mfma = rooflines(args.task, args.policy, args.precision, n, steps_per_launch, launch_us, updates, kernel)
        pol_txt = {"external": "random policy %s pre-generated in HBM as an i.i.d. [steps, N, 3] pool, step() throughput only",
                   "random": "random policy %s generated in-kernel (Philox)",
                   "actor": "TD3 actor forward (exact f32 MFMA) + exploration noise %s fused into the step kernel",
                   "actor_f16x3": "TD3 actor forward (f16 MFMA, 3-pass hi/lo split) + exploration noise %s fused into the step kernel",
                   "datd3": "DATD3_MLP.take_action (two actors, two critics, f16x3 MFMA passes) + exploration noise %s fused into the step kernel",
                   "daddpg": "DADDPG_MLP.take_action (two actors, one critic on both proposals, f16x3 MFMA passes) + exploration noise %s fused "
                             "into the step kernel"}
        noise = "clip(N(0,0.686),+-0.7)" if args.task == "reach" else "N(0,0.392)"
        workload = {"reach": "rl_reach_env %d parallel envs per GPU, %s, KUKA iiwa chain, auto-reset on",
                    "push": "rl_push_env %d parallel envs per GPU (arm FK/IK + cube contact/overlap test), %s, auto-reset on",
                    "pick": "rl_pick_env %d parallel envs per GPU (arm FK/IK + gripper trigger / hold model), %s, auto-reset on",
                    }[args.task] % (n, pol_txt[args.policy] % noise)
        line = {
            "metric": "env-steps/sec at N parallel envs (rl_%s_env)" % args.task,
            "value": value, "value_kernel": value_kernel, "value_steps": total_envs * args.steps / wall_steps_max,
            # rounds 1-5's clock for N > 1 (the logging collective's completion, a device synchronise and RCCL's barrier inside it)
            "value_bracketed": total_envs * args.steps / wall_bracketed_max,
            "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall_max * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64" if args.precision == 64 else "f32", "data": "synthetic",
            "config": {"workload": workload,
                       "envs_per_gpu": n, "total_envs": total_envs, "kernel": kernel, "mode": args.mode, "policy": args.policy,
                       "steps_per_launch": steps_per_launch, "launches": launches, "device_prewarm_ms": args.prewarm_ms,
                       "busy_ahead_ms": args.busy_ahead_ms, "rehearsals": args.rehearsals,
                       # how `value` is taken: 1 = rounds 1-4 (W warm-up steps, then the K steps after host round trips:
                       # this line's ab_host_restore is that regime); 2 = round 5 (--busy-ahead-ms of scratch work in front of every
                       # region); 3 = round 6 (2 + the shared-memory bracket, identical for 1 and N ranks)
                       "procedure_version": 3,
                       "gathers_in_timed_region": gathers, "state_digest": digests, "per_rank": per_rank,
                       "gathered_returns_sha256": gathered["sha256"] if gathered else None,
                       "gathered_returns_mean": gathered["mean"] if gathered else None,
                       # what torch.distributed reports about the job this line was measured in (an 8-GPU node shows 8 / "nccl" = RCCL)
                       "rccl_ranks_seen": {"world_size": dist.get_world_size() if dist.is_initialized() else 1,
                                           "backend": dist.get_backend() if dist.is_initialized() else None},
                       # where the wall clock of the timed region went on the host (us): recording the two HIP events,
                       # enqueueing the launches, waiting for the GPU, the closing stream synchronisation, the barrier; and,
                       # outside the clock, the wait for the logging all-gather
                       "host_us": host_us_main,
                       "bracket": "device synchronise + shared-memory barrier | clock | K steps | launch-stream synchronise + shared-memory barrier "
                                  "| clock (the same code for 1 and N ranks); dist.barrier() before and after, outside the clock",
                       "collective_us": host_us_main.get("collective", 0.0), "barrier_us": host_us_main.get("shm_barrier", 0.0),
                       "rccl_barrier_us": host_us_main.get("barrier", 0.0),
                       "collective_verified": (collective_ok if multi else None),
                       # how the logging all-gather is issued: RCCL's own C API on the side stream (armenv.dist.RcclComm) or torch.distributed
                       "collective_transport": ("rccl-direct" if gather.rccl is not None else ("torch.distributed" if multi else None)),
                       "collective_direct_error": gather.direct_error,
                       "parallelism": "env-sharded x%d, %s all-gather of episode returns every %d steps and at least once per "
                                      "timed region (logging only, side stream; issued INSIDE the clock of `value`, waited for and verified "
                                      "right after it -- config.collective_us; inside the clock of `value_bracketed`)"
                                      % (world, "RCCL" if backend == "nccl" else "gloo (ranks share a GPU: debug)", args.gather_every)
                                      if multi else "single GPU"},
            # `roofline.binding_bound` / `roofline.valu`: the bound that BINDS: 29 flop/B
            # puts the path right of the ridge -- the same launch against the f64 (f32) vector peak
            "roofline": roof,
            "roofline_valu": valu,
            "episodes_finished": counters["episodes"], "nonfinite_states": counters["nonfinite"],
        }
        if mfma:
            line["roofline_mfma"] = mfma
        if repeats:
            vals = sorted([value] + [total_envs * args.steps / w_ for w_, _ in repeats])
            us = [launch_us] + [u_ for _, u_ in repeats]
            line.update({"value_median": vals[len(vals) // 2], "value_min": vals[0], "value_max": vals[-1],
                         "regions": 1 + len(repeats),
                         "launch_us_samples": [round(u_, 2) for u_ in us],
                         "launch_us_median": sorted(us)[len(us) // 2], "launch_us_min": min(us), "launch_us_max": max(us)})
            pr = probes.read()
            aft = [pr.get("r%d:after" % j) for j in range(len(us))]
            bef = [pr.get("r%d:before" % j) for j in range(len(us))]
            # a probe that failed or did not fit costs the probe fields, never the line
            if pr and all(x is not None for x in aft + bef):
                # ns per chained v_fma_f32 of the probe waves (median over all SIMDs) right after each region: ratio of two samples =
                # inverse ratio of the shader clocks they ran at.  launch_us_at_fastest_clock rescales each launch to the run's
                # fastest sample: if the spread of the launches is the clocks', it collapses here.
                fastest = min(v_["ns"] for v_ in pr.values())
                line.update({"clock_probe_ns_samples": [round(x["ns"], 4) for x in aft],
                             "clock_probe_ns_before": [round(x["ns"], 4) for x in bef],
                             "clock_probe_ns_slowest_xcd": [round(x["ns_slowest_xcd"], 4) for x in aft],
                             "clock_probe_memtime_ticks_per_instruction": [round(x["memtime_ticks"], 4) for x in aft],
                             "clock_probe_ns_fastest": fastest, "clock_probe_xcds": probes.xcds,
                             "clock_probe_device_time_us": [round(x["at_us"], 1) for x in bef],
                             "launch_us_at_fastest_clock": [round(u_ * fastest / a_["ns"], 2) for u_, a_ in zip(us, aft)]})
                for mode, res in ab.items():
                    mu = [u_ for _, u_ in res]
                    ma = [pr.get("%s%d:after" % (mode, j)) for j in range(len(res))]
                    if any(x is None for x in ma):
                        continue
                    line["ab_" + mode] = {"launch_us_samples": [round(u_, 2) for u_ in mu], "launch_us_median": sorted(mu)[len(mu) // 2],
                                          "launch_us_min": min(mu), "launch_us_max": max(mu),
                                          "value_median": sorted(total_envs * args.steps / w_ for w_, _ in res)[len(res) // 2],
                                          "clock_probe_ns_samples": [round(x["ns"], 4) for x in ma],
                                          "launch_us_at_fastest_clock": [round(u_ * fastest / a_["ns"], 2) for u_, a_ in zip(mu, ma)]}
        if step_api:
            line["step_api"] = step_api
        if in_kernel:
            line["in_kernel_policy"] = in_kernel
        # the secondary legs never take the headline down with them: a failure is reported in place of the leg
        def leg(name, fn):
            try:
                line[name] = fn()
            except Exception as e:       # noqa: BLE001
                line[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        if not multi and args.fence_steps > 0:
            leg("parity_fence", lambda: parity_fence(Env, n, dev, args.precision, pool, args.fence_steps))
        if not multi and args.large_batch > n and args.task == "reach" and args.policy == "external" and args.mode == "rollout":
            del pool, bufs
            torch.cuda.empty_cache()
            leg("large_batch", lambda: large_batch(Env, dev, args))
        # the other single-GPU BASELINE configs, timed by the same command (each on a fresh handle, after the headline)
        if not multi and args.secondary_legs and args.task == "reach" and args.policy == "external" and args.mode == "rollout":
            # (four timed launches behind three untimed ones: the clocks settle for ~10 ms after the switch to the MFMA-heavy kernel)
            leg("config3_actor_f32", lambda: secondary_leg(envs, dev, "reach", ENVS_PER_GPU, "actor", args.precision, 4, 3))
            leg("config3_actor_f16x3", lambda: secondary_leg(envs, dev, "reach", ENVS_PER_GPU, "actor_f16x3", args.precision, 4, 3))
            leg("config4_push", lambda: secondary_leg(envs, dev, "push", 32768, "external", args.precision, 5, 6, args.fence_steps))
            # beyond the configs: DATD3_MLP.take_action (a consumer north_star names) folded into the same rollout kernel
            leg("datd3_fused", lambda: secondary_leg(envs, dev, "reach", ENVS_PER_GPU, "datd3", args.precision, 2, 2))
            # the reference's DEFAULT agent (config.py:33 opt.algo = 'DADDPG_MLP'): two actors, one critic valuing both proposals
            leg("daddpg_fused", lambda: secondary_leg(envs, dev, "reach", ENVS_PER_GPU, "daddpg", args.precision, 2, 2))
        if not multi and not args.no_cpu_baseline:
            leg("cpu_baseline", lambda: cpu_baseline(args.precision))
        # the scalars a record that keeps only flat `config` values would otherwise lose
        cfg = line["config"]
        cfg["rccl_world_size"] = cfg["rccl_ranks_seen"]["world_size"]
        cfg["rccl_backend"] = cfg["rccl_ranks_seen"]["backend"]
        for k_ in ("value_median", "value_min", "value_max", "launch_us_median", "launch_us_min", "launch_us_max"):
            if k_ in line:
                cfg[k_] = line[k_]
        for k_, leg_ in (("actor_f32", "config3_actor_f32"), ("actor_f16x3", "config3_actor_f16x3"), ("push", "config4_push"), ("datd3", "datd3_fused"), ("daddpg", "daddpg_fused")):
            if isinstance(line.get(leg_), dict) and "us_per_step" in line[leg_]:
                cfg[k_ + "_us_per_step"] = line[leg_]["us_per_step"]
                cfg[k_ + "_env_steps_per_s"] = line[leg_]["value_kernel"]
        for mode in ("host_restore", "device_restore"):
            if "ab_" + mode in line:
                for k_ in ("launch_us_median", "launch_us_min", "launch_us_max"):
                    cfg["ab_%s_%s" % (mode, k_)] = line["ab_" + mode][k_]
        if line.get("clock_probe_ns_samples"):
            cfg["clock_probe_ns_min"] = min(line["clock_probe_ns_samples"])
            cfg["clock_probe_ns_max"] = max(line["clock_probe_ns_samples"])
            lf = line["launch_us_at_fastest_clock"]
            cfg["launch_us_at_fastest_clock_min"], cfg["launch_us_at_fastest_clock_max"] = min(lf), max(lf)
        if step_api:
            cfg["step_api_us"] = step_api["avg_launch_us"]
        if in_kernel:
            cfg["in_kernel_policy_us_per_step"] = cfg["in_kernel_us_per_step"] = in_kernel["us_per_step"]
        if isinstance(line.get("large_batch"), dict) and "us_per_step" in line["large_batch"]:
            cfg["large_batch_env_steps_per_s"] = line["large_batch"]["value"]
        if isinstance(line.get("cpu_baseline"), dict) and "value" in line["cpu_baseline"]:
            cfg["cpu_env_steps_per_s"] = line["cpu_baseline"]["value"]
            cfg["cpu_cores"] = line["cpu_baseline"]["cores"]
        print(json.dumps(line), flush=True)
    env.close()
    gather.close()
    shm.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
