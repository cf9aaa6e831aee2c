"""CPU tests of the host side: C-ABI library loads and exports every symbol include/armenv.h declares,
struct layouts agree with the header, URDF reader, gym-like surface pieces, error behaviour without a GPU."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from conftest import ROOT, golden_json

HEADER = os.path.join(ROOT, "include", "armenv.h")


def test_library_exports_every_declared_symbol():
    from armenv import _lib as L
    lib = L.load()
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    declared = set(re.findall(r"\b(armenv_[a-z_0-9]+)\s*\(", src))
    assert declared, "no declarations parsed"
    assert declared == set(L.SYMBOLS), (declared ^ set(L.SYMBOLS))
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.armenv_abi_version() == L.ABI_VERSION


def test_struct_layout_matches_header():
    from armenv import _lib as L
    prog = r'''
#include "armenv.h"
#include <stdio.h>
#include <stddef.h>
int main(void){
  printf("%zu %zu %zu %zu %zu %zu\n", sizeof(ArmEnvConfig), sizeof(ArmEnvChain), offsetof(ArmEnvConfig, seed),
         offsetof(ArmEnvConfig, q_init), offsetof(ArmEnvConfig, push_success_dis), offsetof(ArmEnvConfig, chain));
  printf("%zu %zu %zu\n", offsetof(ArmEnvConfig, pick_gripper_length), offsetof(ArmEnvConfig, ik_tip_offset),
         offsetof(ArmEnvConfig, rollout_ready_lanes));
  return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(prog)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        out = subprocess.check_output([os.path.join(d, "t")]).decode().split()
    want = [C.sizeof(L.ArmEnvConfig), C.sizeof(L.ArmEnvChain), L.ArmEnvConfig.seed.offset, L.ArmEnvConfig.q_init.offset,
            L.ArmEnvConfig.push_success_dis.offset, L.ArmEnvConfig.chain.offset, L.ArmEnvConfig.pick_gripper_length.offset,
            L.ArmEnvConfig.ik_tip_offset.offset, L.ArmEnvConfig.rollout_ready_lanes.offset]
    assert [int(x) for x in out] == want


def test_default_config_holds_reference_constants():
    from armenv import _lib as L
    c = L.default_config(L.TASK_REACH)
    assert (c.dv, c.reach_dis, c.max_steps) == (0.02, 0.01, 500)                 # config.py:41,42,51
    assert list(c.box_lo) == [0.2, -0.3, 0.0] and list(c.box_hi) == [0.7, 0.3, 0.55]   # rl_reach_env.py:221-223
    assert list(c.q_init) == golden_json("fk_kat.json")["q"]                      # rl_reach_env.py:116-119
    assert (c.ik_lambda, c.ik_residual, c.ik_max_iters) == (1e-5, 1e-4, 20)
    assert c.clamp_joint_limits == 0 and c.precision == 64
    p = L.default_config(L.TASK_PUSH)
    assert p.dv == 0.08 and list(p.box_hi) == [0.7, 0.3, 0.1]                     # rl_push_env.py:314,322
    k = L.default_config(L.TASK_PICK)
    assert k.dv == 0.08 and abs(k.box_hi[2] - (0.55 + 0.257)) < 1e-15              # rl_pick_env.py:313,322
    assert (k.pick_gripper_length, k.pick_trigger_dis) == (0.257, 0.006)          # rl_pick_env.py:79,412
    assert list(k.goal_hi) == [0.7, 0.3, 0.55] and list(k.q_init) == list(c.q_init)   # :61-66, :121-125 (first 7)
    with pytest.raises(L.ArmEnvError):
        L.default_config(7)


def test_builtin_chain_equals_urdf_assets():
    from armenv import _lib as L
    from armenv.urdf import builtin_chain
    for robot, rid in (("kuka", L.ROBOT_KUKA), ("diana", L.ROBOT_DIANA)):
        s = L.ArmEnvChain()
        L.check(L.load().armenv_builtin_chain(rid, C.byref(s)))
        ch = builtin_chain(robot)
        assert np.array_equal(np.array(s.origin_xyz), np.array(ch.origin_xyz))
        assert np.array_equal(np.array(s.origin_rpy), np.array(ch.origin_rpy))
        assert np.array_equal(np.array(s.limit_hi), np.array(ch.limit_hi))
        assert np.array_equal(np.array(s.limit_lo), np.array(ch.limit_lo))


def test_urdf_reader_rejects_what_it_cannot_represent(tmp_path):
    from armenv import urdf
    ch = urdf.builtin_chain("diana")
    assert ch.joint_names[0] == "joint1" and ch.link_names[0] == "base_link" and len(ch.link_names) == 8
    bad = open(os.path.join(urdf.ASSETS, "kuka_iiwa.urdf")).read().replace('<axis xyz="0 0 1"/>', '<axis xyz="0 1 0"/>', 1)
    f = tmp_path / "bad.urdf"; f.write_text(bad)
    with pytest.raises(ValueError, match="axes"):
        urdf.load_urdf(str(f))
    short = re.sub(r'<joint name="lbr_iiwa_joint_7".*?</joint>', "", open(os.path.join(urdf.ASSETS, "kuka_iiwa.urdf")).read(), flags=re.S)
    short = re.sub(r'<link name="lbr_iiwa_link_7">.*?</link>', "", short, flags=re.S)
    f2 = tmp_path / "short.urdf"; f2.write_text(short)
    with pytest.raises(ValueError, match="7-revolute"):
        urdf.load_urdf(str(f2))


def test_box_and_opt_surface():
    from armenv import Box, opt
    b = Box(low=[-0.4, -0.4, -0.6], high=[0.4, 0.4, 0.3])
    assert b.shape == (3,) and b.dtype == np.float32 and float(b.high[0]) + 0.3 == pytest.approx(0.7)  # main.py:87
    b.seed(0)
    assert b.contains(b.sample())
    assert (opt.reach_ctr, opt.reach_dis, opt.max_steps_one_episode, opt.hidden_dim) == (0.02, 0.01, 500, 256)
    with pytest.warns(UserWarning):
        opt._parse({"not_a_field": 1})
    delattr(type(opt), "not_a_field") if hasattr(type(opt), "not_a_field") else None


def test_config_mirrors_every_reference_field():
    """G10: `from armenv.config import opt` can stand in for the reference's `from config import opt`."""
    from armenv.config import opt
    g = golden_json("config_fields.json")["fields"]
    assert len(g) >= 30
    for k, v in g.items():
        assert hasattr(opt, k), k
        if v is not None:
            assert getattr(opt, k) == v, (k, getattr(opt, k), v)


def test_python_random_goal_stream_matches_golden():
    """G4: the N=1 compat class consumes Python's `random` exactly like the reference
    (7 draws per reset, 3 per step)."""
    import random
    from armenv.envs.rl_reach_env import draw_reset_goal, draw_step_unused
    g = golden_json("py_random_targets_seed0.json")
    random.seed(g["seed"])
    for ep in g["episodes"]:
        assert draw_reset_goal() == ep["goal"]
        for _ in range(g["steps_per_episode"]):
            draw_step_unused()


@pytest.mark.parametrize("task", ["push", "pick"])
def test_python_random_placement_stream_matches_reference(task):
    """G9: cube / target placements of successive resets (five steps apart) equal what the reference's own
    rejection-sampling loop produced under random.seed(0) (envs/rl_push_env.py:195-214, envs/rl_pick_env.py:190-208)."""
    import random
    from armenv.envs.rl_push_env import draw_push_placement
    from armenv.envs.rl_pick_env import draw_pick_placement
    from armenv.envs.rl_reach_env import draw_step_unused
    g = golden_json(f"py_random_{task}_seed0.json")
    draw = draw_push_placement if task == "push" else draw_pick_placement
    random.seed(g["seed"])
    for pl in g["placements"]:
        cube, target = draw()
        assert cube == pl["cube"] and target == pl["target"]
        for _ in range(g["steps_between_resets"]):
            draw_step_unused()


def build_c_consumer(out_dir):
    """gcc-compiles tests/c_abi/consumer.c (plain C11) against include/armenv.h and links it with libarmenv.so."""
    from armenv import _lib as L
    exe = os.path.join(out_dir, "consumer")
    libdir = os.path.dirname(L.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           "-I", "/opt/rocm/include", os.path.join(ROOT, "tests", "c_abi", "consumer.c"), "-o", exe,
                           "-L", libdir, "-larmenv", "-L", "/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_plain_c_consumer_links_and_fails_loudly_without_gpu(tmp_path):
    """The boundary is usable from plain C (no torch, no C++): the consumer compiles with -Werror, links against
    libarmenv.so, and without a HIP device armenv_create refuses with ARMENV_ENODEV and a message instead of computing
    anything on the CPU."""
    import torch
    exe = build_c_consumer(str(tmp_path))
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the run is covered by tests/test_gpu_parity.py::test_plain_c_consumer")
    r = subprocess.run([exe, "64", "2"], capture_output=True, text=True)
    assert r.returncode == 3 and "no HIP device" in r.stderr and r.stdout == ""


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from armenv import envs, ArmEnvError
    with pytest.raises(ArmEnvError, match="no HIP device|HIP"):
        envs.BatchedReachEnv(4)
    with pytest.raises(ArmEnvError):
        envs.BatchedReachEnv(4, device="cpu")


def test_product_never_imports_oracle():
    """The product package must not reference oracle/ in any form."""
    pkg = os.path.join(ROOT, "drl-on-robot-arm_amd")
    for dp, dns, fns in os.walk(pkg):
        dns[:] = [d for d in dns if d != "build"]          # build/ holds objects and A/B scratch builds, never product source
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")) or fn == "Makefile":
                txt = open(os.path.join(dp, fn)).read()
                assert "oracle" not in txt.lower(), os.path.join(dp, fn)


def test_td3_learner_matches_reference_golden():
    """G8: the reference's TD3_MLP.train (algo/TD3/TD3_mlp.py:114-161), six updates from torch.manual_seed(0) on fixed
    batches (two delayed actor updates + soft target updates included): same losses and parameters on CPU."""
    import torch
    from conftest import golden_npz
    from armenv.td3 import TD3
    g = golden_npz("td3_train_seed0.npz")
    torch.manual_seed(0)
    agent = TD3(6, 3, 0.7, device="cpu")
    torch.manual_seed(123)
    for i, want in enumerate(g["losses"]):
        b = {k: torch.from_numpy(g[f"b{i}_{k}"]) for k in ("states", "actions", "next_states", "rewards", "dones")}
        loss = float(agent.train(b))
        assert abs(loss - want) < 1e-5 * max(1.0, abs(want)), (i, loss, want)
    for name, net in (("actor", agent.actor), ("critic", agent.critic), ("target_actor", agent.target_actor), ("target_critic", agent.target_critic)):
        for k, v in net.state_dict().items():
            ref = g[f"{name}__{k.replace('.', '_')}"]
            assert np.abs(v.numpy() - ref).max() < 1e-5, (name, k)
    assert agent.total_it == 6


def test_bench_launch_shapes_have_measured_traffic():
    """bench.py reports roofline.traffic only for a launch shape the PMC passes measured (profiles/traffic.json, keyed
    "<kernel>|policy=<p>|T=<steps per launch>|N=<envs>"): both its default shape (--rollout-steps) and the driver's
    (`--steps 20`: one 20-step launch) must be there, or the bench line would silently carry traffic = null -- and the
    measured HBM traffic must equal the algorithmic bytes (no wasted re-reads)."""
    import json
    import re
    src = open(os.path.join(ROOT, "bench.py")).read()
    m = re.search(r'"--rollout-steps", type=int, default=(\d+)', src)
    assert m, "bench.py: --rollout-steps default not found"
    tr = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    for T in (int(m.group(1)), 20):
        t = tr["reach_rollout<f64,kuka>|policy=external|T=%d|N=65536" % T]
        # caller I/O 42 B per step + state (q, cos/sin q, ep_return r+w, goal r, step r+w) 372 B per launch
        algo = (42 * T + 372) * 65536
        assert abs(t["hbm_bytes_per_launch"] - algo) / algo < 0.02, (T, t["hbm_bytes_per_launch"], algo)


def _sd(g, prefix):
    import torch
    return {k: torch.from_numpy(g[f"{prefix}_{k.replace('.', '_')}"]) for k in
            ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc3.weight", "fc3.bias")}


def test_ddpg_and_datd3_take_action_match_reference_golden():
    """G11: the batched take_action of armenv.policies against vectors produced by calling the reference's own
    DDPG_MLP.take_action / DATD3_MLP.take_action one state at a time (algo/DDPG/DDPG_mlp.py:76-91,
    algo/DATD3/DATD3_mlp.py:88-109).  DATD3: the arg-max over (critic1(s, a1), critic2(s, a2)) picks the same actor as the
    reference wherever the two Q values are not within rounding of each other, and both branches occur."""
    import torch
    from conftest import golden_npz
    from armenv.policies import DATD3Policy, DDPGPolicy
    g = golden_npz("ddpg_take_action_seed0.npz")
    pol = DDPGPolicy(6, 3, float(g["action_bound"]), device="cpu").load(_sd(g, "actor"))
    a = pol.take_action(torch.from_numpy(g["states"])).numpy()
    assert np.abs(a - g["actions"]).max() < 1e-6
    g = golden_npz("datd3_take_action_seed0.npz")
    pol = DATD3Policy(6, 3, float(g["action_bound"]), device="cpu").load(*[_sd(g, n) for n in ("actor1", "actor2", "critic1", "critic2")])
    a, q1, q2 = pol.take_action(torch.from_numpy(g["states"]), return_q=True)
    assert np.abs(q1.numpy() - g["q1"]).max() < 1e-5 and np.abs(q2.numpy() - g["q2"]).max() < 1e-5
    clear = np.abs(g["q1"] - g["q2"]) > 1e-4
    assert clear.sum() >= 250 and 100 < g["picked_actor"][clear].sum() < 156
    assert np.abs(a.numpy() - g["actions"])[clear].max() < 1e-6
    assert np.array_equal((q1 < q2).numpy()[clear], g["picked_actor"][clear].astype(bool))


def test_daddpg_learner_reproduces_the_reference_updates():
    """G16: eight DADDPG_MLP.update() calls of the reference's default agent (/root/reference/algo/DADDPG/DADDPG_mlp.py:117-171;
    config.py:33) reproduced by armenv.daddpg.DADDPG on the CPU: the same initial weights under torch.manual_seed(0) (creation order
    actor1, actor2, critic), critic losses to 2e-6, every parameter of the six nets to 2e-6 -- including which target is soft-updated
    on which update (actor 1's on even updates; actor 2's and the critic's on odd ones) -- and take_action's selection."""
    import torch
    from conftest import golden_npz
    from armenv.daddpg import DADDPG
    g = golden_npz("daddpg_train_seed0.npz")
    torch.manual_seed(0)
    agent = DADDPG(6, 3, 0.7, device="cpu")
    for i, want in enumerate(g["losses"]):
        b = {k: torch.from_numpy(g[f"b{i}_{k}"]) for k in ("states", "actions", "next_states", "rewards", "dones")}
        loss = float(agent.train(b))
        assert abs(loss - want) < 2e-6 * max(1.0, abs(want)), (i, loss, want)
    assert agent.total_it == 8
    for name in ("actor1", "actor2", "critic", "target_actor1", "target_actor2", "target_critic"):
        for k, v in getattr(agent, name).state_dict().items():
            d = np.abs(v.numpy() - g[f"{name}__{k.replace('.', '_')}"])
            assert (d < 2e-6).mean() > 0.999 and d.max() < 2e-3, (name, k, (d < 2e-6).mean(), d.max())     # (Adam: see the TD3 test)
    g15 = golden_npz("daddpg_take_action_seed0.npz")
    pol = DADDPG(6, 3, float(g15["action_bound"]), device="cpu")
    for name in ("actor1", "actor2", "critic"):
        getattr(pol, name).load_state_dict(_sd(g15, name))
    clear = np.abs(g15["q1"] - g15["q2"]) > 1e-4
    for i in np.flatnonzero(clear)[:40]:
        assert np.abs(pol.take_action(g15["states"][i]) - g15["actions"][i]).max() < 5e-6
    a1, a2, c = pol.policy_state_dicts()
    assert tuple(a1["fc1.weight"].shape) == (256, 6) and tuple(c["fc1.weight"].shape) == (256, 9) and tuple(c["fc3.weight"].shape) == (1, 256)


def test_capture_safe_linear_is_linear():
    """armenv.td3._CaptureSafeLinear -- the form of y = x W^T + b the learners use under hipGraph capture, whose backward holds no
    multi-block reduction (the bias gradient is a GEMM with a row of ones: round 6, profiles/r06_td3_hipgraph_learning.txt) -- has
    F.linear's value and gradients (f64: to rounding), also when the input needs no gradient; mean_sq / neg_mean are the losses' values."""
    import torch
    import torch.nn.functional as F
    from armenv.td3 import _CaptureSafeLinear, mean_sq, neg_mean
    torch.manual_seed(4)
    for needs_x in (True, False):
        x = torch.randn(37, 9, dtype=torch.float64, requires_grad=needs_x)
        w = torch.randn(5, 9, dtype=torch.float64, requires_grad=True)
        b = torch.randn(5, dtype=torch.float64, requires_grad=True)
        up = torch.randn(37, 5, dtype=torch.float64)
        ins = (x, w, b) if needs_x else (w, b)
        ya, yb = _CaptureSafeLinear.apply(x, w, b), F.linear(x, w, b)
        assert torch.allclose(ya, yb, rtol=0, atol=1e-13)
        for ga, gb in zip(torch.autograd.grad((ya * up).sum(), ins), torch.autograd.grad((yb * up).sum(), ins)):
            assert torch.allclose(ga, gb, rtol=0, atol=1e-12)
    d = torch.randn(64, 1, dtype=torch.float64)
    assert abs(float(mean_sq(d)) - float(F.mse_loss(d, torch.zeros_like(d)))) < 1e-15 and abs(float(neg_mean(d)) + float(d.mean())) < 1e-15


def test_bench_launcher_relays_exit_code_and_refuses_contradictions():
    """`python bench.py --gpus 2` without WORLD_SIZE starts its own two ranks (torch.distributed.run on a free 127.0.0.1 port) and
    relays their exit code -- here, without a GPU, both ranks fail, and the launcher must come back non-zero without a JSON line
    and without leaving a rank behind; WORLD_SIZE that contradicts --gpus is refused by name.  (The successful two-rank run is a
    `-m gpu` test: test_bench_plain_python_launches_its_own_ranks.)"""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    bench = os.path.join(ROOT, "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], capture_output=True, text=True, timeout=300,
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and "--gpus 2" in r.stderr
    import torch
    if torch.cuda.is_available():
        return
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "20", "--warmup", "5", "--envs-per-gpu", "64"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode not in (0, 124), r.stderr[-2000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{\"metric\"")]
    assert "torch.distributed" in r.stderr or "ChildFailedError" in r.stderr      # the ranks were really started
