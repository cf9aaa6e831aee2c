#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the reference checkout.

Runs ONLY in the build container (needs /root/reference); the fixtures it writes are committed and
travel to the GPU box, this script's inputs do not.  Fixtures are data: numbers the reference holds
(known answers, captured getJointInfo tuples) and input/output vectors produced by importing the
reference's importable Python (config, algo.TD3) -- never reference source text.

  G1 fk_kat.json               q = init_joint_positions (envs/rl_reach_env.py:116-119),
                               p = initial_a (main.py:106) = f32(getLinkState(kuka,6)[4])
  G2 joint_info.json           numeric fields of envs/bmirobot_joints_info_pybullet.txt:1-16
  G3 td3_actor_seed0.npz       TD3_MLP(6,3,0.7) actor weights + 1024 states -> actions
                               (algo/TD3/TD3_mlp.py:33-97, algo/TD3/net_mlp.py:29-40)
  G4 py_random_targets_seed0.json   Python `random` stream in the reference's draw pattern
                               (7 draws per reset envs/rl_reach_env.py:180-183,210-212; 3 per step :316-318)
  G5 reward_truth.json         (distance, step_counter) -> (reward, done, success) of
                               envs/rl_reach_env.py:299-309 (strict '>' and '<')
  G6 her_{reach,push}_seed0.npz  ReplayBuffer_Trajectory_{reach,push}.sample outputs + the draws it made
                               (utils/rl_utils.py:108-199), produced by importing the reference
  G8 td3_train_seed0.npz       six TD3_MLP.train() updates of the reference: losses + final parameters
                               (algo/TD3/TD3_mlp.py:114-161), produced by importing the reference
  G7 push_reward_truth.json    (cube, target, d_last, step_counter) -> (reward, done, is_success) of
                               envs/rl_push_env.py:387-432 with its float32 / float64 mix
  G10 config_fields.json       names and literal defaults of config.DefaultConfig (config.py:29-80)
  G9 py_random_{push,pick}_seed0.json  cube / target placements of successive resets (5 steps apart), produced by
                               EXECUTING the reference's own rejection-sampling loop (envs/rl_push_env.py:195-214,
                               envs/rl_pick_env.py:190-208; extracted from the module's AST because the module itself
                               imports pybullet) under random.seed(0)
  G13 visdata_push_td3.json, td3_actor9_seed0.npz   the first 40 per-episode returns of visdata/push/origin_TD3/TD3.json (one
                               train_push_with_TD3 run, main.py:449-515) and the untrained 9-input TD3 actor of torch.manual_seed(0)
  G12 visdata_reach_td3.json   the y values of visdata/reach/TD3_0.01/Reach_TD3.json: per-episode return (351), 10-episode mean
                               return (35), 25-episode success rate (14) of one train_reach_with_TD3 run (main.py:165-231) --
                               the reference's only record of how this env BEHAVES (tests/tools/learning_curve_check.py)
"""
import ast
import json
import os
import random
import re
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def g1_fk_kat():
    src = open(os.path.join(REF, "envs/rl_reach_env.py")).read()
    m = re.search(r"self\.init_joint_positions\s*=\s*\[(.*?)\]", src, re.S)
    q = [float(x) for x in m.group(1).replace("\n", " ").split(",")]
    msrc = open(os.path.join(REF, "main.py")).read()
    m = re.search(r"initial_a\s*=\s*\[(.*?)\]", msrc)
    p = [float(x) for x in m.group(1).split(",")]
    assert len(q) == 7 and len(p) == 3
    json.dump({"robot": "kuka", "q": q, "p_f32": p,
               "source": "envs/rl_reach_env.py:116-119, main.py:106"},
              open(os.path.join(OUT, "fk_kat.json"), "w"), indent=1)


def g2_joint_info():
    rows = {"kuka": [], "diana": []}
    cur = "kuka"
    for ln in open(os.path.join(REF, "envs/bmirobot_joints_info_pybullet.txt")):
        ln = ln.strip()
        if not ln:
            cur = "diana"
            continue
        t = ast.literal_eval(ln)
        rows[cur].append({
            "index": t[0], "name": t[1].decode(), "type": t[2], "damping": t[6], "friction": t[7],
            "lower": t[8], "upper": t[9], "max_force": t[10], "max_velocity": t[11],
            "link": t[12].decode(), "axis": list(t[13]), "parent_frame_pos": list(t[14]),
            "parent_frame_orn": list(t[15]), "parent_index": t[16]})
    assert len(rows["kuka"]) == 7 and len(rows["diana"]) == 8
    rows["source"] = "envs/bmirobot_joints_info_pybullet.txt:1-16 (p.getJointInfo tuples)"
    json.dump(rows, open(os.path.join(OUT, "joint_info.json"), "w"), indent=1)


def g3_td3_actor():
    sys.path.insert(0, REF)
    import torch
    from algo.TD3.TD3_mlp import TD3_MLP  # the reference's own agent
    torch.manual_seed(0)
    agent = TD3_MLP(6, 3, 0.7, device=torch.device("cpu"))
    sd = {k: v.detach().numpy().copy() for k, v in agent.actor.state_dict().items()}
    rng = np.random.default_rng(0)
    lo = np.array([0.2, -0.3, 0.0, 0.2, -0.3, 0.0]); hi = np.array([0.7, 0.3, 0.55, 0.7, 0.3, 0.55])
    states = (lo + (hi - lo) * rng.random((1024, 6))).astype(np.float32)
    states[0] = [0.53205401, -0.00112139, 0.49629840, 0.45, 0.1, 0.3]
    with torch.no_grad():
        actions = agent.actor(torch.from_numpy(states)).numpy()
    # take_action path for row 0 (algo/TD3/TD3_mlp.py:82-97)
    a0 = agent.take_action(states[0])
    assert np.allclose(a0, actions[0], atol=1e-7)
    np.savez(os.path.join(OUT, "td3_actor_seed0.npz"), states=states, actions=actions,
             action_bound=np.float32(0.7), **{k.replace(".", "_"): v for k, v in sd.items()})


def g4_py_random():
    lo = [0.2, -0.3, 0.0]; hi = [0.7, 0.3, 0.55]
    random.seed(0)
    episodes = []
    for ep in range(4):
        goal = [random.uniform(lo[k], hi[k]) for k in range(3)]   # rl_reach_env.py:180-182
        ang = random.random()                                       # :183
        unused = [random.uniform(lo[k], hi[k]) for k in range(3)]  # :210-212
        steps = []
        for _ in range(5):
            steps.append([random.uniform(lo[k], hi[k]) for k in range(3)])  # :316-318
        episodes.append({"goal": goal, "ang_draw": ang, "unused_reset": unused, "unused_step": steps})
    json.dump({"seed": 0, "steps_per_episode": 5, "episodes": episodes,
               "source": "CPython random.seed(0) stream in the draw pattern of envs/rl_reach_env.py"},
              open(os.path.join(OUT, "py_random_targets_seed0.json"), "w"), indent=1)


def g9_py_random_placements():
    import types
    for task, fname in (("push", "envs/rl_push_env.py"), ("pick", "envs/rl_pick_env.py")):
        tree = ast.parse(open(os.path.join(REF, fname)).read())
        cls = next(n for n in tree.body if isinstance(n, ast.ClassDef))
        fns = {f.name: f for f in cls.body if isinstance(f, ast.FunctionDef)}
        # constants the loop reads: plain `self.<name> = <number>` assignments of __init__
        consts = {}
        for st in ast.walk(fns["__init__"]):
            if (isinstance(st, ast.Assign) and len(st.targets) == 1 and isinstance(st.targets[0], ast.Attribute)
                    and isinstance(st.targets[0].value, ast.Name) and st.targets[0].value.id == "self"):
                try:
                    consts[st.targets[0].attr] = ast.literal_eval(st.value)
                except ValueError:
                    pass
        loop = next(n for n in fns["reset"].body if isinstance(n, ast.For) and isinstance(n.iter, ast.Call)
                    and getattr(n.iter.func, "id", "") == "range" and ast.literal_eval(n.iter.args[0]) == 1000)
        names = ("xpos", "ypos", "zpos", "xpos_target", "ypos_target", "zpos_target")
        code = compile(ast.Module(body=[loop], type_ignores=[]), fname, "exec")
        import math
        pstub = types.SimpleNamespace(getQuaternionFromEuler=lambda e: (0.0, 0.0, math.sin(e[2] / 2), math.cos(e[2] / 2)))
        random.seed(0)
        placements = []
        for ep in range(4):
            env = {"random": random, "math": math, "p": pstub, "self": types.SimpleNamespace(**consts)}
            exec(code, env)
            placements.append({"cube": [env[k] for k in names[:3]], "target": [env[k] for k in names[3:]],
                               "dis": env["self"].dis_between_target_block})
            for _ in range(5):                                   # five steps, three unused draws each (push :435-437)
                for lo, hi in ((consts["x_low_obs"], consts["x_high_obs"]), (consts["y_low_obs"], consts["y_high_obs"]),
                               (consts["z_low_obs"], consts["z_high_obs"])):
                    random.uniform(lo, hi)
        json.dump({"seed": 0, "steps_between_resets": 5, "placements": placements,
                   "source": f"the rejection-sampling loop of {fname} executed under random.seed(0)"},
                  open(os.path.join(OUT, f"py_random_{task}_seed0.json"), "w"), indent=1)


def g10_config_fields():
    """field names and literal default values of the reference's DefaultConfig (config.py:29-80)"""
    tree = ast.parse(open(os.path.join(REF, "config.py")).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef))
    fields = {}
    for st in cls.body:
        if isinstance(st, ast.Assign) and isinstance(st.targets[0], ast.Name):
            try:
                fields[st.targets[0].id] = ast.literal_eval(st.value)
            except Exception:
                fields[st.targets[0].id] = None          # computed default (device)
    json.dump({"fields": fields, "source": "config.py DefaultConfig class attributes"},
              open(os.path.join(OUT, "config_fields.json"), "w"), indent=1)


def _reference_env_class(fname, methods, module_funcs=()):
    """The named methods of the env class in /root/reference/<fname> (plus module-level helper functions), compiled from
    the module's AST into a synthetic class -- the module itself cannot be imported (it imports pybullet / gym).  Returns
    (class, constants of __init__'s plain `self.x = <literal>` assignments, globals dict whose `p` the caller stubs)."""
    import types
    tree = ast.parse(open(os.path.join(REF, fname)).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef))
    fns = {f.name: f for f in cls.body if isinstance(f, ast.FunctionDef)}
    consts = {}
    for st in ast.walk(fns["__init__"]):
        if (isinstance(st, ast.Assign) and len(st.targets) == 1 and isinstance(st.targets[0], ast.Attribute)
                and isinstance(st.targets[0].value, ast.Name) and st.targets[0].value.id == "self"):
            try:
                consts[st.targets[0].attr] = ast.literal_eval(st.value)
            except ValueError:
                pass
    body = [f for f in tree.body if isinstance(f, ast.FunctionDef) and f.name in module_funcs]
    body.append(ast.ClassDef(name="Env", bases=[], keywords=[], body=[fns[m] for m in methods], decorator_list=[]))
    mod = ast.fix_missing_locations(ast.Module(body=body, type_ignores=[]))
    sys.path.insert(0, REF)
    from config import opt                       # the reference's own config object (importable: torch + numpy only)
    g = {"np": np, "random": random, "opt": opt, "p": types.SimpleNamespace()}
    exec(compile(mod, fname, "exec"), g)
    return g["Env"], consts, g


def g5_reward_truth():
    """EXECUTES RLReachEnv._reward (/root/reference/envs/rl_reach_env.py:267-319, extracted from the module's AST) with a
    stub `p` that hands it a flange position and a target position, for step counters and distances around the
    thresholds of :299-309.  The row's distance is the one the reference computed (float64 tuple minus float32 array, :281)."""
    import types
    Env, consts, g = _reference_env_class("envs/rl_reach_env.py", ["_reward"])
    from config import opt
    rows = []
    target = np.array([0.45, 0.10, 0.30], dtype=np.float32)
    for step in (1, 250, 499, 500, 501, 502):
        for d in (0.0, 0.005, 0.0099, 0.0099999, 0.01, 0.0100001, 0.0101, 0.05, 0.3):
            robot = (float(target[0]) + d, float(target[1]), float(target[2]))
            g["p"].getLinkState = lambda body, link, _r=robot: (None, None, None, None, _r, (0.0, 0.0, 0.0, 1.0))
            g["p"].getBasePositionAndOrientation = lambda body, _t=target: (tuple(float(x) for x in _t), (0.0, 0.0, 0.0, 1.0))
            env = Env()
            env.__dict__.update(consts)
            env.__dict__.update(kuka_id=0, object_id=1, num_joints=7, step_counter=step,
                                max_steps_one_episode=opt.max_steps_one_episode)
            random.seed(0)
            obs, reward, done, succ = env._reward()
            assert obs.dtype == np.float32 and obs.shape == (6,)
            rows.append({"step_counter": step, "robot": list(robot), "target": [float(x) for x in target],
                         "distance": float(env.distance), "reward": float(reward), "reward_is_int_zero": isinstance(reward, int),
                         "done": bool(done), "success": bool(succ), "obs": [float(x) for x in obs]})
    json.dump({"max_steps": int(opt.max_steps_one_episode), "reach_dis": float(opt.reach_dis), "rows": rows,
               "numpy": np.__version__,
               "source": "RLReachEnv._reward (envs/rl_reach_env.py:267-319) executed from its AST with a stub pybullet"},
              open(os.path.join(OUT, "reward_truth.json"), "w"), indent=1)


def g7_push_reward_truth():
    """EXECUTES RLPushEnv._reward with _get_obs / _is_success / goal_distance (/root/reference/envs/rl_push_env.py:38-41,
    258-308, 368-445, extracted from the module's AST) with a stub `p` that hands it flange, cube and target positions.
    Rows: (cube, target, d_last, step_counter) -> (reward, done, info['is_success'], new d_last).  The float32 / float64
    mix is whatever the reference's expressions produce under the numpy running this script (recorded in the fixture;
    numpy >= 2 keeps float32 scalar x Python number in float32, numpy 1.x -- the reference's era -- promoted it to float64:
    the two differ by one float32 rounding of the timeout reward, < 2e-6, and not at all in the flags)."""
    Env, consts, g = _reference_env_class("envs/rl_push_env.py", ["_reward", "_get_obs", "_is_success"], ["goal_distance"])
    from config import opt
    rows = []
    target = np.array([0.45, 0.10, 0.01])
    robot = (0.53, 0.0, 0.05)
    for step in (1, 499, 500, 501):
        for off, d_last in ((0.2300, 0.2300), (0.2300, 0.2300 + 5e-6), (0.2300, 0.2312), (0.2200, 0.2300),
                            (0.0501, 0.0600), (0.0499, 0.0600), (0.05, 0.07), (0.0500001, 0.07), (0.0499999, 0.07), (0.0, 0.03)):
            cube = target + np.array([off, 0.0, 0.0])
            poses = {1: tuple(cube), 2: tuple(target)}
            g["p"].getLinkState = lambda body, link, computeLinkVelocity=0: (None, None, None, None, robot, (0.0, 0.0, 0.0, 1.0),
                                                                              (0.0, 0.0, 0.0), (0.0, 0.0, 0.0))
            g["p"].getBasePositionAndOrientation = lambda body, _p=poses: (_p[body], (0.0, 0.0, 0.0, 1.0))
            g["p"].getEulerFromQuaternion = lambda q: (0.0, 0.0, 0.0)
            g["p"].getBaseVelocity = lambda body: ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0))
            env = Env()
            env.__dict__.update(consts)
            # the reference keeps the previous step's positions (:243-245, :396-397); a previous cube at d_last from the target
            env.__dict__.update(kuka_id=0, object_id=1, target_object_id=2, num_joints=7, step_counter=step,
                                max_steps_one_episode=opt.max_steps_one_episode,
                                last_object_pos=target + np.array([d_last, 0.0, 0.0]), last_target_pos=target.copy())
            random.seed(0)
            obs, reward, done, info = env._reward()
            assert obs.shape == (9,) and obs.dtype == np.float64
            d_prev = float(np.linalg.norm(env.__dict__["distance_last"]))
            rows.append({"cube": cube.tolist(), "target": target.tolist(), "d_last": d_prev, "step_counter": step,
                         "reward": float(reward), "done": bool(done), "success": bool(info["is_success"] > 0.5),
                         "is_success_dtype": str(np.asarray(info["is_success"]).dtype),
                         "d_new": float(env.distance_current), "obs": [float(x) for x in obs]})
    json.dump({"max_steps": int(opt.max_steps_one_episode), "rows": rows, "numpy": np.__version__,
               "source": "RLPushEnv._reward / _get_obs / _is_success (envs/rl_push_env.py:258-308,368-445) executed from "
                         "their AST with a stub pybullet"},
              open(os.path.join(OUT, "push_reward_truth.json"), "w"), indent=1)


def g6_her_samples():
    """G6: outputs of the reference's ReplayBuffer_Trajectory_{reach,push}.sample (utils/rl_utils.py:108-199) on synthetic
    trajectories, together with the draws it made (trajectory, step, HER coin, future step), recorded by wrapping
    random.sample / np.random.randint / np.random.uniform while the reference runs.  Trajectories are laid out in the
    engine's rollout-chunk layout (time-major [T][N][..], episodes back to back per env column, chunk starts at a reset;
    every column ends with an unfinished episode that must not be sampled)."""
    sys.path.insert(0, REF)
    from utils import rl_utils
    for task, D, Buf in (("reach", 6, rl_utils.ReplayBuffer_Trajectory_reach), ("push", 9, rl_utils.ReplayBuffer_Trajectory_push)):
        rng = np.random.default_rng(6 if D == 6 else 9)
        N, T = 6, 64
        obs0 = np.zeros((N, D), np.float32); obs_after = np.zeros((T, N, D), np.float32)
        next_obs = np.zeros((T, N, D), np.float32); action = np.zeros((T, N, 3), np.float32)
        reward = np.zeros((T, N), np.float32); done = np.zeros((T, N), np.uint8)
        buf = Buf(1000)
        trajs = []
        sdt = np.float32 if D == 6 else np.float64      # reach obs are float32, push obs float64 (rl_push_env.py:308)

        def first_obs():
            o = rng.uniform(0.2, 0.6, D).astype(np.float32)
            return o

        for n in range(N):
            t = 0
            cur = first_obs(); obs0[n] = cur
            while True:
                L = int(rng.integers(3, 14))
                complete = t + L <= T - 2
                if not complete:
                    L = T - t
                traj = rl_utils.Trajectory(cur.astype(sdt))
                for j in range(L):
                    a = rng.normal(0, 0.3, 3).astype(np.float32)
                    nxt = cur.copy(); nxt[:3] += rng.normal(0, 0.03, 3).astype(np.float32)
                    r = np.float32(rng.normal()); d = bool(complete and j == L - 1)
                    traj.store_step(a, nxt.astype(sdt), float(r), d)
                    action[t, n] = a; next_obs[t, n] = nxt; reward[t, n] = r; done[t, n] = d
                    cur = first_obs() if d else nxt            # auto-reset: the returned obs is the next episode's first
                    obs_after[t, n] = cur
                    t += 1
                if not complete:
                    break
                buf.add_trajectory(traj); trajs.append(traj)
        # record the reference's draws
        picks, cur_pick = [], {}
        o_sample, o_randint, o_uniform = random.sample, np.random.randint, np.random.uniform

        def w_sample(pop, k):
            out = o_sample(pop, k)
            cur_pick.clear(); cur_pick["ep"] = next(i for i, tr in enumerate(trajs) if tr is out[0]); cur_pick["n"] = 0
            return out

        def w_randint(*a, **kw):
            v = o_randint(*a, **kw)
            if cur_pick["n"] == 0:
                cur_pick["st"] = int(v)
            else:
                cur_pick["sg"] = int(v)
            cur_pick["n"] += 1
            return v

        def w_uniform(*a, **kw):
            v = o_uniform(*a, **kw)
            cur_pick["u"] = float(v)
            return v
        random.seed(60 + D); np.random.seed(60 + D)
        rl_utils.random.sample, np.random.randint, np.random.uniform = w_sample, w_randint, w_uniform
        try:
            B, ratio, thr = 256, 0.8, 0.1
            # run sample one draw at a time so that each pick can be captured
            outs = dict(states=[], actions=[], next_states=[], rewards=[], dones=[])
            for _ in range(B):
                b = buf.sample(1, use_her=True, dis_threshold=thr, her_ratio=ratio)
                her = cur_pick["u"] <= ratio
                picks.append([cur_pick["ep"], cur_pick["st"], int(her), cur_pick.get("sg", 0) if her else 0])
                cur_pick.pop("sg", None)
                for k in outs:
                    outs[k].append(b[k][0])
        finally:
            rl_utils.random.sample, np.random.randint, np.random.uniform = o_sample, o_randint, o_uniform
        episodes = []
        for n in range(N):
            start = 0
            for t in range(T):
                if done[t, n]:
                    episodes.append([n, start, t - start + 1]); start = t + 1
        assert len(episodes) == len(trajs) and all(e[2] == tr.length for e, tr in zip(episodes, trajs))
        np.savez(os.path.join(OUT, f"her_{task}_seed0.npz"), obs0=obs0, obs_after=obs_after, next_obs=next_obs, action=action,
                 reward=reward, done=done, episodes=np.array(episodes, np.int32), picks=np.array(picks, np.int32),
                 her_ratio=np.float32(ratio), dis_threshold=np.float32(thr),
                 states=np.array(outs["states"], np.float64), next_states=np.array(outs["next_states"], np.float64),
                 actions=np.array(outs["actions"], np.float32), rewards=np.array(outs["rewards"], np.float64),
                 dones=np.array(outs["dones"], np.uint8))


def g8_td3_train():
    """G8: six TD3_MLP.train() updates of the reference (algo/TD3/TD3_mlp.py:114-161) from torch.manual_seed(0)
    initialisation on fixed batches: critic losses and the final actor / critic / target parameters."""
    sys.path.insert(0, REF)
    import torch
    from algo.TD3.TD3_mlp import TD3_MLP
    torch.manual_seed(0)
    agent = TD3_MLP(6, 3, 0.7, device=torch.device("cpu"))
    rng = np.random.default_rng(8)
    B = 64
    batches = []
    for _ in range(6):
        st = rng.uniform(0.2, 0.6, (B, 6)).astype(np.float32)
        ns = st.copy(); ns[:, :3] += rng.normal(0, 0.01, (B, 3)).astype(np.float32)
        batches.append(dict(states=st, actions=rng.uniform(-0.7, 0.7, (B, 3)).astype(np.float32), next_states=ns,
                            rewards=rng.choice([-0.1, 1.0], B).astype(np.float32), dones=rng.integers(0, 2, B).astype(np.uint8)))
    torch.manual_seed(123)          # target-policy smoothing noise stream
    losses = [float(agent.train({k: v.tolist() if k in ("rewards", "dones") else v for k, v in b.items()})) for b in batches]
    out = {"losses": np.array(losses)}
    for i, b in enumerate(batches):
        for k, v in b.items():
            out[f"b{i}_{k}"] = v
    for name, net in (("actor", agent.actor), ("critic", agent.critic), ("target_actor", agent.target_actor), ("target_critic", agent.target_critic)):
        for k, v in net.state_dict().items():
            out[f"{name}__{k.replace('.', '_')}"] = v.detach().numpy().copy()
    np.savez(os.path.join(OUT, "td3_train_seed0.npz"), **out)


def g11_ddpg_datd3_take_action():
    """G11: the consumer contract of the other two agents the reference runs on these envs (north_star: algo/{DDPG,TD3,DATD3}
    consume the env unchanged).  Produced by importing the reference's agents and calling their own take_action one state
    at a time (algo/DDPG/DDPG_mlp.py:76-91, algo/DATD3/DATD3_mlp.py:88-109):
      ddpg_take_action_seed0.npz   actor weights, 256 states, actions
      datd3_take_action_seed0.npz  both actors' and both critics' weights, 256 states, actions, q1, q2, which actor won"""
    sys.path.insert(0, REF)
    import torch
    from algo.DDPG.DDPG_mlp import DDPG_MLP
    from algo.DATD3.DATD3_mlp import DATD3_MLP
    rng = np.random.default_rng(11)
    lo = np.array([0.2, -0.3, 0.0, 0.2, -0.3, 0.0]); hi = np.array([0.7, 0.3, 0.55, 0.7, 0.3, 0.55])
    states = (lo + (hi - lo) * rng.random((256, 6))).astype(np.float32)
    cpu = torch.device("cpu")
    torch.manual_seed(0)
    ddpg = DDPG_MLP(6, 3, 0.7, device=cpu)
    acts = np.stack([ddpg.take_action(s) for s in states])
    out = {"states": states, "actions": acts.astype(np.float32), "action_bound": np.float32(0.7)}
    out.update({"actor_" + k.replace(".", "_"): v.detach().numpy().copy() for k, v in ddpg.actor.state_dict().items()})
    np.savez_compressed(os.path.join(OUT, "ddpg_take_action_seed0.npz"), **out)
    torch.manual_seed(0)
    agent = DATD3_MLP(6, 3, 0.7, device=cpu)
    # freshly initialised critics differ by a near-constant offset over this small observation box (critic1 always wins):
    # widen both output layers and move critic2's output bias by the median gap, so that `action1 if q1 >= q2 else action2`
    # takes both branches across the states, some of them by a narrow margin
    with torch.no_grad():
        for c in (agent.critic1, agent.critic2):
            torch.nn.init.normal_(c.fc3.weight, std=0.5)
            c.fc3.bias.zero_()
        sb = torch.from_numpy(states)
        gap = agent.critic1(sb, agent.actor1(sb)) - agent.critic2(sb, agent.actor2(sb))
        agent.critic2.fc3.bias += gap.median()
    acts, q1s, q2s, pick = [], [], [], []
    for s in states:
        st = torch.tensor([s], dtype=torch.float)
        with torch.no_grad():
            a1, a2 = agent.actor1(st), agent.actor2(st)
            q1, q2 = float(agent.critic1(st, a1)), float(agent.critic2(st, a2))
        a = agent.take_action(s)
        acts.append(a); q1s.append(q1); q2s.append(q2)
        pick.append(0 if np.array_equal(a, a1.numpy().flatten()) else 1)
    out = {"states": states, "actions": np.stack(acts).astype(np.float32), "q1": np.float32(q1s), "q2": np.float32(q2s),
           "picked_actor": np.int32(pick), "action_bound": np.float32(0.7)}
    for name in ("actor1", "actor2", "critic1", "critic2"):
        out.update({name + "_" + k.replace(".", "_"): v.detach().numpy().copy() for k, v in getattr(agent, name).state_dict().items()})
    assert 20 < sum(pick) < 236, sum(pick)        # both branches taken
    np.savez_compressed(os.path.join(OUT, "datd3_take_action_seed0.npz"), **out)


def g12_visdata_reach_td3():
    """The only dynamics data the reference holds for this env: the visdom export of one `train_reach_with_TD3` run at reach_dis = 0.01
    (visdata/reach/TD3_0.01/Reach_TD3.json; main.py:165-231 is the function whose plotting pattern -- "return" every episode,
    "avg_return" every 10, "success_rate" every 25 -- matches the three series' lengths 351 / 35 / 14).  Numbers only."""
    d = json.load(open(os.path.join(REF, "visdata", "reach", "TD3_0.01", "Reach_TD3.json")))["jsons"]
    ys = {k: [float(v) for v in d[k]["content"]["data"][0]["y"]] for k in ("return", "avg_return", "success_rate")}
    assert (len(ys["return"]), len(ys["avg_return"]), len(ys["success_rate"])) == (351, 35, 14)
    out = {"source": "visdata/reach/TD3_0.01/Reach_TD3.json (visdom export; y values of the three plot windows)",
           "protocol": "main.py:165-231 train_reach_with_TD3: one env, a = actor(s) + N(0, 1 * opt.gamma = 0.98) unclipped, n_train = 40 updates "
                       "per episode once 5 episodes are stored, batch 256, HER ratio 0.8 (x 0.75 whenever a 25-episode success rate is a new maximum), "
                       "reach_dis = 0.01 (directory name), 501-step episodes",
           "return_per_episode": ys["return"], "avg_return_every_10_episodes": ys["avg_return"],
           "success_rate_every_25_episodes": ys["success_rate"]}
    json.dump(out, open(os.path.join(OUT, "visdata_reach_td3.json"), "w"))


def g13_visdata_push_td3():
    """The two recorded train_push_with_TD3 runs (main.py:449-515, seed 0), first 40 per-episode returns each:
    visdata/push/updata_TD3/TD3.json fits the reward code the reference ships (an untouched episode returns -504.12: eight of the
    cube's falling steps change its distance to the target by >= 1e-5 and cost -100 x the change instead of -1, rl_push_env.py:388-397,427);
    visdata/push/origin_TD3/TD3.json was recorded with the earlier `reward = -1` (:426): -500 - 50 * final distance.  Same seeds: the
    first five episodes (no network update before five are stored) are the same trajectories under two rewards -- tests/reference_run.py.
    Episode 33 of the origin run is its first success; until then every episode has 501 steps and the `random` stream is a fixed function
    of the seed.  And the untrained 9-input TD3 actor of torch.manual_seed(0)."""
    ys = {}
    for run in ("origin_TD3", "updata_TD3"):
        d = json.load(open(os.path.join(REF, "visdata", "push", run, "TD3.json")))["jsons"]
        ys[run] = [float(v) for v in d["return"]["content"]["data"][0]["y"]]
        assert len(ys[run]) == 5000
    json.dump({"source": "visdata/push/origin_TD3/TD3.json and visdata/push/updata_TD3/TD3.json (visdom exports; y values of the 'return' windows, first 40 of 5000)",
               "protocol": "main.py:449-515 train_push_with_TD3: one env, a = actor(s) + N(0, 0.4 * 0.98) unclipped, seed 0, 501-step episodes",
               "return_per_episode": ys["origin_TD3"][:40], "return_per_episode_updata": ys["updata_TD3"][:40]},
              open(os.path.join(OUT, "visdata_push_td3.json"), "w"))
    import torch
    sys.path.insert(0, REF)
    from algo.TD3.TD3_mlp import TD3_MLP
    torch.manual_seed(0)
    agent = TD3_MLP(9, 3, 0.4, 256, 1e-3, 1e-3, 0.1, 0.005, 0.98, 0.2, 0.5, 3, torch.device("cpu"))
    np.savez_compressed(os.path.join(OUT, "td3_actor9_seed0.npz"),
                        **{k.replace(".", "_"): v.detach().numpy().copy() for k, v in agent.actor.state_dict().items()})


def g14_datd3_take_action_nine_inputs():
    """G14: DATD3_MLP.take_action of the reference for the cube tasks' 9-float observations (actors 9 -> 256 -> 256 -> 3, critics
    12 -> 256 -> 256 -> 1; algo/DATD3/DATD3_mlp.py:88-109, action_bound 0.4 as train_push_with_TD3 sets it, main.py:457), produced like
    G11 by importing the reference's agent and calling its own take_action one state at a time: datd3_take_action9_seed0.npz."""
    sys.path.insert(0, REF)
    import torch
    from algo.DATD3.DATD3_mlp import DATD3_MLP
    rng = np.random.default_rng(14)
    lo = np.array([0.2, -0.3, 0.0, 0.2, -0.3, -0.006, 0.2, -0.3, 0.0]); hi = np.array([0.7, 0.3, 0.1, 0.7, 0.3, 0.01, 0.7, 0.3, 0.011])
    states = (lo + (hi - lo) * rng.random((256, 9))).astype(np.float32)
    cpu = torch.device("cpu")
    torch.manual_seed(0)
    agent = DATD3_MLP(9, 3, 0.4, device=cpu)
    with torch.no_grad():       # as in G11: both branches of `action1 if q1 >= q2 else action2` must occur, some by a narrow margin
        for c in (agent.critic1, agent.critic2):
            torch.nn.init.normal_(c.fc3.weight, std=0.5)
            c.fc3.bias.zero_()
        sb = torch.from_numpy(states)
        gap = agent.critic1(sb, agent.actor1(sb)) - agent.critic2(sb, agent.actor2(sb))
        agent.critic2.fc3.bias += gap.median()
    acts, q1s, q2s, pick = [], [], [], []
    for s in states:
        st = torch.tensor([s], dtype=torch.float)
        with torch.no_grad():
            a1, a2 = agent.actor1(st), agent.actor2(st)
            q1, q2 = float(agent.critic1(st, a1)), float(agent.critic2(st, a2))
        a = agent.take_action(s)
        acts.append(a); q1s.append(q1); q2s.append(q2)
        pick.append(0 if np.array_equal(a, a1.numpy().flatten()) else 1)
    out = {"states": states, "actions": np.stack(acts).astype(np.float32), "q1": np.float32(q1s), "q2": np.float32(q2s),
           "picked_actor": np.uint8(pick), "action_bound": np.float32(0.4)}
    for name in ("actor1", "actor2", "critic1", "critic2"):
        out.update({name + "_" + k.replace(".", "_"): v.detach().numpy().copy() for k, v in getattr(agent, name).state_dict().items()})
    np.savez_compressed(os.path.join(OUT, "datd3_take_action9_seed0.npz"), **out)


def g15_daddpg_darc_take_action():
    """G15: take_action of the reference's DEFAULT agent -- opt.algo = 'DADDPG_MLP' (config.py:33): two actors and ONE critic evaluated
    on both proposals, `action1 if q1 >= q2 else action2` (algo/DADDPG/DADDPG_mlp.py:77-97) -- and of DARC_MLP (two actors, two critics:
    algo/DARC/DARC_mlp.py:92-113, the same selection as DATD3's), produced like G11 / G14 by importing the agents and calling their own
    take_action one state at a time:
      daddpg_take_action_seed0.npz    6-float observations (reach), action_bound 0.7
      daddpg_take_action9_seed0.npz   9-float observations (push / pick), action_bound 0.4 (main.py:457)
      darc_take_action_seed0.npz      6-float observations, action_bound 0.7"""
    sys.path.insert(0, REF)
    import torch
    from algo.DADDPG.DADDPG_mlp import DADDPG_MLP
    from algo.DARC.DARC_mlp import DARC_MLP
    cpu = torch.device("cpu")
    boxes = {6: (np.array([0.2, -0.3, 0.0, 0.2, -0.3, 0.0]), np.array([0.7, 0.3, 0.55, 0.7, 0.3, 0.55])),
             9: (np.array([0.2, -0.3, 0.0, 0.2, -0.3, -0.006, 0.2, -0.3, 0.0]), np.array([0.7, 0.3, 0.1, 0.7, 0.3, 0.01, 0.7, 0.3, 0.011]))}
    for obs, bound, fname, seed in ((6, 0.7, "daddpg_take_action_seed0.npz", 15), (9, 0.4, "daddpg_take_action9_seed0.npz", 16)):
        rng = np.random.default_rng(seed)
        lo, hi = boxes[obs]
        states = (lo + (hi - lo) * rng.random((256, obs))).astype(np.float32)
        torch.manual_seed(0)
        agent = DADDPG_MLP(obs, 3, bound, device=cpu)
        with torch.no_grad():
            # Freshly initialised actors differ by a near-constant offset over this small observation box, and the ONE critic then prefers
            # the same proposal on every state.  So that `action1 if q1 >= q2 else action2` takes both branches, some of them by a narrow
            # margin: a wider critic output layer, an actor 1 whose proposal varies with the state (wider first and last layers), and
            # actor 2 = actor 1 with its hidden layer perturbed (the situation late in training: two actors that nearly agree).
            torch.nn.init.normal_(agent.critic.fc3.weight, std=0.5)
            torch.nn.init.normal_(agent.actor1.fc1.weight, std=2.0)
            torch.nn.init.normal_(agent.actor1.fc3.weight, std=0.2)
            agent.actor2.load_state_dict(agent.actor1.state_dict())
            agent.actor2.fc2.weight += 0.05 * torch.randn_like(agent.actor2.fc2.weight)
        acts, q1s, q2s, pick = [], [], [], []
        for s_ in states:
            st = torch.tensor([s_], dtype=torch.float)
            with torch.no_grad():
                a1, a2 = agent.actor1(st), agent.actor2(st)
                q1, q2 = float(agent.critic(st, a1)), float(agent.critic(st, a2))
            a = agent.take_action(s_)
            acts.append(a); q1s.append(q1); q2s.append(q2)
            pick.append(0 if np.array_equal(a, a1.numpy().flatten()) else 1)
        assert 10 < sum(pick) < 246, sum(pick)        # both branches taken
        out = {"states": states, "actions": np.stack(acts).astype(np.float32), "q1": np.float32(q1s), "q2": np.float32(q2s),
               "picked_actor": np.uint8(pick), "action_bound": np.float32(bound)}
        for name in ("actor1", "actor2", "critic"):
            out.update({name + "_" + k.replace(".", "_"): v.detach().numpy().copy() for k, v in getattr(agent, name).state_dict().items()})
        np.savez_compressed(os.path.join(OUT, fname), **out)
    rng = np.random.default_rng(17)
    lo, hi = boxes[6]
    states = (lo + (hi - lo) * rng.random((256, 6))).astype(np.float32)
    torch.manual_seed(0)
    agent = DARC_MLP(6, 3, 0.7, device=cpu)
    with torch.no_grad():       # as in G11
        for c in (agent.critic1, agent.critic2):
            torch.nn.init.normal_(c.fc3.weight, std=0.5)
            c.fc3.bias.zero_()
        sb = torch.from_numpy(states)
        gap = agent.critic1(sb, agent.actor1(sb)) - agent.critic2(sb, agent.actor2(sb))
        agent.critic2.fc3.bias += gap.median()
    acts, q1s, q2s, pick = [], [], [], []
    for s_ in states:
        st = torch.tensor([s_], dtype=torch.float)
        with torch.no_grad():
            a1, a2 = agent.actor1(st), agent.actor2(st)
            q1, q2 = float(agent.critic1(st, a1)), float(agent.critic2(st, a2))
        a = agent.take_action(s_)
        acts.append(a); q1s.append(q1); q2s.append(q2)
        pick.append(0 if np.array_equal(a, a1.numpy().flatten()) else 1)
    assert 20 < sum(pick) < 236, sum(pick)
    out = {"states": states, "actions": np.stack(acts).astype(np.float32), "q1": np.float32(q1s), "q2": np.float32(q2s),
           "picked_actor": np.uint8(pick), "action_bound": np.float32(0.7)}
    for name in ("actor1", "actor2", "critic1", "critic2"):
        out.update({name + "_" + k.replace(".", "_"): v.detach().numpy().copy() for k, v in getattr(agent, name).state_dict().items()})
    np.savez_compressed(os.path.join(OUT, "darc_take_action_seed0.npz"), **out)


def g16_daddpg_train():
    """G16: eight DADDPG_MLP.train() updates of the reference's default agent (algo/DADDPG/DADDPG_mlp.py:114-171) from
    torch.manual_seed(0) initialisation on fixed batches (four updates of each actor, the critic's target soft-updated on the odd
    ones): critic losses and the final parameters of all six nets."""
    sys.path.insert(0, REF)
    import torch
    from algo.DADDPG.DADDPG_mlp import DADDPG_MLP
    torch.manual_seed(0)
    agent = DADDPG_MLP(6, 3, 0.7, device=torch.device("cpu"))
    rng = np.random.default_rng(16)
    B = 64
    batches = []
    for _ in range(8):
        st = rng.uniform(0.2, 0.6, (B, 6)).astype(np.float32)
        ns = st.copy(); ns[:, :3] += rng.normal(0, 0.01, (B, 3)).astype(np.float32)
        batches.append(dict(states=st, actions=rng.uniform(-0.7, 0.7, (B, 3)).astype(np.float32), next_states=ns,
                            rewards=rng.choice([-0.1, 1.0], B).astype(np.float32), dones=rng.integers(0, 2, B).astype(np.uint8)))
    losses = [float(agent.update({k: v.tolist() if k in ("rewards", "dones") else v for k, v in b.items()}, B)) for b in batches]
    out = {"losses": np.array(losses)}
    for i, b in enumerate(batches):
        for k, v in b.items():
            out[f"b{i}_{k}"] = v
    for name in ("actor1", "actor2", "critic", "target_actor1", "target_actor2", "target_critic"):
        for k, v in getattr(agent, name).state_dict().items():
            out[f"{name}__{k.replace('.', '_')}"] = v.detach().numpy().copy()
    np.savez(os.path.join(OUT, "daddpg_train_seed0.npz"), **out)


if __name__ == "__main__":
    g16_daddpg_train()
    g15_daddpg_darc_take_action()
    g14_datd3_take_action_nine_inputs()
    g13_visdata_push_td3()
    g12_visdata_reach_td3()
    g1_fk_kat(); g2_joint_info(); g3_td3_actor(); g4_py_random(); g5_reward_truth(); g7_push_reward_truth(); g6_her_samples(); g8_td3_train(); g9_py_random_placements(); g10_config_fields(); g11_ddpg_datd3_take_action()
    print("fixtures written to", OUT)
