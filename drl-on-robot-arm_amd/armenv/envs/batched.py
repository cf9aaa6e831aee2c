"""Batched, device-resident counterpart of the reference's gym-style envs.

``BatchedReachEnv`` advances N independent RLReachEnv instances
(/root/reference/envs/rl_reach_env.py:38-322) per call through libarmenv.so's fused HIP kernel.  Inputs
and outputs are PyTorch-ROCm tensors on the env's device; PyTorch is used only for device memory and
streams -- every number is produced by the HIP kernels behind the C ABI (include/armenv.h).
"""
import contextlib
import ctypes as C

import numpy as np
import torch

from .. import _lib as L
from ..spaces import Box
from ..urdf import Chain, builtin_chain


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class BatchedArmEnv:
    """Common plumbing: handle life-cycle, stream handling, state exchange."""

    task = L.TASK_REACH
    obs_dim = 6
    aux_dim = 0

    def __init__(self, num_envs, device="cuda:0", seed=0, auto_reset=True, precision=64, robot="kuka",
                 chain: Chain = None, env_id_offset=0, fk_path=L.FK_AUTO, **overrides):
        self._h = None
        lib = L.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.ArmEnvError(-2, f"device {device!r}: the engine runs on HIP devices only (no CPU fallback)")
        self.num_envs = int(num_envs)
        cfg = L.default_config(self.task)
        cfg.device = self.device.index or 0
        cfg.num_envs = self.num_envs
        cfg.precision = precision
        cfg.fk_path = fk_path
        cfg.auto_reset = 1 if auto_reset else 0
        cfg.seed = seed
        cfg.env_id_offset = env_id_offset
        self.chain = chain if chain is not None else builtin_chain(robot)
        cfg.chain = L.chain_struct(self.chain)
        for k, v in overrides.items():
            if not hasattr(cfg, k):
                raise TypeError(f"unknown config field {k!r}")
            cur = getattr(cfg, k)
            if hasattr(cur, "__len__"):
                cur[:] = v
            else:
                setattr(cfg, k, v)
        self.cfg = cfg
        h = C.c_void_p()
        L.check(lib.armenv_create(C.byref(cfg), C.byref(h)))
        self._h, self._lib = h, lib
        n, dev = self.num_envs, self.device
        self._obs = torch.empty((n, self.obs_dim), dtype=torch.float32, device=dev)
        self._reward = torch.empty(n, dtype=torch.float32, device=dev)
        self._done = torch.empty(n, dtype=torch.uint8, device=dev)
        self._success = torch.empty(n, dtype=torch.uint8, device=dev)
        self._terminal = None
        self._ik_updates = None
        self._diag = None
        self._fixed_stream = None     # set by PipelinedEnv: this handle always launches on its own stream ...
        self._fixed_torch_stream = None   # ... and the same stream as a torch object (ordering against the caller's stream)
        self.action_space = Box(low=[-0.4, -0.4, -0.6], high=[0.4, 0.4, 0.3])       # rl_reach_env.py:87-90
        self.max_steps_one_episode = int(cfg.max_steps)

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        if self._fixed_stream is not None:
            return self._fixed_stream
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @contextlib.contextmanager
    def _ordered(self):
        """For the calls that allocate their tensors here (get_state, episode_stats, fk, ik, set_state): with a fixed launch
        stream the tensors belong to the CALLER's current stream, so the fixed stream first waits for everything the caller has
        enqueued (a recycled block's last reader, the producer of the inputs) and the caller's stream then waits for the kernel
        launched in between.  Without a fixed stream the launch stream IS the current stream and nothing is needed."""
        fs = self._fixed_torch_stream
        if fs is None:
            yield
            return
        cur = torch.cuda.current_stream(self.device)
        fs.wait_stream(cur)
        try:
            yield
        finally:
            cur.wait_stream(fs)     # also when the call in between raised: the caller's tensors stay ordered behind the fixed stream

    @property
    def kernel_name(self):
        return self._lib.armenv_kernel_name(self._h).decode()

    def close(self):
        if self._h is not None:
            torch.cuda.synchronize(self.device)
            self._lib.armenv_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check_action(self, action):
        if action.device != self.device or action.dtype != torch.float32 or tuple(action.shape) != (self.num_envs, 3) \
                or not action.is_contiguous():
            raise ValueError(f"action must be a contiguous float32 tensor [{self.num_envs}, 3] on {self.device}")

    _MLP_KEYS = ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc3.weight", "fc3.bias")

    def _mlp_tensors(self, sd, in_dim, out_dim, what):
        """The six tensors of one of the reference's PolicyNet / QValueNet state dicts (algo/TD3/net_mlp.py:29-71) on the device, f32,
        contiguous -- after checking every shape: the C ABI takes bare pointers and the packing kernel indexes them with the strides
        of [256][in_dim], [256][256], [out_dim][256] (a net of another width would be read out of bounds, silently)."""
        want = {"fc1.weight": (256, in_dim), "fc1.bias": (256,), "fc2.weight": (256, 256), "fc2.bias": (256,),
                "fc3.weight": (out_dim, 256), "fc3.bias": (out_dim,)}
        w = []
        for k in self._MLP_KEYS:
            if k not in sd:
                raise ValueError(f"{what}: state dict has no {k!r}")
            if tuple(sd[k].shape) != want[k]:
                raise ValueError(f"{what}: {k} has shape {tuple(sd[k].shape)}, this task's net needs {want[k]} "
                                 f"(obs_dim {self.obs_dim}, hidden 256)")
            w.append(sd[k].detach().to(device=self.device, dtype=torch.float32).contiguous())
        return w

    # ------------------------------------------------------------------ gym-style API, batched
    def reset(self, mask=None, goal=None):
        """Reset all envs (or those with mask != 0).  Returns obs [N, obs_dim] (rows of envs that were
        not reset keep their previous contents)."""
        m = None
        if mask is not None:
            m = mask.to(device=self.device, dtype=torch.uint8).contiguous()
        if goal is None:
            L.check(self._lib.armenv_reset(self._h, _ptr(m), _ptr(self._obs), self._stream()))
        else:
            g = goal.to(device=self.device, dtype=torch.float32).contiguous()
            want = (self.num_envs, 3 if self.task == L.TASK_REACH else 6)     # push / pick: [cube xyz, target xyz]
            if tuple(g.shape) != want:
                raise ValueError(f"goal must have shape {want}")
            L.check(self._lib.armenv_reset_with_goal(self._h, _ptr(m), _ptr(g), _ptr(self._obs), self._stream()))
        return self._obs

    def step(self, action, want_terminal_obs=False, want_ik_updates=False, want_diag=False):
        """One env step for all N envs; no host synchronisation.  Returns (obs, reward, done, success)
        -- the same preallocated tensors every call (clone them to keep a history).  With
        want_terminal_obs the pre-reset observation is available as ``self.terminal_obs``; with want_ik_updates the number
        of DLS updates of every env's IK call as ``self.ik_updates`` (u8; ik_max_iters = the call did not converge); with
        want_diag ``self.diag`` f64 [N, 4] = the step's end-effector position and reward before any f32 rounding (the last
        two need a handle created with fence_counters >= 1 / = 2)."""
        if action is not None:          # None: the fused policy installed with set_policy() acts
            self._check_action(action)
        if want_terminal_obs and self._terminal is None:
            self._terminal = torch.empty_like(self._obs)
        if want_ik_updates and self._ik_updates is None:
            self._ik_updates = torch.empty(self.num_envs, dtype=torch.uint8, device=self.device)
        if want_diag and self._diag is None:
            self._diag = torch.empty((self.num_envs, 4), dtype=torch.float64, device=self.device)
        term = self._terminal if want_terminal_obs else None
        upd = self._ik_updates if want_ik_updates else None
        dg = self._diag if want_diag else None
        L.check(self._lib.armenv_step(self._h, _ptr(action), _ptr(self._obs), _ptr(self._reward), _ptr(self._done),
                                      _ptr(self._success), _ptr(term), _ptr(upd), _ptr(dg), self._stream()))
        return self._obs, self._reward, self._done.view(torch.bool), self._success.view(torch.bool)

    def bind_step(self, action, stream=None):
        """Everything `step` does except the launch (argument checks, pointer and stream resolution): returns `launch()`, one
        ctypes call into armenv_step on the stream current at bind time (or `stream`); outputs go to the tensors `step` returns.
        action None: the fused policy installed with set_policy() acts (as in `step`)."""
        if action is not None:
            self._check_action(action)
        fn, args = self._lib.armenv_step, (self._h, _ptr(action), _ptr(self._obs), _ptr(self._reward), _ptr(self._done),
                                           _ptr(self._success), None, None, None, stream if stream is not None else self._stream())

        def launch(_fn=fn, _args=args, _check=L.check, _keep=action):
            rc = _fn(*_args)
            if rc:
                _check(rc)
        return launch

    @property
    def ik_updates(self):
        return self._ik_updates

    @property
    def terminal_obs(self):
        return self._terminal

    @property
    def diag(self):
        return self._diag

    def set_policy(self, kind="random", action_bound=0.7, noise_sigma=0.7 * 0.98, noise_clip=0.7, actor_state_dict=None):
        """Install the fused exploration policy of main.py:116-117 for `rollout(actions=None)`:
        a = clip(actor(obs) + N(0, noise_sigma), +-noise_clip).  kind: "external" | "random" | "actor" (exact f32) |
        "actor_f16x3" (layer 2 on the f16 MFMA with hi/lo operand splitting, ~1e-6 from f32)."""
        code = {"external": L.POLICY_EXTERNAL, "random": L.POLICY_RANDOM, "actor": L.POLICY_ACTOR,
                "actor_f16x3": L.POLICY_ACTOR_F16X3}[kind]
        w = [None] * 6
        hidden = 0
        if code in (L.POLICY_ACTOR, L.POLICY_ACTOR_F16X3):
            w = self._mlp_tensors(actor_state_dict, self.obs_dim, 3, "set_policy(%r) actor" % kind)
            hidden = int(w[0].shape[0])
        with self._ordered():
            L.check(self._lib.armenv_set_policy(self._h, code, *[_ptr(t) for t in w], hidden, float(action_bound),
                                                float(noise_sigma), float(noise_clip), self._stream()))
        torch.cuda.current_stream(self.device).synchronize()
        self._policy = kind

    def set_policy_datd3(self, actor1, actor2, critic1, critic2, action_bound=0.7, noise_sigma=0.7 * 0.98, noise_clip=0.7):
        """Install DATD3_MLP.take_action (/root/reference/algo/DATD3/DATD3_mlp.py:88-109) as the fused policy of
        `rollout(actions=None)` / `step(None)`: a1 = actor1(s), a2 = actor2(s), the action whose own critic values it higher
        (q1 = critic1(s, a1) >= q2 = critic2(s, a2) -> a1), then a = clip(a + N(0, noise_sigma), +-noise_clip).  Each argument is
        a state dict of the reference's PolicyNet / QValueNet (fc1.weight ... fc3.bias) for this task's observation width."""
        keep, mlps = [], []
        for name, sd in (("actor1", actor1), ("actor2", actor2), ("critic1", critic1), ("critic2", critic2)):
            critic = name.startswith("critic")
            w = self._mlp_tensors(sd, self.obs_dim + (3 if critic else 0), 1 if critic else 3, "set_policy_datd3 " + name)
            keep.append(w)
            mlps.append(L.ArmEnvMlp(*[t.data_ptr() for t in w]))
        hidden = int(keep[0][0].shape[0])
        with self._ordered():
            L.check(self._lib.armenv_set_policy_datd3(self._h, *[C.byref(m) for m in mlps], hidden, float(action_bound),
                                                      float(noise_sigma), float(noise_clip), self._stream()))
        torch.cuda.current_stream(self.device).synchronize()
        self._policy = "datd3"

    def set_policy_darc(self, actor1, actor2, critic1, critic2, **kw):
        """DARC_MLP.take_action (/root/reference/algo/DARC/DARC_mlp.py:92-113): the same two-actor / two-critic selection as DATD3's
        (`action1 if q1 >= q2 else action2`; the exploration noise the method carries is commented out there, :111) -- one fused
        policy serves both agents."""
        self.set_policy_datd3(actor1, actor2, critic1, critic2, **kw)
        self._policy = "darc"

    def set_policy_daddpg(self, actor1, actor2, critic, action_bound=0.7, noise_sigma=0.7 * 0.98, noise_clip=0.7):
        """Install DADDPG_MLP.take_action (/root/reference/algo/DADDPG/DADDPG_mlp.py:77-97 -- opt.algo's default, config.py:33) as
        the fused policy of `rollout(actions=None)` / `step(None)`: a1 = actor1(s), a2 = actor2(s), the proposal the ONE critic values
        higher (q1 = critic(s, a1) >= q2 = critic(s, a2) -> a1), then a = clip(a + N(0, noise_sigma), +-noise_clip).  Three nets are
        packed; the critic's second pass re-uses what its first left in LDS.  `datd3_forward` evaluates it without noise."""
        keep, mlps = [], []
        for name, sd in (("actor1", actor1), ("actor2", actor2), ("critic", critic)):
            is_c = name == "critic"
            w = self._mlp_tensors(sd, self.obs_dim + (3 if is_c else 0), 1 if is_c else 3, "set_policy_daddpg " + name)
            keep.append(w)
            mlps.append(L.ArmEnvMlp(*[t.data_ptr() for t in w]))
        with self._ordered():
            L.check(self._lib.armenv_set_policy_daddpg(self._h, *[C.byref(m) for m in mlps], 256, float(action_bound),
                                                       float(noise_sigma), float(noise_clip), self._stream()))
        torch.cuda.current_stream(self.device).synchronize()
        self._policy = "daddpg"

    def episode_returns_f32(self, out=None, stream=None):
        """Return of each env's most recently finished episode as ONE f32 vector [N] (armenv_episode_returns_f32: the send buffer
        of the logging all-gather, written by one kernel on the launch stream; `out` to write into a caller's buffer).
        stream: a raw HIP stream (int) to enqueue on instead -- the caller then owns the ordering against the env's launches
        (armenv.dist.ReturnGatherer.launch_into(..., takes_stream=True) does)."""
        if out is None:
            out = torch.empty(self.num_envs, dtype=torch.float32, device=self.device)
        elif out.dtype != torch.float32 or tuple(out.shape) != (self.num_envs,) or out.device != self.device or not out.is_contiguous():
            raise ValueError(f"out must be a contiguous float32 tensor [{self.num_envs}] on {self.device}")
        if stream is not None:
            L.check(self._lib.armenv_episode_returns_f32(self._h, _ptr(out), C.c_void_p(stream)))
            return out
        with self._ordered():
            L.check(self._lib.armenv_episode_returns_f32(self._h, _ptr(out), self._stream()))
        return out

    def datd3_forward(self, states, want_q=False):
        """DATD3_MLP / DARC_MLP / DADDPG_MLP.take_action without noise (whichever set_policy_* installed) for states f32 [n, obs_dim]:
        actions [n, 3] (+ q1, q2 [n], picked_actor u8 [n])."""
        st = states.to(device=self.device, dtype=torch.float32).contiguous().reshape(-1, self.obs_dim)
        n = st.shape[0]
        out = torch.empty((n, 3), dtype=torch.float32, device=self.device)
        q1 = torch.empty(n, dtype=torch.float32, device=self.device) if want_q else None
        q2 = torch.empty(n, dtype=torch.float32, device=self.device) if want_q else None
        pk = torch.empty(n, dtype=torch.uint8, device=self.device) if want_q else None
        with self._ordered():
            L.check(self._lib.armenv_datd3_forward(self._h, n, _ptr(st), _ptr(out), _ptr(q1), _ptr(q2), _ptr(pk), self._stream()))
        return (out, q1, q2, pk) if want_q else out

    def actor_forward(self, states):
        """TD3_MLP.take_action without noise (algo/TD3/TD3_mlp.py:82-97) for states f32 [n, obs_dim]."""
        st = states.to(device=self.device, dtype=torch.float32).contiguous().reshape(-1, self.obs_dim)
        out = torch.empty((st.shape[0], 3), dtype=torch.float32, device=self.device)
        with self._ordered():
            L.check(self._lib.armenv_actor_forward(self._h, st.shape[0], _ptr(st), _ptr(out), self._stream()))
        return out

    def rollout(self, steps, actions=None, out=None, want_actions=False, want_terminal_obs=False, want_ik_updates=False,
                want_diag=False):
        """`steps` env steps of all envs in one kernel launch (the inner loop of main.py:108-128).
        actions: float32 [steps, N, 3] on the device, or None to use the fused policy (set_policy).
        Returns a dict of [steps, N, ...] tensors: obs, reward, done, success (+ actions, terminal_obs, ik_updates, diag)."""
        launch, out = self.bind_rollout(steps, actions, out, want_actions, want_terminal_obs, stream=self._stream(),
                                        want_ik_updates=want_ik_updates, want_diag=want_diag)
        launch()
        return out

    def bind_rollout(self, steps, actions=None, out=None, want_actions=False, want_terminal_obs=False, stream=None,
                     want_ik_updates=False, want_diag=False):
        """Everything `rollout` does except the launch: argument checks, output buffers, pointer and stream resolution.
        Returns (launch, out): `launch()` enqueues the T-step kernel with one ctypes call into armenv_rollout (on the
        stream that was current at bind time, or `stream`), `out` is the dict `rollout` returns.  For callers that issue
        the same launch repeatedly or want nothing but the C call on their critical path (bench.py's timed region)."""
        n, dev, T = self.num_envs, self.device, int(steps)
        if actions is not None:
            if actions.device != dev or actions.dtype != torch.float32 or tuple(actions.shape) != (T, n, 3) \
                    or not actions.is_contiguous():
                raise ValueError(f"actions must be a contiguous float32 tensor [{T}, {n}, 3] on {dev}")
        if out is None:
            out = {}
        def buf(name, shape, dt):
            t = out.get(name)
            if t is None or tuple(t.shape) != shape:
                t = torch.empty(shape, dtype=dt, device=dev)
                out[name] = t
            return t
        obs = buf("obs", (T, n, self.obs_dim), torch.float32)
        rew = buf("reward", (T, n), torch.float32)
        done = buf("done_u8", (T, n), torch.uint8)
        succ = buf("success_u8", (T, n), torch.uint8)
        acts = buf("actions", (T, n, 3), torch.float32) if want_actions else None
        term = buf("terminal_obs", (T, n, self.obs_dim), torch.float32) if want_terminal_obs else None
        upd = buf("ik_updates", (T, n), torch.uint8) if want_ik_updates else None
        dg = buf("diag", (T, n, 4), torch.float64) if want_diag else None
        out["done"] = done.view(torch.bool)
        out["success"] = succ.view(torch.bool)
        fn, h = self._lib.armenv_rollout, self._h
        args = (h, T, _ptr(actions), _ptr(obs), _ptr(rew), _ptr(done), _ptr(succ), _ptr(acts), _ptr(term), _ptr(upd), _ptr(dg),
                stream if stream is not None else self._stream())
        keep = (actions, obs, rew, done, succ, acts, term, upd, dg)     # the closure keeps the tensors alive

        def launch(_fn=fn, _args=args, _check=L.check, _keep=keep):
            rc = _fn(*_args)
            if rc:
                _check(rc)
        return launch, out

    # ------------------------------------------------------------------ engine calls the reference makes
    def fk(self, q):
        """p.getLinkState(body, 6)[4], [5]: q [n,7] f64 -> (pos [n,3], quat xyzw [n,4]) f64."""
        q = q.to(device=self.device, dtype=torch.float64).contiguous().reshape(-1, 7)
        n = q.shape[0]
        pos = torch.empty((n, 3), dtype=torch.float64, device=self.device)
        quat = torch.empty((n, 4), dtype=torch.float64, device=self.device)
        with self._ordered():
            L.check(self._lib.armenv_fk(self._h, n, _ptr(q), _ptr(pos), _ptr(quat), self._stream()))
        return pos, quat

    def ik(self, q, target_pos):
        """p.calculateInverseKinematics: q [n,7], target [n,3] f64 -> (q_out [n,7] f64, updates [n] i32)."""
        q = q.to(device=self.device, dtype=torch.float64).contiguous().reshape(-1, 7)
        t = target_pos.to(device=self.device, dtype=torch.float64).contiguous().reshape(-1, 3)
        n = q.shape[0]
        out = torch.empty_like(q)
        iters = torch.empty(n, dtype=torch.int32, device=self.device)
        with self._ordered():
            L.check(self._lib.armenv_ik(self._h, n, _ptr(q), _ptr(t), _ptr(out), _ptr(iters), self._stream()))
        return out, iters

    # ------------------------------------------------------------------ state exchange / stats
    def get_state(self):
        """Reach: q, goal, step, episode, ep_return, trig.  Push: aux [N,10] (cube xyz, target xyz, d_last, cube velocity xy, 0) replaces goal;
        pick: aux [N,12] (cube xyz, target xyz, d_last, gripper 0/1/2, hold offset xyz, 0).  trig [N,14] = (cos q, sin q) as
        the engine carries them: ``set_state(**get_state())`` restores a checkpoint bit for bit."""
        n, dev = self.num_envs, self.device
        push = self.task != L.TASK_REACH
        st = dict(q=torch.empty((n, 7), dtype=torch.float64, device=dev),
                  step=torch.empty(n, dtype=torch.int32, device=dev),
                  episode=torch.empty(n, dtype=torch.int32, device=dev),   # u32 bits
                  ep_return=torch.empty(n, dtype=torch.float64, device=dev),
                  trig=torch.empty((n, 14), dtype=torch.float64, device=dev))
        if push:
            st["aux"] = torch.empty((n, self.aux_dim), dtype=torch.float64, device=dev)
        else:
            st["goal"] = torch.empty((n, 3), dtype=torch.float32, device=dev)
        with self._ordered():
            L.check(self._lib.armenv_get_state(self._h, _ptr(st["q"]), _ptr(st.get("goal")), _ptr(st["step"]),
                                               _ptr(st["episode"]), _ptr(st["ep_return"]), _ptr(st.get("aux")), _ptr(st["trig"]),
                                               self._stream()))
        return st

    def set_state(self, q=None, goal=None, step=None, episode=None, ep_return=None, aux=None, trig=None, sync=True):
        """Any subset of the fields of get_state().  q without trig: resetJointState semantics, the carried (cos q, sin q)
        are re-derived from the new angles.  sync=False: enqueue only (one device-to-device kernel on the launch stream, no host
        round trip) -- for tensors that already have get_state()'s device, dtypes and layout and that the caller keeps alive
        until the copy has run (a snapshot restored many times); anything that would need a temporary is refused."""
        dev = self.device

        def prep(x, dt, shape):
            if x is None:
                return None
            x = torch.as_tensor(x).to(device=dev, dtype=dt).contiguous()
            assert tuple(x.shape) == shape, (tuple(x.shape), shape)
            return x
        n = self.num_envs
        q_, g_, s_, e_, r_, a_ = (prep(q, torch.float64, (n, 7)), prep(goal, torch.float32, (n, 3)),
                                  prep(step, torch.int32, (n,)), prep(episode, torch.int32, (n,)),
                                  prep(ep_return, torch.float64, (n,)), prep(aux, torch.float64, (n, self.aux_dim)))
        t_ = prep(trig, torch.float64, (n, 14))
        if not sync:
            for given, used in ((q, q_), (goal, g_), (step, s_), (episode, e_), (ep_return, r_), (aux, a_), (trig, t_)):
                if given is not None and (not torch.is_tensor(given) or given.data_ptr() != used.data_ptr()):
                    raise ValueError("set_state(sync=False): every field must already be a contiguous tensor of get_state()'s dtype on the env's device")
        with self._ordered():
            L.check(self._lib.armenv_set_state(self._h, _ptr(q_), _ptr(g_), _ptr(s_), _ptr(e_), _ptr(r_), _ptr(a_), _ptr(t_),
                                               self._stream()))
        if sync:
            # temporaries above must outlive the copy: the current stream (which, with a fixed launch stream, now waits for it)
            torch.cuda.current_stream(dev).synchronize()

    def episode_stats(self):
        """(return, length, success) of each env's most recently finished episode."""
        n, dev = self.num_envs, self.device
        ret = torch.empty(n, dtype=torch.float64, device=dev)
        ln = torch.empty(n, dtype=torch.int32, device=dev)
        su = torch.empty(n, dtype=torch.uint8, device=dev)
        with self._ordered():
            L.check(self._lib.armenv_episode_stats(self._h, _ptr(ret), _ptr(ln), _ptr(su), self._stream()))
        return ret, ln, su

    def summary(self):
        """Device-side logging summary, no host sync: dict of 0-dim tensors (mean / max distance to goal, mean return,
        length and success rate of the envs' last finished episodes)."""
        out = torch.empty(8, dtype=torch.float64, device=self.device)
        with self._ordered():
            L.check(self._lib.armenv_summary(self._h, _ptr(out), self._stream()))
        n = out[5]
        return dict(mean_distance=out[0] / n, max_distance=out[1], mean_last_return=out[2] / n, mean_last_len=out[3] / n,
                    last_success_rate=out[4] / n, raw=out)

    def counters(self):
        out = (C.c_uint64 * 16)()
        L.check(self._lib.armenv_counters(self._h, C.byref(out), self._stream()))
        return dict(episodes=out[0], successes=out[1], env_steps=out[2], nonfinite=out[3], ik_updates=out[4],
                    limit_steps=out[5], low_flange_steps=out[6], cap_steps=out[7], illcond_steps=out[8],
                    wave_trips=out[9], wave_rounds=out[10])


class BatchedReachEnv(BatchedArmEnv):
    """N x RLReachEnv (/root/reference/envs/rl_reach_env.py)."""
    task = L.TASK_REACH
    obs_dim = 6

    def __init__(self, num_envs, **kw):
        super().__init__(num_envs, **kw)
        lo = [0.2, -0.3, 0.0] * 2
        hi = [0.7, 0.3, 0.55] * 2
        self.observation_space = Box(low=lo, high=hi)                                 # rl_reach_env.py:93-96


def diana_cam_reach_kinematics():
    """Keyword arguments that give ``BatchedReachEnv`` the kinematic set-up of the reference's Diana S1 environment
    (/root/reference/envs/diana_cam_reach.py): DianaS1_robot.urdf with the base yawed by pi (:201-204), identity target
    orientation (:156-157), workspace / object box shifted by +0.1 in x (:102-108), dv = 0.005 (:266) and NO Cartesian clip
    of the IK target (:270-274).  Only the kinematics: that environment's camera observation (:355-361) and its
    0 / 1 / -0.1 reward (:318-333) are outside this build's scope, the reach reward and 6-float observation apply."""
    import math
    big = 1.0e9
    return dict(chain=builtin_chain("diana").with_base(rpy=(0.0, 0.0, math.pi)), target_quat=[0.0, 0.0, 0.0, 1.0], dv=0.005,
                goal_lo=[0.3, -0.3, 0.0], goal_hi=[0.8, 0.3, 0.55], box_lo=[-big] * 3, box_hi=[big] * 3)


class BatchedPushEnv(BatchedArmEnv):
    """N x RLPushEnv (/root/reference/envs/rl_push_env.py): arm pipeline exact (dv 0.08, z in [0, 0.1]); the cube falls from its
    spawn height as Bullet lets it (pinned by the reference's recorded runs) and is pushed in the plane by a velocity-level contact
    model with Bullet's step order instead of Bullet's rigid-body step over the KUKA meshes (ArmEnvConfig.push_contact_model,
    DESIGN.md section 2); reward / done / success follow rl_push_env.py:368-445.  obs f32 [N, 9] = [eef, cube, target]."""
    task = L.TASK_PUSH
    obs_dim = 9
    aux_dim = 10

    def __init__(self, num_envs, **kw):
        super().__init__(num_envs, **kw)
        self.observation_space = Box(low=[0.2, -0.3, 0.0], high=[0.7, 0.3, 0.55])       # rl_push_env.py:101-104


class BatchedPickEnv(BatchedArmEnv):
    """N x RLPickEnv (/root/reference/envs/rl_pick_env.py): arm pipeline exact (dv 0.08, start position rounded through
    float32, z in [0, 0.55 + 0.257], IK result applied to joints 0..5 only, :310-351); placement :190-208; reward / done /
    success :358-445.  The gripper (closed by p.getClosestPoints within 6 mm, :412-416) and the cube follow the build's
    own tip-sphere model instead of Bullet's finger contact dynamics: aux[:, 7] is 0 open,
    1 closed, 2 closed and holding the cube.  obs f32 [N, 9] = [eef (link-7 frame), cube, target]."""
    task = L.TASK_PICK
    obs_dim = 9
    aux_dim = 12

    def __init__(self, num_envs, **kw):
        super().__init__(num_envs, **kw)
        gl = float(self.cfg.pick_gripper_length)
        self.observation_space = Box(low=[0.2, -0.3, 0.0 + gl], high=[0.7, 0.3, 0.55 + gl])    # rl_pick_env.py:94-97
