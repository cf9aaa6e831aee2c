"""Device-resident trajectory store with the reference's HER-"future" sampling
(/root/reference/utils/rl_utils.py:91-199: Trajectory, ReplayBuffer_Trajectory_reach / _push), fed directly by the
[T, N, ...] tensors a rollout produces.  Nothing leaves HBM: episodes are indexed and batches are gathered by HIP
kernels behind the C ABI (armenv_count_episodes / armenv_write_episodes / armenv_her_sample)."""
import ctypes as C

import torch

from . import _lib as L


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class Trajectory:
    """One episode collected step by step on the host, as the reference's single-env loops do
    (/root/reference/utils/rl_utils.py:91-105, call sites main.py:109,128): ``Trajectory(init_state)``,
    ``store_step(action, state, reward, done)``; handed to ``TrajectoryStore.add_trajectory``."""

    def __init__(self, init_state):
        self.states = [init_state]
        self.actions = []
        self.rewards = []
        self.dones = []
        self.length = 0

    def store_step(self, action, state, reward, done):
        self.actions.append(action)
        self.states.append(state)
        self.rewards.append(reward)
        self.dones.append(done)
        self.length += 1


class TrajectoryStore:
    """A time-major ring of the last ``capacity_steps`` env steps of N envs (the reference's deque keeps the last
    `capacity` trajectories, rl_utils.py:109-113).  ``size()`` = number of complete episodes currently in the window;
    ``sample(batch_size, use_her, dis_threshold, her_ratio)`` returns device tensors under the reference's keys."""

    def __init__(self, device="cuda:0", seed=0, capacity_steps=None):
        self.device = torch.device(device)
        self._lib = L.load()
        self.seed = int(seed)
        self._draw = 0
        self.capacity = capacity_steps
        self.chunk = None
        self._ring = None

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def add_rollout(self, obs0, out, actions=None, starts_at_reset=True):
        """Append a rollout chunk.  obs0: observation [N, D] before the chunk's first step (only used when the window
        starts at a reset); out: the dict returned by ``env.rollout(..., want_terminal_obs=True)`` (obs, terminal_obs,
        reward, done_u8, and actions unless `actions` is given).  Without ``capacity_steps`` the store holds exactly
        this chunk (zero-copy); with it, chunks accumulate in a ring and the oldest steps fall out."""
        acts = actions if actions is not None else out["actions"]
        src = dict(obs_after=out["obs"], next_obs=out["terminal_obs"], action=acts, reward=out["reward"], done=out["done_u8"])
        Tc, N, D = src["obs_after"].shape
        if self.capacity is None:
            self._ring = dict(cap=Tc, base=0, T=Tc, N=N, D=D, obs0=obs0.contiguous(), at_reset=bool(starts_at_reset),
                              **{k: v.contiguous() for k, v in src.items()})
        else:
            r = self._ring
            if r is None:
                cap = int(self.capacity)
                if cap < Tc:
                    raise ValueError("capacity_steps smaller than one rollout chunk")
                r = dict(cap=cap, base=0, T=0, N=N, D=D, obs0=obs0.clone(), at_reset=bool(starts_at_reset))
                for k, v in src.items():
                    r[k] = torch.empty((cap,) + tuple(v.shape[1:]), dtype=v.dtype, device=self.device)
                self._ring = r
            cap = r["cap"]
            drop = max(0, r["T"] + Tc - cap)                   # oldest steps that fall out of the window
            if drop:
                r["base"] = (r["base"] + drop) % cap
                r["T"] -= drop
                r["at_reset"] = False                         # the window now starts mid-episode
            w = (r["base"] + r["T"]) % cap                     # physical row of the first new step
            first = min(Tc, cap - w)
            for k, v in src.items():
                r[k][w:w + first].copy_(v[:first])
                if first < Tc:
                    r[k][: Tc - first].copy_(v[first:])
            r["T"] += Tc
        self._index()

    def add_trajectory(self, traj):
        """``ReplayBuffer_Trajectory_*.add_trajectory`` (rl_utils.py:112-113) for an episode of ONE env collected on the host: the
        episode becomes a [length, 1, ...] chunk of the device ring (needs ``capacity_steps``).  The episode must end with
        done = True and start at a reset, as the reference's loops produce them (main.py:108-129)."""
        import numpy as np
        if self.capacity is None:
            raise RuntimeError("TrajectoryStore.add_trajectory: construct the store with capacity_steps")
        if traj.length < 1 or not traj.dones[-1]:
            raise ValueError("TrajectoryStore.add_trajectory: a trajectory is one complete episode (last done must be True)")
        dev = self.device
        st = torch.as_tensor(np.asarray(traj.states, dtype=np.float32), device=dev)                    # [L + 1, D]
        out = dict(obs=st[1:, None, :].contiguous(), terminal_obs=st[1:, None, :].contiguous(),
                   reward=torch.as_tensor(np.asarray(traj.rewards, dtype=np.float32), device=dev)[:, None].contiguous(),
                   done_u8=torch.as_tensor(np.asarray(traj.dones, dtype=np.uint8), device=dev)[:, None].contiguous())
        acts = torch.as_tensor(np.asarray(traj.actions, dtype=np.float32), device=dev)[:, None, :].contiguous()
        first = self._ring is None
        if not first:
            # The sampler reads an episode's first state from obs_after of the step before it (auto-reset semantics: the observation
            # after a finishing step is the next episode's first one) and only the window's very first episode from obs0.  A host
            # trajectory brings its own first state: write it where the kernel looks for it, over the previous episode's last
            # obs_after row (that episode's own last next_state lives in next_obs and is not touched).  (without this
            # every episode after the first started, for the sampler, at the previous episode's terminal observation.)
            r = self._ring
            if r["N"] != 1 or r["T"] < 1:
                raise RuntimeError("TrajectoryStore.add_trajectory: the ring holds a batched rollout, not single-env trajectories")
            r["obs_after"][(r["base"] + r["T"] - 1) % r["cap"], 0].copy_(st[0])
        self.add_rollout(st[0:1].contiguous(), out, actions=acts, starts_at_reset=first)

    def _index(self):
        r = self._ring
        dev = self.device.index or 0
        T, N = r["T"], r["N"]
        counts = torch.empty(N, dtype=torch.int32, device=self.device)
        L.check(self._lib.armenv_count_episodes(dev, T, N, r["base"], r["cap"], _p(r["done"]), int(r["at_reset"]), _p(counts),
                                                self._stream()))
        offsets = torch.cumsum(counts, 0, dtype=torch.int64)
        if r.get("episodes") is None or r["episodes"].shape[0] < T * N:
            r["episodes"] = torch.empty((r["cap"] * N, 3), dtype=torch.int32, device=self.device)   # bound: one per step
        L.check(self._lib.armenv_write_episodes(dev, T, N, r["base"], r["cap"], _p(r["done"]), int(r["at_reset"]), _p(counts),
                                                _p(offsets), _p(r["episodes"]), self._stream()))
        r["num_episodes"] = offsets[-1:].contiguous()
        self.chunk = r

    def size(self):
        """number of complete episodes in the window (host sync)"""
        return 0 if self.chunk is None else int(self.chunk["num_episodes"].item())

    def sample(self, batch_size, use_her=True, dis_threshold=0.1, her_ratio=0.8, picks=None, return_picks=False, out=None):
        """`out` (optional): dict of preallocated tensors under the same keys (e.g. the static buffers of
        ``TD3.capture``); the batch is then written in place and no memory is allocated."""
        ch = self.chunk
        if ch is None:
            raise RuntimeError("TrajectoryStore.sample: no rollout stored")
        B, D, dev = int(batch_size), ch["D"], self.device
        a = L.ArmEnvHerArgs()
        a.T, a.N, a.ring_base, a.ring_cap, a.obs_dim, a.use_her = ch["T"], ch["N"], ch["base"], ch["cap"], D, int(bool(use_her))
        a.obs0_dev, a.obs_after_dev, a.next_obs_dev = ch["obs0"].data_ptr(), ch["obs_after"].data_ptr(), ch["next_obs"].data_ptr()
        a.action_dev, a.reward_dev, a.done_dev = ch["action"].data_ptr(), ch["reward"].data_ptr(), ch["done"].data_ptr()
        a.episodes_dev, a.num_episodes_dev = ch["episodes"].data_ptr(), ch["num_episodes"].data_ptr()
        a.batch = B
        pk = None
        if picks is not None:
            pk = torch.as_tensor(picks).to(device=dev, dtype=torch.int32).contiguous()
            assert tuple(pk.shape) == (B, 4)
            a.picks_dev = pk.data_ptr()
        a.seed, a.draw = self.seed, self._draw
        self._draw += 1
        a.her_ratio, a.dis_threshold = float(her_ratio), float(dis_threshold)
        if out is None:
            out = dict(states=torch.empty((B, D), dtype=torch.float32, device=dev),
                       actions=torch.empty((B, 3), dtype=torch.float32, device=dev),
                       next_states=torch.empty((B, D), dtype=torch.float32, device=dev),
                       rewards=torch.empty(B, dtype=torch.float32, device=dev),
                       dones=torch.empty(B, dtype=torch.uint8, device=dev))
        else:
            want = dict(states=((B, D), torch.float32), actions=((B, 3), torch.float32), next_states=((B, D), torch.float32),
                        rewards=((B,), torch.float32), dones=((B,), torch.uint8))
            for k, (shape, dt) in want.items():
                t = out[k]
                if tuple(t.shape) != shape or t.dtype != dt or not t.is_contiguous() or t.device != dev:
                    raise ValueError(f"TrajectoryStore.sample: out[{k!r}] must be a contiguous {dt} tensor of shape {shape} on {dev}")
        a.states_dev, a.actions_dev, a.next_states_dev = out["states"].data_ptr(), out["actions"].data_ptr(), out["next_states"].data_ptr()
        a.rewards_dev, a.dones_dev = out["rewards"].data_ptr(), out["dones"].data_ptr()
        if return_picks:
            out["picks"] = torch.empty((B, 4), dtype=torch.int32, device=dev)
            a.picks_out_dev = out["picks"].data_ptr()
        L.check(self._lib.armenv_her_sample(dev.index or 0, C.byref(a), self._stream()))
        if pk is not None:
            torch.cuda.current_stream(dev).synchronize()
        return out
