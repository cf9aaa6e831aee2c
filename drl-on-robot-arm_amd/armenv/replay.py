"""Device-resident trajectory store with the reference's HER-"future" sampling
(/root/reference/utils/rl_utils.py:91-199: Trajectory, ReplayBuffer_Trajectory_reach / _push), fed directly by the
[T, N, ...] tensors a rollout produces.  Nothing leaves HBM: episodes are indexed and batches are gathered by HIP
kernels behind the C ABI (armenv_count_episodes / armenv_write_episodes / armenv_her_sample)."""
import ctypes as C

import torch

from . import _lib as L


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class TrajectoryStore:
    """One rollout chunk of N envs x T steps.  ``size()`` = number of complete episodes (what the reference calls
    trajectories); ``sample(batch_size, use_her, dis_threshold, her_ratio)`` returns a dict of device tensors with the
    reference's keys."""

    def __init__(self, device="cuda:0", seed=0):
        self.device = torch.device(device)
        self._lib = L.load()
        self.seed = int(seed)
        self._draw = 0
        self.chunk = None

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def add_rollout(self, obs0, out, actions=None, starts_at_reset=True):
        """obs0: observation [N, D] before the rollout's first step (what reset()/the previous step returned);
        out: the dict returned by ``env.rollout(..., want_terminal_obs=True)`` (needs obs, terminal_obs, reward, done_u8
        and actions -- pass ``actions`` when the policy was external)."""
        acts = actions if actions is not None else out["actions"]
        obs_after, next_obs = out["obs"], out["terminal_obs"]
        T, N, D = obs_after.shape
        done = out["done_u8"]
        dev = self.device.index or 0
        counts = torch.empty(N, dtype=torch.int32, device=self.device)
        L.check(self._lib.armenv_count_episodes(dev, T, N, _p(done), int(bool(starts_at_reset)), _p(counts), self._stream()))
        offsets = torch.cumsum(counts, 0, dtype=torch.int64)
        episodes = torch.empty((T * N, 3), dtype=torch.int32, device=self.device)   # upper bound: one episode per step
        L.check(self._lib.armenv_write_episodes(dev, T, N, _p(done), int(bool(starts_at_reset)), _p(counts), _p(offsets),
                                                _p(episodes), self._stream()))
        self.chunk = dict(T=T, N=N, D=D, obs0=obs0.contiguous(), obs_after=obs_after, next_obs=next_obs, action=acts.contiguous(),
                          reward=out["reward"], done=done, episodes=episodes, num_episodes=offsets[-1:].contiguous(),
                          counts=counts, offsets=offsets)

    def size(self):
        """number of complete episodes (host sync)"""
        return 0 if self.chunk is None else int(self.chunk["num_episodes"].item())

    def sample(self, batch_size, use_her=True, dis_threshold=0.1, her_ratio=0.8, picks=None, return_picks=False):
        ch = self.chunk
        if ch is None:
            raise RuntimeError("TrajectoryStore.sample: no rollout stored")
        B, D, dev = int(batch_size), ch["D"], self.device
        a = L.ArmEnvHerArgs()
        a.T, a.N, a.obs_dim, a.use_her = ch["T"], ch["N"], D, int(bool(use_her))
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
        out = dict(states=torch.empty((B, D), dtype=torch.float32, device=dev),
                   actions=torch.empty((B, 3), dtype=torch.float32, device=dev),
                   next_states=torch.empty((B, D), dtype=torch.float32, device=dev),
                   rewards=torch.empty(B, dtype=torch.float32, device=dev),
                   dones=torch.empty(B, dtype=torch.uint8, device=dev))
        a.states_dev, a.actions_dev, a.next_states_dev = out["states"].data_ptr(), out["actions"].data_ptr(), out["next_states"].data_ptr()
        a.rewards_dev, a.dones_dev = out["rewards"].data_ptr(), out["dones"].data_ptr()
        if return_picks:
            out["picks"] = torch.empty((B, 4), dtype=torch.int32, device=dev)
            a.picks_out_dev = out["picks"].data_ptr()
        L.check(self._lib.armenv_her_sample(dev.index or 0, C.byref(a), self._stream()))
        if pk is not None:
            torch.cuda.current_stream(dev).synchronize()
        return out
