"""The gym-style step path for closed-loop consumers, split over HIP streams.

One ``armenv_step`` launch over N envs ends with its slowest env (65 536 reach envs: a launch's slowest lane needs 5.8 IK trips,
the mean wave 4.0), and a policy that has to see step t's observations before it can produce step t+1's actions
(``DATD3Policy``, any unmodified ``algo/*`` agent: /root/reference/main.py:111-128, algo/DATD3/DATD3_mlp.py:88-109) cannot use
``armenv_rollout``.  Envs never interact, so the batch can be cut into P independent parts, each with its own engine handle
and its own HIP stream: part A's launch t+1 (and the policy kernels that produce its actions) run while part B's launch t is
still waiting for its slowest env.  The parts own consecutive env-index ranges with ``env_id_offset`` = the range's first
index, so the P handles reproduce the single handle's trajectory bit for bit (goals and placements are Philox draws keyed
by the GLOBAL env index; the same invariance ``bench.py --gpus N`` relies on) -- tested.

    env = PipelinedEnv(BatchedReachEnv, 65536, parts=2, device="cuda:0", seed=0)
    obs = env.reset()
    obs, rew, done, succ = env.step(actions)                     # same contract as BatchedArmEnv.step, [N, ...] tensors
    env.run_closed_loop(policy, steps)                           # policy(obs_part) -> actions_part, per part, on the part's stream

PyTorch supplies streams and events only; every env number comes from the HIP kernels behind the C ABI.
"""
import ctypes as C

import torch


class PipelinedEnv:
    def __init__(self, env_cls, num_envs, parts=2, device="cuda:0", seed=0, env_id_offset=0, **kw):
        self.device = torch.device(device)
        self.num_envs = int(num_envs)
        self.parts = int(parts)
        if self.parts < 1 or self.num_envs % self.parts:
            raise ValueError("PipelinedEnv: num_envs must be a multiple of parts")
        m = self.num_envs // self.parts
        self.bounds = [(p * m, (p + 1) * m) for p in range(self.parts)]
        self.envs = [env_cls(m, device=device, seed=seed, env_id_offset=env_id_offset + lo, **kw) for lo, _ in self.bounds]
        e0 = self.envs[0]
        self.obs_dim, self.task = e0.obs_dim, e0.task
        self.action_space, self.observation_space = e0.action_space, e0.observation_space
        self.max_steps_one_episode = e0.max_steps_one_episode
        n, dev = self.num_envs, self.device
        # the parts write straight into consecutive row ranges of the full-batch output tensors
        self._obs = torch.empty((n, self.obs_dim), dtype=torch.float32, device=dev)
        self._reward = torch.empty(n, dtype=torch.float32, device=dev)
        self._done = torch.empty(n, dtype=torch.uint8, device=dev)
        self._success = torch.empty(n, dtype=torch.uint8, device=dev)
        self.streams = [torch.cuda.Stream(dev) for _ in range(self.parts)]
        for e, (lo, hi), s in zip(self.envs, self.bounds, self.streams):
            e._obs, e._reward, e._done, e._success = self._obs[lo:hi], self._reward[lo:hi], self._done[lo:hi], self._success[lo:hi]
            e._fixed_stream = C.c_void_p(s.cuda_stream)           # BatchedArmEnv._stream(): this part always launches here
            e._fixed_torch_stream = s                             # BatchedArmEnv._ordered(): allocating calls order against the caller
        self._fork = torch.cuda.Event()
        self._join = [torch.cuda.Event() for _ in range(self.parts)]

    # ------------------------------------------------------------------ stream plumbing
    def _fork_from_current(self):
        """the part streams wait for everything enqueued on the caller's stream so far (the producer of the actions)"""
        cur = torch.cuda.current_stream(self.device)
        self._fork.record(cur)
        for s in self.streams:
            s.wait_event(self._fork)

    def join(self):
        """the caller's stream waits for every part's work so far (before it reads the [N, ...] outputs)"""
        cur = torch.cuda.current_stream(self.device)
        for s, ev in zip(self.streams, self._join):
            ev.record(s)
            cur.wait_event(ev)

    # ------------------------------------------------------------------ gym-style API
    def reset(self):
        self._fork_from_current()
        for e in self.envs:
            e.reset()
        self.join()
        return self._obs

    def step(self, action, fork=True, join=True):
        """One env step of all N envs as `parts` launches on `parts` streams.  fork / join = False: the caller vouches that the
        action tensor is complete / will call join() itself before reading the outputs (an open-loop driver that enqueues many
        steps needs neither per step)."""
        if action.device != self.device or action.dtype != torch.float32 or tuple(action.shape) != (self.num_envs, 3) \
                or not action.is_contiguous():
            raise ValueError(f"action must be a contiguous float32 tensor [{self.num_envs}, 3] on {self.device}")
        if fork:
            self._fork_from_current()
        for e, (lo, hi) in zip(self.envs, self.bounds):
            e.step(action[lo:hi])
        if join:
            self.join()
        return self._obs, self._reward, self._done.view(torch.bool), self._success.view(torch.bool)

    def bind_steps(self, actions):
        """actions f32 [K, N, 3], complete before the first call: returns a list of K closures, closure t = step t of every part
        (`parts` bare C calls on the parts' own streams, no stream or event traffic).  Outputs of the last executed step are in the
        [N, ...] tensors after join().  For open-loop drivers.  (Not for hipGraph capture: the closures launch on `parts` streams the
        capturing stream knows nothing about; capture a single handle's `bind_step` closures instead, as bench.py's step_api leg does.)"""
        fns = []
        for t in range(actions.shape[0]):
            calls = []
            for e, (lo, hi) in zip(self.envs, self.bounds):
                calls.append(e.bind_step(actions[t, lo:hi]))
            fns.append(calls)

        def make(calls):
            def run():
                for c in calls:
                    c()
            return run
        return [make(c) for c in fns]

    def run_closed_loop(self, policy, steps, obs=None, on_step=None):
        """`steps` iterations of  a = policy(obs); obs, r, d, s = step(a)  per part: part p's policy kernels and its step launch
        are enqueued on stream p, so the GPU overlaps policy(A) with step(B).  `policy` maps a [m, obs_dim] observation tensor
        to a contiguous f32 [m, 3] action tensor with torch ops on the current stream (e.g. DATD3Policy.take_action).
        Returns the [N, ...] output tensors of the last step, joined into the caller's stream."""
        self._fork_from_current()
        for _ in range(int(steps)):
            for e, s in zip(self.envs, self.streams):
                with torch.cuda.stream(s):
                    a = policy(e._obs)
                    e.step(a)
                    if on_step is not None:
                        on_step(e)
        self.join()
        return self._obs, self._reward, self._done.view(torch.bool), self._success.view(torch.bool)

    # ------------------------------------------------------------------ state / stats over all parts
    def get_state(self):
        # every part's get_state orders its own stream against the caller's on both sides (BatchedArmEnv._ordered)
        sts = [e.get_state() for e in self.envs]
        return {k: torch.cat([s[k] for s in sts], dim=0) for k in sts[0]}

    def counters(self):
        tot = {}
        for e in self.envs:
            for k, v in e.counters().items():
                tot[k] = tot.get(k, 0) + v
        return tot

    def episode_stats(self):
        parts = [e.episode_stats() for e in self.envs]
        return tuple(torch.cat([p[i] for p in parts]) for i in range(3))

    def close(self):
        for e in self.envs:
            e.close()
