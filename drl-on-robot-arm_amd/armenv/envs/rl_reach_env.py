"""N=1 drop-in for ``envs.RLReachEnv`` (/root/reference/envs/rl_reach_env.py:38-322).

Same constructor, attributes and return types as the reference class, so an unmodified
``main.py``-style loop (/root/reference/main.py:83-128) runs on it; the arithmetic runs on the GPU
through the C ABI (one-env handle, one host sync per call).  Quirks kept on purpose
(SURVEY.md Appendix D): the constructor performs a reset; goals come from Python's global ``random``
with the reference's draw counts (7 per reset, 3 per step); ``seed()`` does not affect sampling;
an episode lasts 501 steps; the success reward is ``0``.
"""
import math
import random

import numpy as np
import torch

from ..config import opt
from ..spaces import Box
from .batched import BatchedReachEnv


_LO, _HI = (0.2, -0.3, 0), (0.7, 0.3, 0.55)     # rl_reach_env.py:65-70


def draw_reset_goal():
    """The seven `random` draws of reset(): target xyz (:180-182), cube yaw (:183, unobservable), and the
    three unused draws of :210-212.  Returns the target position."""
    goal = [random.uniform(_LO[k], _HI[k]) for k in range(3)]
    random.random()
    for k in range(3):
        random.uniform(_LO[k], _HI[k])
    return goal


def draw_step_unused():
    """The three unused draws at the end of _reward() (:316-318)."""
    for k in range(3):
        random.uniform(_LO[k], _HI[k])


class RLReachEnv:
    metadata = {'render.modes': ['human', 'rgb_array'], 'video.frames_per_second': 50}

    def __init__(self, is_render=False, is_good_view=False, device="cuda:0"):
        self.is_render = is_render            # accepted and ignored: there is no GUI
        self.is_good_view = is_good_view
        self.max_steps_one_episode = opt.max_steps_one_episode          # rl_reach_env.py:57 (ctor time)
        self.x_low_obs, self.x_high_obs = 0.2, 0.7                      # :65-70
        self.y_low_obs, self.y_high_obs = -0.3, 0.3
        self.z_low_obs, self.z_high_obs = 0, 0.55
        self.action_space = Box(low=[-0.4, -0.4, -0.6], high=[0.4, 0.4, 0.3])          # :87-90
        self.observation_space = Box(low=[0.2, -0.3, 0, 0.2, -0.3, 0],                 # :93-96
                                     high=[0.7, 0.3, 0.55, 0.7, 0.3, 0.55])
        self.step_counter = 0
        self.init_joint_positions = [0.006418, 0.413184, -0.011401, -1.589317, 0.005379, 1.137684, -0.006539]
        self._device = device
        self._make_engine()
        self.seed()
        self.reset()                                                    # :125

    def _make_engine(self):
        self._dv, self._dis = float(opt.reach_ctr), float(opt.reach_dis)
        # fence_counters=2: the bookkeeping build of the kernels whose step also hands out the f64 end-effector position and reward
        self._eng = BatchedReachEnv(1, device=self._device, auto_reset=False, precision=64, fk_path=1, fence_counters=2,
                                    dv=self._dv, reach_dis=self._dis, max_steps=int(self.max_steps_one_episode))

    def _sync_opt(self):
        # the reference reads opt.reach_ctr / opt.reach_dis at call time (:231,303)
        if float(opt.reach_ctr) != self._dv or float(opt.reach_dis) != self._dis:
            st = self._eng.get_state()
            self._eng.close()
            self._make_engine()
            self._eng.set_state(**st)

    def seed(self, seed=None):
        """:127-130 -- sets np_random, which nothing reads."""
        self.np_random = np.random.RandomState(seed)
        return [seed]

    def reset(self):
        """:132-217"""
        self._sync_opt()
        self.step_counter = 0
        self.terminated = False
        goal = torch.tensor([draw_reset_goal()], dtype=torch.float64).to(torch.float32)
        obs = self._eng.reset(goal=goal)
        self.object_state = goal[0].numpy().copy()
        return obs[0].cpu().numpy()                                      # np.float32[6]  :217

    def step(self, action):
        """:219-319 -> (np.float32[6], float reward, bool done, bool is_success).  One armenv_step launch.  The reward buffer of the C
        ABI is f32, the reference returns a Python float computed in f64: position and reward are taken from the step's own f64
        diagnostics (armenv_step diag_dev) -- reward, done and is_success all come from the kernel's one f64 distance.
        `self.distance` (an attribute the reference sets at :281, read by nobody) is recomputed here from that position as the
        reference's plain sum of squares; the kernel's distance is FMA-contracted, so the two can differ in the last place and
        `self.distance < reach_dis` may disagree with is_success for a distance within one ulp of the threshold."""
        self._sync_opt()
        a = torch.as_tensor(np.asarray(action, dtype=np.float64).reshape(1, 3), dtype=torch.float32).to(self._eng.device)
        obs, reward, done, success = self._eng.step(a, want_diag=True)
        self.step_counter += 1
        draw_step_unused()
        packed = torch.cat([obs[0].double(), done.double(), success.double(), self._eng.diag[0]]).cpu().numpy()     # one host sync
        self.terminated = bool(packed[6])
        self.is_success = bool(packed[7])
        self.robot_state = tuple(float(x) for x in packed[8:11])                                 # :271 getLinkState(...)[4]
        goal = self.object_state.astype(np.float64)                                              # :276-278 float32 cube position
        self.distance = float(np.sqrt(np.sum((np.asarray(self.robot_state) - goal) ** 2)))       # :281 (f64)
        reward = float(packed[11])                                                               # :299-309, computed in f64 by the kernel
        return packed[:6].astype(np.float32), reward, self.terminated, self.is_success

    def close(self):
        self._eng.close()
