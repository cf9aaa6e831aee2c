"""N=1 drop-in for ``envs.RLPushEnv`` (/root/reference/envs/rl_push_env.py:43-448): same constructor, attributes,
``reset() -> np.float64[9]``, ``step(a) -> (np.float64[9], reward, done, {'is_success': np.float32})`` and the same
consumption of Python's global ``random`` (6 draws per placement try :197-209, 3 unused draws per step :435-437).
The arm pipeline, the cube and the reward logic run in the HIP kernels.  The cube's free fall after reset() is Bullet's (pinned
by the reference's recorded runs); its contact with the tool is ArmEnvConfig.push_contact_model's stand-in for Bullet's rigid-body
step over the KUKA meshes (not parity-checkable here)."""
import math
import random

import numpy as np
import torch

from ..config import opt
from ..spaces import Box
from .batched import BatchedPushEnv

_LO, _HI = (0.2, -0.3, 0), (0.7, 0.3, 0.55)


def draw_push_placement():
    """Rejection sampling of cube / target positions, rl_push_env.py:195-214 (<= 1000 tries, last one kept)."""
    xpos = ypos = xt = yt = 0.0
    for _ in range(1000):
        xpos = random.uniform(_LO[0], _HI[0])
        ypos = random.uniform(_LO[1], _HI[1])
        random.random()                                  # cube yaw :200
        xt = random.uniform(_LO[0], _HI[0])
        yt = random.uniform(_LO[1], _HI[1])
        random.random()                                  # target yaw :207
        d = math.sqrt((xpos - xt) ** 2 + (ypos - yt) ** 2 + (0.01 - 0.01) ** 2)
        if 0.22 <= d <= 0.25:
            break
    return [xpos, ypos, 0.01], [xt, yt, 0.01]


class RLPushEnv:
    metadata = {'render.modes': ['human', 'rgb_array'], 'video.frames_per_second': 50}

    def __init__(self, is_render=False, is_good_view=False, device="cuda:0"):
        self.is_render, self.is_good_view = is_render, is_good_view
        self.max_steps_one_episode = opt.max_steps_one_episode                  # :62
        self.distance_threshold = 0.05                                          # :85
        self.action_space = Box(low=[-0.4, -0.4, -0.6], high=[0.4, 0.4, 0.3])  # :94-97
        self.observation_space = Box(low=[0.2, -0.3, 0], high=[0.7, 0.3, 0.55])  # :100-103
        self.step_counter = 0
        self._eng = BatchedPushEnv(1, device=device, auto_reset=False, precision=64, fk_path=1,
                                   fence_counters=2,
                                   max_steps=int(self.max_steps_one_episode))
        self.seed()
        self.reset()                                                            # :137

    def seed(self, seed=None):
        self.np_random = np.random.RandomState(seed)
        return [seed]

    def _obs64(self, obs_f32):
        aux = self._eng.get_state()["aux"][0].cpu().numpy()
        self._aux = aux
        return np.hstack((obs_f32[:3].astype(np.float32), aux[0:3], aux[3:6]))  # :308 (f32 eef, f64 cube, f64 target)

    def reset(self):
        self.step_counter = 0
        cube, target = draw_push_placement()
        # the cube is spawned at z = 0.01 (the placement test above sees it there); its height is the engine's from there on: one
        # step into its fall after reset()'s own stepSimulation (:241), on the table 13 steps later (ArmEnvConfig.push_contact_model);
        # the target is a fixed body and stays
        goal = torch.tensor([cube + target], dtype=torch.float64).to(torch.float32)
        obs = self._eng.reset(goal=goal)[0].cpu().numpy()
        # keep the f64 placement exactly (the engine's reset_with_goal takes f32)
        st = self._eng.get_state()["aux"].cpu().numpy()
        st[0, 0:2], st[0, 3:6] = cube[0:2], target
        cube = [cube[0], cube[1], float(st[0, 2])]
        st[0, 6] = math.sqrt(sum((a - b) ** 2 for a, b in zip(cube, target)))
        self._eng.set_state(aux=st)       # aux[6] = d_last: last_object_pos / last_target_pos (:243-245) live in the engine's state
        return self._obs64(obs)

    def step(self, action):
        """One armenv_step launch.  Reward, done and is_success are the kernel's (push_step's f64 diagnostics, armenv_step diag_dev:
        the C ABI's reward buffer is f32, the reference returns a Python float computed in f64 -- the shaped -100 * (d_now - d_last)
        of :388-397 / :427 with d_last carried in the engine's state, +100 on success :422-424, the float32-state time-limit reward
        :400 / :418-420); nothing of the reward is computed on the host."""
        a = torch.as_tensor(np.asarray(action, dtype=np.float64).reshape(1, 3), dtype=torch.float32).to(self._eng.device)
        obs, reward, done, success = self._eng.step(a, want_diag=True)
        self.step_counter += 1
        for k in range(3):
            random.uniform(_LO[k], _HI[k])                                      # :435-437 unused draws
        aux = self._eng.get_state()["aux"][0]
        packed = torch.cat([obs[0, :3].double(), aux[:6], done.double(), success.double(), self._eng.diag[0, 3:4]]).cpu().numpy()   # one host sync
        self._aux = packed[3:9]
        self.terminated = bool(packed[9])
        info = {'is_success': np.float32(bool(packed[10]))}                     # :430-432
        obs64 = np.hstack((packed[:3].astype(np.float32), packed[3:6], packed[6:9]))   # :308 (f32 eef, f64 cube, f64 target)
        return obs64, float(packed[11]), self.terminated, info

    def close(self):
        self._eng.close()
