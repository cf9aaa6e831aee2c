"""N=1 drop-in for ``envs.RLPickEnv`` (/root/reference/envs/rl_pick_env.py:44-448): same constructor, attributes,
``reset() -> np.float64[9]``, ``step(a) -> (np.float64[9], reward, done, {'is_success': np.float32})`` and the same
consumption of Python's global ``random`` (7 draws per placement try :192-203, 3 unused draws per step :437-439).
The arm pipeline (:310-351) and the reward logic (:358-435) run in the HIP kernels; the gripper and the cube follow the
build's tip-sphere model, not Bullet's finger / cube contact dynamics."""
import math
import random

import numpy as np
import torch

from ..config import opt
from ..spaces import Box
from .batched import BatchedPickEnv

_LO, _HI = (0.2, -0.3, 0), (0.7, 0.3, 0.55)


def draw_pick_placement():
    """Rejection sampling of cube / target positions, rl_pick_env.py:190-208 (<= 1000 tries, last one kept): the cube
    is spawned on the table (z = 0.01; it then settles to ArmEnvConfig.push_rest_z), the target floats anywhere in the workspace box, 0.22 <= 3-D distance <= 0.25."""
    xpos = ypos = xt = yt = zt = 0.0
    for _ in range(1000):
        xpos = random.uniform(_LO[0], _HI[0])
        ypos = random.uniform(_LO[1], _HI[1])
        random.random()                                  # cube yaw :195
        xt = random.uniform(_LO[0], _HI[0])
        yt = random.uniform(_LO[1], _HI[1])
        zt = random.uniform(_LO[2], _HI[2])
        random.random()                                  # target yaw :202
        d = math.sqrt((xpos - xt) ** 2 + (ypos - yt) ** 2 + (0.01 - zt) ** 2)
        if 0.22 <= d <= 0.25:
            break
    return [xpos, ypos, 0.01], [xt, yt, zt]


class RLPickEnv:
    metadata = {'render.modes': ['human', 'rgb_array'], 'video.frames_per_second': 50}

    def __init__(self, is_render=False, is_good_view=False, device="cuda:0"):
        self.is_render, self.is_good_view = is_render, is_good_view
        self.max_steps_one_episode = opt.max_steps_one_episode                  # :55
        self.distance_threshold = 0.05                                          # :78
        self.gripper_length = 0.257                                             # :79
        self.action_space = Box(low=[-0.4, -0.4, -0.6], high=[0.4, 0.4, 0.3])  # :88-91
        gl = self.gripper_length
        self.observation_space = Box(low=[0.2, -0.3, 0 + gl], high=[0.7, 0.3, 0.55 + gl])   # :94-97
        self.step_counter = 0
        self.end_effector_index = 6                                             # :101
        self._eng = BatchedPickEnv(1, device=device, auto_reset=False, precision=64, fk_path=1,
                                   fence_counters=2,
                                   max_steps=int(self.max_steps_one_episode))
        self.seed()
        self.reset()                                                            # :133

    def seed(self, seed=None):
        self.np_random = np.random.RandomState(seed)
        return [seed]

    def _obs64(self, obs_f32):
        aux = self._eng.get_state()["aux"][0].cpu().numpy()
        self._aux = aux
        return np.hstack((obs_f32[:3].astype(np.float32), aux[0:3], aux[3:6]))  # :308 (f32 eef, f64 cube, f64 target)

    @property
    def gripper_state(self):
        """0 open, 1 closed, 2 closed and holding the cube (build-defined gripper model)."""
        return int(self._eng.get_state()["aux"][0, 7].item())

    def reset(self):
        self.step_counter = 0
        cube, target = draw_pick_placement()
        # the cube (the push task's body, :210) is spawned at z = 0.01 (the placement test above sees it there); its height is the
        # engine's from there on: one step into its fall after reset()'s own stepSimulation (:242), on the table within seven env steps
        # (two stepSimulation calls per env step, :348 and :417; ArmEnvConfig.push_contact_model); the target stays
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
        """One armenv_step launch.  Reward, done and is_success are the kernel's (pick_step's f64 diagnostics, armenv_step diag_dev:
        the C ABI's reward buffer is f32, the reference returns a Python float computed in f64 -- the shaped -100 * (d_now - d_last)
        of :388-397 / :427 with d_last carried in the engine's state, +100 on success :422-424, the float32-state time-limit reward
        :400 / :418-420); nothing of the reward is computed on the host."""
        a = torch.as_tensor(np.asarray(action, dtype=np.float64).reshape(1, 3), dtype=torch.float32).to(self._eng.device)
        obs, reward, done, success = self._eng.step(a, want_diag=True)
        self.step_counter += 1
        for k in range(3):
            random.uniform(_LO[k], _HI[k])                                      # :437-439 unused draws
        aux = self._eng.get_state()["aux"][0]
        packed = torch.cat([obs[0, :3].double(), aux[:6], done.double(), success.double(), self._eng.diag[0, 3:4]]).cpu().numpy()   # one host sync
        self._aux = packed[3:9]
        self.terminated = bool(packed[9])
        info = {'is_success': np.float32(bool(packed[10]))}                     # :430-432
        obs64 = np.hstack((packed[:3].astype(np.float32), packed[3:6], packed[6:9]))   # :308 (f32 eef, f64 cube, f64 target)
        return obs64, float(packed[11]), self.terminated, info

    def close(self):
        self._eng.close()
