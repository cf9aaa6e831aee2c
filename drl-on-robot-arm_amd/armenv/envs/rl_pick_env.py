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
        self._eng.set_state(aux=st)
        self._d_last = float(np.linalg.norm(np.asarray(cube) - np.asarray(target), axis=-1))   # last_object_pos / last_target_pos (:243-245)
        return self._obs64(obs)

    def step(self, action):
        a = torch.as_tensor(np.asarray(action, dtype=np.float64).reshape(1, 3), dtype=torch.float32).to(self._eng.device)
        obs, reward, done, success = self._eng.step(a)
        self.step_counter += 1
        for k in range(3):
            random.uniform(_LO[k], _HI[k])                                      # :437-439 unused draws
        o = obs[0].cpu().numpy()
        r = float(reward[0].item())
        self.terminated = bool(done[0].item())
        info = {'is_success': np.float32(bool(success[0].item()))}              # :430-432
        obs64 = self._obs64(o)
        # The shaped reward -100 * (d_now - d_last) (:388-397, :427) recomputed in f64 from the env's f64 cube / target state with
        # the reference's own numpy expressions (armenv_step's reward buffer is f32); the two other branches are exact in f32:
        # +100 on success (:422-424), and the time-limit reward, which the reference computes from float32 states (:400, :418-420).
        d_cur = float(np.linalg.norm(obs64[3:6] - obs64[6:9], axis=-1))
        test = d_cur - self._d_last
        if abs(test) < 1e-5:
            test = 0.01
        self._d_last = d_cur
        if r == 100.0:
            reward64 = 100
        elif self.step_counter > self.max_steps_one_episode:
            reward64 = r
        else:
            reward64 = -test * 100
        return obs64, reward64, self.terminated, info

    def close(self):
        self._eng.close()
