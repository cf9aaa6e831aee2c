"""BASELINE.json configs[0] plumbing: the reference's single-env loop (main.py:100-128, take_action -> noise -> step ->
store_step) on the N=1 drop-in class with the build's TD3 counterpart, 1000 steps.  Prints steps/s (one host sync per
env call: a compatibility path, not a throughput path)."""
import os, sys, time, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "drl-on-robot-arm_amd"))
import numpy as np, torch
from armenv import envs, opt
from armenv.td3 import TD3

env = envs.RLReachEnv(is_render=False, is_good_view=False)
state_dim, action_dim = env.observation_space.shape[0], env.action_space.shape[0]
action_bound = float(env.action_space.high[0]) + 0.3
random.seed(opt.random_seed); np.random.seed(opt.random_seed); torch.manual_seed(opt.random_seed)
agent = TD3(state_dim, action_dim, action_bound)
steps, episodes, t0 = 0, 0, None
while steps < 1100:
    state = env.reset(); done = False; traj = [state]
    while not done and steps < 1100:
        if steps == 100:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.no_grad():
            a = agent.actor(torch.tensor(state[None], device="cuda")).cpu().numpy()[0]        # take_action
        a = (a + np.random.normal(0, action_bound * opt.gamma, size=action_dim)).clip(-action_bound, action_bound)
        state, reward, done, is_success = env.step(a)
        traj.append(state); steps += 1
    episodes += 1
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"config0 plumbing: 1000 single-env steps (actor forward + env.step, host-synchronous) in {dt:.3f}s = {1000/dt:.0f} steps/s; "
      f"env-only: ", end="")
t0 = time.perf_counter()
for _ in range(1000):
    env.step(np.zeros(3))
print(f"{1000/(time.perf_counter()-t0):.0f} steps/s")
env.close()
