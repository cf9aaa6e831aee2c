"""On-device training loop: the reference's ``run()`` (/root/reference/main.py:77-162) with its four stages kept on the
GPU -- fused-policy rollouts (armenv_rollout), trajectory store + HER batches (armenv_her_sample), the agent's updates (torch),
success accounting (armenv_counters).  One iteration = `rollout_steps` env steps of `num_envs` envs followed by
`updates` updates; the reference does 40 updates of 256 samples after every (<= 501-step) episode of its single env.
The agent: `--algo td3` (train_reach_with_TD3's, main.py:165-231) or `--algo daddpg` -- opt.algo's default, what `run()` itself
instantiates (config.py:33, main.py:93): two actors and one critic, take_action fused into the rollout kernel.

    python -m armenv.train --iterations 200
    python -m armenv.train --iterations 200 --algo daddpg
"""
import argparse
import json
import time

import torch

from . import envs
from .replay import TrajectoryStore
from .daddpg import DADDPG
from .td3 import TD3


def train_reach(num_envs=1024, iterations=200, rollout_steps=32, updates=48, batch_size=2048, her_ratio=0.8, seed=0,
                device="cuda:0", actor_kind="actor_f16x3", expl_sigma=0.7 * 0.98, log_every=10, log=print,
                window_steps=1536, minimal_episodes=5, max_steps=500, use_graphs=True, algo="td3"):
    # use_graphs: the agent's update replayed from hipGraphs (GraphedLearner.capture): the update is ~130 small kernels, launch-bound
    # when issued one by one (160 iterations: 7 s against 14 s).  Round 6 found the replayed updates no longer learning and why: a
    # hipMemsetAsync captured into a hipGraph works on the first replay only on this ROCm build, torch's multi-block reductions
    # initialise their semaphores with one, so every captured bias gradient went wrong from the second replay on.  The captured update
    # now contains no such reduction (armenv.td3._CaptureSafeLinear; profiles/r06_td3_hipgraph_learning.txt) and learns like the eager one.
    torch.manual_seed(seed)
    action_bound = 0.7                                            # main.py:87
    env = envs.BatchedReachEnv(num_envs, device=device, seed=seed, max_steps=max_steps)
    agent = (DADDPG if algo == "daddpg" else TD3)(6, 3, action_bound, device=device)      # getattr(algo, opt.algo)(...), main.py:93
    static = agent.capture(batch_size) if use_graphs else None     # TD3 update as hipGraphs: launch-bound otherwise
    store = TrajectoryStore(device=device, seed=seed, capacity_steps=window_steps)   # last `window_steps` steps of every env
    ready = False
    obs = env.reset()
    history = []
    c_prev = env.counters()
    t0 = time.perf_counter()
    bufs = {}
    for it in range(iterations):
        # take_action + exploration noise + step, fused (main.py:114-124)
        if algo == "daddpg":
            env.set_policy_daddpg(*agent.policy_state_dicts(), action_bound=action_bound, noise_sigma=expl_sigma, noise_clip=action_bound)
        else:
            env.set_policy(actor_kind, action_bound=action_bound, noise_sigma=expl_sigma, noise_clip=action_bound,
                           actor_state_dict=agent.actor_state_dict())
        obs0 = obs.clone()
        out = env.rollout(rollout_steps, None, out=bufs, want_actions=True, want_terminal_obs=True)
        obs = out["obs"][-1]
        store.add_rollout(obs0, out, starts_at_reset=(it == 0))    # traj.store_step / add_trajectory (main.py:128-129)
        # replay_buffer.size() >= minimal_episodes (main.py:135), re-checked every iteration: the ring window can lose its
        # complete episodes again, and the sampler then returns inert all-zero batches that must not be trained on
        ready = store.size() >= minimal_episodes
        if ready:
            for _ in range(updates):                              # main.py:136-138
                if use_graphs:     # HER batch written straight into the captured update's static buffers
                    agent.train_graphed(store.sample(batch_size, use_her=True, her_ratio=her_ratio, out=static))
                else:
                    agent.train(store.sample(batch_size, use_her=True, her_ratio=her_ratio))
        if (it + 1) % log_every == 0:
            c = env.counters()
            ep = c["episodes"] - c_prev["episodes"]
            rate = (c["successes"] - c_prev["successes"]) / max(1, ep)
            c_prev = c
            rec = dict(iteration=it + 1, env_steps=c["env_steps"], episodes=c["episodes"], success_rate=rate,
                       wall_s=time.perf_counter() - t0)
            history.append(rec)
            log(json.dumps(rec))
    env.close()
    return agent, history


def train_push(num_envs=1024, iterations=300, rollout_steps=32, updates=48, batch_size=2048, her_ratio=0.8, seed=0,
               device="cuda:0", actor_kind="actor_f16x3", log_every=10, log=print, window_steps=1536, minimal_episodes=5,
               max_steps=500, task="push", use_graphs=True, algo="td3"):
    """``train_push_with_TD3`` (/root/reference/main.py:449-515) on the device: state_dim 9, action_bound 0.4 (:455-457),
    unclipped exploration noise N(0, 0.4 * 0.98) (:484), push HER relabel rule (utils/rl_utils.py:171-188).  The cube
    follows the build's simplified push-out model, so learning curves are not comparable with the reference's.
    ``task="pick"`` is ``train_pick_with_TD3`` (main.py:518-585), the same loop around RLPickEnv."""
    torch.manual_seed(seed)
    action_bound = 0.4
    Env = envs.BatchedPushEnv if task == "push" else envs.BatchedPickEnv
    env = Env(num_envs, device=device, seed=seed, max_steps=max_steps)
    agent = (DADDPG if algo == "daddpg" else TD3)(9, 3, action_bound, device=device)
    static = agent.capture(batch_size) if use_graphs else None
    store = TrajectoryStore(device=device, seed=seed, capacity_steps=window_steps)
    obs = env.reset()
    history, ready, bufs = [], False, {}
    c_prev = env.counters()
    t0 = time.perf_counter()
    for it in range(iterations):
        if algo == "daddpg":
            env.set_policy_daddpg(*agent.policy_state_dicts(), action_bound=action_bound, noise_sigma=action_bound * 0.98, noise_clip=1e9)
        else:
            env.set_policy(actor_kind, action_bound=action_bound, noise_sigma=action_bound * 0.98, noise_clip=1e9,
                           actor_state_dict=agent.actor_state_dict())
        obs0 = obs.clone()
        out = env.rollout(rollout_steps, None, out=bufs, want_actions=True, want_terminal_obs=True)
        obs = out["obs"][-1]
        store.add_rollout(obs0, out, starts_at_reset=(it == 0))
        ready = store.size() >= minimal_episodes         # re-checked every iteration, see train_reach
        if ready:
            for _ in range(updates):
                if use_graphs:     # HER batch written straight into the captured update's static buffers
                    agent.train_graphed(store.sample(batch_size, use_her=True, her_ratio=her_ratio, out=static))
                else:
                    agent.train(store.sample(batch_size, use_her=True, her_ratio=her_ratio))
        if (it + 1) % log_every == 0:
            c = env.counters()
            ep = c["episodes"] - c_prev["episodes"]
            rec = dict(iteration=it + 1, env_steps=c["env_steps"], episodes=c["episodes"],
                       success_rate=(c["successes"] - c_prev["successes"]) / max(1, ep), wall_s=time.perf_counter() - t0)
            c_prev = c
            history.append(rec)
            log(json.dumps(rec))
    env.close()
    return agent, history


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="reach", choices=["reach", "push", "pick"])
    ap.add_argument("--num-envs", type=int, default=1024)
    ap.add_argument("--iterations", type=int, default=200)
    ap.add_argument("--rollout-steps", type=int, default=32)
    ap.add_argument("--updates", type=int, default=48)
    ap.add_argument("--batch-size", type=int, default=2048)
    ap.add_argument("--actor", default="actor_f16x3", choices=["actor", "actor_f16x3"])
    ap.add_argument("--sigma", type=float, default=0.7 * 0.98)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--window-steps", type=int, default=1536)
    ap.add_argument("--max-steps", type=int, default=500, help="opt.max_steps_one_episode")
    ap.add_argument("--graphs", type=int, default=1, help="1: the agent's updates replayed from hipGraphs (default); 0: issued eagerly")
    ap.add_argument("--algo", default="td3", choices=["td3", "daddpg"], help="the agent (config.py:33's default is DADDPG_MLP)")
    a = ap.parse_args()
    if a.task != "reach":
        train_push(a.num_envs, a.iterations, a.rollout_steps, a.updates, a.batch_size, seed=a.seed, actor_kind=a.actor,
                   window_steps=a.window_steps, max_steps=a.max_steps, task=a.task, use_graphs=a.graphs == 1, algo=a.algo)
        return
    train_reach(a.num_envs, a.iterations, a.rollout_steps, a.updates, a.batch_size, seed=a.seed, actor_kind=a.actor,
                expl_sigma=a.sigma, window_steps=a.window_steps, max_steps=a.max_steps, algo=a.algo,
                use_graphs=a.graphs == 1)


if __name__ == "__main__":
    main()
