"""Numpy restatement of the reference's trajectory replay + HER-"future" sampler for rollout chunks
(/root/reference/utils/rl_utils.py:91-199).  TEST INFRASTRUCTURE ONLY (pure-Python loops: small cases).

Chunk layout as written by the engine's rollout: obs0 [N,D], obs_after [T,N,D], next_obs [T,N,D], action [T,N,3],
reward [T,N], done [T,N].  A trajectory = one complete episode of one env column (rl_utils.py:91-105)."""
import numpy as np


def index_episodes(done, starts_at_reset=True):
    """(env, t_start, length) of every complete episode, env-major then time order."""
    T, N = done.shape
    out = []
    for n in range(N):
        start = 0 if starts_at_reset else -1
        for t in range(T):
            if done[t, n]:
                if start >= 0:
                    out.append((n, start, t - start + 1))
                start = t + 1
    return np.array(out, dtype=np.int32).reshape(-1, 3)


def _state(ch, n, t0, j):
    """traj.states[j] of the episode of env n starting at chunk step t0"""
    tt = t0 + j
    if j == 0:
        return ch["obs0"][n] if tt == 0 else ch["obs_after"][tt - 1, n]
    return ch["next_obs"][tt - 1, n]


def sample_with_picks(ch, episodes, picks, dis_threshold=0.1):
    """ReplayBuffer_Trajectory_{reach,push}.sample with the draws given: picks [B,4] = (episode, step_state, use_her,
    step_goal).  Reach (D=6): rl_utils.py:125-141; push (D=9): :171-188."""
    D = ch["obs0"].shape[1]
    B = len(picks)
    states = np.zeros((B, D), np.float32); nexts = np.zeros((B, D), np.float32)
    actions = np.zeros((B, 3), np.float32); rewards = np.zeros(B, np.float32); dones = np.zeros(B, np.uint8)
    for b, (ep, st, her, sg) in enumerate(picks):
        n, t0, L = episodes[ep]
        assert 0 <= st < L
        s = _state(ch, n, t0, st).copy(); s2 = _state(ch, n, t0, st + 1).copy()
        t = t0 + st
        r = ch["reward"][t, n]; d = ch["done"][t, n]
        if her:
            assert st + 1 <= sg <= L
            goal = _state(ch, n, t0, sg)[:3]
            if D == 6:      # float32 observations: float32 arithmetic like numpy's
                dis = np.sqrt(np.sum(np.square(s2[:3] - goal)))
            else:           # push observations are float64 in the reference
                dis = np.sqrt(np.sum(np.square(s2[:3].astype(np.float64) - goal.astype(np.float64))))
            far = float(dis) > dis_threshold
            r = -0.1 if far else 1.0
            d = 0 if far else 1
            if D == 9:
                s2[6:9] = s[6:9]            # next_state = hstack(next_state[:3], goal, state[6:10])   :188
            s[3:6] = goal; s2[3:6] = goal
        states[b], nexts[b], actions[b], rewards[b], dones[b] = s, s2, ch["action"][t, n], r, d
    return dict(states=states, actions=actions, next_states=nexts, rewards=rewards, dones=dones)
