"""Batched `take_action` of the reference's other agents that consume these envs (north_star: algo/{DDPG,TD3,DATD3}) as stock torch
modules -- the EXTERNAL-ACTIONS form (`policy.take_action(obs)` feeding `armenv_step`, e.g. through `PipelinedEnv.run_closed_loop`):

  DDPG_MLP.take_action   /root/reference/algo/DDPG/DDPG_mlp.py:76-91     a = actor(s)
  DATD3_MLP.take_action  /root/reference/algo/DATD3/DATD3_mlp.py:88-109  a = actor1(s) if critic1(s, a1) >= critic2(s, a2) else actor2(s)

for [N, D] observation tensors on the env's device, returning the [N, 3] float32 action tensor `BatchedArmEnv.step` takes -- no host
round trip.  Every one of them also has a FUSED form inside the rollout kernel (no torch, MFMA passes between two env steps):

  * DDPG's actor IS the TD3 `PolicyNet` (algo/DDPG/net_mlp.py:29-40 == algo/TD3/net_mlp.py:29-40): its state_dict goes straight into
    `BatchedArmEnv.set_policy("actor" | "actor_f16x3", actor_state_dict=...)`;
  * DATD3 / DARC (algo/DARC/DARC_mlp.py:92-113, the same selection): `BatchedArmEnv.set_policy_datd3` / `set_policy_darc`;
  * DADDPG -- the reference's default agent (config.py:33; two actors, ONE critic: algo/DADDPG/DADDPG_mlp.py:77-97):
    `BatchedArmEnv.set_policy_daddpg`.

Parameter names follow the reference modules (fc1 / fc2 / fc3), so their state_dicts load unchanged."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .td3 import Actor, linear


class QValueNet(nn.Module):
    """q = fc3(relu(fc2(relu(fc1(cat(s, a))))))   (algo/DATD3/net_mlp.py:46-58, algo/DDPG/net_mlp.py:43-55)"""

    def __init__(self, state_dim, hidden_dim, action_dim):
        super().__init__()
        self.fc1 = nn.Linear(state_dim + action_dim, hidden_dim)
        self.fc2 = nn.Linear(hidden_dim, hidden_dim)
        self.fc3 = nn.Linear(hidden_dim, 1)

    def forward(self, s, a):
        return linear(self.fc3, F.relu(linear(self.fc2, F.relu(linear(self.fc1, torch.cat([s, a], dim=1))))))


class DDPGPolicy:
    def __init__(self, state_dim, action_dim, action_bound, hidden_dim=256, device="cuda:0"):
        self.device = torch.device(device)
        self.actor = Actor(state_dim, hidden_dim, action_dim, action_bound).to(self.device)

    def load(self, actor_state_dict):
        self.actor.load_state_dict(actor_state_dict)
        return self

    @torch.no_grad()
    def take_action(self, states):
        """states [N, D] -> actions f32 [N, action_dim] (no exploration noise: the caller adds it, main.py:116-117)"""
        return self.actor(states.to(self.device, torch.float32)).contiguous()


class DATD3Policy:
    def __init__(self, state_dim, action_dim, action_bound, hidden_dim=256, device="cuda:0"):
        self.device = torch.device(device)
        mk_a = lambda: Actor(state_dim, hidden_dim, action_dim, action_bound).to(self.device)
        mk_q = lambda: QValueNet(state_dim, hidden_dim, action_dim).to(self.device)
        self.actor1, self.actor2, self.critic1, self.critic2 = mk_a(), mk_a(), mk_q(), mk_q()

    def load(self, actor1, actor2, critic1, critic2):
        for m, sd in ((self.actor1, actor1), (self.actor2, actor2), (self.critic1, critic1), (self.critic2, critic2)):
            m.load_state_dict(sd)
        return self

    @torch.no_grad()
    def take_action(self, states, return_q=False):
        """Per env: action1 if q1 >= q2 else action2 (DATD3_mlp.py:100-107)."""
        s = states.to(self.device, torch.float32)
        a1, a2 = self.actor1(s), self.actor2(s)
        q1, q2 = self.critic1(s, a1), self.critic2(s, a2)
        a = torch.where(q1 >= q2, a1, a2).contiguous()
        return (a, q1.squeeze(1), q2.squeeze(1)) if return_q else a
