"""The reference's DEFAULT agent on PyTorch-ROCm tensors: DADDPG_MLP (/root/reference/algo/DADDPG/DADDPG_mlp.py:33-173;
`opt.algo = 'DADDPG_MLP'`, /root/reference/config.py:33, is what `python main.py run` instantiates, main.py:93) -- two actors,
ONE critic, alternating actor updates.  Like armenv.td3 it consumes device-resident HER batches (armenv.replay.TrajectoryStore.sample)
and hands its three nets to the env engine for fused rollouts (BatchedArmEnv.set_policy_daddpg: take_action inside the rollout
kernel).  Stock torch ops: the learner is integration, not a kernel."""
import torch

from .policies import QValueNet
from .td3 import Actor, GraphedLearner, mean_sq, neg_mean


class DADDPG(GraphedLearner):
    """Hyper-parameters default to config.py:55-62 (hidden 256, lr 1e-3, tau 0.005, gamma 0.98)."""

    def __init__(self, state_dim, action_dim, action_bound, hidden_dim=256, actor_lr=1e-3, critic_lr=1e-3, tau=0.005, gamma=0.98,
                 device="cuda:0"):
        self.device = torch.device(device)
        cap = self.device.type == "cuda"      # step counters on the device: the update can be captured in a hipGraph
        mk_a = lambda: Actor(state_dim, hidden_dim, action_dim, action_bound).to(self.device)
        # the three learning nets in the reference's creation order (DADDPG_mlp.py:57, 61, 65: their initial weights are a function of
        # it under torch.manual_seed; the reference's targets are deep copies, which draw nothing) -- the targets after them
        self.actor1, self.actor2 = mk_a(), mk_a()
        self.critic = QValueNet(state_dim, hidden_dim, action_dim).to(self.device)
        self.target_actor1, self.target_actor2 = mk_a(), mk_a()
        self.target_critic = QValueNet(state_dim, hidden_dim, action_dim).to(self.device)
        for t_, n_ in ((self.target_actor1, self.actor1), (self.target_actor2, self.actor2), (self.target_critic, self.critic)):
            t_.load_state_dict(n_.state_dict())
        self.actor1_opt = torch.optim.Adam(self.actor1.parameters(), lr=actor_lr, capturable=cap)
        self.actor2_opt = torch.optim.Adam(self.actor2.parameters(), lr=actor_lr, capturable=cap)
        self.critic_opt = torch.optim.Adam(self.critic.parameters(), lr=critic_lr, capturable=cap)
        self.tau, self.gamma, self.action_bound = tau, gamma, action_bound
        self.total_it = 0
        self._graphs = None

    @property
    def actor(self):          # (GraphedLearner.capture reads the observation / action widths off `actor`)
        return self.actor1

    def _flag(self):
        return self.total_it % 2 == 0          # update_a1, DADDPG_mlp.py:119

    def _nets(self):
        return (self.actor1, self.actor2, self.critic, self.target_actor1, self.target_actor2, self.target_critic)

    def _opts(self):
        return (self.actor1_opt, self.actor2_opt, self.critic_opt)

    def _update(self, s, a, r, s2, d, update_a1):
        """DADDPG_mlp.py:117-171: the critic regresses on r + gamma (1 - done) min over the two target actors' proposals as the ONE
        target critic values them; then one of the two actors ascends the critic -- actor 1 on even updates (with its own target's soft
        update only), actor 2 on odd ones (with the soft updates of its target AND the critic's)."""
        with torch.no_grad():
            tq = torch.min(self.target_critic(s2, self.target_actor1(s2)), self.target_critic(s2, self.target_actor2(s2)))
            target_q = r + (1 - d) * self.gamma * tq
        critic_loss = mean_sq(self.critic(s, a) - target_q)                      # F.mse_loss, DADDPG_mlp.py:142
        self._step(critic_loss, self.critic_opt)
        if update_a1:
            self._step(neg_mean(self.critic(s, self.actor1(s))), self.actor1_opt)
            self._soft_update(self.actor1, self.target_actor1)
        else:
            self._step(neg_mean(self.critic(s, self.actor2(s))), self.actor2_opt)
            self._soft_update(self.actor2, self.target_actor2)
            self._soft_update(self.critic, self.target_critic)
        return critic_loss.detach()

    @torch.no_grad()
    def take_action(self, state):
        """DADDPG_MLP.take_action (DADDPG_mlp.py:77-97): one state -> np.float32[action_dim], no exploration noise (the caller adds
        it, main.py:116-117); one host round trip, like the reference.  Batched and fused: BatchedArmEnv.set_policy_daddpg."""
        import numpy as np
        s = torch.tensor(np.asarray([state], dtype=np.float32), device=self.device)
        a1, a2 = self.actor1(s), self.actor2(s)
        q1, q2 = self.critic(s, a1), self.critic(s, a2)
        return (a1 if bool(q1 >= q2) else a2).cpu().numpy()[0]

    def policy_state_dicts(self):
        """(actor1, actor2, critic) for BatchedArmEnv.set_policy_daddpg"""
        return tuple({k: v.detach() for k, v in n.state_dict().items()} for n in (self.actor1, self.actor2, self.critic))
