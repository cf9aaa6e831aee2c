"""Is it the random-number call inside the captured TD3 update?  (round 6; profiles/r06_td3_hipgraph_learning.txt)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "drl-on-robot-arm_amd"))
import torch
import torch.nn.functional as F
from armenv import train, td3
class NoRng(td3.TD3):
    """target-policy smoothing off AND no randn call in the update"""
    def _update(self, s, a, r, s2, d, with_actor):
        with torch.no_grad():
            a2 = self.target_actor(s2).clamp(-self.action_bound, self.action_bound)
            tq1, tq2 = self.target_critic(s2, a2)
            target_q = r + (1 - d) * self.gamma * torch.min(tq1, tq2)
        q1, q2 = self.critic(s, a)
        critic_loss = F.mse_loss(q1, target_q) + F.mse_loss(q2, target_q)
        self._step(critic_loss, self.critic_opt)
        if with_actor:
            self._step(-self.critic.q1(s, self.actor(s)).mean(), self.actor_opt)
            self._soft_update(self.actor, self.target_actor)
            self._soft_update(self.critic, self.target_critic)
        return critic_loss.detach()
class NoiseOutside(td3.TD3):
    """the noise drawn eagerly into a static buffer before every replay"""
    def _update(self, s, a, r, s2, d, with_actor):
        if not hasattr(self, "_noise") or self._noise.shape != a.shape:
            self._noise = torch.zeros_like(a)
        with torch.no_grad():
            noise = (self._noise * self.policy_noise).clamp(-self.noise_clip, self.noise_clip)
            a2 = (self.target_actor(s2) + noise).clamp(-self.action_bound, self.action_bound)
            tq1, tq2 = self.target_critic(s2, a2)
            target_q = r + (1 - d) * self.gamma * torch.min(tq1, tq2)
        q1, q2 = self.critic(s, a)
        critic_loss = F.mse_loss(q1, target_q) + F.mse_loss(q2, target_q)
        self._step(critic_loss, self.critic_opt)
        if with_actor:
            self._step(-self.critic.q1(s, self.actor(s)).mean(), self.actor_opt)
            self._soft_update(self.actor, self.target_actor)
            self._soft_update(self.critic, self.target_critic)
        return critic_loss.detach()
    def train_graphed(self, batch):
        self._noise.normal_()
        return super().train_graphed(batch)
    def train(self, batch):
        if hasattr(self, "_noise"):
            self._noise.normal_()
        return super().train(batch)
for cls in (NoRng, NoiseOutside):
    train.TD3 = cls
    for seed in (0, 1):
        for graphs in (True, False):
            hist = []
            t0 = time.perf_counter()
            train.train_reach(iterations=160, log_every=20, log=lambda s: hist.append(json.loads(s)), use_graphs=graphs, seed=seed)
            print("%-13s %-9s seed %d: %s  %.1f s" % (cls.__name__, "hipGraphs" if graphs else "eager", seed, [round(h["success_rate"], 2) for h in hist], time.perf_counter() - t0), flush=True)
