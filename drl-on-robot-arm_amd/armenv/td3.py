"""Batched TD3 learner on PyTorch-ROCm tensors: the counterpart of
/root/reference/algo/TD3/TD3_mlp.py:33-161 + net_mlp.py:29-71 that consumes device-resident HER batches
(armenv.replay.TrajectoryStore.sample) without a host round trip and hands its actor to the env engine for fused
rollouts (BatchedArmEnv.set_policy).  Stock torch ops: the learner is integration, not a kernel."""
import copy

import torch
import torch.nn as nn
import torch.nn.functional as F


class _CaptureSafeLinear(torch.autograd.Function):
    """y = x W^T + b whose backward contains no multi-block REDUCTION: the bias gradient is a GEMM with a row of ones.

    Why: torch's reductions initialise their cross-block semaphores with cudaMemsetAsync (ATen/native/cuda/Reduce.cuh), and on this
    ROCm 7.2 / PyTorch 2.10 build a hipMemsetAsync captured into a hipGraph does its work on the FIRST replay only -- later replays
    write pointer-like garbage to the destination (tests/tools/exp/graph_memset_probe.py: 49 of 50 replays wrong for every size from
    64 B to 64 KB; the eager call is right 50 of 50).  A captured `grad_output.sum(0)` therefore goes wrong from the second replay
    on: in the captured TD3 / DADDPG updates every gradient was bit-identical to the eager one EXCEPT the bias gradients, off by 0.4-0.7
    on 103 of 150 updates, and the replayed learners did not learn (profiles/r06_td3_hipgraph_learning.txt).  Used by `linear()` below
    under stream capture only; the eager path stays `F.linear` (bit-identical to the reference's modules)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return torch.addmm(b, x, w.t())

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gx = g.mm(w) if ctx.needs_input_grad[0] else None
        gw = g.t().mm(x) if ctx.needs_input_grad[1] else None
        gb = torch.ones(1, g.shape[0], dtype=g.dtype, device=g.device).mm(g)[0] if ctx.needs_input_grad[2] else None
        return gx, gw, gb


def linear(layer, x):
    """`layer(x)` for an nn.Linear: F.linear eagerly, _CaptureSafeLinear while the current stream is being captured into a hipGraph"""
    if x.is_cuda and torch.cuda.is_current_stream_capturing():
        return _CaptureSafeLinear.apply(x, layer.weight, layer.bias)
    return layer(x)


def mean_sq(d):
    """mean(d^2) of a [B, 1] column -- F.mse_loss's value -- as a 1 x 1 GEMM under capture (no multi-block reduction, see above)"""
    if d.is_cuda and torch.cuda.is_current_stream_capturing():
        return (d.t().mm(d) / d.shape[0])[0, 0]
    return (d * d).mean()


def neg_mean(q):
    """-mean(q) of a [B, 1] column (the actors' loss), as a GEMM with a row of ones under capture"""
    if q.is_cuda and torch.cuda.is_current_stream_capturing():
        return -(torch.ones(1, q.shape[0], dtype=q.dtype, device=q.device).mm(q) / q.shape[0])[0, 0]
    return -q.mean()


class Actor(nn.Module):
    """a = bound * tanh(fc3(relu(fc2(relu(fc1(s))))))   (net_mlp.py:29-40; parameter names fc1/fc2/fc3 so that a
    reference state_dict loads unchanged)"""

    def __init__(self, state_dim, hidden_dim, action_dim, action_bound):
        super().__init__()
        self.fc1, self.fc2, self.fc3 = nn.Linear(state_dim, hidden_dim), nn.Linear(hidden_dim, hidden_dim), nn.Linear(hidden_dim, action_dim)
        self.action_bound = action_bound

    def forward(self, s):
        return torch.tanh(linear(self.fc3, F.relu(linear(self.fc2, F.relu(linear(self.fc1, s)))))) * self.action_bound


class TwinCritic(nn.Module):
    """two Q heads over cat(state, action) in one module (net_mlp.py:43-71; fc1-3 = Q1, fc4-6 = Q2)"""

    def __init__(self, state_dim, hidden_dim, action_dim):
        super().__init__()
        d = state_dim + action_dim
        self.fc1, self.fc2, self.fc3 = nn.Linear(d, hidden_dim), nn.Linear(hidden_dim, hidden_dim), nn.Linear(hidden_dim, 1)
        self.fc4, self.fc5, self.fc6 = nn.Linear(d, hidden_dim), nn.Linear(hidden_dim, hidden_dim), nn.Linear(hidden_dim, 1)

    def q1(self, s, a):
        x = torch.cat([s, a], dim=1)
        return linear(self.fc3, F.relu(linear(self.fc2, F.relu(linear(self.fc1, x)))))

    def forward(self, s, a):
        x = torch.cat([s, a], dim=1)
        return (linear(self.fc3, F.relu(linear(self.fc2, F.relu(linear(self.fc1, x))))),
                linear(self.fc6, F.relu(linear(self.fc5, F.relu(linear(self.fc4, x))))))


class GraphedLearner:
    """What the learners of this package share: soft updates, `train(batch)` from a dict of device tensors, and the update replayed
    from hipGraphs.  A subclass provides `_update(s, a, r, s2, d, flag)` (one update; `flag` selects which of its two variants runs),
    `_flag()` (the variant of update number self.total_it), `_nets()` and `_opts()` (everything an update may write)."""

    tau = 0.005
    total_it = 0
    _graphs = None

    @torch.no_grad()
    def _soft_update(self, net, target):
        for pt, p in zip(target.parameters(), net.parameters()):
            pt.mul_(1.0 - self.tau).add_(p, alpha=self.tau)

    @staticmethod
    def _step(loss, opt):
        """loss.backward() + opt.step() of the reference's updates with the .grad bookkeeping made explicit: the gradients of `loss` with
        respect to the parameters `opt` owns -- and nothing else -- go into PERSISTENT .grad buffers (made once, written by a copy), so
        that the two captured variants of an update and the eager path share one set of gradient tensors in ordinary memory.
        (`loss.backward()` also leaves gradients on every other parameter the loss touches -- the actor loss on the critic's, which the
        reference wipes with the next zero_grad -- and under capture re-makes the .grad tensors inside the capturing graph's private
        pool.)  Bit-identical to backward() + step() eagerly."""
        params = [p for g in opt.param_groups for p in g["params"]]
        grads = torch.autograd.grad(loss, params)
        with torch.no_grad():
            for p, g in zip(params, grads):
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                p.grad.copy_(g)
        opt.step()

    def train(self, batch):
        """One update from a dict of device tensors: states [B,D], actions [B,3], next_states [B,D], rewards [B], dones [B] (any
        dtype).  Returns the critic loss as a 0-dim tensor (no host sync)."""
        s = batch["states"].to(self.device, torch.float32)
        a = batch["actions"].to(self.device, torch.float32)
        r = batch["rewards"].to(self.device, torch.float32).view(-1, 1)
        s2 = batch["next_states"].to(self.device, torch.float32)
        d = batch["dones"].to(self.device, torch.float32).view(-1, 1)
        self.total_it += 1
        return self._update(s, a, r, s2, d, self._flag())

    # ---- hipGraph path: one update is ~130 small kernels (1.7 ms of launch latency at any batch size up to 16 k);
    # replayed from a captured graph it costs its kernel time only.
    def capture(self, batch_size):
        """Captures the two update variants as hipGraphs over static input buffers of `batch_size` rows; ``train_graphed`` then
        replays them.  Parameters and optimiser state are left exactly as they were (the warm-up and capture passes run on a
        snapshot that is restored)."""
        dev, B = self.device, int(batch_size)
        D, A = self.actor.fc1.in_features, self.actor.fc3.out_features
        buf = dict(states=torch.zeros(B, D, device=dev), actions=torch.zeros(B, A, device=dev),
                   next_states=torch.zeros(B, D, device=dev), rewards=torch.zeros(B, device=dev),
                   dones=torch.zeros(B, dtype=torch.uint8, device=dev))
        loss = torch.zeros((), device=dev)
        nets, opts = self._nets(), self._opts()

        def run(flag):
            out = self._update(buf["states"], buf["actions"], buf["rewards"].view(-1, 1), buf["next_states"],
                               buf["dones"].to(torch.float32).view(-1, 1), flag)
            loss.copy_(out)

        def snapshot():
            return ([{k: v.clone() for k, v in n.state_dict().items()} for n in nets], [copy.deepcopy(o.state_dict()) for o in opts])

        def restore(snap):
            with torch.no_grad():
                for n, sd in zip(nets, snap[0]):
                    for k, v in n.state_dict().items():
                        v.copy_(sd[k])
                # in place (the graphs alias these tensors); state that did not exist before the warm-up goes back to a
                # fresh Adam's: zero moments, step 0
                for opt, saved in zip(opts, (x["state"] for x in snap[1])):
                    for pid, p in enumerate(opt.param_groups[0]["params"]):
                        for k, v in opt.state[p].items():
                            if pid in saved:
                                v.copy_(saved[pid][k])
                            else:
                                v.zero_()

        first = snapshot()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                       # warm-up: allocates grads and optimiser state
            for _ in range(2):
                run(True)
                run(False)
        torch.cuda.current_stream(dev).wait_stream(side)
        graphs = {}
        for flag in (False, True):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                run(flag)
            graphs[flag] = g
        # optimiser state may not have existed before the warm-up: "restore" then means zero moments and step 0
        restore(first)
        torch.cuda.synchronize(dev)
        self._graphs = dict(buf=buf, loss=loss, g=graphs, B=B)
        return buf

    def train_graphed(self, batch):
        """``train`` through the captured graphs: `batch` is copied into the static buffers (or IS the dict returned by
        ``capture`` / filled in place by ``TrajectoryStore.sample(out=...)``).  Returns the loss tensor of the replay."""
        g = self._graphs
        if g is None:
            raise RuntimeError("%s.train_graphed: call capture(batch_size) first" % type(self).__name__)
        if batch is not g["buf"]:
            for k, v in g["buf"].items():
                v.copy_(batch[k].view_as(v))
        self.total_it += 1
        g["g"][self._flag()].replay()
        return g["loss"]


class TD3(GraphedLearner):
    """Hyper-parameters default to config.py:55-73 (hidden 256, lr 1e-3, tau 0.005, gamma 0.98, policy noise 0.2,
    clip 0.5, delayed actor update every 3 critic updates)."""

    def __init__(self, state_dim, action_dim, action_bound, hidden_dim=256, actor_lr=1e-3, critic_lr=1e-3, tau=0.005,
                 gamma=0.98, policy_noise=0.2, noise_clip=0.5, policy_freq=3, device="cuda:0"):
        self.device = torch.device(device)
        self.actor = Actor(state_dim, hidden_dim, action_dim, action_bound).to(self.device)       # creation order as
        self.critic = TwinCritic(state_dim, hidden_dim, action_dim).to(self.device)               # TD3_mlp.py:61-64
        self.target_actor = Actor(state_dim, hidden_dim, action_dim, action_bound).to(self.device)
        self.target_critic = TwinCritic(state_dim, hidden_dim, action_dim).to(self.device)
        self.target_critic.load_state_dict(self.critic.state_dict())
        self.target_actor.load_state_dict(self.actor.state_dict())
        cap = self.device.type == "cuda"      # step counters on the device: the update can be captured in a hipGraph
        self.actor_opt = torch.optim.Adam(self.actor.parameters(), lr=actor_lr, capturable=cap)
        self.critic_opt = torch.optim.Adam(self.critic.parameters(), lr=critic_lr, capturable=cap)
        self.tau, self.gamma, self.action_bound = tau, gamma, action_bound
        self.policy_noise, self.noise_clip, self.policy_freq = policy_noise, noise_clip, policy_freq
        self.total_it = 0
        self._graphs = None

    def _flag(self):
        return self.total_it % self.policy_freq == 0        # delayed actor + soft updates, TD3_mlp.py:144

    def _nets(self):
        return (self.actor, self.critic, self.target_actor, self.target_critic)

    def _opts(self):
        return (self.actor_opt, self.critic_opt)

    def _update(self, s, a, r, s2, d, with_actor):
        """TD3_mlp.py:114-161"""
        with torch.no_grad():
            noise = (torch.randn_like(a) * self.policy_noise).clamp(-self.noise_clip, self.noise_clip)
            a2 = (self.target_actor(s2) + noise).clamp(-self.action_bound, self.action_bound)
            tq1, tq2 = self.target_critic(s2, a2)
            target_q = r + (1 - d) * self.gamma * torch.min(tq1, tq2)
        q1, q2 = self.critic(s, a)
        critic_loss = mean_sq(q1 - target_q) + mean_sq(q2 - target_q)          # F.mse_loss + F.mse_loss, TD3_mlp.py:137
        self._step(critic_loss, self.critic_opt)
        if with_actor:
            actor_loss = neg_mean(self.critic.q1(s, self.actor(s)))               # -mean(Q1), TD3_mlp.py:147
            self._step(actor_loss, self.actor_opt)
            self._soft_update(self.actor, self.target_actor)
            self._soft_update(self.critic, self.target_critic)
        return critic_loss.detach()

    def take_action(self, state):
        """TD3_MLP.take_action (TD3_mlp.py:82-97): one state (sequence of floats) -> np.float32[action_dim], no exploration noise
        (the caller adds it, main.py:116); one host round trip, like the reference."""
        import numpy as np
        with torch.no_grad():
            s = torch.tensor(np.asarray([state], dtype=np.float32), device=self.device)
            return self.actor(s).detach().cpu().numpy()[0]

    def actor_state_dict(self):
        return {k: v.detach() for k, v in self.actor.state_dict().items()}
