"""Minimal asymmetric-actor-critic PPO for the TriFinger env (BASELINE config 5 when `rl_games` is absent).

Follows the agent configuration the reference ships for RL-Games (resources/config/rlg/asymm.yaml): continuous
A2C/PPO, actor MLP [400, 200, 100] ELU on `obs`, central value MLP [400, 200, 100] ELU on `states`, state-independent
log-std (`fixed_sigma`), horizon `steps_num` 32, 4 mini-epochs, minibatch = num_envs, gamma 0.99, GAE tau 0.95,
lr 3e-4 adaptive on a KL threshold of 0.008, e_clip 0.2, critic_coef 4, reward scale 0.01, grad-norm 1.0,
bounds loss 1e-4, normalised advantages.

This is host-side training glue, NOT part of the measured hot path: networks are plain `torch.nn` (rocBLAS GEMMs).
Data parallelism: every rank owns an env shard (leibnizgym_amd.sharding) and its own rollout; gradients are averaged
with ONE all-reduce of a flat buffer per minibatch (`torch.distributed`, backend nccl = RCCL over xGMI on the GPU
box, gloo in the CPU tests) - ~1 MB, latency-bound, so a single fused collective is the right shape.
"""
import math
from dataclasses import dataclass, field
from typing import List

import torch
import torch.nn as nn


@dataclass
class PPOConfig:
    units: List[int] = field(default_factory=lambda: [400, 200, 100])
    horizon: int = 32                 # steps_num
    mini_epochs: int = 4
    minibatches: int = 32             # minibatch_size = num_envs  ->  horizon minibatches per epoch
    gamma: float = 0.99
    tau: float = 0.95
    lr: float = 3e-4
    lr_value: float = 5e-4            # central_value_config.lr
    kl_threshold: float = 0.008       # lr_schedule: adaptive
    e_clip: float = 0.2
    critic_coef: float = 4.0
    reward_scale: float = 0.01
    grad_norm: float = 1.0
    bounds_loss_coef: float = 1e-4
    entropy_coef: float = 0.0
    normalize_advantage: bool = True
    seed: int = 7


def mlp(inp, units, out):
    layers, last = [], inp
    for u in units:
        layers += [nn.Linear(last, u), nn.ELU()]
        last = u
    layers.append(nn.Linear(last, out))
    return nn.Sequential(*layers)


class ActorCritic(nn.Module):
    def __init__(self, obs_dim, state_dim, act_dim, units):
        super().__init__()
        self.actor = mlp(obs_dim, units, act_dim)
        self.critic = mlp(state_dim if state_dim > 0 else obs_dim, units, 1)
        self.log_std = nn.Parameter(torch.zeros(act_dim))          # sigma_init const 0, fixed_sigma
        self.central = state_dim > 0

    def value(self, obs, states):
        return self.critic(states if self.central else obs).squeeze(-1)

    def dist(self, obs):
        mu = self.actor(obs)
        return mu, self.log_std.expand_as(mu)


def neglogp(x, mu, log_std):
    return (0.5 * ((x - mu) / log_std.exp()).pow(2) + log_std + 0.5 * math.log(2 * math.pi)).sum(-1)


class PPOTrainer:
    """`env` is an RlGamesGpuEnvAdapter-like object: reset() -> {"obs","states"} or obs; step(a) -> (same, r, d, info)."""

    def __init__(self, env, obs_dim, state_dim, act_dim, cfg: PPOConfig = None, device="cuda:0", group=None):
        self.env, self.cfg, self.device, self.group = env, cfg or PPOConfig(), torch.device(device), group
        torch.manual_seed(self.cfg.seed)
        self.net = ActorCritic(obs_dim, state_dim, act_dim, self.cfg.units).to(self.device)
        # fused multi-tensor Adam on the GPU: the update is launch-bound (tiny MLPs), one kernel instead of ~60
        fused = self.device.type == "cuda"
        self.opt = torch.optim.Adam(self.net.parameters(), lr=self.cfg.lr, eps=1e-8, **({"fused": True} if fused else {}))
        self.lr = self.cfg.lr
        self.dist_on = False
        try:
            import torch.distributed as dist
            self.dist = dist
            self.dist_on = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        except Exception:
            self.dist = None
        if self.dist_on:                       # identical initial weights on every rank
            for p in self.net.parameters():
                self.dist.broadcast(p.data, src=0, group=group)
        self.last = self._unpack(env.reset())
        self.frames = 0

    @staticmethod
    def _unpack(o):
        if isinstance(o, dict):
            return o["obs"].clone(), o["states"].clone()
        return o.clone(), None

    @torch.no_grad()
    def rollout(self):
        c, T = self.cfg, self.cfg.horizon
        obs, states = self.last
        n = obs.shape[0]
        dev = self.device
        buf = dict(obs=torch.zeros(T, n, obs.shape[1], device=dev),
                   states=torch.zeros(T, n, states.shape[1], device=dev) if states is not None else None,
                   act=None, nlp=torch.zeros(T, n, device=dev), val=torch.zeros(T + 1, n, device=dev),
                   rew=torch.zeros(T, n, device=dev), done=torch.zeros(T, n, device=dev), mu=None)
        for t in range(T):
            mu, ls = self.net.dist(obs)
            a = mu + ls.exp() * torch.randn_like(mu)
            if buf["act"] is None:
                buf["act"] = torch.zeros(T, n, a.shape[1], device=dev)
                buf["mu"] = torch.zeros(T, n, a.shape[1], device=dev)
            buf["obs"][t], buf["act"][t], buf["mu"][t] = obs, a, mu
            if states is not None:
                buf["states"][t] = states
            buf["nlp"][t] = neglogp(a, mu, ls)
            buf["val"][t] = self.net.value(obs, states)
            out, r, d, _ = self.env.step(a)
            obs, states = self._unpack(out)
            buf["rew"][t] = r.to(dev) * c.reward_scale
            buf["done"][t] = d.to(dev).float()
        buf["val"][T] = self.net.value(obs, states)
        self.last = (obs, states)
        adv = torch.zeros(T, n, device=dev)
        last = torch.zeros(n, device=dev)
        for t in reversed(range(T)):
            nd = 1.0 - buf["done"][t]
            delta = buf["rew"][t] + c.gamma * buf["val"][t + 1] * nd - buf["val"][t]
            last = delta + c.gamma * c.tau * nd * last
            adv[t] = last
        buf["ret"] = adv + buf["val"][:T]
        buf["adv"] = adv
        self.frames += T * n
        return buf

    def _allreduce_grads(self):
        if not self.dist_on:
            return
        grads = [p.grad for p in self.net.parameters() if p.grad is not None]
        flat = torch.cat([g.reshape(-1) for g in grads])
        self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, group=self.group)
        flat /= self.dist.get_world_size(self.group)
        off = 0
        for g in grads:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()

    def update(self, buf):
        c = self.cfg
        T, n = buf["nlp"].shape
        flat = lambda x: x.reshape(T * n, *x.shape[2:]) if x is not None else None  # noqa: E731
        obs, states, act = flat(buf["obs"]), flat(buf["states"]), flat(buf["act"])
        old_nlp, ret, adv, old_mu = flat(buf["nlp"]), flat(buf["ret"]), flat(buf["adv"]), flat(buf["mu"])
        if c.normalize_advantage:
            adv = (adv - adv.mean()) / (adv.std() + 1e-8)
        total = T * n
        mb = max(1, total // c.minibatches)
        zero = torch.zeros((), device=obs.device)
        acc = {"loss": zero.clone(), "a_loss": zero.clone(), "c_loss": zero.clone()}     # device-side, no per-minibatch sync
        stats = {"kl": 0.0}
        count = 0
        for _ in range(c.mini_epochs):
            perm = torch.randperm(total, device=obs.device)
            kls = []
            for s in range(0, total - mb + 1, mb):
                idx = perm[s:s + mb]
                mu, ls = self.net.dist(obs[idx])
                nlp = neglogp(act[idx], mu, ls)
                ratio = (old_nlp[idx] - nlp).exp()
                a = adv[idx]
                a_loss = torch.max(-a * ratio, -a * ratio.clamp(1 - c.e_clip, 1 + c.e_clip)).mean()
                v = self.net.value(obs[idx], states[idx] if states is not None else None)
                c_loss = (v - ret[idx]).pow(2).mean()
                b_loss = ((mu - 1.1).clamp(min=0).pow(2) + (-1.1 - mu).clamp(min=0).pow(2)).sum(-1).mean()
                ent = (ls + 0.5 + 0.5 * math.log(2 * math.pi)).sum(-1).mean()
                loss = a_loss + 0.5 * c.critic_coef * c_loss - c.entropy_coef * ent + c.bounds_loss_coef * b_loss
                self.opt.zero_grad(set_to_none=True)
                loss.backward()
                self._allreduce_grads()
                nn.utils.clip_grad_norm_(self.net.parameters(), c.grad_norm, foreach=True)
                self.opt.step()
                with torch.no_grad():      # KL between the old and new diagonal Gaussians (same sigma)
                    kl = (0.5 * ((mu - old_mu[idx]) / ls.exp()).pow(2)).sum(-1).mean()
                kls.append(kl)
                acc["loss"] += loss.detach(); acc["a_loss"] += a_loss.detach(); acc["c_loss"] += c_loss.detach()
                count += 1
            kl = torch.stack(kls).mean()
            if self.dist_on:
                self.dist.all_reduce(kl, op=self.dist.ReduceOp.SUM, group=self.group)
                kl /= self.dist.get_world_size(self.group)
            kl = float(kl)
            stats["kl"] = kl
            if kl > 2.0 * c.kl_threshold:      # rl_games AdaptiveScheduler
                self.lr = max(self.lr / 1.5, 1e-6)
            elif kl < 0.5 * c.kl_threshold:
                self.lr = min(self.lr * 1.5, 1e-2)
            for g in self.opt.param_groups:
                g["lr"] = self.lr
        for k in ("loss", "a_loss", "c_loss"):
            stats[k] = float(acc[k]) / max(count, 1)
        stats["lr"] = self.lr
        stats["mean_reward"] = float(buf["rew"].mean() / c.reward_scale)
        return stats

    def train(self, epochs, log=None):
        out = []
        for e in range(epochs):
            st = self.update(self.rollout())
            st["epoch"], st["frames"] = e, self.frames
            out.append(st)
            if log:
                log(st)
        return out
